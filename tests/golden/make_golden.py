"""Regenerates the golden fixtures under tests/golden/ from the CPU oracle (oracle/).

The reference has no tests, fixtures or CPU path, so these vectors are produced by
the build's restatement of shaders/frag.glsl, which is itself pinned by the hand-derived KATs of
SURVEY.md 8(c) (kat.json).  Inputs come from the deterministic host generators (rt_gen_*), which need no
GPU.  Run:  python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import oracle                      # noqa: E402
import raytracer_amd as rt         # noqa: E402
from raytracer_amd import scenes   # noqa: E402


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    meta = {}
    # KATs, hand-derived from the reference's scene data; exact in fp32
    kat = {"KAT-1": {"o": [0, 4, 1], "d": [0, -1, 0], "prim": 2, "t": 6.0, "u": 0.5, "v": 0.5},
           "KAT-2": {"o": [0.5, 1, 1], "d": [0, 0, -1], "prim": 0, "t": 1.0, "u": 0.625, "v": 0.25}}
    json.dump(kat, open(os.path.join(HERE, "kat.json"), "w"), indent=1)

    tris, sph, _ = scenes.cornell()
    cam = rt.default_camera(1.0)
    sc = oracle.Scene(tris, sph)
    np.save(os.path.join(HERE, "c1_cornell_64x64_hits.npy"), oracle.trace_primary(sc, cam, 64, 64))
    meta["c1_cornell_512x512_sha256"] = sha(oracle.trace_primary(sc, cam, 512, 512))
    meta["c1_cornell_rays_64x64_sha256"] = sha(oracle.primary_rays(cam, 64, 64))

    t32, s32, _ = scenes.cornell32()
    np.save(os.path.join(HERE, "c1_32_cornell_64x64_hits.npy"), oracle.trace_primary(oracle.Scene(t32, s32), cam, 64, 64))

    # ragged image (not a multiple of the 8x8 tile) with the reference window aspect
    camw = rt.default_camera()
    np.save(os.path.join(HERE, "c1_cornell_61x37_hits.npy"), oracle.trace_primary(sc, camw, 61, 37))

    # small soup + random rays
    st = scenes.soup(2000, half_extent=0.05)
    rays = scenes.random_rays(4096)
    scs = oracle.Scene(st)
    hits = oracle.trace(scs, rays)
    np.save(os.path.join(HERE, "soup2k_rays4k_hits.npy"), hits)
    meta["soup2k_tris_sha256"] = sha(st)
    meta["rays4k_sha256"] = sha(rays)
    # shadow + bounce stages on the Cornell 64x64 frame (one light above the box centre)
    pr = oracle.primary_rays(cam, 64, 64)
    ph = oracle.trace(sc, pr)
    lit, sr = oracle.shadow_lit(sc, pr, ph, (0.0, 0.0, 1.9), want_rays=True)
    np.save(os.path.join(HERE, "c1_cornell_64x64_lit.npy"), lit)
    br = oracle.bounce(sc, pr, ph, seed=7)
    np.save(os.path.join(HERE, "c1_cornell_64x64_bounce_rays.npy"), br)
    # jittered 4-spp primary rays
    meta["c1_cornell_rays_32x32x4_seed1_sha256"] = sha(oracle.primary_rays(cam, 32, 32, spp=4, seed=1, jitter=True))
    # shading + path loop: linear framebuffer of a small Cornell render (lights: none, as shipped)
    lin, _ = oracle.render(sc, cam, 32, 32, 2, 3, 1)
    meta["c1_render_32x32_spp2_b3_seed1_linear_sha256"] = sha(lin)
    json.dump(meta, open(os.path.join(HERE, "meta.json"), "w"), indent=1)
    print(json.dumps(meta, indent=1))


if __name__ == "__main__":
    main()
