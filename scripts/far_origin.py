"""Far-origin / offset-scene parity sweep: rays that start 10^2 .. 10^5 scene extents away, aimed at
vertices, edge midpoints and box-edge-touching triangle edges, and the same scenes translated by 10^3 and 10^4.

The box pad applied at upload only knows the vertex magnitudes; the slab test's rounding and Moeller-Trumbore's own slop
grow with the ORIGIN's magnitude.  Every ray therefore widens the boxes by 2^-pad_log2 * max|o| (kernels.hip make_slab).
`run(pad_on)` counts hit records that differ between every BVH layout / kernel and the brute-force HIP kernel (= the
reference loop, frag.glsl:265-301), and between that kernel and the CPU oracle on a sample.  With the per-ray pad the
count must be 0 (tests/test_gpu_parity.py); without it (RT_OPT_RAY_PAD 0) the script prints how many hits are lost.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import oracle
import raytracer_amd as rt
from raytracer_amd import scenes

SCALES = (1e2, 1e3, 1e4, 1e5)
OFFSETS = ((0.0, 0.0, 0.0), (1e3, -1e3, 1e3), (1e4, 1e4, -1e4))


def _targets(tris):
    p = np.stack([tris[v]["position"].astype(np.float64) for v in ("v0", "v1", "v2")], 1)      # (n, 3, 3)
    verts = p.reshape(-1, 3)
    mids = ((p + np.roll(p, 1, axis=1)) / 2).reshape(-1, 3)                                     # edge midpoints
    near = (p * 0.999 + np.roll(p, 1, axis=1) * 0.001).reshape(-1, 3)                           # on an edge, next to a vertex
    cent = p.mean(axis=1)
    return np.unique(np.concatenate([verts, mids, near, cent]), axis=0)


def _dirs(rng, k):
    d = rng.standard_normal((k, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    fixed = np.array([(1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1), (1, 1, 1), (-1, 1, -1), (1, 1e-3, 0), (0, 1, 1e-4)], dtype=np.float64)
    fixed /= np.linalg.norm(fixed, axis=1, keepdims=True)
    return np.concatenate([fixed, d])


def _far_rays(tris, rng, dirs_per_scale):
    tg = _targets(tris)
    lo, hi = tg.min(axis=0), tg.max(axis=0)
    ctr, ext = (lo + hi) / 2, float(np.max(hi - lo))
    parts = []
    for s in SCALES:
        for d in _dirs(rng, dirs_per_scale):
            org = ctr + d * (s * ext)
            parts.append(scenes.rays_at(tg, np.broadcast_to(org, tg.shape)))
    return np.concatenate(parts), ctr, ext


def _diff(a, b):
    return int((a.view(np.uint32).reshape(-1, 4) != b.view(np.uint32).reshape(-1, 4)).any(axis=1).sum())


def run(pad_on=True, verbose=True, quick=False):
    rng = np.random.default_rng(777)
    cornell, sph, _ = scenes.cornell()
    g_flat, _ = scenes.grid_scene(13 if quick else 37, 0.0, rng)
    g_bump, _ = scenes.grid_scene(13 if quick else 37, 0.01, rng)
    base = [("cornell", cornell, sph, 22), ("grid coplanar + duplicates", scenes.cat([g_flat, g_flat[::3]]), None, 2 if quick else 6),
            ("grid jittered", g_bump, None, 2 if quick else 6)]
    tot = {"rays": 0, "bad": 0, "oracle": 0, "oracle_bad": 0, "tile_rays": 0, "tile_bad": 0}
    for name, tris0, sph0, k in base:
        for off in OFFSETS:
            tris = scenes.translated(tris0, off)
            s_off = None
            if sph0 is not None:
                s_off = sph0.copy()
                s_off["center"] = (s_off["center"] + np.asarray(off, dtype=np.float32)).astype(np.float32)
            rays, ctr, ext = _far_rays(tris, rng, k)
            with rt.Renderer((0,)) as r:
                r.set_option(rt.OPT_RAY_PAD, 1 if pad_on else 0)
                r.set_option(rt.OPT_TRAVERSAL, rt.TRAVERSAL_BRUTE)
                r.create_data(tris, s_off)
                ref = r.trace(rays)
                # tile kernel: far cameras with a narrow field of view that just frames the scene
                cams = []
                for s in SCALES[:3]:
                    for d in _dirs(rng, 0)[[0, 3, 6, 8]]:
                        pos = ctr + d * (s * ext)
                        fov = float(np.degrees(2 * np.arctan(0.75 / s)))
                        cams.append(rt.camera_look_at(tuple(pos), tuple(ctr), fov, 1.0))
                tile_ref = []
                for cam in cams:
                    r.update_camera(cam)
                    tile_ref.append(r.trace_primary(48, 48))
                bad = tile_bad = 0
                for mode in (rt.TRAVERSAL_BVH2, rt.TRAVERSAL_BVH4Q):
                    for strip in (0, 256):
                        r.set_option(rt.OPT_TRAVERSAL, mode)
                        r.set_option(rt.OPT_STRIP, strip)
                        r.create_data(tris, s_off)
                        bad += _diff(r.trace(rays), ref)
                        tot["rays"] += len(rays)
                    for cam, want in zip(cams, tile_ref):
                        r.update_camera(cam)
                        tile_bad += _diff(r.trace_primary(48, 48), want)
                        tot["tile_rays"] += want.size
                tot["bad"] += bad
                tot["tile_bad"] += tile_bad
            m = min(len(rays), 1500)
            idx = rng.choice(len(rays), m, replace=False)
            ob = _diff(oracle.trace(oracle.Scene(tris, s_off), rays[idx]), ref[idx])
            tot["oracle"] += m
            tot["oracle_bad"] += ob
            if verbose:
                hit = float((ref["prim"] != rt.MISS).mean())
                print(f"{name} offset {off}: {len(tris)} tris, {len(rays)} rays (hit {hit:.2f}): ray-buffer kernels vs brute {bad}, "
                      f"tile kernel vs brute {tile_bad}, brute vs oracle {ob}/{m}", flush=True)
    if verbose:
        print(f"TOTAL (per-ray pad {'on' if pad_on else 'OFF'}): {tot['rays']} ray-buffer results, {tot['bad']} differing records "
              f"(rate {tot['bad'] / max(tot['rays'], 1):.2e}); {tot['tile_rays']} tile-kernel results, {tot['tile_bad']} differing "
              f"(rate {tot['tile_bad'] / max(tot['tile_rays'], 1):.2e}); brute-force kernel vs CPU oracle {tot['oracle_bad']}/{tot['oracle']}", flush=True)
    return tot


if __name__ == "__main__":
    on = run(True)
    off = run(False, verbose="-v" in sys.argv)
    print(f"without the per-ray pad: {off['bad']} of {off['rays']} ray-buffer records and {off['tile_bad']} of {off['tile_rays']} "
          f"tile-kernel records differ from the reference loop", flush=True)
    sys.exit(1 if (on["bad"] or on["tile_bad"] or on["oracle_bad"]) else 0)
