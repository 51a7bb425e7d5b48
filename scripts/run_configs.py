"""Runs the BASELINE.json configs C1..C5 on ONE MI355X and prints one JSON line per config (results table of
DESIGN.md).  Parity gates (prim-ID mismatches against the oracle / the brute-force HIP kernel) ride along.

usage: python scripts/run_configs.py [c0 c1 c2 c3 c3b c4 c5] [--soup N] [--rays N]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import oracle
import raytracer_amd as rt
from raytracer_amd import scenes

dev = torch.device("cuda", 0)
f32 = lambda n: torch.empty(n, dtype=torch.float32, device=dev)


def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def counters_of(r, fn):
    r.set_option(rt.OPT_COUNTERS, 1); r.counters(); fn(); torch.cuda.synchronize()
    c = r.counters(); r.set_option(rt.OPT_COUNTERS, 0)
    rays = max(c["rays"], 1)
    return {"B": round(c["box_tests"] / rays, 2), "N": round(c["nodes"] / rays, 2), "T": round(c["tri_tests"] / rays, 2)}


def sample_parity(tris, sph, rays_np, hits_np, k, seed=0):
    idx = np.sort(np.random.default_rng(seed).choice(len(rays_np), min(k, len(rays_np)), replace=False))
    t = time.time()
    ref = oracle.trace(oracle.Scene(tris, sph), rays_np[idx])
    dt = time.time() - t
    got = hits_np[idx]
    return {"sample": int(len(idx)), "prim_id_mismatches": int((ref["prim"] != got["prim"]).sum()),
            "bit_exact": bool(ref.tobytes() == got.tobytes()), "oracle_s": round(dt, 1),
            "oracle_Mrays_s": round(len(idx) / dt / 1e6, 5), "oracle_threads": oracle.num_threads()}


def out(d):
    print(json.dumps(d), flush=True)


def c0():
    """The reference's own frame: the Cornell box of VkRenderer.cpp:171-250 in its window (1200x1000, globals.h:12-13) at the
    shader's SAMPLES 10 / BOUNCES 8 (frag.glsl:3-4) through rt_render, host RGBA8 out -- the workload behind the ms/frame (FPS)
    text the reference shows (VkRenderer.cpp:1405).  Beside it the oracle's frame on all host cores, and a gate at a tenth of
    the size: RGBA8 within 1 LSB of orc_render (device powf vs libm powf; the linear frame is bit-exact, tests/)."""
    tris, sph, _ = scenes.cornell()
    w, h, spp, bounces = 1200, 1000, 10, 8
    cam = rt.default_camera(w / h)
    res = {"config": f"C0 reference frame: Cornell 12 tris + 3 spheres, {w}x{h}, {spp} spp x {bounces} bounces, rt_render -> host RGBA8"}
    with rt.Renderer((0,)) as r:
        r.create_data(tris, sph); r.update_camera(cam)
        r.render(w, h, spp, bounces, 1)
        t = time.perf_counter(); K = 10
        for k in range(K):
            fr = r.render(w, h, spp, bounces, 1 + k)
        ms = (time.perf_counter() - t) / K * 1e3
        res.update({"ms_per_frame": round(ms, 3), "fps": round(1e3 / ms, 1), "path_segments_per_frame_max": w * h * spp * bounces,
                    "mean_rgb": round(float(fr[..., :3].mean()), 2)})
        sw, sh = w // 10, h // 10
        got = r.render(sw, sh, spp, bounces, 3)
        t = time.perf_counter()
        _, want = oracle.render(oracle.Scene(tris, sph), rt.default_camera(sw / sh), sw, sh, spp, bounces, 3)
        dt = time.perf_counter() - t
        r.update_camera(rt.default_camera(sw / sh)); got = r.render(sw, sh, spp, bounces, 3)
        res["gate"] = {"size": [sw, sh], "max_abs_rgba8_diff": int(np.abs(got.astype(np.int32) - want.astype(np.int32)).max()),
                       "identical_pixels_frac": round(float((got == want).all(axis=-1).mean()), 5)}
        res["cpu_oracle"] = {"ms_per_frame_extrapolated": round(dt * 1e3 * (w * h) / (sw * sh), 1), "threads": oracle.num_threads(),
                             "note": f"orc_render on {sw}x{sh} ({dt:.2f} s), scaled by the pixel ratio"}
    out(res)


def c1():
    tris, sph, _ = scenes.cornell()
    cam = rt.default_camera(1.0)
    res = {"config": "C1 Cornell 12 tris + 3 spheres, 512x512 primary"}
    ref = oracle.trace_primary(oracle.Scene(tris, sph), cam, 512, 512)
    for name, mode in (("brute(auto)", rt.TRAVERSAL_AUTO), ("bvh2", rt.TRAVERSAL_BVH2), ("bvh4q", rt.TRAVERSAL_BVH4Q)):
        with rt.Renderer((0,)) as r:
            r.set_option(rt.OPT_TRAVERSAL, mode); r.create_data(tris, sph); r.update_camera(cam)
            hits = f32(512 * 512 * 4)
            ms = timeit(lambda: r.trace_primary_dev(512, 512, hits), 20)
            got = hits.cpu().numpy().view(rt.HIT).reshape(512, 512)
            res[name] = {"ms": round(ms, 4), "Mrays_s": round(512 * 512 / ms / 1e3, 1),
                         "full_frame_bit_exact": bool(got.tobytes() == ref.tobytes()),
                         "prim_id_mismatches": int((got["prim"] != ref["prim"]).sum())}
    out(res)


def c2():
    tris, sph, lights, cam = scenes.bunny_class()
    w, h = 1920, 1080
    n = w * h
    res = {"config": f"C2 bunny-class stand-in {len(tris)} tris, 1920x1080 primary + 1 shadow ray per hit pixel"}
    for name, mode in (("bvh2", rt.TRAVERSAL_BVH2), ("bvh4q", rt.TRAVERSAL_BVH4Q)):
        with rt.Renderer((0,)) as r:
            r.set_option(rt.OPT_TRAVERSAL, mode); st = r.create_data(tris, sph, lights); r.update_camera(cam)
            rays, hits = f32(n * 8), f32(n * 4)
            lit = torch.empty(n, dtype=torch.uint8, device=dev)
            lp = tuple(float(x) for x in lights[0]["position"])
            ms_p = timeit(lambda: r.trace_primary_dev(w, h, hits, rays))
            ms_s = timeit(lambda: r.shadow_dev(rays, hits, n, lp, lit))
            ms_sp = None
            if mode == rt.TRAVERSAL_BVH4Q:
                # the shadow rays start at the hits of primary rays in pixel order: coherent, so they may walk as packets
                # (RT_OPT_RAY_PACKETS); the flags must not change
                lit_p = torch.empty(n, dtype=torch.uint8, device=dev)
                r.set_option(rt.OPT_RAY_PACKETS, 1)
                ms_sp = timeit(lambda: r.shadow_dev(rays, hits, n, lp, lit_p))
                r.set_option(rt.OPT_RAY_PACKETS, 0)
                torch.cuda.synchronize()
                assert torch.equal(lit, lit_p), "shadow flags differ between per-lane and packet walks"
            hh = hits.cpu().numpy().view(rt.HIT)
            n_shadow = int((hh["prim"] != rt.MISS).sum())
            par = sample_parity(tris, sph, rays.cpu().numpy().view(rt.RAY), hh, 16384)
            idx = np.sort(np.random.default_rng(1).choice(n, 16384, replace=False))
            o_lit = oracle.shadow_lit(oracle.Scene(tris, sph), rays.cpu().numpy().view(rt.RAY)[idx], hh[idx], lp)
            par["shadow_flag_mismatches"] = int((o_lit != lit.cpu().numpy()[idx]).sum())
            res[name] = {"primary_ms": round(ms_p, 4), "shadow_ms": round(ms_s, 4), "shadow_rays": n_shadow,
                         **({"shadow_packets_ms": round(ms_sp, 4), "Mrays_s_with_shadow_packets": round((n + n_shadow) / (ms_p + ms_sp) / 1e3, 1)} if ms_sp else {}),
                         "Mrays_s": round((n + n_shadow) / (ms_p + ms_s) / 1e3, 1),
                         "primary_Mrays_s": round(n / ms_p / 1e3, 1), "shadow_Mrays_s": round(n_shadow / ms_s / 1e3, 1),
                         "work_primary": counters_of(r, lambda: r.trace_primary_dev(w, h, hits, rays)),
                         "bvh": {"nodes": st["n_nodes"], "depth": st["max_depth"], "build_ms": round(st["build_ms"], 1)},
                         "parity": par}
    out(res)


def c3(which):
    """The bench frame (4K x 4 spp: primary + segment 0, bounce closest hit, last segment + resolve) on the sphere-grid
    stand-in of SURVEY 8(d) ("sponza") or on the harder colonnade hall ("colonnade"), with the work counters and parity gates."""
    tris, sph, lights, cam = scenes.sponza_class() if which == "sponza" else scenes.colonnade()
    w, h, spp = 3840, 2160, 4
    n = w * h * spp
    label = ("C3 sponza-class stand-in (8x8 sphereGeometry(45,45) in a room)" if which == "sponza" else
             "C3b colonnade hall (long thin floor/wall/beam triangles, fluted columns, arches, vault, balusters, drapes; edge lengths 0.014..36)")
    res = {"config": f"{label}, {len(tris)} tris, 3840x2160 x 4 spp primary + 1 bounce + shading"}
    with rt.Renderer((0,)) as r:
        st = r.create_data(tris, sph, lights); r.update_camera(cam)
        rays, brays, phits, bhits, thr, rad = f32(n * 8), f32(n * 8), f32(n * 4), f32(n * 4), f32(n * 4), f32(n * 4)
        fb = torch.empty(w * h, dtype=torch.int32, device=dev)
        prim = lambda: r.primary_shade_stripes_dev(w, h, 0, h, h, 1, thr, rad, brays, spp=spp, seed=1, jitter=True, shade_seed=2)
        bounce = lambda: r.trace_dev(brays, n, bhits)
        last = lambda: r.shade_resolve_dev(brays, bhits, n, spp, thr, rad, None, fb)
        prim(); bounce(); last(); torch.cuda.synchronize()
        ms_p, ms_b, ms_l = timeit(prim, 10), timeit(bounce, 10), timeit(last, 10)
        unf = lambda: r.trace_primary_dev(w, h, phits, rays, spp=spp, seed=1, jitter=True)
        wp = counters_of(r, unf)
        wb = counters_of(r, bounce)
        live_b = int((brays.view(-1, 8)[:, 7] > 0).sum())
        res.update({"primary_shade_ms": round(ms_p, 3), "bounce_ms": round(ms_b, 3), "shade_resolve_ms": round(ms_l, 3),
                    "frame_ms": round(ms_p + ms_b + ms_l, 3), "Mrays_s": round((n + live_b) / (ms_p + ms_b + ms_l) / 1e3, 1),
                    "primary_Mrays_s": round(n / ms_p / 1e3, 1), "bounce_Mrays_s": round(live_b / ms_b / 1e3, 1), "live_bounce_rays": live_b,
                    "work_primary": wp, "work_bounce": wb,
                    "bvh": {"nodes": st["n_nodes"], "depth": st["max_depth"], "sah_cost": round(st["sah_cost"], 1), "build_ms": round(st["build_ms"], 1)},
                    "nonblack_pixels": int((fb.cpu().numpy() & 0xFFFFFF != 0).sum())})
        # parity: 64 k primary + 64 k bounce rays against the reference loop on the GPU, 4 k + 4 k against the CPU oracle
        K = 1 << 16
        idx = torch.from_numpy(np.sort(np.random.default_rng(7).choice(n, K, replace=False))).to(dev)
        r.set_option(rt.OPT_TRAVERSAL, rt.TRAVERSAL_BRUTE)
        par = {"sample_rays": 2 * K, "prim_id_mismatches_vs_gpu_brute": 0, "bit_exact_vs_gpu_brute": True, "oracle_sample": 8192, "prim_id_mismatches_vs_oracle": 0}
        sc = oracle.Scene(tris, sph)
        for rb, hb in ((rays, phits), (brays, bhits)):
            s_r, s_h = rb.view(-1, 8)[idx].contiguous(), hb.view(-1, 4)[idx].contiguous()
            bh = f32(K * 4)
            r.trace_dev(s_r.view(-1), K, bh); torch.cuda.synchronize()
            got, ref = s_h.cpu().numpy().view(rt.HIT).ravel(), bh.cpu().numpy().view(rt.HIT).ravel()
            par["prim_id_mismatches_vs_gpu_brute"] += int((got["prim"] != ref["prim"]).sum())
            par["bit_exact_vs_gpu_brute"] = bool(par["bit_exact_vs_gpu_brute"] and got.tobytes() == ref.tobytes())
            o = oracle.trace(sc, s_r.cpu().numpy().view(rt.RAY).ravel()[:4096])
            par["prim_id_mismatches_vs_oracle"] += int((o["prim"] != got["prim"][:4096]).sum())
        res["parity"] = par
    out(res)


def c4(n_tris, n_rays):
    res = {"config": f"C4 random soup {n_tris} tris (half extent 0.005), {n_rays} random rays, one GPU"}
    t = time.time(); tris = scenes.soup(n_tris); res["gen_tris_s"] = round(time.time() - t, 1)
    t = time.time(); rays_np = scenes.random_rays(n_rays); res["gen_rays_s"] = round(time.time() - t, 1)
    rays = torch.from_numpy(rays_np.view(np.float32).reshape(-1)).to(dev)
    hits = f32(n_rays * 4)
    keep = {}
    for name, mode in (("bvh2", rt.TRAVERSAL_BVH2), ("bvh4q", rt.TRAVERSAL_BVH4Q)):
        with rt.Renderer((0,)) as r:
            r.set_option(rt.OPT_TRAVERSAL, mode)
            t = time.time(); st = r.create_data(tris); up = time.time() - t
            ms = timeit(lambda: r.trace_dev(rays, n_rays, hits), 3)
            keep[name] = hits.cpu().numpy().view(rt.HIT).copy()
            work = counters_of(r, lambda: r.trace_dev(rays, n_rays, hits))
            res[name] = {"ms": round(ms, 3), "Mrays_s": round(n_rays / ms / 1e3, 1), "work": work,
                         "alg_GBps": round((64 * work["N"] + 48 * work["T"] + 48) * n_rays / ms / 1e6, 1),
                         "hit_fraction": round(float((keep[name]["prim"] != rt.MISS).mean()), 4),
                         "bvh": {"nodes": st["n_nodes"], "depth": st["max_depth"], "build_ms": round(st["build_ms"], 1),
                                 "upload_call_s": round(up, 2), "device_MB": round(st["device_bytes"] / 1e6, 1)}}
            if name == "bvh2":
                # mismatch rate against the reference loop on the GPU (brute force over all triangles)
                m = min(n_rays, 1 << 17)
                r.set_option(rt.OPT_TRAVERSAL, rt.TRAVERSAL_BRUTE)
                bh = f32(m * 4)
                t = time.time(); r.trace_dev(rays, m, bh); torch.cuda.synchronize(); bs = time.time() - t
                b = bh.cpu().numpy().view(rt.HIT)
                res["gpu_brute_check"] = {"rays": m, "seconds": round(bs, 2),
                                          "prim_id_mismatches_bvh2": int((b["prim"] != keep["bvh2"][:m]["prim"]).sum()),
                                          "bit_exact_bvh2": bool(b.tobytes() == keep["bvh2"][:m].tobytes())}
                keep["brute"] = b
    # default path (BVH4q + ray binning) and the PCIe-inclusive rate of the host-pointer entry point
    with rt.Renderer((0,)) as r:
        r.create_data(tris)
        for name, b in (("bvh4q_binned", 1), ("bvh4q_unbinned", 0)):
            r.set_option(rt.OPT_RAY_BINNING, b)
            ms = timeit(lambda: r.trace_dev(rays, n_rays, hits), 3)
            res[name] = {"ms": round(ms, 3), "Mrays_s": round(n_rays / ms / 1e3, 1),
                         "same_as_bvh2": bool(hits.cpu().numpy().view(rt.HIT).tobytes() == keep["bvh2"].tobytes())}
        r.set_option(rt.OPT_RAY_BINNING, 2)
        mh = min(n_rays, 1 << 25)
        hp = {"rays": mh, "note": "rt_trace (second call): host rays in + kernels + host hits out, 4 Mi-ray pieces on this structure, copies and kernels on their own streams; pinned = rt_host_alloc "
                                  "blocks (DMA direct), pageable = staged through pinned buffers by host threads"}
        p_rays, p_hits = r.host_alloc(mh, rt.RAY), r.host_alloc(mh, rt.HIT)
        p_rays[:] = rays_np[:mh]
        for name, src, dst in (("pinned", p_rays, p_hits), ("pageable", rays_np[:mh], np.empty(mh, dtype=rt.HIT))):
            r.trace(src, out=dst)                                        # warm: staging buffers, ring, binning scratch of a full call
            t = time.time(); hh = r.trace(src, out=dst); dt = time.time() - t
            hp[name + "_seconds"] = round(dt, 4); hp[name + "_Mrays_s"] = round(mh / dt / 1e6, 1)
            hp[name + "_same"] = bool(hh.tobytes() == keep["bvh2"][:mh].tobytes())
        res["host_ptr"] = hp
    m = len(keep["brute"])
    res["gpu_brute_check"]["prim_id_mismatches_bvh4q"] = int((keep["brute"]["prim"] != keep["bvh4q"][:m]["prim"]).sum())
    res["layouts_agree"] = bool(keep["bvh2"].tobytes() == keep["bvh4q"].tobytes())
    res["parity_vs_oracle"] = sample_parity(tris, None, rays_np, keep["bvh2"], 4096)
    out(res)


def c5(n_tris):
    w, h, spp, bounces = 3840, 2160, 16, 3
    res = {"config": f"C5 stand-in: C4 soup {n_tris} tris with a pinhole camera, 3840x2160 x 16 spp x 3 bounces, one GPU"}
    tris = scenes.soup(n_tris)
    e = tris["material"]["emission"]; e[::50] = (1.0, 0.9, 0.8); tris["material"]["emissionStrength"][::50] = 2.0   # 2 % emitters
    cam = rt.camera_look_at((0.5, 2.2, 0.5), (0.5, 0.5, 0.5), 45.0, w / h)
    n = w * h * spp
    K = 1 << 15                                                          # rays per generation in the parity sample
    idx = torch.from_numpy(np.sort(np.random.default_rng(5).choice(n, K, replace=False))).to(dev)
    with rt.Renderer((0,)) as r:
        r.set_option(rt.OPT_TRAVERSAL, rt.TRAVERSAL_BVH)
        r.create_data(tris); r.update_camera(cam)
        ra, rb, hits, thr, rad = f32(n * 8), f32(n * 8), f32(n * 4), f32(n * 4), f32(n * 4)
        fb = torch.empty(w * h, dtype=torch.int32, device=dev)
        live, sample = [], []

        def frame(count=False, capture=False):
            a, b = ra, rb
            if capture:                                                  # keep the primary rays / hits too (optional outputs)
                pr, ph = f32(n * 8), f32(n * 4)
                r.primary_shade_stripes_dev(w, h, 0, h, h, 1, thr, rad, a, spp=spp, seed=1, jitter=True, shade_seed=2, rays=pr, hits=ph)
                sample.append((pr.view(-1, 8)[idx].clone(), ph.view(-1, 4)[idx].clone()))
                del pr, ph
            else:
                r.primary_shade_stripes_dev(w, h, 0, h, h, 1, thr, rad, a, spp=spp, seed=1, jitter=True, shade_seed=2)
            if count:
                live.append(n)
            for k in range(1, bounces):
                if count:
                    live.append(int((a.view(-1, 8)[:, 7] > 0).sum()))
                r.trace_dev(a, n, hits)
                if capture:
                    sample.append((a.view(-1, 8)[idx].clone(), hits.view(-1, 4)[idx].clone()))
                if k + 1 == bounces:
                    r.shade_resolve_dev(a, hits, n, spp, thr, rad, None, fb)
                else:
                    r.shade_dev(a, hits, n, thr, rad, b, key0=0, seed=2 + k)
                    a, b = b, a
        frame(True, True); torch.cuda.synchronize()
        ms = timeit(frame, 2)
        traced = sum(live)
        res.update({"frame_ms": round(ms, 2), "closest_hit_rays": traced, "Mrays_s": round(traced / ms / 1e3, 1),
                    "live_rays_per_segment": live, "nonblack_pixels": int((fb.cpu().numpy() & 0xFFFFFF != 0).sum())})
        # parity gate: a seeded sample of every ray generation (primary + each bounce) against the reference loop run as
        # the brute-force HIP kernel over all triangles, and a sub-sample against the CPU oracle
        r.set_option(rt.OPT_TRAVERSAL, rt.TRAVERSAL_BRUTE)
        gate = {"sample_rays_per_generation": K, "generations": len(sample), "prim_id_mismatches_vs_gpu_brute": 0, "bit_exact_vs_gpu_brute": True,
                "oracle_sample_per_generation": 512, "prim_id_mismatches_vs_oracle": 0}
        sc = oracle.Scene(tris)
        for s_rays, s_hits in sample:
            bh = f32(K * 4)
            r.trace_dev(s_rays.contiguous().view(-1), K, bh); torch.cuda.synchronize()
            got, ref = s_hits.cpu().numpy().view(rt.HIT).ravel(), bh.cpu().numpy().view(rt.HIT).ravel()
            gate["prim_id_mismatches_vs_gpu_brute"] += int((got["prim"] != ref["prim"]).sum())
            gate["bit_exact_vs_gpu_brute"] = bool(gate["bit_exact_vs_gpu_brute"] and got.tobytes() == ref.tobytes())
            o = oracle.trace(sc, s_rays.cpu().numpy().view(rt.RAY).ravel()[:512])
            gate["prim_id_mismatches_vs_oracle"] += int((o["prim"] != got["prim"][:512]).sum())
        res["parity"] = gate
    out(res)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("which", nargs="*", default=["c1", "c2", "c4", "c5"])
    ap.add_argument("--soup", type=int, default=10_000_000)
    ap.add_argument("--rays", type=int, default=64 * (1 << 20))
    a = ap.parse_args()
    for w in a.which:
        {"c0": c0, "c1": c1, "c2": c2, "c3": lambda: c3("sponza"), "c3b": lambda: c3("colonnade"), "c4": lambda: c4(a.soup, a.rays), "c5": lambda: c5(a.soup)}[w]()


if __name__ == "__main__":
    main()
