"""rank_share_probe.py with F frames in flight: frame k runs on stream k % F with its own buffers, so the tail of one
frame's launches is filled by the next frame's.  Prints ms per frame (throughput) for rank 0 of N, F = 1, 2, 3, 4, 6."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracer_amd as rt
from raytracer_amd import dist as rd, scenes
W, H, SPP, STRIPE = 3840, 2160, 4, 8
tris, sph, lights, cam = scenes.sponza_class()
r = rt.Renderer((0,)); r.create_data(tris, sph, lights); r.update_camera(cam)
for world in (1, 2, 4, 8):
    plan = rd.stripe_plan(H, world, 0, STRIPE)
    n = plan.padded_rows * W * SPP
    for F in (1, 2, 3, 4, 6):
        f = lambda k: torch.empty(n * k, dtype=torch.float32, device="cuda")
        sets = [dict(thr=f(4), rad=f(4), brays=f(8), bhits=f(4), fb=torch.empty(plan.padded_rows * W, dtype=torch.int32, device="cuda"),
                     st=torch.cuda.Stream()) for _ in range(F)]
        def step(k):
            s = sets[k % F]
            r.primary_shade_stripes_dev(W, H, plan.y0, plan.padded_rows, STRIPE, world, s["thr"], s["rad"], s["brays"], spp=SPP, seed=1, jitter=True, shade_seed=2, stream=s["st"])
            r.trace_dev(s["brays"], n, s["bhits"], stream=s["st"])
            r.shade_resolve_dev(s["brays"], s["bhits"], n, SPP, s["thr"], s["rad"], None, s["fb"], stream=s["st"])
        for k in range(6): step(k)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        K = 60
        for k in range(K): step(k)
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / K * 1e3
        print(f"rank 0 of {world}, {F} frame(s) in flight: {ms:.3f} ms/frame", flush=True)
        del sets
