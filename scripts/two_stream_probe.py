"""Does splitting the bench frame into S interleaved shares on S streams fill the kernels' tails?  (one GPU)
Each share runs the three stages of bench.py on its own stream; shares use the stripe plan of an S-rank job."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import raytracer_amd as rt
from raytracer_amd import dist as rd, scenes

W, H, SPP, STRIPE, SEED = 3840, 2160, 4, 8, 1
tris, sph, lights, cam = scenes.sponza_class()
r = rt.Renderer((0,))
r.set_option(rt.OPT_TRAVERSAL, 3)
r.create_data(tris, sph, lights); r.update_camera(cam)
dev = torch.device("cuda", 0)
f32 = lambda n: torch.empty(n, dtype=torch.float32, device=dev)

def frame_fn(S):
    shares = []
    for s in range(S):
        plan = rd.stripe_plan(H, S, s, STRIPE)
        n = plan.padded_rows * W * SPP
        shares.append(dict(plan=plan, n=n, brays=f32(n * 8), bhits=f32(n * 4), thr=f32(n * 4), rad=f32(n * 4),
                           fb=torch.empty(plan.padded_rows * W, dtype=torch.int32, device=dev),
                           st=torch.cuda.Stream() if S > 1 else torch.cuda.current_stream()))
    def run():
        main = torch.cuda.current_stream()
        for sh in shares:
            if S > 1: sh["st"].wait_stream(main)
        for sh in shares:            # stage by stage, so that launches of the shares alternate in submission order
            p = sh["plan"]
            r.primary_shade_stripes_dev(W, H, p.y0, p.padded_rows, STRIPE, S, sh["thr"], sh["rad"], sh["brays"], spp=SPP, seed=SEED,
                                        jitter=True, shade_seed=SEED + 1, stream=sh["st"])
        for sh in shares:
            r.trace_dev(sh["brays"], sh["n"], sh["bhits"], stream=sh["st"])
        for sh in shares:
            r.shade_resolve_dev(sh["brays"], sh["bhits"], sh["n"], SPP, sh["thr"], sh["rad"], None, sh["fb"], stream=sh["st"])
        for sh in shares:
            if S > 1: main.wait_stream(sh["st"])
    return run, shares

for S in (1, 2, 3, 4, 1):
    run, shares = frame_fn(S)
    for _ in range(3): run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 20
    for _ in range(K): run()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / K * 1e3
    print(f"S={S}: {ms:.3f} ms/frame  ({2 * W * H * SPP / ms / 1e3:.0f} Mrays/s)")
