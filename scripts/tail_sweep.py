"""Second pass over the tail of a ray-buffer launch (RT_OPT_TAIL_LANES) -- an experiment that was NOT adopted; apply
profiles/r04f_tail_second_pass.patch to run it.  Bounce rays of the bench frame, whole and as the share of one rank of 8, and
the 10 M-triangle soup at 8 / 64 Mi rays, for several thresholds.  Hit buffers must not change (they do not); every threshold
is slower than one launch (profiles/r04f_tail_sweep.txt): the handed-over rays walk again from the root in a launch too small
to hide its memory latency."""
import sys, os, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import raytracer_amd as rt
from raytracer_amd import dist as rd, scenes
W, H, SPP, STRIPE = 3840, 2160, 4, 8
dev = torch.device("cuda", 0)
def timeit(fn, reps):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
sha = lambda t: hashlib.sha256(t.cpu().numpy().tobytes()).hexdigest()[:12]
lanes = [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else "0,8,16,24,32,40".split(","))]
tris, sph, lights, cam = scenes.sponza_class()
r = rt.Renderer((0,)); r.create_data(tris, sph, lights); r.update_camera(cam)
for world in (1, 8):
    plan = rd.stripe_plan(H, world, 0, STRIPE)
    n = plan.padded_rows * W * SPP
    f = lambda k: torch.empty(n * k, dtype=torch.float32, device=dev)
    thr, rad, brays, bhits = f(4), f(4), f(8), f(4)
    r.primary_shade_stripes_dev(W, H, plan.y0, plan.padded_rows, STRIPE, world, thr, rad, brays, spp=SPP, seed=1, jitter=True, shade_seed=2)
    for L in lanes:
        r.set_option(rt.OPT_TAIL_LANES, L)
        ms = timeit(lambda: r.trace_dev(brays, n, bhits), 30)
        print(f"bench bounce rays, rank 0 of {world} ({n} rays), tail lanes {L:2d}: {ms:.4f} ms  hits {sha(bhits)}", flush=True)
r.close()
tris = scenes.soup(10_000_000)
r = rt.Renderer((0,)); r.create_data(tris)
for n in (8 << 20, 64 << 20):
    rays = torch.from_numpy(scenes.random_rays(n).view(np.float32).reshape(-1)).to(dev)
    hits = torch.empty(n * 4, dtype=torch.float32, device=dev)
    for L in lanes:
        r.set_option(rt.OPT_TAIL_LANES, L)
        ms = timeit(lambda: r.trace_dev(rays, n, hits), 5)
        print(f"soup 10 M, {n >> 20} Mi rays, tail lanes {L:2d}: {ms:.3f} ms  {n / ms / 1e3:.1f} Mrays/s  hits {sha(hits)}", flush=True)
