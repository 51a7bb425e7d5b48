"""What one rank of N does per step on its own (no gather): emulates rank 0 of N on one GPU to expose fixed overheads
(launch gaps, kernel tails) that limit strong scaling of the 7.7 ms C3 frame."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracer_amd as rt
from raytracer_amd import dist as rd, scenes
W, H, SPP, STRIPE = 3840, 2160, 4, 8
tris, sph, lights, cam = scenes.sponza_class()
r = rt.Renderer((0,)); r.create_data(tris, sph, lights); r.update_camera(cam)
stream = torch.cuda.current_stream()
for world in (1, 2, 4, 8):
    plan = rd.stripe_plan(H, world, 0, STRIPE)
    n = plan.padded_rows * W * SPP
    f = lambda k: torch.empty(n * k, dtype=torch.float32, device="cuda")
    thr, rad, brays, bhits = f(4), f(4), f(8), f(4)
    fb = torch.empty(plan.padded_rows * W, dtype=torch.int32, device="cuda")
    def step():
        r.primary_shade_stripes_dev(W, H, plan.y0, plan.padded_rows, STRIPE, world, thr, rad, brays, spp=SPP, seed=1, jitter=True, shade_seed=2, stream=stream)
        r.trace_dev(brays, n, bhits, stream=stream)
        r.shade_resolve_dev(brays, bhits, n, SPP, thr, rad, None, fb, stream=stream)
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    K = 50
    for _ in range(K): step()
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / K * 1e3
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    ev[0].record(stream)
    r.primary_shade_stripes_dev(W, H, plan.y0, plan.padded_rows, STRIPE, world, thr, rad, brays, spp=SPP, seed=1, jitter=True, shade_seed=2, stream=stream); ev[1].record(stream)
    r.trace_dev(brays, n, bhits, stream=stream); ev[2].record(stream)
    r.shade_resolve_dev(brays, bhits, n, SPP, thr, rad, None, fb, stream=stream); ev[3].record(stream)
    torch.cuda.synchronize()
    st = [ev[i].elapsed_time(ev[i + 1]) for i in range(3)]
    print(f"   stages: primary+seg0 {st[0]:.3f}  bounce {st[1]:.3f}  last+resolve {st[2]:.3f} ms", flush=True)
    print(f"rank 0 of {world}: {ms:.3f} ms/step  (ideal: the N = 1 step / N)", flush=True)
