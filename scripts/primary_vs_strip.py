"""Is the persistent strip kernel faster than the tile-per-wave primary kernel on coherent (primary) rays?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracer_amd as rt
from raytracer_amd import scenes
tris, sph, _, cam = scenes.sponza_class()
r = rt.Renderer((0,)); r.set_option(rt.OPT_TRAVERSAL, rt.TRAVERSAL_BVH); r.create_data(tris, sph); r.update_camera(cam)
W, H, SPP = 3840, 2160, 4; n = W * H * SPP
f = lambda k: torch.empty(n * k, dtype=torch.float32, device='cuda')
rays, hits, h2 = f(8), f(4), f(4)
def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
print("k_primary (gen + trace + write rays): %.3f ms" % timeit(lambda: r.trace_primary_dev(W, H, hits, rays, spp=SPP, seed=1, jitter=True)))
print("k_primary (no ray output):           %.3f ms" % timeit(lambda: r.trace_primary_dev(W, H, hits, None, spp=SPP, seed=1, jitter=True)))
for strip, tth, refill in ((256, 24, 16), (256, 32, 16), (256, 16, 16), (256, 24, 32), (64, 24, 16)):
    r.set_option(rt.OPT_STRIP, strip); r.set_option(rt.OPT_REFILL, refill | (tth << 8))
    ms = timeit(lambda: r.trace_dev(rays, n, h2))
    print(f"strip kernel on the primary rays, strip {strip} tth {tth} refill {refill}: {ms:.3f} ms  same={torch.equal(hits.view(torch.int32), h2.view(torch.int32))}")
