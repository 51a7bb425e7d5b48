import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import raytracer_amd as rt
from raytracer_amd import scenes
tris, sph, _, cam = scenes.sponza_class()
r = rt.Renderer((0,)); r.set_option(rt.OPT_TRAVERSAL, int(sys.argv[1]) if len(sys.argv) > 1 else rt.TRAVERSAL_BVH); r.create_data(tris, sph); r.update_camera(cam)
W,H,SPP=3840,2160,4; n=W*H*SPP
f=lambda k: torch.empty(n*k, dtype=torch.float32, device='cuda')
rays,hits,brays,bhits,ref=f(8),f(4),f(8),f(4),f(4)
r.trace_primary_dev(W,H,hits,rays,spp=SPP,seed=1,jitter=True); r.bounce_dev(rays,hits,n,brays,seed=1)
r.set_option(rt.OPT_STRIP, 0); r.trace_dev(brays,n,ref); torch.cuda.synchronize()
def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/reps
print("plain k_trace: %.3f ms" % timeit(lambda: r.trace_dev(brays,n,bhits)))
for strip in (256, 512, 1024):
  for tth in (8, 16, 24, 32, 40, 48):
    for refill in (16, 32):
        r.set_option(rt.OPT_STRIP, strip); r.set_option(rt.OPT_REFILL, refill | (tth << 8))
        ms = timeit(lambda: r.trace_dev(brays,n,bhits))
        ok = torch.equal(bhits.view(torch.int32), ref.view(torch.int32))
        print(f"strip {strip:6d} tth {tth:2d} refill {refill:2d}: {ms:.3f} ms  {n/ms/1e3:.0f} Mrays/s  exact={ok}", flush=True)
