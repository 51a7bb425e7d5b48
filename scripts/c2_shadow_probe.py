"""C2's shadow pass: k_shadow (one ray per lane, fused generation) against shadow rays written once + the ray-buffer any-hit kernel."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracer_amd as rt
from raytracer_amd import scenes
tris, sph, lights, cam = scenes.bunny_class()
w, h = 1920, 1080
n = w * h
dev = torch.device("cuda", 0)
r = rt.Renderer((0,)); r.create_data(tris, sph, lights); r.update_camera(cam)
f = lambda k: torch.empty(n * k, dtype=torch.float32, device=dev)
rays, hits, srays = f(8), f(4), f(8)
lit, occ = torch.empty(n, dtype=torch.uint8, device=dev), torch.empty(n, dtype=torch.uint8, device=dev)
lp = tuple(float(x) for x in lights[0]["position"])
r.trace_primary_dev(w, h, hits, rays)
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
print("k_shadow (fused):            %.3f ms" % timeit(lambda: r.shadow_dev(rays, hits, n, lp, lit)))
r.shadow_dev(rays, hits, n, lp, lit, shadow_rays=srays); torch.cuda.synchronize()
print("k_shadow writing the rays:   %.3f ms" % timeit(lambda: r.shadow_dev(rays, hits, n, lp, lit, shadow_rays=srays)))
for opts in ((), ((rt.OPT_RAY_QUEUE, 0),), ((rt.OPT_STRIP, 0),)):
    for k, v in opts: r.set_option(k, v)
    print("any-hit ray-buffer kernel on the written shadow rays %s: %.3f ms" % (opts, timeit(lambda: r.occluded_dev(srays, n, occ))))
    for k, v in opts: r.set_option(k, {rt.OPT_RAY_QUEUE: 2, rt.OPT_STRIP: 256}[k])
live = (lit != 2)
print("flags agree where a shadow ray exists:", bool(((occ == 0) == (lit == 1))[live].all().item()), " shadow rays", int(live.sum()))
