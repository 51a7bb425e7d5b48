"""C5: how fast does the ray-buffer (strip) kernel trace the PRIMARY rays of the 10 M-triangle soup, against k_primary?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracer_amd as rt
from raytracer_amd import scenes
w, h, spp = 3840, 2160, 16
tris = scenes.soup(10_000_000)
cam = rt.camera_look_at((0.5, 2.2, 0.5), (0.5, 0.5, 0.5), 45.0, w / h)
n = w * h * spp
f32 = lambda k: torch.empty(n * k, dtype=torch.float32, device="cuda")
def timeit(fn, reps=2):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
with rt.Renderer((0,)) as r:
    r.create_data(tris); r.update_camera(cam)
    rays, hits, h2 = f32(8), f32(4), f32(4)
    print("k_primary (rays + hits out): %.2f ms" % timeit(lambda: r.trace_primary_dev(w, h, hits, rays, spp=spp, seed=1, jitter=True)), flush=True)
    for binning in (2, 0):
        r.set_option(rt.OPT_RAY_BINNING, binning)
        print("strip kernel on the same rays (binning %d): %.2f ms  same=%s" % (binning, timeit(lambda: r.trace_dev(rays, n, h2)),
              torch.equal(hits.view(torch.int32), h2.view(torch.int32))), flush=True)
