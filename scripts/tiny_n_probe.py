"""Fixed cost of a launch: ray-buffer kernel and tile kernel on tiny workloads (C3 scene)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracer_amd as rt
from raytracer_amd import scenes
tris, sph, _, cam = scenes.sponza_class()
r = rt.Renderer((0,)); r.create_data(tris, sph); r.update_camera(cam)
W, H, SPP = 3840, 2160, 1; n = W * H
f = lambda k: torch.empty(n * k, dtype=torch.float32, device='cuda')
rays, hits, brays, bhits = f(8), f(4), f(8), f(4)
r.trace_primary_dev(W, H, hits, rays); r.bounce_dev(rays, hits, n, brays, seed=1)
def timeit(fn, reps=50):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for m in (64, 4096, 65536, 1 << 20):
    t_strip = timeit(lambda: r.trace_dev(brays, m, bhits))
    r.set_option(rt.OPT_STRIP, 0); t_plain = timeit(lambda: r.trace_dev(brays, m, bhits)); r.set_option(rt.OPT_STRIP, 256)
    print(f"{m} bounce rays: ray-buffer kernel {t_strip:.1f} us, one-ray-per-lane kernel {t_plain:.1f} us", flush=True)
for w, h in ((64, 64), (512, 512), (1920, 1080)):
    print(f"primary {w}x{h}: {timeit(lambda: r.trace_primary_dev(w, h, hits, None)):.1f} us", flush=True)
