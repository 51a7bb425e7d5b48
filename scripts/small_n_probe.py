"""Strip kernel on shrinking ray counts (what a rank of N gets): time vs the ideal n/N scaling, for several chunk sizes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracer_amd as rt
from raytracer_amd import scenes
tris, sph, _, cam = scenes.sponza_class()
r = rt.Renderer((0,)); r.create_data(tris, sph); r.update_camera(cam)
W, H, SPP = 3840, 2160, 4; n = W * H * SPP
f = lambda k: torch.empty(n * k, dtype=torch.float32, device='cuda')
thr, rad, brays, bhits = f(4), f(4), f(8), f(4)
r.primary_shade_stripes_dev(W, H, 0, H, H, 1, thr, rad, brays, spp=SPP, seed=1, jitter=True, shade_seed=2)
def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
base = None
for div in (1, 2, 4, 8, 16):
    m = n // div
    row = []
    for strip in (64, 128, 256, 512):
        r.set_option(rt.OPT_STRIP, strip)
        row.append(timeit(lambda: r.trace_dev(brays, m, bhits)))
    if base is None: base = row[2]
    print(f"n/{div}: " + "  ".join(f"strip {s}: {t:.3f} ms" for s, t in zip((64, 128, 256, 512), row)) + f"   ideal {base / div:.3f}", flush=True)
