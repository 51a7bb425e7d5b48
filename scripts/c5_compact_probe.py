"""C5: how much does tracing only the LIVE bounce rays (order-preserving compaction) save over tracing the buffer with
its dead rays (paths that ended) in place?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracer_amd as rt
from raytracer_amd import scenes
w, h, spp = 3840, 2160, 16
tris = scenes.soup(10_000_000)
tris["material"]["emission"][::50] = (1.0, 0.9, 0.8); tris["material"]["emissionStrength"][::50] = 2.0
cam = rt.camera_look_at((0.5, 2.2, 0.5), (0.5, 0.5, 0.5), 45.0, w / h)
n = w * h * spp
f32 = lambda k: torch.empty(n * k, dtype=torch.float32, device="cuda")
def timeit(fn, reps=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
with rt.Renderer((0,)) as r:
    r.create_data(tris); r.update_camera(cam)
    ra, hits, thr, rad = f32(8), f32(4), f32(4), f32(4)
    r.primary_shade_stripes_dev(w, h, 0, h, h, 1, thr, rad, ra, spp=spp, seed=1, jitter=True, shade_seed=2)
    torch.cuda.synchronize()
    live = ra.view(-1, 8)[:, 7] > 0
    nl = int(live.sum())
    compact = ra.view(-1, 8)[live].contiguous().view(-1)
    h2 = torch.empty(nl * 4, dtype=torch.float32, device="cuda")
    print(f"bounce rays: {n} slots, {nl} live", flush=True)
    print("trace with dead rays in place: %.2f ms" % timeit(lambda: r.trace_dev(ra, n, hits)), flush=True)
    print("trace live rays only:          %.2f ms" % timeit(lambda: r.trace_dev(compact, nl, h2)), flush=True)
