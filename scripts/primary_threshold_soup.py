"""Tile kernel vs ray-buffer kernel for PRIMARY rays on random soups of growing size (SAH cost as the predictor)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracer_amd as rt
from raytracer_amd import scenes
W, H, SPP = 3840, 2160, 4; n = W * H * SPP
f = lambda k: torch.empty(n * k, dtype=torch.float32, device='cuda')
thr, rad, nxt = f(4), f(4), f(8)
cam = rt.camera_look_at((0.5, 2.2, 0.5), (0.5, 0.5, 0.5), 45.0, W / H)
def timeit(fn, reps=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for nt, half in ((30000, 0.03), (100000, 0.02), (300000, 0.015), (1000000, 0.01), (3000000, 0.007), (10000000, 0.005)):
    tris = scenes.soup(nt, half_extent=half)
    with rt.Renderer((0,)) as r:
        r.set_option(rt.OPT_PRIMARY, 3)            # (at upload: builds the wide node copy the packet kernel reads)
        st = r.create_data(tris); r.update_camera(cam)
        out = []
        for path in (1, 2, 3):
            r.set_option(rt.OPT_PRIMARY, path)
            out.append(timeit(lambda: r.primary_shade_stripes_dev(W, H, 0, H, H, 1, thr, rad, nxt, spp=SPP, seed=1, jitter=True, shade_seed=2)))
        print(f"soup {nt} half {half}: sah {st['sah_cost']:.0f}: tile (ray per lane) {out[0]:.2f} ms, ray-buffer {out[1]:.2f} ms, tile (packet) {out[2]:.2f} ms", flush=True)
