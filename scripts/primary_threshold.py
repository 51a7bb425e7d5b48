"""Where does the ray-buffer kernel overtake the tile kernels (one ray per lane / packet) for PRIMARY rays?  Same scene at growing tessellation."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracer_amd as rt
from raytracer_amd import scenes
W, H, SPP = 3840, 2160, 4; n = W * H * SPP
f = lambda k: torch.empty(n * k, dtype=torch.float32, device='cuda')
thr, rad, nxt = f(4), f(4), f(8)
def timeit(fn, reps=4):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for tess in (45, 64, 90, 128, 180, 256):
    tris, sph, _, cam = scenes.sponza_class(tess)
    with rt.Renderer((0,)) as r:
        r.set_option(rt.OPT_PRIMARY, 3)            # (at upload: builds the wide node copy the packet kernel reads)
        st = r.create_data(tris, sph); r.update_camera(cam)
        mb = (st["n_nodes"] * 64 + st["n_tris"] * 48) / 1e6
        out = []
        for path in (1, 2, 3):
            r.set_option(rt.OPT_PRIMARY, path)
            out.append(timeit(lambda: r.primary_shade_stripes_dev(W, H, 0, H, H, 1, thr, rad, nxt, spp=SPP, seed=1, jitter=True, shade_seed=2)))
        print(f"tess {tess}: {len(tris)} tris, nodes+tris {mb:.0f} MB: tile (ray per lane) {out[0]:.2f} ms, ray-buffer {out[1]:.2f} ms, tile (packet) {out[2]:.2f} ms  SAH {st['sah_cost']:.0f}", flush=True)
