"""Fused primary + first shading segment on the bench scene WITH point lights (shadow rays inside the kernel):
one ray per lane (RT_OPT_PRIMARY 1: shadow rays walk per lane with stacks in LDS) against the packet kernel (3: the
wave's shadow rays walk together).  4K x 4 spp.  Outputs must be bit-identical."""
import os, sys
sys.path.insert(0, os.environ.get("RT_REPO") or os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import raytracer_amd as rt
from raytracer_amd import scenes
W, H, SPP = 3840, 2160, 4
tris, sph, _, cam = scenes.sponza_class() if (len(sys.argv) < 2 or sys.argv[1] == "sponza") else scenes.colonnade()
lights = np.zeros(2, dtype=rt.LIGHT)
lights["position"] = [(0.0, 0.0, 4.5), (-6.0, 5.0, 3.0)]
lights["color"] = [(1.0, 0.9, 0.8), (0.6, 0.7, 1.0)]
lights["intensity"] = [30.0, 20.0]
n = W * H * SPP
f32 = lambda k: torch.empty(n * k, dtype=torch.float32, device="cuda")
outs = {}
for path in (1, 3):
    r = rt.Renderer((0,)); r.set_option(rt.OPT_PRIMARY, path); r.create_data(tris, sph, lights); r.update_camera(cam)
    thr, rad, nxt = f32(4), f32(4), f32(8)
    go = lambda: r.primary_shade_stripes_dev(W, H, 0, H, H, 1, thr, rad, nxt, spp=SPP, seed=1, jitter=True, shade_seed=2)
    for _ in range(3): go()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): go()
    e1.record(); torch.cuda.synchronize()
    outs[path] = (thr.clone(), rad.clone(), nxt.clone())
    print(f"RT_OPT_PRIMARY {path}: primary + segment 0 with {len(lights)} lights: {e0.elapsed_time(e1)/10:.3f} ms  lit radiance entries {int((rad.view(-1,4)[:, :3].sum(1) > 0).sum())}", flush=True)
    r.close()
same = all(torch.equal(a.view(torch.int32), b.view(torch.int32)) for a, b in zip(outs[1], outs[3]))
print("throughput / radiance / next rays bit-identical between the two kernels:", same)
sys.exit(0 if same else 1)
