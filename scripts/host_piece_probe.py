"""rt_trace on host pointers against the 10 M-triangle soup (kernel-bound pieces) and the bench scene: Mrays/s, pinned and pageable."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import raytracer_amd as rt
from raytracer_amd import scenes

which = sys.argv[1] if len(sys.argv) > 1 else "soup"
r = rt.Renderer((0,)); r.set_option(rt.OPT_TRAVERSAL, 3)
if which == "soup":
    r.create_data(scenes.soup())
    m = 1 << 25
    rays = scenes.random_rays(m)
else:
    tris, sph, lights, cam = scenes.sponza_class(); r.create_data(tris, sph, lights)
    m = 1 << 24
    rays = scenes.random_rays(m); rays["o"] = rays["o"] * 20 - 10
p_rays, p_hits = r.host_alloc(m, rt.RAY), r.host_alloc(m, rt.HIT)
p_rays[:] = rays
out = np.empty(m, dtype=rt.HIT)
for name, src, dst in (("pinned", p_rays, p_hits), ("pageable", rays, out)):
    ts = []
    for _ in range(3):
        t = time.perf_counter(); r.trace(src, out=dst); ts.append(time.perf_counter() - t)
    print(f"{which:6s} {name:9s}", " ".join(f"{m / t / 1e6:7.0f}" for t in ts), "Mrays/s", flush=True)
assert p_hits.tobytes() == out.tobytes()
