"""rt_trace on host pointers: pinned (rt_host_alloc) against pageable memory in both directions, several repetitions each (one GPU)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import raytracer_amd as rt
from raytracer_amd import scenes

tris, sph, lights, cam = scenes.sponza_class()
r = rt.Renderer((0,)); r.set_option(rt.OPT_TRAVERSAL, 3); r.create_data(tris, sph, lights); r.update_camera(cam)
m = 1 << 24
rays = scenes.random_rays(m); rays["o"] = rays["o"] * 20 - 10
p_rays, p_hits = r.host_alloc(m, rt.RAY), r.host_alloc(m, rt.HIT)
p_rays[:] = rays
out = np.empty(m, dtype=rt.HIT)
for name, src, dst in (("pinned", p_rays, p_hits), ("pageable", rays, out), ("pinned in, pageable out", p_rays, out),
                       ("pageable in, pinned out", rays, p_hits)):
    ts = []
    for _ in range(4):
        t = time.perf_counter(); r.trace(src, out=dst); ts.append(time.perf_counter() - t)
    print(f"{name:26s}", " ".join(f"{m / t / 1e6:7.0f}" for t in ts), "Mrays/s", flush=True)
assert p_hits.tobytes() == out.tobytes()
