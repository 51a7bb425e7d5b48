"""Probe (variant build -DRT_PROBE_DEADPOP): share of BVH4q node steps in which all four children miss, on the bench
frame's primary and bounce rays.  Such a step is mostly a stack entry that became useless after a closer hit was found."""
import os, sys
sys.path.insert(0, os.environ.get("RT_REPO") or os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracer_amd as rt
from raytracer_amd import scenes
W, H, SPP = 1920, 1080, 4
which = sys.argv[1] if len(sys.argv) > 1 else "sponza"
tris, sph, lights, cam = scenes.sponza_class() if which == "sponza" else scenes.colonnade()
r = rt.Renderer((0,)); r.create_data(tris, None, lights); r.update_camera(cam)      # no spheres: the probe reuses their counter
n = W * H * SPP
f32 = lambda k: torch.empty(n * k, dtype=torch.float32, device="cuda")
thr, rad, brays, bhits, prays, phits = f32(4), f32(4), f32(8), f32(4), f32(8), f32(4)
r.primary_shade_stripes_dev(W, H, 0, H, H, 1, thr, rad, brays, spp=SPP, seed=1, jitter=True, shade_seed=2)
r.trace_primary_dev(W, H, phits, prays, spp=SPP, seed=1, jitter=True)
for name, rays in (("primary", prays), ("bounce", brays)):
    r.set_option(rt.OPT_COUNTERS, 1); r.counters()
    r.trace_dev(rays, n, bhits); torch.cuda.synchronize()
    c = r.counters(); r.set_option(rt.OPT_COUNTERS, 0)
    print(f"{which} {name}: rays {c['rays']} N {c['nodes']/c['rays']:.2f} all-miss steps {c['sphere_tests']/c['rays']:.2f} per ray ({100*c['sphere_tests']/max(c['nodes'],1):.1f} % of node steps)  T {c['tri_tests']/c['rays']:.2f}", flush=True)
