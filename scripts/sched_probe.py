"""Scheduling probes of the strip kernel on the C3 bounce rays: lanes active per interior / triangle block."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracer_amd as rt
from raytracer_amd import scenes
tris, sph, _, cam = scenes.sponza_class()
r = rt.Renderer((0,)); r.set_option(rt.OPT_TRAVERSAL, rt.TRAVERSAL_BVH); r.create_data(tris, sph); r.update_camera(cam)
W, H, SPP = 3840, 2160, 4; n = W * H * SPP
f = lambda k: torch.empty(n * k, dtype=torch.float32, device='cuda')
rays, hits, brays, bhits = f(8), f(4), f(8), f(4)
r.trace_primary_dev(W, H, hits, rays, spp=SPP, seed=1, jitter=True); r.bounce_dev(rays, hits, n, brays, seed=1)
for strip, tth, refill, binning, cap in ((256, 24, 16, 0, 20), (256, 16, 16, 0, 20), (256, 32, 16, 0, 20), (256, 24, 8, 0, 20), (256, 24, 24, 0, 20), (128, 24, 16, 0, 20), (512, 24, 16, 0, 20), (256, 24, 16, 0, 16), (256, 24, 16, 0, 24)):
    r.set_option(rt.OPT_STRIP, strip); r.set_option(rt.OPT_REFILL, refill | (tth << 8)); r.set_option(rt.OPT_RAY_BINNING, binning); r.set_option(rt.OPT_LDS_STACK, cap)
    r.trace_dev(brays, n, bhits); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): r.trace_dev(brays, n, bhits)
    e1.record(); torch.cuda.synchronize(); ms = e0.elapsed_time(e1) / 5
    r.set_option(rt.OPT_COUNTERS, 1); r.counters(); r.trace_dev(brays, n, bhits); torch.cuda.synchronize()
    c = r.counters(); r.set_option(rt.OPT_COUNTERS, 0)
    rounds = max(c["waves"], 1)
    blk, box = c["box_tests"] >> 32, c["box_tests"] & 0xffffffff
    tb, tt = c["tri_tests"] >> 32, c["tri_tests"] & 0xffffffff
    print(f"strip {strip} tth {tth} refill {refill} cap {cap}: {ms:.3f} ms N/ray {c['nodes']/c['rays']:.2f} T/ray {tt/c['rays']:.2f} "
          f"rounds/ray {rounds*64/c['rays']:.2f} | per round: interior {c['wave_max_nodes']/rounds:.1f} no-ray {c['sphere_tests']/rounds:.1f} "
          f"blocked {blk/rounds:.1f} | tri blocks/round {tb/rounds:.2f} lanes/tri-block {c['wave_max_tris']/max(tb,1):.1f}")
