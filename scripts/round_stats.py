"""Scheduling statistics of the ray-buffer kernel on the bench scene's bounce rays (instrumented build): rounds per wave,
lanes taking part in interior steps and in triangle blocks, node / triangle tests per ray.  1920x1080 x 4 spp."""
import os, sys
sys.path.insert(0, os.environ.get("RT_REPO") or os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracer_amd as rt
from raytracer_amd import scenes
W, H, SPP = 1920, 1080, 4
tris, sph, lights, cam = scenes.sponza_class()
r = rt.Renderer((0,)); r.create_data(tris, sph, lights); r.update_camera(cam)
n = W * H * SPP
f32 = lambda k: torch.empty(n * k, dtype=torch.float32, device="cuda")
thr, rad, brays, bhits = f32(4), f32(4), f32(8), f32(4)
r.primary_shade_stripes_dev(W, H, 0, H, H, 1, thr, rad, brays, spp=SPP, seed=1, jitter=True, shade_seed=2)
for opt in sys.argv[1:] or ["default"]:
    if opt != "default":
        k, v = opt.split("="); r.set_option(int(k), int(v))
    r.set_option(rt.OPT_COUNTERS, 1); r.counters()
    r.trace_dev(brays, n, bhits); torch.cuda.synchronize()
    c = r.counters(); r.set_option(rt.OPT_COUNTERS, 0)
    for k_ in ('idle_lane_rounds', 'blocked_lane_rounds', 'drained_lane_rounds', 'tri_rounds'): c.setdefault(k_, 0)
    rays = c["rays"]; waves_worth = rays / 64
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    r.trace_dev(brays, n, bhits); torch.cuda.synchronize()
    for _ in range(10): r.trace_dev(brays, n, bhits)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(20): r.trace_dev(brays, n, bhits)
    e1.record(); torch.cuda.synchronize()
    print(f"{opt}: {e0.elapsed_time(e1)/20:.3f} ms  rays {rays}  N {c['nodes']/rays:.2f}  B {c['box_tests']/rays:.2f}  T {c['tri_tests']/rays:.2f}  "
          f"rounds/64 rays {c['waves']/waves_worth:.1f}  lanes per interior round {c['wave_max_nodes']/max(c['waves'],1):.1f}  "
          f"lanes per round in tri blocks {c['wave_max_tris']/max(c['waves'],1):.1f}  "
          f"(interior lane-steps {c['wave_max_nodes']/rays:.2f}/ray, tri lane-steps {c['wave_max_tris']/rays:.2f}/ray)  "
          f"per round: idle {c['idle_lane_rounds']/max(c['waves'],1):.1f}  blocked on 2nd leaf {c['blocked_lane_rounds']/max(c['waves'],1):.1f}  "
          f"drained {c['drained_lane_rounds']/max(c['waves'],1):.1f}  tri-block rounds {c['tri_rounds']/max(c['waves'],1):.2f} "
          f"(lanes per tri block {c['wave_max_tris']/max(c['tri_rounds'],1):.1f})", flush=True)
