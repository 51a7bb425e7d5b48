import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import raytracer_amd as rt
from raytracer_amd import scenes
tris, sph, _, cam = scenes.sponza_class()
r = rt.Renderer((0,)); r.set_option(rt.OPT_TRAVERSAL, 2); r.create_data(tris, sph); r.update_camera(cam)
W,H,SPP=3840,2160,4; n=W*H*SPP
f=lambda k: torch.empty(n*k, dtype=torch.float32, device='cuda')
rays,hits,brays,bhits=f(8),f(4),f(8),f(4)
r.set_option(rt.OPT_COUNTERS,1); r.counters()
r.trace_primary_dev(W,H,hits,rays,spp=SPP,seed=1,jitter=True); torch.cuda.synchronize()
for name in ("primary","bounce"):
    c=r.counters()
    if name=="primary":
        r.bounce_dev(rays,hits,n,brays,seed=1); r.trace_dev(brays,n,bhits); torch.cuda.synchronize()
    else: pass
    if name=='bounce': print('rounds/wave-strip: rounds %d  I-lanes/round %.1f  T-runs %d (%.2f of rounds) T-lanes/run %.1f' % (c['waves'], c['wave_max_nodes']/c['waves'], c['sphere_tests'], c['sphere_tests']/c['waves'], c['wave_max_tris']/max(c['sphere_tests'],1)))
    print(name, c, "mean nodes/ray %.1f  wave-max nodes %.1f  mean tris %.2f wave-max tris %.1f" % (c['nodes']/c['rays'], c['wave_max_nodes']/max(c['waves'],1), c['tri_tests']/c['rays'], c['wave_max_tris']/max(c['waves'],1)), 'both-miss nodes/ray %.2f' % (c['sphere_tests']/c['rays']))
