import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import raytracer_amd as rt
from raytracer_amd import scenes
tris, sph, _, cam = scenes.sponza_class()
W,H,SPP=3840,2160,4; n=W*H*SPP
f=lambda k: torch.empty(n*k, dtype=torch.float32, device='cuda')
rays,hits,brays,bhits=f(8),f(4),f(8),f(4)
def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/reps
r = rt.Renderer((0,)); r.set_option(rt.OPT_TRAVERSAL, rt.TRAVERSAL_BVH)
for leaf in (1, 2, 3, 4, 6, 8):
    r.set_option(rt.OPT_MAX_LEAF, leaf)
    st = r.create_data(tris, sph); r.update_camera(cam)
    tp = timeit(lambda: r.trace_primary_dev(W,H,hits,rays,spp=SPP,seed=1,jitter=True))
    r.bounce_dev(rays,hits,n,brays,seed=1)
    tb = timeit(lambda: r.trace_dev(brays,n,bhits))
    r.set_option(rt.OPT_COUNTERS,1); r.counters(); r.trace_dev(brays,n,bhits); torch.cuda.synchronize(); c=r.counters(); r.set_option(rt.OPT_COUNTERS,0)
    print(f"leaf {leaf}: nodes {st['n_nodes']} depth {st['max_depth']} sah {st['sah_cost']:.1f} primary {tp:.3f} ms bounce {tb:.3f} ms  N/ray {c['nodes']/c['rays']:.1f} T/ray {c['tri_tests']/c['rays']:.2f}", flush=True)
