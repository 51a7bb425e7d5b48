"""A/B of the two node layouts on C3 (primary + bounce) and a 2M-tri soup, one process, interleaved."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import raytracer_amd as rt
from raytracer_amd import scenes
def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/reps
f=lambda k: torch.empty(k, dtype=torch.float32, device='cuda')
tris, sph, _, cam = scenes.sponza_class()
W,H,SPP=3840,2160,4; n=W*H*SPP
rays,hits,brays,bhits=f(n*8),f(n*4),f(n*8),f(n*4)
soup = scenes.soup(2_000_000); sr_np = scenes.random_rays(8_000_000)
sr = torch.from_numpy(sr_np.view(np.float32).reshape(-1)).cuda(); sh = f(8_000_000*4)
R = {}
for name, mode in (("bvh2", 2), ("bvh4q", 3)):
    r = rt.Renderer((0,)); r.set_option(rt.OPT_TRAVERSAL, mode); r.create_data(tris, sph); r.update_camera(cam)
    r2 = rt.Renderer((0,)); r2.set_option(rt.OPT_TRAVERSAL, mode); r2.create_data(soup)
    R[name] = (r, r2)
ref = {}
for rnd in range(2):
    for name,(r,r2) in R.items():
        tp = timeit(lambda: r.trace_primary_dev(W,H,hits,rays,spp=SPP,seed=1,jitter=True))
        r.bounce_dev(rays,hits,n,brays,seed=1)
        tb = timeit(lambda: r.trace_dev(brays,n,bhits))
        ts = timeit(lambda: r2.trace_dev(sr, 8_000_000, sh), 3)
        key = (hits.cpu().numpy().tobytes()[:1<<24], bhits.cpu().numpy().tobytes()[:1<<24], sh.cpu().numpy().tobytes()[:1<<24])
        same = ref.setdefault("k", key) == key
        print(f"round {rnd} {name}: primary {tp:.3f} ms  bounce {tb:.3f} ms  soup2M/8Mrays {ts:.3f} ms ({8000/ts:.0f} Mrays/s)  same-as-first={same}", flush=True)
