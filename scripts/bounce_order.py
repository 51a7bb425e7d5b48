"""Experiment: does handing out the bench frame's bounce rays grouped by direction octant (within windows of W rays, so
origins stay close) raise the ray-buffer kernel's rate?  Rays are physically reordered with torch; the kernel is the same."""
import os, sys
sys.path.insert(0, os.environ.get("RT_REPO") or os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracer_amd as rt
from raytracer_amd import scenes
W, H, SPP = 3840, 2160, 4
which = sys.argv[1] if len(sys.argv) > 1 else "sponza"
tris, sph, lights, cam = scenes.sponza_class() if which == "sponza" else scenes.colonnade()
r = rt.Renderer((0,)); r.create_data(tris, sph, lights); r.update_camera(cam)
n = W * H * SPP
f32 = lambda k: torch.empty(n * k, dtype=torch.float32, device="cuda")
thr, rad, brays, bhits = f32(4), f32(4), f32(8), f32(4)
r.primary_shade_stripes_dev(W, H, 0, H, H, 1, thr, rad, brays, spp=SPP, seed=1, jitter=True, shade_seed=2)
torch.cuda.synchronize()
R = brays.view(-1, 8)
octant = ((R[:, 4] < 0).long() | ((R[:, 5] < 0).long() << 1) | ((R[:, 6] < 0).long() << 2))
dead = (R[:, 7] <= R[:, 3]).long()
idx = torch.arange(n, device="cuda")
def timed(rays, label):
    for _ in range(3): r.trace_dev(rays, n, bhits)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): r.trace_dev(rays, n, bhits)
    e1.record(); torch.cuda.synchronize()
    r.set_option(rt.OPT_COUNTERS, 1); r.counters(); r.trace_dev(rays, n, bhits); torch.cuda.synchronize(); c = r.counters(); r.set_option(rt.OPT_COUNTERS, 0)
    print(f"{which} {label}: {e0.elapsed_time(e1)/10:.3f} ms  N {c['nodes']/c['rays']:.2f}  lanes per interior round {c['wave_max_nodes']/max(c['waves'],1):.1f}", flush=True)
timed(brays, "pixel order")
for win in (256, 1024, 4096, 65536, 1 << 20):
    key = (idx // win) * 16 + dead * 8 + octant
    perm = torch.argsort(key, stable=True)
    timed(R[perm].contiguous().view(-1), f"octant within {win}")
