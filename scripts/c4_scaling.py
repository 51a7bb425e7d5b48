"""C4 soup: closest-hit time against the number of rays (fixed cost of a launch vs cost per ray), binned and unbinned."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import raytracer_amd as rt
from raytracer_amd import scenes
n_max = 64 << 20
tris = scenes.soup(10_000_000); rays_np = scenes.random_rays(n_max)
dev = torch.device("cuda", 0)
rays = torch.from_numpy(rays_np.view(np.float32).reshape(-1)).to(dev)
hits = torch.empty(n_max * 4, dtype=torch.float32, device=dev)
r = rt.Renderer((0,)); r.create_data(tris)
for binning in (2, 0):
    r.set_option(rt.OPT_RAY_BINNING, binning)
    for n in (1 << 20, 2 << 20, 4 << 20, 8 << 20, 16 << 20, 32 << 20, 64 << 20):
        for _ in range(2): r.trace_dev(rays, n, hits)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4): r.trace_dev(rays, n, hits)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 4
        print(f"binning {binning} rays {n >> 20:3d} Mi: {ms:8.3f} ms  {n / ms / 1e3:7.1f} Mrays/s  {ms / (n >> 20) * 1e3:7.1f} us per Mi rays", flush=True)
