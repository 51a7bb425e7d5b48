"""Experiment: how much does ray binning (origin cell in Morton order [+ direction octant]) buy on C4?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import raytracer_amd as rt
from raytracer_amd import scenes
n_tris, n_rays = 10_000_000, 64 << 20
tris = scenes.soup(n_tris); rays_np = scenes.random_rays(n_rays)
dev = torch.device("cuda", 0)
r = rt.Renderer((0,)); r.create_data(tris)
hits = torch.empty(n_rays * 4, dtype=torch.float32, device=dev)
def timeit(rays):
    r.trace_dev(rays, n_rays, hits); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): r.trace_dev(rays, n_rays, hits)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 3
def part1by2(x):
    x = x.astype(np.uint32) & 0x3ff
    x = (x | (x << 16)) & 0x030000FF; x = (x | (x << 8)) & 0x0300F00F
    x = (x | (x << 4)) & 0x030C30C3; x = (x | (x << 2)) & 0x09249249
    return x
rays_t = torch.from_numpy(rays_np.view(np.float32).reshape(-1)).to(dev)
print("unsorted: %.2f ms" % timeit(rays_t), flush=True)
o, d = rays_np["o"], rays_np["d"]
for bits, use_oct in ((4, False), (5, False), (6, False), (6, True), (7, False), (7, True)):
    q = np.clip((o * (1 << bits)).astype(np.int32), 0, (1 << bits) - 1)
    key = part1by2(q[:, 0]) | (part1by2(q[:, 1]) << 1) | (part1by2(q[:, 2]) << 2)
    if use_oct:
        key = (key << 3) | ((d[:, 0] < 0).astype(np.uint32) | ((d[:, 1] < 0).astype(np.uint32) << 1) | ((d[:, 2] < 0).astype(np.uint32) << 2))
    perm = np.argsort(key, kind="stable")
    srt = torch.from_numpy(rays_np[perm].view(np.float32).reshape(-1)).to(dev)
    print(f"grid {1<<bits}^3 octant={use_oct}: {timeit(srt):.2f} ms", flush=True)
    del srt
