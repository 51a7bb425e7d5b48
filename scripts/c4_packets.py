"""Experiment: C4's random rays sorted by a fine Morton code of their origin (2^B cells per axis) and traced as packets of 64
(RT_OPT_RAY_PACKETS) against the default path (origin binning + ray-buffer kernel).  Sorting is done with torch and not timed."""
import os, sys
sys.path.insert(0, os.environ.get("RT_REPO") or os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import raytracer_amd as rt
from raytracer_amd import scenes
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16 << 20
tris = scenes.soup(10_000_000); rays_np = scenes.random_rays(n)
rays = torch.from_numpy(rays_np.view(np.float32).reshape(-1)).cuda()
r = rt.Renderer((0,)); st = r.create_data(tris)
def timed(buf, label, packets, ref=None, perm=None):
    r.set_option(rt.OPT_RAY_PACKETS, packets)
    hits = torch.empty(n * 4, dtype=torch.float32, device="cuda")
    r.trace_dev(buf, n, hits); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): r.trace_dev(buf, n, hits)
    e1.record(); torch.cuda.synchronize()
    r.set_option(rt.OPT_COUNTERS, 1); r.counters(); r.trace_dev(buf, n, hits); torch.cuda.synchronize(); c = r.counters(); r.set_option(rt.OPT_COUNTERS, 0)
    same = "" if ref is None else "  same records: %s" % bool(torch.equal(hits.view(-1, 4).view(torch.int32), torch.cat([torch.index_select(ref.view(-1, 4), 0, perm[a:a + (1 << 22)]) for a in range(0, n, 1 << 22)]).view(torch.int32)))
    print(f"{label}: {e0.elapsed_time(e1)/3:.3f} ms  {n/(e0.elapsed_time(e1)/3)/1e3:.0f} Mrays/s  N {c['nodes']/max(c['rays'],1):.1f}  T {c['tri_tests']/max(c['rays'],1):.1f}{same}", flush=True)
    r.set_option(rt.OPT_RAY_PACKETS, 0)
    return hits
ref = timed(rays, "default (binning + ray-buffer kernel)", 0)
R = rays.view(-1, 8)
def spread(x):          # 10 bits -> every third bit
    x = x & 0x3ff
    x = (x | (x << 16)) & 0x30000ff
    x = (x | (x << 8)) & 0x300f00f
    x = (x | (x << 4)) & 0x30c30c3
    x = (x | (x << 2)) & 0x9249249
    return x
for B in (4, 5, 6):
    q = (R[:, 0:3].clamp(0, 0.999999) * (1 << B)).long()
    key = spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)
    perm = torch.argsort(key, stable=True)
    buf = torch.cat([torch.index_select(R, 0, perm[a:a + (1 << 22)]) for a in range(0, n, 1 << 22)]).view(-1)   # (in pieces: one gather of 64 Mi rows exceeds a torch launch limit)
    timed(buf, f"sorted by {B}-bit Morton of the origin, ray-buffer kernel", 0, ref, perm)
    if n <= (16 << 20): timed(buf, f"sorted by {B}-bit Morton of the origin, packets", 1, ref, perm)
    del buf, perm, key, q
    torch.cuda.empty_cache()
