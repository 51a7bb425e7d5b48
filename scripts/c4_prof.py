"""C4 workload for profiling runs: 10 M-triangle soup, 64 Mi random rays, a few closest-hit launches."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import argparse
import numpy as np, torch
import raytracer_amd as rt
from raytracer_amd import scenes
ap = argparse.ArgumentParser()
ap.add_argument("--soup", type=int, default=10_000_000); ap.add_argument("--rays", type=int, default=64 << 20)
ap.add_argument("--reps", type=int, default=3); ap.add_argument("--traversal", type=int, default=rt.TRAVERSAL_BVH)
ap.add_argument("--strip", type=int, default=-1); ap.add_argument("--refill", type=int, default=-1); ap.add_argument("--leaf", type=int, default=-1); ap.add_argument("--binning", type=int, default=-1); ap.add_argument("--cap", type=int, default=-1); ap.add_argument("--opts", default="", help="key=value,... extra rt_set_option pairs")
a = ap.parse_args()
tris = scenes.soup(a.soup); rays_np = scenes.random_rays(a.rays)
dev = torch.device("cuda", 0)
rays = torch.from_numpy(rays_np.view(np.float32).reshape(-1)).to(dev)
hits = torch.empty(a.rays * 4, dtype=torch.float32, device=dev)
r = rt.Renderer((0,)); r.set_option(rt.OPT_TRAVERSAL, a.traversal)
if a.leaf > 0: r.set_option(rt.OPT_MAX_LEAF, a.leaf)
if a.strip >= 0: r.set_option(rt.OPT_STRIP, a.strip)
if a.cap >= 0: r.set_option(rt.OPT_LDS_STACK, a.cap)
if a.binning >= 0: r.set_option(rt.OPT_RAY_BINNING, a.binning)
if a.refill >= 0: r.set_option(rt.OPT_REFILL, a.refill)
for kv in filter(None, a.opts.split(",")):
    k_, v_ = kv.split("="); r.set_option(int(k_), int(v_))
st = r.create_data(tris)
r.trace_dev(rays, a.rays, hits); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(a.reps): r.trace_dev(rays, a.rays, hits)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / a.reps
r.set_option(rt.OPT_COUNTERS, 1); r.counters(); r.trace_dev(rays, a.rays, hits); torch.cuda.synchronize(); c = r.counters()
print(f"C4 soup {a.soup} rays {a.rays} leaf {a.leaf} opts {a.opts}: {ms:.3f} ms  {a.rays/ms/1e3:.1f} Mrays/s  nodes {st['n_nodes']} sah {st['sah_cost']:.0f} build {st['build_ms']:.0f} ms  "
      f"N {c['nodes']/c['rays']:.1f} T {c['tri_tests']/c['rays']:.1f}", flush=True)
