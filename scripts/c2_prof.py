"""C2 workload for profiling runs: bunny-class stand-in (69 k triangles), 1920x1080 primary rays + one shadow ray per hit pixel."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracer_amd as rt
from raytracer_amd import scenes
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
tris, sph, lights, cam = scenes.bunny_class()
w, h = 1920, 1080
n = w * h
dev = torch.device("cuda", 0)
r = rt.Renderer((0,)); r.create_data(tris, sph, lights); r.update_camera(cam)
rays, hits = torch.empty(n * 8, dtype=torch.float32, device=dev), torch.empty(n * 4, dtype=torch.float32, device=dev)
lit = torch.empty(n, dtype=torch.uint8, device=dev)
lp = tuple(float(x) for x in lights[0]["position"])
for _ in range(reps):
    r.trace_primary_dev(w, h, hits, rays)
    r.shadow_dev(rays, hits, n, lp, lit)
torch.cuda.synchronize()
e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
e[0].record(); r.trace_primary_dev(w, h, hits, rays); e[1].record(); r.shadow_dev(rays, hits, n, lp, lit); e[2].record(); torch.cuda.synchronize()
print(f"C2: primary {e[0].elapsed_time(e[1]):.3f} ms  shadow {e[1].elapsed_time(e[2]):.3f} ms  shadow rays {int((lit != 2).sum())}", flush=True)
