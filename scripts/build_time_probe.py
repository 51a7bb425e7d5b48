"""Host half of rt_scene_upload on the 10 M-triangle soup for several thread counts (RT_HOST_TIMING=1 prints the phases)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as rt
from raytracer_amd import scenes
tris = scenes.soup(int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000)
for th in (16, 8, 32, 16):
    with rt.Renderer((0,)) as r:
        r.set_option(rt.OPT_BVH_THREADS, th)
        t = time.time(); st = r.create_data(tris); dt = time.time() - t
        print(f"threads {th}: upload call {dt:.3f} s  build_ms {st['build_ms']:.1f}  upload_ms {st['upload_ms']:.1f}  nodes {st['n_nodes']} sah {st['sah_cost']:.3f}", flush=True)
