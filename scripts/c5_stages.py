"""Per-stage timing of the C5 stand-in frame (10 M-triangle soup, 3840x2160 x 16 spp x 3 bounces)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracer_amd as rt
from raytracer_amd import scenes
w, h, spp, bounces = 3840, 2160, 16, 3
tris = scenes.soup(int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000)
tris["material"]["emission"][::50] = (1.0, 0.9, 0.8); tris["material"]["emissionStrength"][::50] = 2.0
cam = rt.camera_look_at((0.5, 2.2, 0.5), (0.5, 0.5, 0.5), 45.0, w / h)
n = w * h * spp
dev = torch.device("cuda", 0)
f32 = lambda k: torch.empty(n * k, dtype=torch.float32, device=dev)
with rt.Renderer((0,)) as r:
    for kv in filter(None, os.environ.get('RT_OPTS', '').split(',')):      # tuning sweeps: key=value pairs of rt_set_option
        r.set_option(int(kv.split('=')[0]), int(kv.split('=')[1]))
    r.create_data(tris); r.update_camera(cam)
    ra, rb, hits, thr, rad = f32(8), f32(8), f32(4), f32(4), f32(4)
    fb = torch.empty(w * h, dtype=torch.int32, device=dev)
    names, evs = [], []
    def mark(name):
        e = torch.cuda.Event(enable_timing=True); e.record(); names.append(name); evs.append(e)
    for rep in range(2):
        names, evs = [], []
        mark("start")
        r.primary_shade_stripes_dev(w, h, 0, h, h, 1, thr, rad, ra, spp=spp, seed=1, jitter=True, shade_seed=2); mark("primary+shade")
        a, b = ra, rb
        for k in range(1, bounces):
            r.trace_dev(a, n, hits); mark(f"trace {k}")
            if k + 1 == bounces:
                r.shade_resolve_dev(a, hits, n, spp, thr, rad, None, fb); mark("shade+resolve")
            else:
                r.shade_dev(a, hits, n, thr, rad, b, key0=0, seed=2 + k); mark(f"shade {k}")
                a, b = b, a
        torch.cuda.synchronize()
    tot = evs[0].elapsed_time(evs[-1])
    print("C5 stages (ms):", {names[i + 1]: round(evs[i].elapsed_time(evs[i + 1]), 2) for i in range(len(evs) - 1)}, "total", round(tot, 2))
