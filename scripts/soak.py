"""Soak: the same frame / ray buffer traced many times must give the same bytes every time (persistent kernels, cursor
rings, spill slots and binning scratch are reused across launches; a race would show up here)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import raytracer_amd as rt
from raytracer_amd import scenes
dev = torch.device("cuda", 0)
W, H, SPP = 3840, 2160, 4; n = W * H * SPP
f = lambda k: torch.empty(n * k, dtype=torch.float32, device=dev)
tris, sph, lights, cam = scenes.sponza_class()
with rt.Renderer((0,)) as r:
    r.create_data(tris, sph, lights); r.update_camera(cam)
    thr, rad, brays, bhits = f(4), f(4), f(8), f(4)
    fb = torch.empty(W * H, dtype=torch.int32, device=dev)
    s2 = torch.cuda.Stream()
    ref, bad = None, 0
    for it in range(300):
        st = torch.cuda.current_stream() if it % 3 else s2          # alternate streams (stream-ordered via events below)
        with torch.cuda.stream(st):
            r.primary_shade_stripes_dev(W, H, 0, H, H, 1, thr, rad, brays, spp=SPP, seed=1, jitter=True, shade_seed=2, stream=st)
            r.trace_dev(brays, n, bhits, stream=st)
            r.shade_resolve_dev(brays, bhits, n, SPP, thr, rad, None, fb, stream=st)
        torch.cuda.synchronize()
        if ref is None: ref = fb.clone()
        elif not torch.equal(fb, ref): bad += 1
    print(f"C3 frame x300: {bad} differing frames", flush=True)
n4 = 16 << 20
tris = scenes.soup(3_000_000); rays_np = scenes.random_rays(n4)
rays_np["tmax"][::17] = 0.0
rays = torch.from_numpy(rays_np.view(np.float32).reshape(-1)).to(dev)
with rt.Renderer((0,)) as r:
    r.create_data(tris)
    hits = torch.empty(n4 * 4, dtype=torch.float32, device=dev); occ = torch.empty(n4, dtype=torch.uint8, device=dev)
    ref, bad = None, 0
    for it in range(60):
        r.set_option(rt.OPT_RAY_BINNING, (2, 1, 0)[it % 3]); r.set_option(rt.OPT_LDS_STACK, (20, 4, 0, 12)[it % 4]); r.set_option(rt.OPT_RAY_QUEUE, (2, 1, 0, 1, 2)[it % 5])
        r.trace_dev(rays, n4, hits); r.occluded_dev(rays, n4, occ); torch.cuda.synchronize()
        cur = (hits.clone(), occ.clone())
        if ref is None: ref = cur
        elif not (torch.equal(cur[0].view(torch.int32), ref[0].view(torch.int32)) and torch.equal(cur[1], ref[1])): bad += 1
    print(f"soup 3M x 16Mi rays x60 (binning / stack cap / ray queue cycled): {bad} differing results", flush=True)
