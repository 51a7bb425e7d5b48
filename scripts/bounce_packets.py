"""Experiment: bounce rays as packets.  The bench frame's bounce rays are grouped by (screen patch of PxP pixels, direction
bin: cube-map face x KxK cells), physically reordered with torch, and traced (a) by the ray-buffer kernel in pixel order
(b) by the packet kernel (RT_OPT_RAY_PACKETS) in the grouped order.  Hit records must agree ray for ray."""
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
thr, rad, brays, bhits, phits = f32(4), f32(4), f32(8), f32(4), f32(4)
r.primary_shade_stripes_dev(W, H, 0, H, H, 1, thr, rad, brays, spp=SPP, seed=1, jitter=True, shade_seed=2)
torch.cuda.synchronize()
R = brays.view(-1, 8)
idx = torch.arange(n, device="cuda")
px, py = (idx // SPP) % W, (idx // SPP) // W
dead = (R[:, 7] <= R[:, 3]).long()
d = R[:, 4:7]
ad = d.abs()
face = torch.argmax(ad, dim=1)                                   # dominant axis
sign = (torch.gather(d, 1, face[:, None])[:, 0] < 0).long()
u = torch.gather(d, 1, ((face + 1) % 3)[:, None])[:, 0] / torch.gather(ad, 1, face[:, None])[:, 0].clamp_min(1e-20)
v = torch.gather(d, 1, ((face + 2) % 3)[:, None])[:, 0] / torch.gather(ad, 1, face[:, None])[:, 0].clamp_min(1e-20)
def timed(rays, label, packets, ref=None, perm=None):
    r.set_option(rt.OPT_RAY_PACKETS, packets)
    hits = f32(4)
    for _ in range(2): r.trace_dev(rays, n, hits)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): r.trace_dev(rays, n, hits)
    e1.record(); torch.cuda.synchronize()
    r.set_option(rt.OPT_COUNTERS, 1); r.counters(); r.trace_dev(rays, n, hits); torch.cuda.synchronize(); c = r.counters(); r.set_option(rt.OPT_COUNTERS, 0)
    same = ""
    if ref is not None:
        same = "  same records: %s" % bool(torch.equal(hits.view(-1, 4).view(torch.int32), ref.view(-1, 4)[perm].view(torch.int32)))
    print(f"{which} {label}: {e0.elapsed_time(e1)/5:.3f} ms  N {c['nodes']/max(c['rays'],1):.2f}  T {c['tri_tests']/max(c['rays'],1):.2f}{same}", flush=True)
    r.set_option(rt.OPT_RAY_PACKETS, 0)
    return hits
ref = timed(brays, "pixel order, ray-buffer kernel", 0)
timed(brays, "pixel order, packets", 1, ref, idx)
for P, K in ((16, 2), (32, 2), (32, 4), (64, 4), (64, 8), (128, 8)):
    cell = ((u * 0.5 + 0.5) * K).long().clamp(0, K - 1) * K + ((v * 0.5 + 0.5) * K).long().clamp(0, K - 1)
    dbin = (face * 2 + sign) * K * K + cell
    patch = (py // P) * ((W + P - 1) // P) + (px // P)
    key = (dead * (1 << 40)) + patch * (6 * K * K) + dbin
    perm = torch.argsort(key, stable=True)
    timed(R[perm].contiguous().view(-1), f"patch {P}x{P} px, {6*K*K} direction bins, packets", 1, ref, perm)
