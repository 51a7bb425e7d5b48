"""Developer check: HIP path vs oracle on small cases + first timings.  Not part of the test suite."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import raytracer_amd as rt
from raytracer_amd import scenes
import oracle

def cmp(a, b, name):
    same = a.tobytes() == b.tobytes()
    npr = int((a['prim'] != b['prim']).sum())
    print(f"{name}: bit-exact={same} prim-mismatch={npr}/{a.size}", flush=True)
    if not same:
        idx = np.flatnonzero((a['prim'] != b['prim']).ravel() | (a['t'] != b['t']).ravel())[:5]
        for i in idx: print("   ", i, a.ravel()[i], b.ravel()[i])
    return same

r = rt.Renderer((0,))
tris, sph, _ = scenes.cornell()
cam = rt.default_camera(1.0)
sc = oracle.Scene(tris, sph)
ref = oracle.trace_primary(sc, cam, 512, 512)
for mode in (1, 2):
    r.set_option(rt.OPT_TRAVERSAL, mode)
    print(r.create_data(tris, sph))
    r.update_camera(cam)
    got = r.trace_primary(512, 512)
    cmp(got, ref, f"C1 cornell 512x512 mode={mode}")
kat = np.zeros(2, dtype=rt.RAY)
kat[0]['o']=(0,4,1); kat[0]['d']=(0,-1,0); kat[0]['tmax']=1e20
kat[1]['o']=(0.5,1,1); kat[1]['d']=(0,0,-1); kat[1]['tmax']=1e20
print("KAT", r.trace(kat))

# soup
r.set_option(rt.OPT_TRAVERSAL, 2)
st = scenes.soup(20000, half_extent=0.02)
rays = scenes.random_rays(50000)
print(r.create_data(st))
t=time.time(); ref = oracle.trace(oracle.Scene(st), rays); print("oracle s", time.time()-t)
got = r.trace(rays)
cmp(got, ref, "soup20k bvh")
r.set_option(rt.OPT_TRAVERSAL, 1); r.create_data(st)
cmp(r.trace(rays), ref, "soup20k brute")
occ = r.occluded(rays); print("occluded match", np.array_equal(occ, (ref['prim'] != rt.MISS).astype(np.uint8)))

# C3 timing
r.set_option(rt.OPT_TRAVERSAL, 2)
tr, _, _, cam3 = scenes.sponza_class()
t=time.time(); print(r.create_data(tr)); print("upload wall", time.time()-t)
r.update_camera(cam3)
W,H,SPP=3840,2160,4
n = W*H*SPP
rays_d = torch.empty(n*8, dtype=torch.float32, device='cuda')
hits_d = torch.empty(n*4, dtype=torch.float32, device='cuda')
brays_d = torch.empty(n*8, dtype=torch.float32, device='cuda')
bhits_d = torch.empty(n*4, dtype=torch.float32, device='cuda')
s = torch.cuda.current_stream()
for it in range(3):
    torch.cuda.synchronize(); t0=time.time()
    r.trace_primary_dev(W,H,hits_d,rays_d,spp=SPP,seed=1,jitter=True,stream=s)
    torch.cuda.synchronize(); t1=time.time()
    r.bounce_dev(rays_d,hits_d,n,brays_d,seed=1,stream=s)
    torch.cuda.synchronize(); t2=time.time()
    r.trace_dev(brays_d,n,bhits_d,stream=s)
    torch.cuda.synchronize(); t3=time.time()
    print(f"C3 primary {n/(t1-t0)/1e6:.1f} Mrays/s  bouncegen {(t2-t1)*1e3:.2f} ms  bounce-trace {n/(t3-t2)/1e6:.1f} Mrays/s", flush=True)
# parity sample for C3: 20000 random primary + bounce rays vs oracle
hr = rays_d.cpu().numpy().view(rt.RAY); hh = hits_d.cpu().numpy().view(rt.HIT)
br = brays_d.cpu().numpy().view(rt.RAY); bh = bhits_d.cpu().numpy().view(rt.HIT)
rng = np.random.default_rng(0); idx = rng.choice(n, 20000, replace=False)
sc3 = oracle.Scene(tr)
t=time.time(); ref = oracle.trace(sc3, hr[idx]); print("oracle C3 sample s", time.time()-t)
cmp(hh[idx], ref, "C3 primary sample")
ob = oracle.bounce(sc3, hr[idx], hh[idx], seed=1)
# keys differ (index-based): regenerate on GPU with compact buffers instead
ref = oracle.trace(sc3, br[idx]); cmp(bh[idx], ref, "C3 bounce sample")
r.set_option(rt.OPT_COUNTERS, 1)
r.trace_primary_dev(W,H,hits_d,rays_d,spp=SPP,seed=1,jitter=True,stream=s); torch.cuda.synchronize()
print("primary counters", r.counters())
r.trace_dev(brays_d,n,bhits_d,stream=s); torch.cuda.synchronize()
print("bounce counters", r.counters())
