"""Adversarial parity fuzz (north_star: "mismatch rate on epsilon/tie cases stated"): scenes built to provoke ties and
epsilon decisions -- coplanar triangle grids with shared edges and vertices, duplicated triangles (same geometry, two
indices), zero-area and sliver triangles, axis-aligned geometry -- and rays aimed exactly at vertices, edge midpoints and
along edges, plus axis-parallel and grazing rays.  Every BVH mode / kernel must return the brute-force HIP kernel's
records bit for bit (that kernel IS the reference loop, and is itself checked against the CPU oracle on a sample)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oracle
import raytracer_amd as rt
from raytracer_amd import scenes

rng = np.random.default_rng(12345)
total_rays = total_bad = total_oracle = total_oracle_bad = 0

tri_array = scenes.tri_array
rays_at = scenes.rays_at
def grid_scene(n, jitter_z):
    return scenes.grid_scene(n, jitter_z, rng)

def check(name, tris, rays, sph=None):
    global total_rays, total_bad, total_oracle, total_oracle_bad
    with rt.Renderer((0,)) as r:
        r.set_option(rt.OPT_TRAVERSAL, rt.TRAVERSAL_BRUTE); r.create_data(tris, sph)
        ref = r.trace(rays)
        bad = 0
        for mode in (rt.TRAVERSAL_BVH2, rt.TRAVERSAL_BVH4Q):
            for strip in (0, 256):
                for leaf in (1, 4):
                    r.set_option(rt.OPT_TRAVERSAL, mode); r.set_option(rt.OPT_STRIP, strip); r.set_option(rt.OPT_MAX_LEAF, leaf)
                    r.create_data(tris, sph)
                    got = r.trace(rays)
                    bad += int((got.view(np.uint32).reshape(-1, 4) != ref.view(np.uint32).reshape(-1, 4)).any(axis=1).sum())
                    total_rays += len(rays)
        total_bad += bad
        k = min(len(rays), 2000)
        idx = rng.choice(len(rays), k, replace=False)
        o = oracle.trace(oracle.Scene(tris, sph), rays[idx])
        ob = int((o.view(np.uint32).reshape(-1, 4) != ref[idx].view(np.uint32).reshape(-1, 4)).any(axis=1).sum())
        total_oracle += k; total_oracle_bad += ob
        hit = float((ref["prim"] != rt.MISS).mean())
        print(f"{name}: {len(tris)} tris, {len(rays)} rays, hit {hit:.2f}: BVH-vs-brute differing records {bad}, brute-vs-oracle {ob}/{k}", flush=True)

for n, jz in ((8, 0.0), (8, 0.05), (37, 0.0), (37, 0.01)):
    tris, verts = grid_scene(n, jz)
    dup = scenes.cat([tris, tris[::3]])                                  # duplicates: same geometry under a second index
    e_mid = (verts[:-1] + verts[1:]) / 2
    targets = np.concatenate([verts, e_mid, (verts[:-n - 2] + verts[n + 2:]) / 2])
    for label, origin in (("above", np.array([0.5, 0.5, 2.0])), ("oblique", np.array([-0.7, 0.3, 0.4])), ("in-plane", np.array([-1.0, 0.5, 0.0]))):
        origins = np.broadcast_to(origin, targets.shape) + 0.0
        check(f"grid{n} jz{jz} {label}", dup, rays_at(targets, origins))
    # straight down onto every vertex / edge midpoint: exact axis-parallel rays
    r = np.zeros(len(targets), dtype=rt.RAY)
    r["o"] = targets + np.array([0, 0, 1.5], dtype=np.float32); r["d"] = (0, 0, -1); r["tmax"] = 1e20
    check(f"grid{n} jz{jz} axis-parallel", dup, r)
# slivers, zero-area triangles, coincident spheres
p0 = rng.random((3000, 3)); e = rng.standard_normal((3000, 3)) * 0.05
t = tri_array(p0, p0 + e, p0 + e * (1 + 1e-6 * rng.standard_normal((3000, 1))))          # slivers
z = tri_array(p0[:200], p0[:200], p0[:200] + e[:200])                                    # zero area
sph = np.zeros(4, dtype=rt.SPHERE); sph["center"] = [(0.5, 0.5, 0.5)] * 2 + [(0.2, 0.2, 0.2)] * 2; sph["radius"] = [0.2, 0.2, 0.1, 0.1]
check("slivers + zero-area + coincident spheres", scenes.cat([t, z]), scenes.random_rays(60000), sph)
print(f"TOTAL: {total_rays} BVH ray results compared with the brute-force kernel: {total_bad} differing records "
      f"(rate {total_bad / max(total_rays, 1):.2e}); brute-force kernel vs CPU oracle: {total_oracle_bad}/{total_oracle}", flush=True)
sys.exit(1 if (total_bad or total_oracle_bad) else 0)
