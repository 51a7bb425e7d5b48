"""CPU tests of the host BVH builder and both flattened layouts (rt_bvh_build_host: the host half of
rt_scene_upload).  Build-original code (the reference has no acceleration structure); the invariant that
matters for parity is conservativeness: every child box, as the KERNEL decodes it, contains the exact bounds
of every triangle below it, so traversal can never cull a triangle the reference loop would test."""
import numpy as np
import pytest

import raytracer_amd as rt
from raytracer_amd import scenes


def tri_bounds(tris):
    p = np.stack([tris[k]["position"] for k in ("v0", "v1", "v2")], axis=1).astype(np.float64)   # [n,3,3]
    return p.min(axis=1), p.max(axis=1)


def decode_children(layout, node):
    """-> list of (lo[3], hi[3], ref) exactly as kernels.hip sees them."""
    out = []
    if layout == rt.TRAVERSAL_BVH2:
        f = node.view(np.float32).astype(np.float64)
        refs = node[12:14].view(np.int32)
        out.append((np.array([f[0], f[2], f[8]]), np.array([f[1], f[3], f[9]]), int(refs[0])))
        out.append((np.array([f[4], f[6], f[10]]), np.array([f[5], f[7], f[11]]), int(refs[1])))
        return out
    org = node[0:3].view(np.float32).astype(np.float64)
    scale = np.array([node[3:4].view(np.float32)[0], node[14:15].view(np.float32)[0], node[15:16].view(np.float32)[0]], dtype=np.float64)
    refs = node[4:8].view(np.int32)
    lo_words = [int(node[8]), int(node[9]), int(node[10])]
    hi_words = [int(node[11]), int(node[12]), int(node[13])]
    # empty slots are stored inverted (lo bytes 255, hi bytes 0); real children fill the slots from 0 up
    n = sum(1 for k in range(4) if not (((lo_words[0] >> (8 * k)) & 0xff) == 255 and ((hi_words[0] >> (8 * k)) & 0xff) == 0))
    for k in range(n):
        ql = np.array([(lo_words[a] >> (8 * k)) & 0xff for a in range(3)], dtype=np.float64)
        qh = np.array([(hi_words[a] >> (8 * k)) & 0xff for a in range(3)], dtype=np.float64)
        out.append((org + scale * ql, org + scale * qh, int(refs[k])))
    return out


def check_tree(tris, layout, max_leaf=4):
    stats, nodes, order = rt.bvh_build_host(tris, layout=layout, max_leaf=max_leaf, threads=2)
    n = len(tris)
    assert sorted(order.tolist()) == list(range(n)), "every triangle exactly once"
    tlo, thi = tri_bounds(tris)
    seen = np.zeros(n, dtype=int)
    visited_nodes = set()
    max_stack = 0

    def walk(node_i, lo, hi, depth, stack):
        nonlocal max_stack
        assert node_i not in visited_nodes
        visited_nodes.add(node_i)
        kids = decode_children(layout, nodes[node_i])
        assert 1 <= len(kids) <= (2 if layout == rt.TRAVERSAL_BVH2 else 4)
        max_stack = max(max_stack, stack + len(kids) - 1)
        sub_lo, sub_hi = np.full(3, np.inf), np.full(3, -np.inf)
        for klo, khi, ref in kids:
            if layout == rt.TRAVERSAL_BVH2 and klo[0] > khi[0]:
                continue                                   # inverted = empty slot (single-leaf root)
            if ref >= 0:
                clo, chi = walk(ref, klo, khi, depth + 1, stack + len(kids) - 1)
            else:
                x = ~ref
                first, cnt = x >> 3, (x & 7) + 1
                assert cnt <= max(max_leaf, 1) or cnt <= 8
                ids = order[first:first + cnt]
                seen[ids] += 1
                clo, chi = tlo[ids].min(axis=0), thi[ids].max(axis=0)
            assert (klo <= clo).all() and (khi >= chi).all(), "child box must contain everything below it"
            sub_lo, sub_hi = np.minimum(sub_lo, clo), np.maximum(sub_hi, chi)
        return sub_lo, sub_hi

    if n:
        import sys
        sys.setrecursionlimit(10000)
        walk(0, None, None, 1, 0)
        assert (seen == 1).all()
        assert len(visited_nodes) == stats["n_nodes"]
        if layout == rt.TRAVERSAL_BVH4Q:
            assert max_stack <= stats["stack_entries"] - 2, "reported stack bound must cover the tree"
            assert stats["n_lights"] == 0
    return stats


@pytest.mark.parametrize("layout", [rt.TRAVERSAL_BVH2, rt.TRAVERSAL_BVH4Q])
def test_layouts_are_conservative_and_complete(layout):
    check_tree(scenes.soup(3000, half_extent=0.05), layout)
    check_tree(scenes.soup(1, half_extent=0.3), layout)
    check_tree(scenes.soup(5, half_extent=0.3), layout)
    check_tree(scenes.cornell32()[0], layout)
    check_tree(scenes.soup(800, half_extent=0.2), layout, max_leaf=1)
    check_tree(scenes.soup(800, half_extent=0.2), layout, max_leaf=8)
    big = scenes.sphere_geometry((100.0, -50.0, 7.0), 3.0, 20, 30)          # far from the origin, thin pole slivers
    check_tree(big, layout)


def test_degenerate_inputs():
    same = np.zeros(64, dtype=rt.TRIANGLE)                                   # 64 identical zero triangles: no split plane
    st = check_tree(same, rt.TRAVERSAL_BVH2)
    assert st["max_depth"] <= 30
    check_tree(same, rt.TRAVERSAL_BVH4Q)
    st, nodes, order = rt.bvh_build_host(np.zeros(0, dtype=rt.TRIANGLE))
    assert st["n_nodes"] == 0 and len(order) == 0


def test_depth_is_bounded_for_adversarial_input():
    """Exponentially spaced triangles defeat SAH balance; the builder must fall back to median splits so that
    the traversal stack (LDS, sized at upload) stays bounded."""
    n = 4096
    t = np.zeros(n, dtype=rt.TRIANGLE)
    x = (1.0001 ** np.arange(n)).astype(np.float32) * np.linspace(1, 2000, n).astype(np.float32)
    for k, dx in (("v0", 0.0), ("v1", 0.01), ("v2", 0.0)):
        t[k]["position"][:, 0] = x + dx
        t[k]["position"][:, 1] = 0.01 if k == "v2" else 0.0
    st = check_tree(t, rt.TRAVERSAL_BVH2)
    assert st["max_depth"] <= 30
    st4 = check_tree(t, rt.TRAVERSAL_BVH4Q)
    assert st4["stack_entries"] <= 64


def test_build_is_deterministic_across_thread_counts():
    tris = scenes.soup(100000, half_extent=0.01)
    a = rt.bvh_build_host(tris, threads=1)
    b = rt.bvh_build_host(tris, threads=8)
    assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])


def test_cpu_bvh_walk_equals_the_reference_loop():
    """SURVEY 8(d) second CPU row: the oracle's walk over the product's flattened BVH2 (handed over as data) returns
    the brute-force reference loop's records bit for bit -- soups, Cornell (incl. KAT-1's tie), rays from outside,
    axis-parallel rays, intervals.  This also pins the builder: no accepted triangle's leaf may be culled."""
    import oracle
    for n_tris, half, leaf in ((1, 0.3, 4), (7, 0.3, 1), (3000, 0.05, 4), (20000, 0.02, 8)):
        tris = scenes.soup(n_tris, half_extent=half)
        rays = scenes.random_rays(20000)
        rays["o"][:5000] = rays["o"][:5000] * 4 - 2
        rays["d"][5000:5100] = (0.0, 0.0, 1.0)
        rays["tmax"][::9] = 0.1
        _, nodes, order = rt.bvh_build_host(tris, layout=rt.TRAVERSAL_BVH2, max_leaf=leaf)
        sc = oracle.Scene(tris)
        assert oracle.trace_bvh2(sc, nodes, order, rays).tobytes() == oracle.trace(sc, rays).tobytes(), (n_tris, leaf)
    tris, sph, _ = scenes.cornell()
    rays = oracle.primary_rays(rt.default_camera(1.0), 96, 96)
    kat = np.zeros(1, dtype=rt.RAY); kat["o"], kat["d"], kat["tmax"] = (0, 4, 1), (0, -1, 0), 1e20
    rays = np.concatenate([rays.ravel(), kat])
    _, nodes, order = rt.bvh_build_host(tris, layout=rt.TRAVERSAL_BVH2)
    sc = oracle.Scene(tris, sph)
    got = oracle.trace_bvh2(sc, nodes, order, rays)
    assert got.tobytes() == oracle.trace(sc, rays).tobytes()
    assert got[-1]["prim"] == 2 and got[-1]["t"] == 6.0                      # KAT-1: the tie keeps the lower index
    assert oracle.trace_bvh2(oracle.Scene(None, sph), np.zeros((0, 16), np.uint32), np.zeros(0, np.uint32), rays).tobytes() == \
        oracle.trace(oracle.Scene(None, sph), rays).tobytes()
