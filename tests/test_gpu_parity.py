"""GPU parity tests: the HIP path, called through the C ABI, against the CPU oracle and the golden fixtures.

Bar (BASELINE.json north_star): primitive IDs bit-exact, t/u/v within 1e-5 relative.  Because the kernels
and the oracle fix the same fp32 evaluation order (DESIGN.md "Arithmetic contract"), these tests ask for
MORE: the whole 16-byte hit record must be bit-identical; `assert_parity` reports the prim-ID mismatch
rate and the worst relative error when it is not.
"""
import hashlib
import json
import os

import numpy as np
import pytest

import oracle
import raytracer_amd as rt
from raytracer_amd import dist as rd
from raytracer_amd import scenes

from conftest import GOLDEN

pytestmark = pytest.mark.gpu

REL_TOL = 1e-5      # north_star tolerance for t, u, v (we require bit-exactness, this is only reported)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def assert_parity(got, ref, what):
    got, ref = got.ravel(), ref.ravel()
    assert got.shape == ref.shape, what
    if got.tobytes() == ref.tobytes():
        return
    bad = got["prim"] != ref["prim"]
    rel = 0.0
    for f in ("t", "u", "v"):
        a, b = got[f][~bad].astype(np.float64), ref[f][~bad].astype(np.float64)
        rel = max(rel, float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-30), initial=0.0)))
    first = np.flatnonzero(got.view(np.uint32).reshape(-1, 4) != ref.view(np.uint32).reshape(-1, 4))[:1]
    raise AssertionError(f"{what}: prim-ID mismatches {int(bad.sum())}/{got.size} "
                         f"({bad.mean():.3e}), max rel err t/u/v {rel:.3e} (tol {REL_TOL}); first diff at word {first}")


def mk_rays(o, d, tmin=0.0, tmax=1e20):
    o, d = np.atleast_2d(np.asarray(o, dtype=np.float32)), np.atleast_2d(np.asarray(d, dtype=np.float32))
    r = np.zeros(len(o), dtype=rt.RAY)
    r["o"], r["d"], r["tmin"], r["tmax"] = o, d, tmin, tmax
    return r


@pytest.fixture(scope="module")
def torch_dev():
    import torch
    assert torch.cuda.is_available()
    return torch


@pytest.fixture()
def renderer():
    r = rt.Renderer((0,))
    yield r
    r.close()


def dev_buf(torch, n_items, itemsize):
    return torch.empty(n_items * itemsize // 4, dtype=torch.float32, device="cuda")


def to_np(t, dtype):
    return t.cpu().numpy().view(dtype)


def to_dev(torch, a):
    return torch.from_numpy(np.ascontiguousarray(a).view(np.float32).reshape(-1).copy()).cuda()


# ---- C1: Cornell box, the reference's own scene ------------------------------------------------------
@pytest.mark.parametrize("mode", [rt.TRAVERSAL_AUTO, rt.TRAVERSAL_BRUTE, rt.TRAVERSAL_BVH2, rt.TRAVERSAL_BVH4Q])
def test_c1_cornell_512_full_frame_bit_exact(renderer, mode):
    tris, sph, _ = scenes.cornell()
    cam = rt.default_camera(1.0)
    renderer.set_option(rt.OPT_TRAVERSAL, mode)
    renderer.create_data(tris, sph)
    renderer.update_camera(cam)
    got = renderer.trace_primary(512, 512)
    assert_parity(got, oracle.trace_primary(oracle.Scene(tris, sph), cam, 512, 512), f"C1 512x512 mode {mode}")
    meta = json.load(open(os.path.join(GOLDEN, "meta.json")))
    assert sha(got) == meta["c1_cornell_512x512_sha256"]
    assert (got["prim"] != rt.MISS).all()                      # closed box: every primary ray hits


@pytest.mark.parametrize("mode", [rt.TRAVERSAL_BRUTE, rt.TRAVERSAL_BVH2, rt.TRAVERSAL_BVH4Q])
def test_kats_on_gpu(renderer, mode):
    tris, sph, _ = scenes.cornell()
    renderer.set_option(rt.OPT_TRAVERSAL, mode)
    renderer.create_data(tris, sph)
    kat = json.load(open(os.path.join(GOLDEN, "kat.json")))
    for name, k in kat.items():
        h = renderer.trace(mk_rays(k["o"], k["d"]))[0]
        assert int(h["prim"]) == k["prim"], f"{name}: a BVH that returns the other triangle of the tie fails here"
        assert (float(h["t"]), float(h["u"]), float(h["v"])) == (k["t"], k["u"], k["v"]), name


@pytest.mark.parametrize("mode", [rt.TRAVERSAL_BRUTE, rt.TRAVERSAL_BVH2, rt.TRAVERSAL_BVH4Q])
def test_golden_fixtures_on_gpu(renderer, mode):
    renderer.set_option(rt.OPT_TRAVERSAL, mode)
    tris, sph, _ = scenes.cornell()
    renderer.create_data(tris, sph)
    renderer.update_camera(rt.default_camera(1.0))
    assert_parity(renderer.trace_primary(64, 64), np.load(os.path.join(GOLDEN, "c1_cornell_64x64_hits.npy")), "golden 64x64")
    renderer.update_camera(rt.default_camera())
    assert_parity(renderer.trace_primary(61, 37), np.load(os.path.join(GOLDEN, "c1_cornell_61x37_hits.npy")), "golden ragged 61x37")
    t32, s32, _ = scenes.cornell32()
    renderer.create_data(t32, s32)
    renderer.update_camera(rt.default_camera(1.0))
    assert_parity(renderer.trace_primary(64, 64), np.load(os.path.join(GOLDEN, "c1_32_cornell_64x64_hits.npy")), "golden C1-32")
    st, rays = scenes.soup(2000, half_extent=0.05), scenes.random_rays(4096)
    renderer.create_data(st)
    assert_parity(renderer.trace(rays), np.load(os.path.join(GOLDEN, "soup2k_rays4k_hits.npy")), "golden soup")


def test_shadow_and_bounce_stages_vs_golden(renderer, torch_dev):
    torch = torch_dev
    tris, sph, _ = scenes.cornell()
    cam = rt.default_camera(1.0)
    renderer.set_option(rt.OPT_TRAVERSAL, rt.TRAVERSAL_BVH)
    renderer.create_data(tris, sph)
    renderer.update_camera(cam)
    n = 64 * 64
    rays, hits = dev_buf(torch, n, 32), dev_buf(torch, n, 16)
    srays, brays = dev_buf(torch, n, 32), dev_buf(torch, n, 32)
    lit = torch.empty(n, dtype=torch.uint8, device="cuda")
    s = torch.cuda.current_stream()
    renderer.trace_primary_dev(64, 64, hits, rays, stream=s)
    renderer.shadow_dev(rays, hits, n, (0.0, 0.0, 1.9), lit, srays, stream=s)
    renderer.bounce_dev(rays, hits, n, brays, key0=0, seed=7, stream=s)
    torch.cuda.synchronize()
    sc = oracle.Scene(tris, sph)
    assert to_np(rays, rt.RAY).tobytes() == oracle.primary_rays(cam, 64, 64).tobytes()
    assert_parity(to_np(hits, rt.HIT), np.load(os.path.join(GOLDEN, "c1_cornell_64x64_hits.npy")), "primary_dev")
    assert lit.cpu().numpy().tobytes() == np.load(os.path.join(GOLDEN, "c1_cornell_64x64_lit.npy")).tobytes()
    _, sr = oracle.shadow_lit(sc, to_np(rays, rt.RAY), to_np(hits, rt.HIT), (0.0, 0.0, 1.9), want_rays=True)
    assert to_np(srays, rt.RAY).tobytes() == sr.tobytes()
    assert to_np(brays, rt.RAY).tobytes() == np.load(os.path.join(GOLDEN, "c1_cornell_64x64_bounce_rays.npy")).tobytes()
    assert 0 < int((lit == 0).sum()) < n                        # the spheres do cast shadows


def test_jittered_spp_primary_rays_bit_exact(renderer, torch_dev):
    torch = torch_dev
    tris, sph, _ = scenes.cornell()
    cam = rt.default_camera(1.0)
    renderer.create_data(tris, sph)
    renderer.update_camera(cam)
    w, h, spp = 32, 32, 4
    rays, hits = dev_buf(torch, w * h * spp, 32), dev_buf(torch, w * h * spp, 16)
    renderer.trace_primary_dev(w, h, hits, rays, spp=spp, seed=1, jitter=True, stream=torch.cuda.current_stream())
    torch.cuda.synchronize()
    ref_rays = oracle.primary_rays(cam, w, h, spp=spp, seed=1, jitter=True)
    meta = json.load(open(os.path.join(GOLDEN, "meta.json")))
    assert to_np(rays, rt.RAY).tobytes() == ref_rays.tobytes()
    assert sha(to_np(rays, rt.RAY)) == meta["c1_cornell_rays_32x32x4_seed1_sha256"]
    assert_parity(to_np(hits, rt.HIT), oracle.trace(oracle.Scene(tris, sph), ref_rays), "jittered primary hits")


@pytest.mark.parametrize("mode", [rt.TRAVERSAL_BRUTE, rt.TRAVERSAL_BVH])
@pytest.mark.parametrize("w,h,spp", [(37, 21, 1), (37, 21, 2), (29, 13, 6), (21, 10, 10), (33, 9, 8), (19, 7, 5)])
def test_sample_grouping_of_the_tile_kernel(renderer, torch_dev, mode, w, h, spp):
    """A wave of the tile kernel takes 4, 2 or 1 samples of a 4x4, 8x4 or 8x8 pixel tile, whichever divides spp (the
    reference's own SAMPLES is 10): every grouping must give the oracle's rays and hits in (pixel, sample) order, on
    ragged image sizes too."""
    torch = torch_dev
    tris, sph, _ = scenes.cornell()
    cam = rt.default_camera(w / h)
    renderer.set_option(rt.OPT_TRAVERSAL, mode)
    renderer.create_data(tris, sph)
    renderer.update_camera(cam)
    rays, hits = dev_buf(torch, w * h * spp, 32), dev_buf(torch, w * h * spp, 16)
    renderer.trace_primary_dev(w, h, hits, rays, spp=spp, seed=7, jitter=True, stream=torch.cuda.current_stream())
    torch.cuda.synchronize()
    ref_rays = oracle.primary_rays(cam, w, h, spp=spp, seed=7, jitter=True)
    assert to_np(rays, rt.RAY).tobytes() == ref_rays.tobytes()
    assert_parity(to_np(hits, rt.HIT), oracle.trace(oracle.Scene(tris, sph), ref_rays), f"spp {spp}")


# ---- edge cases -----------------------------------------------------------------------------------------
def test_empty_scene_and_empty_ray_buffer(renderer):
    renderer.create_data()
    h = renderer.trace(mk_rays([(0, 0, 0), (1, 2, 3)], [(0, 0, 1), (1, 0, 0)]))
    assert (h["prim"] == rt.MISS).all() and (h["t"] == np.float32(1e20)).all() and (h["u"] == 0).all()
    assert renderer.trace(np.zeros(0, dtype=rt.RAY)).size == 0
    assert renderer.occluded(np.zeros(0, dtype=rt.RAY)).size == 0
    renderer.update_camera(rt.default_camera())
    assert (renderer.trace_primary(9, 5)["prim"] == rt.MISS).all()


def test_spheres_only_and_sphere_tie_rule(renderer):
    sph = np.zeros(2, dtype=rt.SPHERE)
    sph[0]["center"], sph[0]["radius"] = (0, 0, 0), 1.0
    sph[1]["center"], sph[1]["radius"] = (0, 0, 0), 1.0          # identical twin: the lower index wins the tie
    tri = np.zeros(1, dtype=rt.TRIANGLE)
    tri[0]["v0"]["position"], tri[0]["v1"]["position"], tri[0]["v2"]["position"] = (-4, -4, 1), (4, -4, 1), (0, 8, 1)
    rays = mk_rays([(0, 0, 3), (0, 0, 0), (5, 5, 5)], [(0, 0, -1), (0, 0, 1), (0, 0, 1)])
    for mode in (rt.TRAVERSAL_BRUTE, rt.TRAVERSAL_BVH2, rt.TRAVERSAL_BVH4Q):
        renderer.set_option(rt.OPT_TRAVERSAL, mode)
        renderer.create_data(None, sph)
        assert_parity(renderer.trace(rays), oracle.trace(oracle.Scene(None, sph), rays), "spheres only")
        renderer.create_data(tri, sph)
        got = renderer.trace(rays)
        assert_parity(got, oracle.trace(oracle.Scene(tri, sph), rays), "sphere vs triangle tie")
        assert int(got[0]["prim"]) == rt.SPHERE_BIT and float(got[0]["t"]) == 2.0


def test_intervals_degenerates_and_axis_aligned_rays(renderer):
    tris, sph, _ = scenes.cornell()
    z = np.zeros(3, dtype=rt.TRIANGLE)                            # zero triangles (VkRenderer.cpp:262-270 placeholder)
    tris = scenes.cat([tris, z])
    sc = oracle.Scene(tris, sph)
    o = [(0, 4, 1), (0, 4, 1), (0, 4, 1), (0.5, 1, 1), (0, 0, 1), (2, 0, 1), (-2, -2, 0), (0, 0, 1), (0, 0, 1), (100, 100, 100)]
    d = [(0, -1, 0), (0, -1, 0), (0, -1, 0), (0, 0, -1), (0, 0, 0), (1, 0, 0), (1, 1, 1), (1, 0, 0), (0, 0, 1), (-1, -1, -1)]
    rays = mk_rays(o, d)
    rays["tmax"][1] = 6.0           # hit exactly at tmax is rejected
    rays["tmin"][2] = 6.5           # interval beyond the hit
    rays["tmax"][3] = 0.0           # empty interval
    for mode in (rt.TRAVERSAL_BRUTE, rt.TRAVERSAL_BVH2, rt.TRAVERSAL_BVH4Q):
        renderer.set_option(rt.OPT_TRAVERSAL, mode)
        renderer.create_data(tris, sph)
        assert_parity(renderer.trace(rays), oracle.trace(sc, rays), f"edge rays mode {mode}")
        assert np.array_equal(renderer.occluded(rays), oracle.occluded(sc, rays))


def special_value_rays(n, seed):
    """Random rays in which a third of the records carry IEEE special values (signed zeros, NaN, infinities, denormals,
    the largest and smallest normals) in random fields: nothing at the boundary validates ray buffers, so every path
    has to give the reference loop's answer (mostly: no hit) for them, and must not hang."""
    rng = np.random.default_rng(seed)
    rays = scenes.random_rays(n, seed=seed)
    pool = np.array([0.0, -0.0, np.nan, np.inf, -np.inf, 1e-45, -1e-45, 1.1754944e-38, 3.4028235e38, -3.4028235e38,
                     1e-20, -1e-20, 1e20, 1.0, -1.0], dtype=np.float32)
    flat = rays.view(np.float32).reshape(n, 8)                     # o.xyz tmin d.xyz tmax
    for _ in range(2):
        rows = rng.choice(n, n // 5, replace=False)
        flat[rows, rng.integers(0, 8, rows.size)] = pool[rng.integers(0, pool.size, rows.size)]
    rows = rng.choice(n, n // 50, replace=False)                   # axis-parallel with signed zeros, from inside
    flat[rows, 4:7] = 0.0
    flat[rows, 4 + rng.integers(0, 3, rows.size)] = rng.choice(np.float32([1, -1]), rows.size)
    flat[rows[::2], 4 + rng.integers(0, 3, rows[::2].size)] *= np.float32(-1.0)   # 0 -> -0 (or flips the axis)
    return rays


def test_special_value_rays_in_every_path(renderer, torch_dev):
    n = 24000 + 13
    st, rays = scenes.soup(6000, half_extent=0.04), special_value_rays(n, 0xBAD5EED)
    sph = np.zeros(2, dtype=rt.SPHERE); sph["center"] = [(0.3, 0.3, 0.3), (0.7, 0.6, 0.5)]; sph["radius"] = [0.05, 0.08]
    sc = oracle.Scene(st, sph)
    ref, ref_occ = oracle.trace(sc, rays), oracle.occluded(sc, rays)
    hit = ref["prim"] != rt.MISS
    assert hit.sum() > n // 4 and (~hit).sum() > n // 4
    d_rays = to_dev(torch_dev, rays)
    for mode in (rt.TRAVERSAL_BRUTE, rt.TRAVERSAL_BVH2, rt.TRAVERSAL_BVH4Q):
        renderer.set_option(rt.OPT_TRAVERSAL, mode)
        renderer.create_data(st, sph)
        for binning, packets in ((0, 0), (1, 0), (0, 1)):
            renderer.set_option(rt.OPT_RAY_BINNING, binning)
            renderer.set_option(rt.OPT_RAY_PACKETS, packets)
            hits, occ = dev_buf(torch_dev, n, 16), torch_dev.zeros(n, dtype=torch_dev.uint8, device="cuda")
            renderer.trace_dev(d_rays, n, hits)
            renderer.occluded_dev(d_rays, n, occ)
            torch_dev.cuda.synchronize()
            what = f"special-value rays mode {mode} binning {binning} packets {packets}"
            assert_parity(to_np(hits, rt.HIT), ref, what)
            assert np.array_equal(occ.cpu().numpy(), ref_occ), what
    renderer.set_option(rt.OPT_RAY_BINNING, 2)
    renderer.set_option(rt.OPT_RAY_PACKETS, 0)


@pytest.mark.parametrize("scale,offset", [(1e3, 0.0), (1e6, 3e6), (1e12, -7e12), (2.0 ** -4, 40.0)])
def test_scaled_and_shifted_scenes_keep_parity(renderer, scale, offset):
    """The same soup in other units and away from the world origin: node grids, per-ray pads and the binning grid all
    derive from the scene bounds, so nothing may depend on coordinates being of order one.  (Below ~1e-2 the reference's
    absolute 1e-6 determinant threshold rejects every triangle, above ~1e15 its products overflow: both ends give empty
    images in the reference too and are not interesting.)"""
    n = 30000
    st, rays = scenes.soup(8000, half_extent=0.04), scenes.random_rays(n)
    rays["o"][: n // 4] = rays["o"][: n // 4] * 6 - 2.5
    s, off = np.float32(scale), np.float32(offset)
    for v in ("v0", "v1", "v2"):
        st[v]["position"] = st[v]["position"] * s + off
    rays["o"] = rays["o"] * s + off
    rays["tmax"] = 1e20 if scale < 1e6 else 3e38
    ref = oracle.trace(oracle.Scene(st), rays)
    assert (ref["prim"] != rt.MISS).mean() > 0.1
    for mode in (rt.TRAVERSAL_BRUTE, rt.TRAVERSAL_BVH2, rt.TRAVERSAL_BVH4Q):
        renderer.set_option(rt.OPT_TRAVERSAL, mode)
        renderer.create_data(st)
        for binning, packets in ((0, 0), (1, 0), (0, 1)):
            renderer.set_option(rt.OPT_RAY_BINNING, binning)
            renderer.set_option(rt.OPT_RAY_PACKETS, packets)
            assert_parity(renderer.trace(rays), ref, f"scale {scale} offset {offset} mode {mode} binning {binning} packets {packets}")
    renderer.set_option(rt.OPT_RAY_BINNING, 2)
    renderer.set_option(rt.OPT_RAY_PACKETS, 0)


def test_error_conventions(renderer):
    with pytest.raises(RuntimeError, match="no scene"):
        renderer.trace(mk_rays((0, 0, 0), (0, 0, 1)))
    renderer.create_data(*scenes.cornell()[:2])
    with pytest.raises(RuntimeError, match="no camera"):
        renderer.trace_primary(8, 8)
    with pytest.raises(RuntimeError):
        renderer.set_option(99, 0)
    bad = np.zeros(4, dtype=rt.TRIANGLE)
    bad[1]["v0"]["position"] = (np.nan, 0, 0)
    with pytest.raises(RuntimeError, match="non-finite"):
        renderer.create_data(bad)
    with pytest.raises(RuntimeError):
        rt.Renderer((99,))


# ---- random soups: BVH vs the reference loop ---------------------------------------------------------------
@pytest.mark.parametrize("n_tris,half,n_rays,leaf", [(1, 0.3, 2000, 4), (5, 0.3, 2000, 4), (20000, 0.02, 60000, 4),
                                                     (20000, 0.02, 30000, 1), (3000, 0.2, 30000, 8)])
def test_soup_bvh_and_brute_vs_oracle(renderer, n_tris, half, n_rays, leaf):
    st, rays = scenes.soup(n_tris, half_extent=half), scenes.random_rays(n_rays)
    rays["o"][: n_rays // 4] = rays["o"][: n_rays // 4] * 4 - 2          # a quarter start outside the unit cube
    ref = oracle.trace(oracle.Scene(st), rays)
    assert (ref["prim"] != rt.MISS).any()
    renderer.set_option(rt.OPT_MAX_LEAF, leaf)
    for mode in (rt.TRAVERSAL_BRUTE, rt.TRAVERSAL_BVH2, rt.TRAVERSAL_BVH4Q):
        renderer.set_option(rt.OPT_TRAVERSAL, mode)
        stats = renderer.create_data(st)
        assert stats["max_depth"] <= 31
        assert_parity(renderer.trace(rays), ref, f"soup {n_tris} mode {mode} leaf {leaf}")
        assert np.array_equal(renderer.occluded(rays), (ref["prim"] != rt.MISS).astype(np.uint8))


def test_ray_binning_changes_the_order_not_the_answer(renderer, torch_dev):
    """RT_OPT_RAY_BINNING hands rays out grouped by origin cell; every hit must land in its own slot, bit for bit,
    including dead rays (tmax <= tmin), rays that start outside the scene bounds, any-hit, and a ragged count."""
    n = (1 << 18) + 4321
    st, rays = scenes.soup(30000, half_extent=0.02), scenes.random_rays(n)
    rays["o"][::7] = rays["o"][::7] * 4 - 2                  # outside the grid: clamped to border cells
    rays["tmax"][3::11] = 0.05                               # short intervals
    rays["tmax"][::5] = 0.0                                  # dead rays: their own bin
    renderer.set_option(rt.OPT_TRAVERSAL, rt.TRAVERSAL_BVH)
    renderer.create_data(st)
    d_rays = to_dev(torch_dev, rays)
    out = {}
    for binning in (0, 1):
        renderer.set_option(rt.OPT_RAY_BINNING, binning)
        hits, occ = dev_buf(torch_dev, n, 16), torch_dev.zeros(n, dtype=torch_dev.uint8, device="cuda")
        renderer.trace_dev(d_rays, n, hits)
        renderer.occluded_dev(d_rays, n, occ)
        torch_dev.cuda.synchronize()
        out[binning] = (to_np(hits, rt.HIT).copy(), occ.cpu().numpy().copy())
    assert out[0][0].tobytes() == out[1][0].tobytes()
    assert out[0][1].tobytes() == out[1][1].tobytes()
    assert (out[1][0]["prim"][::5] == rt.MISS).all() and (out[1][0]["prim"] != rt.MISS).any()
    idx = np.sort(np.random.default_rng(3).choice(n, 4096, replace=False))
    assert_parity(out[1][0][idx], oracle.trace(oracle.Scene(st), rays[idx]), "binned soup sample vs oracle")
    # the host-pointer entry point goes through the same path (4 Mi-ray pieces)
    assert renderer.trace(rays).tobytes() == out[0][0].tobytes()


def test_two_level_binning_of_large_buffers(renderer, torch_dev):
    """Buffers of 2^22 rays and more are binned in two levels (upper bits of the cell first).  Same contract as above:
    the order changes, no record does -- for shuffled rays and for a buffer sorted by x (very uneven first-level runs)."""
    n = (1 << 22) + 1237
    st, rays = scenes.soup(30000, half_extent=0.02), scenes.random_rays(n)
    rays["o"][::9] = rays["o"][::9] * 4 - 2
    rays["tmax"][::6] = 0.0
    rays["tmax"][1::17] = 0.03
    renderer.set_option(rt.OPT_TRAVERSAL, rt.TRAVERSAL_BVH)
    renderer.create_data(st)
    for label, r_np in (("shuffled", rays), ("sorted by x", rays[np.argsort(rays["o"][:, 0], kind="stable")])):
        d_rays = to_dev(torch_dev, r_np)
        out = {}
        for binning in (0, 1):
            renderer.set_option(rt.OPT_RAY_BINNING, binning)
            hits = dev_buf(torch_dev, n, 16)
            renderer.trace_dev(d_rays, n, hits)
            torch_dev.cuda.synchronize()
            out[binning] = to_np(hits, rt.HIT).copy()
        assert out[0].tobytes() == out[1].tobytes(), label
        dead = ~(r_np["tmax"] > r_np["tmin"])
        assert (out[1]["prim"][dead] == rt.MISS).all() and (out[1]["prim"] != rt.MISS).any()
        idx = np.sort(np.random.default_rng(4).choice(n, 2048, replace=False))
        assert_parity(out[1][idx], oracle.trace(oracle.Scene(st), r_np[idx]), f"two-level binned sample vs oracle ({label})")
    renderer.set_option(rt.OPT_RAY_BINNING, 2)


def test_ray_buffer_as_packets_gives_the_same_records(renderer, torch_dev):
    """RT_OPT_RAY_PACKETS: 64 consecutive rays of a buffer walk the tree as one packet.  Whatever the rays -- here random
    ones, the worst case for speed, with dead and short rays and a ragged count -- the hit records are the per-ray ones."""
    n = 20000 + 37
    st, rays = scenes.soup(20000, half_extent=0.03), scenes.random_rays(n)
    rays["tmax"][::5] = 0.0
    rays["tmax"][3::11] = 0.05
    sph = np.zeros(2, dtype=rt.SPHERE); sph["center"] = [(0.3, 0.3, 0.3), (0.7, 0.6, 0.5)]; sph["radius"] = [0.05, 0.08]
    renderer.set_option(rt.OPT_TRAVERSAL, rt.TRAVERSAL_BVH4Q)
    renderer.create_data(st, sph)
    d_rays = to_dev(torch_dev, rays)
    out = {}
    for packets in (0, 1):
        renderer.set_option(rt.OPT_RAY_PACKETS, packets)
        hits = dev_buf(torch_dev, n, 16)
        renderer.trace_dev(d_rays, n, hits)
        torch_dev.cuda.synchronize()
        out[packets] = to_np(hits, rt.HIT).copy()
    assert out[0].tobytes() == out[1].tobytes()
    idx = np.sort(np.random.default_rng(8).choice(n, 1024, replace=False))
    assert_parity(out[1][idx], oracle.trace(oracle.Scene(st, sph), rays[idx]), "packets of random rays vs oracle")
    renderer.set_option(rt.OPT_RAY_PACKETS, 0)


def test_shadow_rays_as_packets_give_the_same_flags(renderer, torch_dev):
    """rt_shadow_dev under RT_OPT_RAY_PACKETS: lit flags and shadow rays of the packet walk equal the per-lane ones, for the
    coherent case it is meant for (hits of primary rays in pixel order, ragged frame) and for shuffled hits."""
    tris, sph, lights, cam = scenes.bunny_class()
    w, h = 123, 77
    n = w * h
    renderer.set_option(rt.OPT_TRAVERSAL, rt.TRAVERSAL_BVH4Q)
    renderer.create_data(tris, sph, lights); renderer.update_camera(cam)
    rays, hits = dev_buf(torch_dev, n, 32), dev_buf(torch_dev, n, 16)
    renderer.trace_primary_dev(w, h, hits, rays)
    torch_dev.cuda.synchronize()
    lp = tuple(float(x) for x in lights[0]["position"])
    perm = torch_dev.randperm(n, device="cuda")
    for label, rr, hh in (("pixel order", rays, hits),
                          ("shuffled", rays.view(-1, 8)[perm].contiguous().view(-1), hits.view(-1, 4)[perm].contiguous().view(-1))):
        out = {}
        for packets in (0, 1):
            renderer.set_option(rt.OPT_RAY_PACKETS, packets)
            lit = torch_dev.zeros(n, dtype=torch_dev.uint8, device="cuda")
            sr = dev_buf(torch_dev, n, 32)
            renderer.shadow_dev(rr, hh, n, lp, lit, sr)
            torch_dev.cuda.synchronize()
            out[packets] = (lit.cpu().numpy().copy(), sr.cpu().numpy().copy())
        assert np.array_equal(out[0][0], out[1][0]), label
        assert out[0][1].tobytes() == out[1][1].tobytes(), label
        assert (out[1][0] == 1).any() and (out[1][0] == 0).any() and (out[1][0] == 2).any()
    renderer.set_option(rt.OPT_RAY_PACKETS, 0)


def test_capped_lds_stack_with_spill_changes_nothing(renderer, torch_dev):
    """RT_OPT_LDS_STACK: the ray-buffer kernel keeps only the first entries of a lane's stack in LDS and spills deeper ones
    to HBM.  Caps 2 and 3 force nearly every push through the spill path; every cap must give the uncapped records."""
    n = 150001
    st, rays = scenes.soup(40000, half_extent=0.03), scenes.random_rays(n)
    rays["tmax"][::13] = 0.2
    renderer.set_option(rt.OPT_TRAVERSAL, rt.TRAVERSAL_BVH4Q)
    renderer.create_data(st)
    d_rays = to_dev(torch_dev, rays)
    ref = None
    for cap in (0, 2, 3, 7, 16, 64):
        renderer.set_option(rt.OPT_LDS_STACK, cap)
        hits, occ = dev_buf(torch_dev, n, 16), torch_dev.zeros(n, dtype=torch_dev.uint8, device="cuda")
        renderer.trace_dev(d_rays, n, hits)
        renderer.occluded_dev(d_rays, n, occ)
        torch_dev.cuda.synchronize()
        got = (to_np(hits, rt.HIT).tobytes(), occ.cpu().numpy().tobytes())
        if ref is None:
            ref = got
            idx = np.sort(np.random.default_rng(5).choice(n, 2048, replace=False))
            assert_parity(to_np(hits, rt.HIT)[idx], oracle.trace(oracle.Scene(st), rays[idx]), "uncapped soup sample vs oracle")
        assert got == ref, f"cap {cap}"
    with pytest.raises(RuntimeError):
        renderer.set_option(rt.OPT_LDS_STACK, 1)


def test_staged_ray_queue_changes_nothing(renderer, torch_dev):
    """RT_OPT_RAY_QUEUE: idle lanes of the ray-buffer kernel start their next ray from a queue of prepared rays in LDS (filled by
    half the wave) instead of loading and preparing it themselves.  Closest-hit records and any-hit flags must be the ones of the
    kernel without the queue -- with dead rays (empty intervals) in the buffer, with spheres in the scene, with the hand-out order
    permuted by ray binning, with the capped and the uncapped stack, for every refill threshold, and for a buffer smaller than
    one wave."""
    tris, sph, _ = scenes.cornell()
    soup = scenes.soup(40000, half_extent=0.03)
    for st, spheres, n in ((soup, None, 300007), (tris, sph, 70001), (soup, None, 37)):
        rays = scenes.random_rays(n)
        rays["tmax"][::13] = 0.2
        rays["tmax"][5::17] = rays["tmin"][5::17]                     # dead rays: handed out, written as misses, never traced
        renderer.set_option(rt.OPT_TRAVERSAL, rt.TRAVERSAL_BVH4Q)
        renderer.create_data(st, spheres)
        d_rays = to_dev(torch_dev, rays)

        def run():
            hits, occ = dev_buf(torch_dev, n, 16), torch_dev.zeros(n, dtype=torch_dev.uint8, device="cuda")
            renderer.trace_dev(d_rays, n, hits)
            renderer.occluded_dev(d_rays, n, occ)
            torch_dev.cuda.synchronize()
            return to_np(hits, rt.HIT).tobytes(), occ.cpu().numpy().tobytes()

        renderer.set_option(rt.OPT_RAY_QUEUE, 0)
        renderer.set_option(rt.OPT_RAY_BINNING, 0)
        ref = run()
        k = min(n, 1024)
        assert_parity(np.frombuffer(ref[0], dtype=rt.HIT)[:k], oracle.trace(oracle.Scene(st, spheres), rays[:k]), "no queue vs oracle")
        renderer.set_option(rt.OPT_RAY_QUEUE, 1)
        for binning, cap, refill in ((0, 20, 16 | (12 << 8)), (0, 20, 1 | (0 << 8)), (0, 0, 6 | (1 << 8)), (0, 3, 64 | (64 << 8)),
                                     (1, 20, 16 | (12 << 8)), (1, 2, 4 | (1 << 8))):
            renderer.set_option(rt.OPT_RAY_BINNING, binning)
            renderer.set_option(rt.OPT_LDS_STACK, cap)
            renderer.set_option(rt.OPT_REFILL, refill)
            assert run() == ref, f"queue on, binning {binning}, cap {cap}, refill {refill:#x}, {n} rays"
        renderer.set_option(rt.OPT_LDS_STACK, 20)
        renderer.set_option(rt.OPT_REFILL, 16 | (12 << 8))
        renderer.set_option(rt.OPT_RAY_BINNING, 0)
        for strip in (64, 1 << 14):                                   # chunks of one wave's worth, and chunks larger than small buffers
            renderer.set_option(rt.OPT_STRIP, strip)
            assert run() == ref, f"queue on, chunks of {strip} rays, {n} rays"
        renderer.set_option(rt.OPT_STRIP, 256)
        renderer.set_option(rt.OPT_RAY_QUEUE, 2)                      # auto: on for this launch (nothing is binned at this size)
        renderer.set_option(rt.OPT_RAY_BINNING, 2)
        assert run() == ref, "queue auto"
    with pytest.raises(RuntimeError):
        renderer.set_option(rt.OPT_RAY_QUEUE, 3)


def test_two_slots_on_one_device_equal_one_slot():
    st, rays = scenes.soup(5000, half_extent=0.05), scenes.random_rays(10001)
    tris, sph, _ = scenes.cornell()
    with rt.Renderer((0,)) as a, rt.Renderer((0, 0, 0)) as b:
        assert b.num_slots == 3
        for r in (a, b):
            r.set_option(rt.OPT_TRAVERSAL, rt.TRAVERSAL_BVH)
            r.create_data(st)
        assert a.trace(rays).tobytes() == b.trace(rays).tobytes()
        assert a.occluded(rays).tobytes() == b.occluded(rays).tobytes()
        for r in (a, b):
            r.create_data(tris, sph)
            r.update_camera(rt.default_camera())
        assert a.trace_primary(61, 37).tobytes() == b.trace_primary(61, 37).tobytes()


@pytest.mark.parametrize("w,h", [(61, 37), (96, 72), (40, 8), (33, 3)])
def test_frames_of_several_slots_are_interleaved_shares_gathered_on_slot_0(w, h):
    """rt_trace_primary / rt_render on a context of several slots: 8-row stripes dealt round-robin, gathered on slot 0's
    device, one D2H.  Slots that repeat an ordinal take the peer-copy exchange; the frame must equal the
    one-slot frame bit for bit (RNG keys are path indices in the full frame), and the oracle's."""
    tris, sph, _ = scenes.cornell()
    cam = rt.default_camera(w / h)
    _, want = oracle.render(oracle.Scene(tris, sph, _lights2()), cam, w, h, 4, 3, 7)
    ref_hits = oracle.trace_primary(oracle.Scene(tris, sph), cam, w, h)
    frames = []
    for devs in ((0,), (0, 0), (0, 0, 0), (0,) * 5):
        with rt.Renderer(devs) as r:
            r.set_option(rt.OPT_TRAVERSAL, rt.TRAVERSAL_BVH)
            r.create_data(tris, sph, _lights2())
            r.update_camera(cam)
            hits = r.trace_primary(w, h)
            assert hits.tobytes() == ref_hits.tobytes(), devs
            frames.append(r.render(w, h, spp=4, bounces=3, seed=7))
            frames.append(r.render(w, h, spp=3, bounces=2, seed=7))            # spp not a power of two: separate resolve
    for k in range(2, len(frames)):
        assert np.array_equal(frames[k], frames[k % 2]), f"slots variant {k // 2} differs from one slot"
    assert np.abs(frames[0].astype(np.int32) - want.astype(np.int32)).max() <= 1


@pytest.mark.parametrize("primary", [1, 2, 3])
def test_host_frames_on_every_primary_path_and_slot_count(primary):
    """rt_trace_primary / rt_render with RT_OPT_PRIMARY = 1 (tile kernel), 2 (ray generation + ray-buffer kernel, also behind
    the host entry points since round 3) and 3 (packet walk), on one slot and on three: the oracle's hit buffer, one frame."""
    tris, sph, _ = scenes.cornell32()[0], None, None
    w, h = 75, 43
    cam = rt.default_camera(w / h)
    ref = oracle.trace_primary(oracle.Scene(tris, None), cam, w, h)
    frames = []
    for devs in ((0,), (0, 0, 0)):
        with rt.Renderer(devs) as r:
            r.set_option(rt.OPT_TRAVERSAL, rt.TRAVERSAL_BVH)
            r.set_option(rt.OPT_PRIMARY, primary)
            r.create_data(tris, None, _lights2())
            r.update_camera(cam)
            assert r.trace_primary(w, h).tobytes() == ref.tobytes(), (primary, devs)
            frames.append(r.render(w, h, spp=2, bounces=3, seed=11))
    assert np.array_equal(frames[0], frames[1])


def test_rccl_frame_exchange_inside_the_library():
    """RT_OPT_FRAME_EXCHANGE = 1: the gather goes through RCCL (librccl.so.1 loaded by the library, ncclCommInitAll, grouped
    ncclSend / ncclRecv on the slots' streams).  One GPU here, so the communicator has ONE rank and the slot sends its
    share to itself before the frame is assembled from the received copy; N > 1 ranks in one process are unexecuted.
    Repeated ordinals cannot form a communicator: a clear error, and auto (0) falls back to peer copies."""
    tris, sph, _ = scenes.cornell()
    w, h = 96, 52
    cam = rt.default_camera(w / h)
    with rt.Renderer((0,)) as a, rt.Renderer((0,)) as b, rt.Renderer((0, 0)) as two:
        for r in (a, b, two):
            r.set_option(rt.OPT_TRAVERSAL, rt.TRAVERSAL_BVH)
            r.create_data(tris, sph, _lights2())
            r.update_camera(cam)
        b.set_option(rt.OPT_FRAME_EXCHANGE, 1)
        for _ in range(2):                                                    # second call: communicator and staging reused
            assert np.array_equal(a.render(w, h, spp=4, bounces=3, seed=5), b.render(w, h, spp=4, bounces=3, seed=5))
            assert a.trace_primary(w, h).tobytes() == b.trace_primary(w, h).tobytes()
        assert b.frame_exchange_used == 1 and a.frame_exchange_used == 0      # RCCL really carried b's frames
        two.set_option(rt.OPT_FRAME_EXCHANGE, 1)
        with pytest.raises(RuntimeError, match="distinct device ordinals"):
            two.render(w, h, spp=4, bounces=3, seed=5)
        two.set_option(rt.OPT_FRAME_EXCHANGE, 0)
        assert np.array_equal(two.render(w, h, spp=4, bounces=3, seed=5), a.render(w, h, spp=4, bounces=3, seed=5))
        assert two.frame_exchange_used == 2                                   # repeated ordinals: peer copies
        with pytest.raises(RuntimeError):
            two.set_option(rt.OPT_FRAME_EXCHANGE, 3)


# ---- C2 / C3 stand-ins: sampled against the oracle, full frames against the GPU brute-force loop --------------
def test_c2_bunny_class_primary_and_shadow_sample(renderer, torch_dev):
    torch = torch_dev
    tris, sph, lights, cam = scenes.bunny_class()
    renderer.set_option(rt.OPT_TRAVERSAL, rt.TRAVERSAL_BVH)
    renderer.create_data(tris, sph, lights)
    renderer.update_camera(cam)
    w, h = 1920, 1080
    n = w * h
    rays, hits = dev_buf(torch, n, 32), dev_buf(torch, n, 16)
    lit = torch.empty(n, dtype=torch.uint8, device="cuda")
    s = torch.cuda.current_stream()
    renderer.trace_primary_dev(w, h, hits, rays, stream=s)
    renderer.shadow_dev(rays, hits, n, tuple(lights[0]["position"]), lit, stream=s)
    torch.cuda.synchronize()
    hr, hh, hl = to_np(rays, rt.RAY), to_np(hits, rt.HIT), lit.cpu().numpy()
    idx = np.random.default_rng(2).choice(n, 12000, replace=False)
    sc = oracle.Scene(tris, sph)
    assert hr[idx].tobytes() == oracle.primary_rays(cam, w, h)[idx].tobytes()
    assert_parity(hh[idx], oracle.trace(sc, hr[idx]), "C2 primary sample")
    assert np.array_equal(hl[idx], oracle.shadow_lit(sc, hr[idx], hh[idx], tuple(lights[0]["position"])))
    assert set(np.unique(hl)) == {0, 1, 2}                     # lit, shadowed and sky pixels all occur


def test_c3_sponza_class_primary_bounce_sample(renderer, torch_dev):
    torch = torch_dev
    tris, sph, _, cam = scenes.sponza_class()
    renderer.set_option(rt.OPT_TRAVERSAL, rt.TRAVERSAL_BVH)
    renderer.create_data(tris, sph)
    renderer.update_camera(cam)
    w, h, spp = 960, 540, 4
    n = w * h * spp
    rays, hits, brays, bhits = dev_buf(torch, n, 32), dev_buf(torch, n, 16), dev_buf(torch, n, 32), dev_buf(torch, n, 16)
    s = torch.cuda.current_stream()
    renderer.trace_primary_dev(w, h, hits, rays, spp=spp, seed=1, jitter=True, stream=s)
    renderer.bounce_dev(rays, hits, n, brays, key0=0, seed=1, stream=s)
    renderer.trace_dev(brays, n, bhits, stream=s)
    torch.cuda.synchronize()
    hr, hh, br, bh = to_np(rays, rt.RAY), to_np(hits, rt.HIT), to_np(brays, rt.RAY), to_np(bhits, rt.HIT)
    sc = oracle.Scene(tris, sph)
    idx = np.sort(np.random.default_rng(3).choice(n, 6000, replace=False))
    assert_parity(hh[idx], oracle.trace(sc, hr[idx]), "C3 primary sample")
    assert_parity(bh[idx], oracle.trace(sc, br[idx]), "C3 bounce sample")
    # bounce generation: the oracle keys its RNG on the buffer index, so compare a contiguous head
    head = 5000
    assert br[:head].tobytes() == oracle.bounce(sc, hr[:head], hh[:head], seed=1).tobytes()
    assert (hh["prim"] != rt.MISS).all() and (bh["prim"] != rt.MISS).mean() > 0.99     # closed room


def test_full_frame_bvh_equals_reference_loop_on_gpu(renderer, torch_dev):
    """Size-independent property at a size the CPU oracle cannot reach: the BVH kernel must return exactly
    what the reference's brute-force loop (frag.glsl:282-298) returns, here run as the brute-force HIP
    kernel (itself pinned bit-exactly to the oracle by the tests above).  1.0 M rays x 259 k triangles."""
    torch = torch_dev
    tris, sph, _, cam = scenes.sponza_class()
    renderer.update_camera(cam)
    w, h = 1280, 720
    n = w * h
    out = {}
    for mode in (rt.TRAVERSAL_BVH4Q, rt.TRAVERSAL_BVH2, rt.TRAVERSAL_BRUTE):
        renderer.set_option(rt.OPT_TRAVERSAL, mode)
        renderer.create_data(tris, sph)
        rays, hits, brays, bhits = dev_buf(torch, n, 32), dev_buf(torch, n, 16), dev_buf(torch, n, 32), dev_buf(torch, n, 16)
        s = torch.cuda.current_stream()
        renderer.trace_primary_dev(w, h, hits, rays, seed=5, jitter=True, stream=s)
        renderer.bounce_dev(rays, hits, n, brays, seed=5, stream=s)
        renderer.trace_dev(brays, n, bhits, stream=s)
        torch.cuda.synchronize()
        out[mode] = (to_np(hits, rt.HIT).copy(), to_np(bhits, rt.HIT).copy())
    for m in (rt.TRAVERSAL_BVH4Q, rt.TRAVERSAL_BVH2):
        assert_parity(out[m][0], out[rt.TRAVERSAL_BRUTE][0], f"full-frame primary: layout {m} vs reference loop")
        assert_parity(out[m][1], out[rt.TRAVERSAL_BRUTE][1], f"full-frame bounce: layout {m} vs reference loop")


def test_counters_and_idempotence(renderer, torch_dev):
    torch = torch_dev
    st = scenes.soup(20000, half_extent=0.02)
    rays = scenes.random_rays(50000)
    renderer.set_option(rt.OPT_TRAVERSAL, rt.TRAVERSAL_BVH)
    renderer.create_data(st)
    d_rays, d_hits = to_dev(torch, rays), dev_buf(torch, len(rays), 16)
    s = torch.cuda.current_stream()
    renderer.trace_dev(d_rays, len(rays), d_hits, stream=s)
    torch.cuda.synchronize()
    first = to_np(d_hits, rt.HIT).copy()
    renderer.set_option(rt.OPT_COUNTERS, 1)
    renderer.counters()
    renderer.trace_dev(d_rays, len(rays), d_hits, stream=s)
    torch.cuda.synchronize()
    c = renderer.counters()
    assert to_np(d_hits, rt.HIT).tobytes() == first.tobytes()            # instrumented kernel, same answers
    assert c["rays"] == len(rays) and 2 * c["nodes"] <= c["box_tests"] <= 4 * c["nodes"] and c["tri_tests"] > 0
    assert c["nodes"] / c["rays"] < 200


# ---- multi-GPU sharding, both ranks emulated on the one GPU of this box ------------------------------------------
def test_stripe_sharding_reassembles_the_full_frame(renderer, torch_dev):
    torch = torch_dev
    tris, sph, _ = scenes.cornell()
    cam = rt.default_camera()
    renderer.set_option(rt.OPT_TRAVERSAL, rt.TRAVERSAL_BVH)
    renderer.create_data(tris, sph)
    renderer.update_camera(cam)
    w, h, spp = 200, 150, 2
    s = torch.cuda.current_stream()
    full = dev_buf(torch, w * h * spp, 16)
    renderer.trace_primary_dev(w, h, full, spp=spp, seed=3, jitter=True, stream=s)
    for world, sh in ((2, 16), (3, 8)):
        parts = []
        for rank in range(world):
            p = rd.stripe_plan(h, world, rank, sh)
            loc = torch.zeros(p.padded_rows * w * spp * 4, dtype=torch.float32, device="cuda")
            renderer.trace_primary_stripes_dev(w, h, p.y0, p.padded_rows, sh, world, loc, spp=spp, seed=3, jitter=True, stream=s)
            parts.append(loc.view(p.padded_rows, w * spp * 4))
        cat = torch.cat(parts, 0)
        img = cat.index_select(0, rd.gather_index(rd.stripe_plan(h, world, 0, sh), "cuda"))
        torch.cuda.synchronize()
        assert to_np(img.reshape(-1), rt.HIT).tobytes() == to_np(full, rt.HIT).tobytes(), (world, sh)


def test_cpp_headless_harness(tmp_path):
    """The C++ host layer (raytracer_amd/host): same ABI, std::runtime_error convention, PPM out."""
    import subprocess
    exe = os.path.join(os.path.dirname(GOLDEN), "..", "raytracer_amd", "host", "rt_headless")
    if not os.path.exists(exe):
        pytest.skip("rt_headless not built")
    out = tmp_path / "f.ppm"
    p = subprocess.run([exe, "64", "48", str(out)], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr
    assert "3072 hits" in p.stdout and "12 triangles" in p.stdout
    data = out.read_bytes()
    assert data.startswith(b"P6\n64 48\n255\n") and len(data) == len(b"P6\n64 48\n255\n") + 64 * 48 * 3
    # whole shader (rt_render) + an OBJ mesh in place of the spheres: the headless presentation hook (f2, f3)
    obj = tmp_path / "tet.obj"
    obj.write_text("v -0.5 0.2 0.5\nv 0.5 0.2 0.5\nv 0 0.2 1.5\nv 0 1.0 1.0\nf 1 2 4\nf 2 3 4\nf 3 1 4\nf 1 3 2\n")
    p = subprocess.run([exe, "64", "48", str(out), "2", "2", str(obj)], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr
    assert "rendered 64x48 at 2 spp, 2 bounces" in p.stdout and "16 triangles" in p.stdout
    lit = int(p.stdout.split(" non-black")[0].split()[-1])
    assert 100 < lit <= 64 * 48              # light patch + surfaces whose one bounce reaches it
    assert len(out.read_bytes()) == len(b"P6\n64 48\n255\n") + 64 * 48 * 3
    # the same frame from an engine of three slots (interleaved stripes gathered on the first device), through the C++ wrapper
    one = out.read_bytes()
    p = subprocess.run([exe, "64", "48", str(out), "2", "2", str(obj)], capture_output=True, text=True, timeout=120,
                       env=dict(os.environ, RT_HEADLESS_DEVICES="0,0,0"))
    assert p.returncode == 0, p.stderr
    assert out.read_bytes() == one
    p = subprocess.run([exe, "64", "48", str(out)], capture_output=True, text=True, timeout=120, env=dict(os.environ, RT_HEADLESS_DEVICES="0,x"))
    assert p.returncode == 1 and "RT_HEADLESS_DEVICES" in p.stderr
    p = subprocess.run([exe, "64", "48", str(out), "2", "2", str(tmp_path / "missing.obj")], capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and "rt_load_obj failed" in p.stderr
    p = subprocess.run([exe, "0", "0", str(out)], capture_output=True, text=True, timeout=120)
    assert p.returncode == 1 and "empty image" in p.stderr          # EXIT_FAILURE + message, as src/main.cpp:8-14


# ---- shading + path loop -----------------------------------------------------------------------
def _lights2():
    l = np.zeros(2, dtype=rt.LIGHT)
    l[0]["position"], l[0]["color"], l[0]["intensity"] = (0.0, 0.0, 1.9), (1.0, 1.0, 1.0), 1.0
    l[1]["position"], l[1]["color"], l[1]["intensity"] = (-1.5, 1.5, 0.6), (1.0, 0.6, 0.3), 0.5
    return l


@pytest.mark.parametrize("with_lights", [False, True])
@pytest.mark.parametrize("mode", [rt.TRAVERSAL_BRUTE, rt.TRAVERSAL_BVH2, rt.TRAVERSAL_BVH4Q])
def test_path_loop_stages_bit_exact(renderer, torch_dev, mode, with_lights):
    """rt_path_init_dev / rt_shade_dev / rt_resolve_dev against orc_shade / orc_resolve: throughput, radiance, next
    rays and the linear framebuffer are bit-identical; RGBA8 (device powf vs libm powf, frag.glsl:358) within 1 LSB."""
    torch = torch_dev
    tris, sph, _ = scenes.cornell()
    lights = _lights2() if with_lights else None
    cam = rt.default_camera(1.0)
    renderer.set_option(rt.OPT_TRAVERSAL, mode)
    renderer.create_data(tris, sph, lights)
    renderer.update_camera(cam)
    w, h, spp, bounces, seed = 40, 24, 3, 4, 9
    n = w * h * spp
    f = lambda k: torch.empty(n * k, dtype=torch.float32, device="cuda")
    ra, rb, hits, thr, rad = f(8), f(8), f(4), f(4), f(4)
    lin = torch.empty(w * h * 4, dtype=torch.float32, device="cuda")
    rgba = torch.empty(w * h * 4, dtype=torch.uint8, device="cuda")
    s = torch.cuda.current_stream()
    renderer.path_init_dev(n, thr, rad, stream=s)
    renderer.trace_primary_dev(w, h, hits, ra, spp=spp, seed=seed, jitter=True, stream=s)
    sc = oracle.Scene(tris, sph, lights)
    o_rays = oracle.primary_rays(cam, w, h, spp=spp, seed=seed, jitter=True)
    o_thr = np.ones((n, 4), dtype=np.float32); o_thr[:, 3] = 0
    o_rad = np.zeros((n, 4), dtype=np.float32)
    for b in range(bounces):
        o_hits = oracle.trace(sc, o_rays)
        torch.cuda.synchronize()
        assert_parity(to_np(hits, rt.HIT), o_hits, f"bounce {b} hits")
        renderer.shade_dev(ra, hits, n, thr, rad, rb, key0=0, seed=seed + 1 + b, stream=s)
        renderer.trace_dev(rb, n, hits, stream=s)
        o_rays = oracle.shade(sc, o_rays, o_hits, o_thr, o_rad, key0=0, seed=seed + 1 + b)
        torch.cuda.synchronize()
        assert to_np(rb, rt.RAY).tobytes() == o_rays.tobytes(), f"bounce {b}: next rays differ"
        assert thr.cpu().numpy().tobytes() == o_thr.tobytes(), f"bounce {b}: throughput differs"
        assert rad.cpu().numpy().tobytes() == o_rad.tobytes(), f"bounce {b}: radiance differs"
        ra, rb = rb, ra
    renderer.resolve_dev(rad, w * h, spp, lin, rgba, stream=s)
    torch.cuda.synchronize()
    o_lin, o_rgba = oracle.resolve(o_rad, w * h, spp)
    assert lin.cpu().numpy().tobytes() == o_lin.tobytes()
    assert np.abs(rgba.cpu().numpy().astype(np.int32).reshape(-1, 4) - o_rgba.astype(np.int32)).max() <= 1
    assert o_rad[:, :3].sum() > 0 and (not with_lights or o_rad[:, :3].mean() > 0.05)


@pytest.mark.parametrize("with_lights", [False, True])
@pytest.mark.parametrize("mode", [rt.TRAVERSAL_BRUTE, rt.TRAVERSAL_BVH2, rt.TRAVERSAL_BVH4Q])
def test_fused_primary_shade_equals_the_three_separate_stages(renderer, torch_dev, mode, with_lights):
    """rt_primary_shade_stripes_dev = rt_path_init_dev + rt_trace_primary_stripes_dev + rt_shade_stripes_dev, bit for bit:
    whole frame, and a striped share with a padding stripe (dead slots must read throughput 1, radiance 0, dead ray)."""
    torch = torch_dev
    tris, sph, _ = scenes.cornell()
    renderer.set_option(rt.OPT_TRAVERSAL, mode)
    renderer.create_data(tris, sph, _lights2() if with_lights else None)
    renderer.update_camera(rt.default_camera(52 / 40))
    w, h, spp, seed = 52, 40, 3, 5
    s = torch.cuda.current_stream()
    for world, rank, sh in ((1, 0, 8), (3, 2, 8), (2, 1, 16)):
        p = rd.stripe_plan(h, world, rank, sh)
        n = p.padded_rows * w * spp
        f = lambda k: torch.full((n * k,), 7.0, dtype=torch.float32, device="cuda")
        ra, hits, thr, rad, nxt = f(8), f(4), f(4), f(4), f(8)
        renderer.path_init_dev(n, thr, rad, stream=s)
        renderer.trace_primary_stripes_dev(w, h, p.y0, p.padded_rows, sh, world, hits, ra, spp=spp, seed=seed, jitter=True, stream=s)
        renderer.shade_stripes_dev(ra, hits, n, thr, rad, nxt, p.y0 * w * spp, w * spp, sh, world, seed=seed + 1, stream=s)
        ra2, hits2, thr2, rad2, nxt2 = f(8), f(4), f(4), f(4), f(8)
        renderer.primary_shade_stripes_dev(w, h, p.y0, p.padded_rows, sh, world, thr2, rad2, nxt2, spp=spp, seed=seed,
                                           jitter=True, shade_seed=seed + 1, rays=ra2, hits=hits2, stream=s)
        thr3, rad3 = f(4), f(4)                       # last-segment form: no next rays, no copies of rays / hits
        renderer.primary_shade_stripes_dev(w, h, p.y0, p.padded_rows, sh, world, thr3, rad3, None, spp=spp, seed=seed,
                                           jitter=True, shade_seed=seed + 1, stream=s)
        torch.cuda.synchronize()
        for a, b, what in ((ra, ra2, "rays"), (hits, hits2, "hits"), (thr, thr2, "throughput"), (rad, rad2, "radiance"),
                           (nxt, nxt2, "next rays"), (rad, rad3, "radiance, last-segment form")):
            assert torch.equal(a.view(torch.int32), b.view(torch.int32)), (world, rank, sh, what)
        assert (not with_lights) or float(rad.view(-1, 4)[:, :3].sum()) > 0


def _open_scene_emissive_first():
    """A floor and a floating panel under open sky: most primary rays of the default camera miss.  Triangle 0 is an
    EMITTER, so a lane that only follows the packet and reads primitive 0 as a placeholder would show."""
    t = np.zeros(6, dtype=rt.TRIANGLE)
    quads = [((-0.6, 0.3, 1.6), (0.6, 0.3, 1.6), (0.6, -0.3, 1.6), (-0.6, -0.3, 1.6)),       # panel (emissive), tris 0-1
             ((-2.0, 2.0, 0.0), (2.0, 2.0, 0.0), (2.0, -2.0, 0.0), (-2.0, -2.0, 0.0)),        # floor, tris 2-3
             ((-0.4, -1.0, 0.2), (0.4, -1.0, 0.2), (0.4, -1.0, 1.0), (-0.4, -1.0, 1.0))]      # small wall, tris 4-5
    for q, (a, b, c, d) in enumerate(quads):
        for k, (p0, p1, p2) in enumerate(((a, b, c), (a, c, d))):
            tri = t[2 * q + k]
            e1, e2 = np.subtract(p1, p0), np.subtract(p2, p0)
            nrm = np.cross(e1, e2); nrm = (nrm / np.linalg.norm(nrm)).astype(np.float32)
            for name, p in (("v0", p0), ("v1", p1), ("v2", p2)):
                tri[name]["position"], tri[name]["normal"] = p, nrm
            tri["material"]["albedo"] = (0.8, 0.7, 0.6)
            tri["material"]["roughness"], tri["material"]["metallic"] = 0.6, 0.1
            if q == 0:
                tri["material"]["emission"], tri["material"]["emissionStrength"] = (1.0, 0.9, 0.8), 3.0
    return t


@pytest.mark.parametrize("mode", [rt.TRAVERSAL_BVH2, rt.TRAVERSAL_BVH4Q])
def test_fused_primary_shade_leaves_missing_rays_dark_when_triangle_0_emits(renderer, torch_dev, mode):
    """Open scene, point lights, emissive triangle 0: the fused packet kernel (BVH4Q) keeps its missing lanes in the wave
    for the shadow walk; they must not pick up primitive 0's emission.  Fused = separate stages = oracle, bit for bit."""
    torch = torch_dev
    tris, lights = _open_scene_emissive_first(), _lights2()
    cam = rt.default_camera(48 / 32)
    renderer.set_option(rt.OPT_TRAVERSAL, mode)
    renderer.create_data(tris, None, lights)
    renderer.update_camera(cam)
    w, h, spp, seed = 48, 32, 4, 3
    n = w * h * spp
    s = torch.cuda.current_stream()
    f = lambda k: torch.full((n * k,), 7.0, dtype=torch.float32, device="cuda")
    ra, hits, thr, rad, nxt = f(8), f(4), f(4), f(4), f(8)
    renderer.path_init_dev(n, thr, rad, stream=s)
    renderer.trace_primary_dev(w, h, hits, ra, spp=spp, seed=seed, jitter=True, stream=s)
    renderer.shade_dev(ra, hits, n, thr, rad, nxt, key0=0, seed=seed + 1, stream=s)
    thr2, rad2, nxt2 = f(4), f(4), f(8)
    renderer.primary_shade_stripes_dev(w, h, 0, h, h, 1, thr2, rad2, nxt2, spp=spp, seed=seed, jitter=True,
                                       shade_seed=seed + 1, stream=s)
    torch.cuda.synchronize()
    g_hits = to_np(hits, rt.HIT)
    missed = g_hits["prim"] == rt.MISS
    assert 0.2 < missed.mean() < 0.95, "the scene is meant to be open"
    r2 = rad2.cpu().numpy().reshape(-1, 4)
    assert not r2[missed, :3].any(), "a ray that misses carries no radiance (frag.glsl:351-353)"
    for a, b, what in ((thr, thr2, "throughput"), (rad, rad2, "radiance"), (nxt, nxt2, "next rays")):
        assert torch.equal(a.view(torch.int32), b.view(torch.int32)), what
    sc = oracle.Scene(tris, None, lights)
    o_rays = oracle.primary_rays(cam, w, h, spp=spp, seed=seed, jitter=True)
    o_hits = oracle.trace(sc, o_rays)
    assert_parity(g_hits, o_hits, "primary hits")
    o_thr = np.ones((n, 4), dtype=np.float32); o_thr[:, 3] = 0
    o_rad = np.zeros((n, 4), dtype=np.float32)
    oracle.shade(sc, o_rays, o_hits, o_thr, o_rad, key0=0, seed=seed + 1)
    assert rad2.cpu().numpy().tobytes() == o_rad.tobytes() and o_rad[:, :3].sum() > 0


@pytest.mark.parametrize("with_lights", [False, True])
@pytest.mark.parametrize("spp", [1, 3, 4, 16])
def test_fused_last_segment_resolve_equals_the_two_stages(renderer, torch_dev, spp, with_lights):
    """rt_shade_resolve_dev = rt_shade_dev(out = NULL) + rt_resolve_dev, bit for bit (spp 3: the documented fallback)."""
    torch = torch_dev
    tris, sph, _ = scenes.cornell()
    renderer.set_option(rt.OPT_TRAVERSAL, rt.TRAVERSAL_BVH)
    renderer.create_data(tris, sph, _lights2() if with_lights else None)
    renderer.update_camera(rt.default_camera(1.0))
    w, h, seed = 37, 29, 4                          # 37 * 29 * spp rays: ragged last block
    n = w * h * spp
    f = lambda k: torch.empty(n * k, dtype=torch.float32, device="cuda")
    ra, hits, thr, rad, nxt = f(8), f(4), f(4), f(4), f(8)
    s = torch.cuda.current_stream()
    renderer.primary_shade_stripes_dev(w, h, 0, h, h, 1, thr, rad, nxt, spp=spp, seed=seed, jitter=True, shade_seed=seed + 1, stream=s)
    renderer.trace_dev(nxt, n, hits, stream=s)
    lin_a, lin_b = torch.zeros(w * h * 4, device="cuda"), torch.zeros(w * h * 4, device="cuda")
    px_a, px_b = torch.zeros(w * h, dtype=torch.int32, device="cuda"), torch.zeros(w * h, dtype=torch.int32, device="cuda")
    rad_b = rad.clone()
    renderer.shade_resolve_dev(nxt, hits, n, spp, thr, rad, lin_a, px_a, stream=s)
    renderer.shade_dev(nxt, hits, n, thr, rad_b, None, stream=s)
    renderer.resolve_dev(rad_b, w * h, spp, lin_b, px_b, stream=s)
    torch.cuda.synchronize()
    assert torch.equal(lin_a.view(torch.int32), lin_b.view(torch.int32))
    assert torch.equal(px_a, px_b)
    assert float(lin_a.view(-1, 4)[:, :3].sum()) > 0
    with pytest.raises(RuntimeError):                # n must be a whole number of pixels; spp 0 is rejected
        if spp > 1:
            renderer.shade_resolve_dev(nxt, hits, n - 1, spp, thr, rad, lin_a, px_a)
        else:
            renderer.shade_resolve_dev(nxt, hits, n, 0, thr, rad, lin_a, px_a)


@pytest.mark.parametrize("with_lights", [False, True])
def test_primary_through_the_ray_buffer_kernel_equals_the_tile_kernel(renderer, torch_dev, with_lights):
    """RT_OPT_PRIMARY: 1 = tile kernel, 2 = ray generation + ray-buffer kernel (+ first-segment shade), what big trees
    get automatically.  Rays, hits, throughput, radiance and bounce rays must not differ by a bit; also with a striped
    share that has a padding stripe, and without the optional ray / hit outputs (library scratch)."""
    torch = torch_dev
    tris, sph, _ = scenes.cornell()
    extra = scenes.soup(3000, half_extent=0.05)
    renderer.set_option(rt.OPT_TRAVERSAL, rt.TRAVERSAL_BVH)
    renderer.create_data(scenes.cat([tris, extra]), sph, _lights2() if with_lights else None)
    renderer.update_camera(rt.default_camera(52 / 40))
    w, h, spp, seed = 52, 40, 4, 6
    s = torch.cuda.current_stream()
    for world, rank, sh in ((1, 0, 8), (3, 2, 8)):
        p = rd.stripe_plan(h, world, rank, sh)
        n = p.padded_rows * w * spp
        f = lambda k: torch.full((n * k,), 3.0, dtype=torch.float32, device="cuda")
        got = {}
        for path in (1, 2, 3):
            renderer.set_option(rt.OPT_PRIMARY, path)
            ra, hits, ra2, hits2, thr, rad, nxt, thr3, rad3, nxt3 = f(8), f(4), f(8), f(4), f(4), f(4), f(8), f(4), f(4), f(8)
            renderer.trace_primary_stripes_dev(w, h, p.y0, p.padded_rows, sh, world, hits, ra, spp=spp, seed=seed, jitter=True, stream=s)
            renderer.primary_shade_stripes_dev(w, h, p.y0, p.padded_rows, sh, world, thr, rad, nxt, spp=spp, seed=seed,
                                               jitter=True, shade_seed=seed + 1, rays=ra2, hits=hits2, stream=s)
            renderer.primary_shade_stripes_dev(w, h, p.y0, p.padded_rows, sh, world, thr3, rad3, nxt3, spp=spp, seed=seed,
                                               jitter=True, shade_seed=seed + 1, stream=s)
            hits4 = f(4)
            renderer.trace_primary_stripes_dev(w, h, p.y0, p.padded_rows, sh, world, hits4, None, spp=spp, seed=seed, jitter=True, stream=s)
            torch.cuda.synchronize()
            got[path] = [t.clone() for t in (ra, hits, ra2, hits2, thr, rad, nxt, thr3, rad3, nxt3, hits4)]
        names = ("rays", "hits", "rays (fused call)", "hits (fused call)", "throughput", "radiance", "next rays",
                 "throughput (no copies)", "radiance (no copies)", "next rays (no copies)", "hits (no ray copy)")
        for other in (2, 3):      # 2 = ray generation + ray-buffer kernel, 3 = packet traversal in the tile kernel
            for a, b, what in zip(got[1], got[other], names):
                assert torch.equal(a.view(torch.int32), b.view(torch.int32)), (world, rank, other, what)
        assert torch.equal(got[2][0].view(torch.int32), got[2][2].view(torch.int32)) and torch.equal(got[2][1].view(torch.int32), got[2][10].view(torch.int32))
    renderer.set_option(rt.OPT_PRIMARY, 0)
    with pytest.raises(RuntimeError):
        renderer.set_option(rt.OPT_PRIMARY, 4)


def test_rt_render_matches_oracle_frame():
    """rt_render (host RGBA8, two replicas on the one GPU = the in-process multi-device path) against orc_render;
    the reference's own SAMPLES/BOUNCES are 10/8 (frag.glsl:3-4), here 4/5 on 96x72 to keep the oracle quick."""
    tris, sph, _ = scenes.cornell()
    cam = rt.default_camera(96 / 72)
    _, want = oracle.render(oracle.Scene(tris, sph), cam, 96, 72, 4, 5, 2)
    for devs in ((0,), (0, 0)):
        with rt.Renderer(devs) as r:
            r.create_data(tris, sph)
            r.update_camera(cam)
            got = r.render(96, 72, spp=4, bounces=5, seed=2)
        assert got.shape == (72, 96, 4) and (got[..., 3] == 255).all()
        assert np.abs(got.astype(np.int32) - want.astype(np.int32)).max() <= 1, devs
        assert got[..., :3].mean() > 20                                   # the emissive panel lights the box


@pytest.mark.parametrize("with_lights", [False, True])
def test_fused_path_kernel_equals_the_staged_pipeline(with_lights):
    """rt_render on a brute-force scene (the reference's Cornell box: 12 triangles + 3 spheres, frag.glsl's own loop): the one-kernel
    path loop (RT_OPT_FUSED_PATHS 1, default) against the staged trace / shade kernels (0), byte for byte, on one slot and on
    interleaved shares, for a ragged frame, one bounce and many; and against orc_render within the gamma LSB."""
    tris, sph, _ = scenes.cornell()
    lights = _lights2() if with_lights else None
    w, h = 83, 37
    cam = rt.default_camera(w / h)
    for spp, bounces in ((3, 1), (4, 5), (10, 8)):
        frames = []
        for devs, fused in (((0,), 1), ((0,), 0), ((0, 0, 0), 1)):
            with rt.Renderer(devs) as r:
                r.set_option(rt.OPT_FUSED_PATHS, fused)
                r.create_data(tris, sph, lights)
                r.update_camera(cam)
                frames.append(r.render(w, h, spp=spp, bounces=bounces, seed=4))
        assert np.array_equal(frames[0], frames[1]), (spp, bounces, "fused vs staged")
        assert np.array_equal(frames[0], frames[2]), (spp, bounces, "three slots")
        _, want = oracle.render(oracle.Scene(tris, sph, lights), cam, w, h, spp, bounces, 4)
        assert np.abs(frames[0].astype(np.int32) - want.astype(np.int32)).max() <= 1
        assert frames[0][..., :3].mean() > 2                              # one bounce, no lights: only the emissive panel shows


def test_adversarial_ties_and_epsilon_cases():
    """scripts/fuzz_ties.py: coplanar grids with shared edges / vertices, duplicated triangles, slivers, zero-area triangles,
    coincident spheres; rays aimed exactly at vertices and edge midpoints, in-plane and axis-parallel.  0 differing records
    between every BVH mode / kernel and the brute-force kernel, and between that kernel and the CPU oracle."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "scripts", "fuzz_ties.py")], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-2000:]
    assert "0 differing records (rate 0.00e+00)" in p.stdout


def test_far_origins_and_offset_scenes_lose_no_hits():
    """scripts/far_origin.py: rays that start 10^2 .. 10^5 scene extents away, aimed at vertices, edge midpoints and the
    box-edge-touching wall edges of the Cornell box and of coplanar / jittered grids, and the same scenes translated by
    10^3 and 10^4.  The slab test's rounding grows with the ORIGIN's magnitude, the upload-time box pad only knows the
    vertices'; the per-ray pad (kernels.hip make_slab) closes that: every BVH layout x kernel (ray-buffer strip kernel,
    one-ray-per-lane kernel, tile kernel) must return the brute-force kernel's records bit for bit, and that kernel the
    oracle's (frag.glsl:265-301: every triangle is tested; a BVH may not cull one the loop accepts)."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("far_origin", os.path.join(root, "scripts", "far_origin.py"))
    far = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(far)
    tot = far.run(pad_on=True, verbose=False)
    assert tot["rays"] > 1_000_000 and tot["tile_rays"] > 100_000
    assert tot["bad"] == 0, f"ray-buffer kernels: {tot['bad']} of {tot['rays']} records differ from the reference loop"
    assert tot["tile_bad"] == 0, f"tile kernel: {tot['tile_bad']} of {tot['tile_rays']} records differ from the reference loop"
    assert tot["oracle_bad"] == 0, f"brute-force kernel vs oracle: {tot['oracle_bad']} of {tot['oracle']}"


# ---- the paths the big configs (C4, C5) take automatically ---------------------------------------------------------
@pytest.fixture(scope="module")
def big_soup():
    """1.5 M random triangles: nodes + triangles exceed 48 MB (auto ray binning) and the tree's SAH cost exceeds 250
    (auto primary-via-ray-buffer); its 4-wide stack bound exceeds the 20 LDS entries (capped stack with spill)."""
    return scenes.soup(1_500_000, half_extent=0.008)


def test_big_soup_takes_the_auto_paths_and_matches_the_reference_loop(big_soup, torch_dev):
    torch = torch_dev
    tris = big_soup
    n = (1 << 20) + 12345
    rays = scenes.random_rays(n)
    rays["tmax"][::97] = 0.0                                    # some dead rays (their own bin)
    hstats, _, _ = rt.bvh_build_host(tris, layout=rt.TRAVERSAL_BVH4Q, want_nodes=False)
    with rt.Renderer((0,)) as r:                                # no options: everything below is what "auto" picks
        st = r.create_data(tris)
        assert st["n_nodes"] * 64 + st["n_tris"] * 48 > 48 << 20, "scene too small to trigger auto ray binning"
        assert st["sah_cost"] > 250.0, "tree too coherent to trigger the ray-buffer primary path"
        assert hstats["stack_entries"] > 20, "stack bound too small to trigger the capped LDS stack"
        d_rays = to_dev(torch, rays)
        hits, occ = dev_buf(torch, n, 16), torch.zeros(n, dtype=torch.uint8, device="cuda")
        r.trace_dev(d_rays, n, hits)
        r.occluded_dev(d_rays, n, occ)
        torch.cuda.synchronize()
        got, got_occ = to_np(hits, rt.HIT).copy(), occ.cpu().numpy().copy()
        assert (got["prim"][::97] == rt.MISS).all() and 0.05 < (got["prim"] != rt.MISS).mean() < 0.999
        # (a) no binning, no capped stack, BVH2 nodes: every scheduling / layout choice gives the same records
        r.set_option(rt.OPT_RAY_BINNING, 0)
        r.set_option(rt.OPT_LDS_STACK, 0)
        h2 = dev_buf(torch, n, 16)
        r.trace_dev(d_rays, n, h2)
        torch.cuda.synchronize()
        assert to_np(h2, rt.HIT).tobytes() == got.tobytes(), "auto (binned, capped stack) vs plain strip kernel"
        r.set_option(rt.OPT_RAY_BINNING, 2)
        r.set_option(rt.OPT_LDS_STACK, 20)
        # (b) the reference loop on the GPU over a 64 k sample, the CPU oracle over 2 k
        idx = np.sort(np.random.default_rng(11).choice(n, 1 << 16, replace=False))
        r.set_option(rt.OPT_TRAVERSAL, rt.TRAVERSAL_BRUTE)
        d_s = to_dev(torch, rays[idx])
        hb, ob = dev_buf(torch, len(idx), 16), torch.zeros(len(idx), dtype=torch.uint8, device="cuda")
        r.trace_dev(d_s, len(idx), hb)
        r.occluded_dev(d_s, len(idx), ob)
        torch.cuda.synchronize()
        assert_parity(got[idx], to_np(hb, rt.HIT), "1.5 M soup: auto path vs brute-force kernel (64 k sample)")
        assert np.array_equal(got_occ[idx], ob.cpu().numpy())
        assert np.array_equal(got_occ, (got["prim"] != rt.MISS).astype(np.uint8))
        sub = idx[:: len(idx) // 2048][:2048]
        assert_parity(got[sub], oracle.trace(oracle.Scene(tris), rays[sub]), "1.5 M soup: auto path vs oracle (2 k sample)")
        r.set_option(rt.OPT_TRAVERSAL, rt.TRAVERSAL_AUTO)
        # (c) primary rays: auto = packet tile kernel (every BVH4q tree); 2 = ray generation + ray-buffer kernel with library
        #     scratch (what auto picks for a soup like this one without the packet form); 1 = tile kernel, one ray per lane
        cam = rt.camera_look_at((0.5, 2.2, 0.5), (0.5, 0.5, 0.5), 45.0, 256 / 192)
        r.update_camera(cam)
        w, h, spp = 256, 192, 4
        m = w * h * spp
        out = {}
        for path in (0, 1, 2):
            r.set_option(rt.OPT_PRIMARY, path)
            ph, pr = dev_buf(torch, m, 16), dev_buf(torch, m, 32)
            r.trace_primary_dev(w, h, ph, None if path == 2 else pr, spp=spp, seed=3, jitter=True)
            torch.cuda.synchronize()
            out[path] = to_np(ph, rt.HIT).copy()
        assert out[0].tobytes() == out[1].tobytes(), "primary rays: packet tile kernel vs one ray per lane"
        assert out[2].tobytes() == out[1].tobytes(), "primary rays: ray-buffer path vs tile kernel"
        o_rays = oracle.primary_rays(cam, w, h, spp=spp, seed=3, jitter=True)
        pidx = np.sort(np.random.default_rng(12).choice(m, 1024, replace=False))
        assert_parity(out[0][pidx], oracle.trace(oracle.Scene(tris), o_rays[pidx]), "1.5 M soup: primary sample vs oracle")


def test_c5_shaped_path_traced_frame_stage_by_stage():
    """C5's workload in small: a soup (SAH cost > 250) with emitters, a
    pinhole camera from rt_camera_look_at, 16 spp, 3 bounces, through the fused entry points C5 uses
    (rt_primary_shade_stripes_dev, rt_trace_dev, rt_shade_dev, rt_shade_resolve_dev).  Every stage against orc_shade /
    orc_resolve: bounce rays, throughput, radiance and the linear frame bit-exact, RGBA8 within 1 LSB."""
    import torch
    tris = scenes.soup(40_000, half_extent=0.04)             # SAH cost ~320
    e = tris["material"]["emission"]; e[::50] = (1.0, 0.9, 0.8); tris["material"]["emissionStrength"][::50] = 2.0
    w, h, spp, bounces, seed = 40, 24, 16, 3, 1
    cam = rt.camera_look_at((0.5, 2.2, 0.5), (0.5, 0.5, 0.5), 45.0, w / h)
    n = w * h * spp
    f = lambda k: torch.empty(n * k, dtype=torch.float32, device="cuda")
    sc = oracle.Scene(tris)
    o_rays = oracle.primary_rays(cam, w, h, spp=spp, seed=seed, jitter=True)
    o_thr = np.ones((n, 4), dtype=np.float32); o_thr[:, 3] = 0
    o_rad = np.zeros((n, 4), dtype=np.float32)
    with rt.Renderer((0,)) as r:
        st = r.create_data(tris)
        assert st["sah_cost"] > 250.0
        r.update_camera(cam)
        ra, rb, hits, thr, rad = f(8), f(8), f(4), f(4), f(4)
        lin = torch.empty(w * h * 4, dtype=torch.float32, device="cuda")
        fb = torch.empty(w * h, dtype=torch.int32, device="cuda")
        r.primary_shade_stripes_dev(w, h, 0, h, h, 1, thr, rad, ra, spp=spp, seed=seed, jitter=True, shade_seed=seed + 1)
        o_hits = oracle.trace(sc, o_rays)
        o_rays = oracle.shade(sc, o_rays, o_hits, o_thr, o_rad, key0=0, seed=seed + 1)
        for k in range(1, bounces):
            torch.cuda.synchronize()
            assert to_np(ra, rt.RAY).tobytes() == o_rays.tobytes(), f"segment {k}: rays differ"
            assert thr.cpu().numpy().tobytes() == o_thr.tobytes(), f"segment {k}: throughput differs"
            assert rad.cpu().numpy().tobytes() == o_rad.tobytes(), f"segment {k}: radiance differs"
            r.trace_dev(ra, n, hits)
            o_hits = oracle.trace(sc, o_rays)
            torch.cuda.synchronize()
            assert_parity(to_np(hits, rt.HIT), o_hits, f"segment {k} hits")
            if k + 1 == bounces:
                r.shade_resolve_dev(ra, hits, n, spp, thr, rad, lin, fb)
                oracle.shade(sc, o_rays, o_hits, o_thr, o_rad, key0=0, seed=0)      # radiance only matters now (:313-338)
            else:
                r.shade_dev(ra, hits, n, thr, rad, rb, key0=0, seed=seed + 1 + k)
                o_rays = oracle.shade(sc, o_rays, o_hits, o_thr, o_rad, key0=0, seed=seed + 1 + k)
                ra, rb = rb, ra
        torch.cuda.synchronize()
        o_lin, o_rgba = oracle.resolve(o_rad, w * h, spp)
        assert lin.cpu().numpy().tobytes() == o_lin.tobytes()
        got = fb.cpu().numpy().view(np.uint8).reshape(-1, 4).astype(np.int32)
        assert np.abs(got - o_rgba.astype(np.int32)).max() <= 1
        assert (got[:, :3].sum(axis=1) > 0).mean() > 0.02           # emitters are seen


def test_host_pointer_path_pipelined_and_pinned(torch_dev):
    """rt_trace / rt_occluded stream the rays through pinned staging buffers in 2 Mi-ray pieces on two streams; a block
    of rt_host_alloc (pinned) is read and written by DMA directly.  More than one ring of pieces, a ragged
    tail, pageable and pinned buffers: all must give the device-resident entry point's records."""
    torch = torch_dev
    n = 7 * (1 << 20) + 77                                       # 4 pieces: ring of 3 wraps, last piece ragged
    st_, rays = scenes.soup(20000, half_extent=0.02), scenes.random_rays(n)
    rays["tmax"][::9] = 0.3
    with rt.Renderer((0,)) as r:
        r.create_data(st_)
        d_rays, d_hits = to_dev(torch, rays), dev_buf(torch, n, 16)
        d_occ = torch.zeros(n, dtype=torch.uint8, device="cuda")
        r.trace_dev(d_rays, n, d_hits)
        r.occluded_dev(d_rays, n, d_occ)
        torch.cuda.synchronize()
        want, want_occ = to_np(d_hits, rt.HIT).tobytes(), d_occ.cpu().numpy().tobytes()
        assert r.trace(rays).tobytes() == want                   # pageable in, pageable out
        assert r.occluded(rays).tobytes() == want_occ
        p_rays, p_out = r.host_alloc(n, rt.RAY), r.host_alloc(n, rt.HIT)      # pinned: DMA straight from / to them
        p_rays[:] = rays
        p_out[:] = 0
        assert r.trace(p_rays, out=p_out) is p_out
        assert p_out.tobytes() == want                           # pinned in, pinned out: no staging
        assert r.trace(p_rays).tobytes() == want                 # pinned in, pageable out
        out = np.zeros(n, dtype=rt.HIT)
        assert r.trace(rays, out=out) is out and out.tobytes() == want
        r.host_free(p_out)
        with pytest.raises(RuntimeError, match="not a block"):
            r.host_free(out)
        assert r.trace(rays[:1000]).tobytes() == want[:16000]    # smaller than a piece, after a big call


def test_launches_on_two_streams_share_scratch_safely(torch_dev):
    """Per-slot scratch (cursor sets, spill columns, the binning permutation) is shared by launches; launches on different
    streams are ordered by events inside the library, so un-synchronised use from two streams gives the right records."""
    torch = torch_dev
    n = 300_000
    st_, rays = scenes.soup(60000, half_extent=0.02), scenes.random_rays(2 * n)
    with rt.Renderer((0,)) as r:
        r.set_option(rt.OPT_RAY_BINNING, 1)                      # both launches permute through the one binning buffer
        r.create_data(st_)
        d_a, d_b = to_dev(torch, rays[:n]), to_dev(torch, rays[n:])
        h_ref_a, h_ref_b = dev_buf(torch, n, 16), dev_buf(torch, n, 16)
        r.trace_dev(d_a, n, h_ref_a); r.trace_dev(d_b, n, h_ref_b)
        torch.cuda.synchronize()
        s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
        outs = [(dev_buf(torch, n, 16), dev_buf(torch, n, 16)) for _ in range(6)]
        for ha, hb in outs:                                     # 12 launches in flight on two streams, no host sync
            r.trace_dev(d_a, n, ha, stream=s1)
            r.trace_dev(d_b, n, hb, stream=s2)
        torch.cuda.synchronize()
        for ha, hb in outs:
            assert torch.equal(ha.view(torch.int32), h_ref_a.view(torch.int32))
            assert torch.equal(hb.view(torch.int32), h_ref_b.view(torch.int32))


def test_rccl_collectives_on_gpu_tensors_world_1(tmp_path):
    """The exchange step of raytracer_amd/dist.py on the "nccl" (= RCCL) backend with GPU tensors.  One GPU here, so a
    group of one rank: gather_rows_to (dist.gather into views of the frame), gather_rows / gather_ranges
    (all_gather_into_tensor) must run on the device and return the rank's own stripes in image order.  Own process: the
    process group must not leak into the other tests."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = f"""
import os, sys
sys.path.insert(0, {root!r})
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1")
import torch, torch.distributed as dist
from raytracer_amd import dist as rd
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
assert dist.get_backend() == "nccl"
for h, sh in ((37, 8), (64, 16)):
    plan = rd.stripe_plan(h, 1, 0, sh)
    local = torch.arange(plan.padded_rows * 5, dtype=torch.int32, device="cuda").view(plan.padded_rows, 5)
    a = rd.gather_rows_to(local, plan, 0)
    b = rd.gather_rows(local, plan)
    assert a.is_cuda and b.is_cuda and a.shape == (h, 5)
    assert torch.equal(a, local[:h]) and torch.equal(b, local[:h])
    # the pipelined form bench.py uses: the gather of frame k is in flight while frame k + 1 is produced
    fg = rd.FrameGather(plan, (5,), torch.int32, "cuda")
    bufs, prev, frames = [torch.empty_like(local), torch.empty_like(local)], None, []
    for k in range(5):
        bufs[k % 2].copy_(local + 1000 * k)
        t = fg.start(bufs[k % 2])
        if prev is not None: frames.append(fg.finish(prev))
        prev = t
    frames.append(fg.finish(prev))
    torch.cuda.synchronize()
    for k, f in enumerate(frames):
        assert f.is_cuda and torch.equal(f, local[:h] + 1000 * k), k
r = torch.arange(1000, dtype=torch.float32, device="cuda").view(250, 4)
assert torch.equal(rd.gather_ranges(r, 240, 1), r[:240])
dist.barrier(); torch.cuda.synchronize()
dist.destroy_process_group()
print("rccl world-1 ok")
"""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0 and "rccl world-1 ok" in p.stdout, p.stdout[-2000:] + p.stderr[-3000:]

