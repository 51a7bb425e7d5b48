"""CPU tests: the oracle against the hand-derived KATs and the committed golden fixtures.

The reference has no tests of its own; the KATs are derived by hand from the
reference's scene data (src/vulkan/VkRenderer.cpp:171-250) and its intersection routine
(shaders/frag.glsl:130-156), and are exact in fp32.
"""
import ctypes as C
import hashlib
import json
import os

import numpy as np
import pytest

import oracle
import raytracer_amd as rt
from raytracer_amd import scenes

from conftest import GOLDEN


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def ray(o, d, tmin=0.0, tmax=1e20):
    r = np.zeros(1, dtype=oracle.RAY)
    r[0]["o"], r[0]["d"], r[0]["tmin"], r[0]["tmax"] = o, d, tmin, tmax
    return r


@pytest.fixture(scope="module")
def cornell():
    tris, sph, _ = scenes.cornell()
    return tris, sph, oracle.Scene(tris, sph)


def test_kats(cornell):
    _, _, sc = cornell
    kat = json.load(open(os.path.join(GOLDEN, "kat.json")))
    for name, k in kat.items():
        h = oracle.trace(sc, ray(k["o"], k["d"]))[0]
        assert int(h["prim"]) == k["prim"], name
        # exact in fp32: compare bits, not tolerances
        assert float(h["t"]) == k["t"] and float(h["u"]) == k["u"] and float(h["v"]) == k["v"], name


def test_kat1_is_a_tie_kept_by_the_lower_index(cornell):
    """KAT-1: triangles 2 and 3 share the diagonal; both give t = 6; the scan keeps index 2
    (strict '<' at frag.glsl:285).  Triangle 3 alone must also report t = 6 with v = -0.0."""
    tris, _, _ = cornell
    L = oracle.lib()
    r = ray((0, 4, 1), (0, -1, 0))
    t, u, v = C.c_float(), C.c_float(), C.c_float()
    for idx, (eu, ev) in ((2, (0.5, 0.5)), (3, (0.5, 0.0))):
        p = [np.ascontiguousarray(tris[idx][k]["position"], dtype=np.float32) for k in ("v0", "v1", "v2")]
        fp = [a.ctypes.data_as(C.POINTER(C.c_float)) for a in p]
        hit = L.orc_ray_triangle(C.c_void_p(r.ctypes.data), fp[0], fp[1], fp[2], C.byref(t), C.byref(u), C.byref(v))
        assert hit == 1 and t.value == 6.0 and u.value == eu and v.value == ev
    assert np.signbit(np.float32(v.value))          # v = -0.0 passes `v < 0.0` (frag.glsl:147)


def test_sphere_first_positive_root(cornell):
    """frag.glsl:174-181: t0 if t0 > 0 else t1 if t1 > 0; origin inside a sphere takes the far root."""
    _, sph, sc = cornell
    c = sph[1]["center"]
    h = oracle.trace(sc, ray((c[0], c[1], c[2] + 1.0), (0, 0, -1)))[0]
    assert int(h["prim"]) == (rt.SPHERE_BIT | 1) and abs(float(h["t"]) - 0.8) < 1e-6
    h = oracle.trace(oracle.Scene(None, sph), ray(tuple(c), (0, 0, 1)))[0]
    assert int(h["prim"]) == (rt.SPHERE_BIT | 1) and abs(float(h["t"]) - 0.2) < 1e-6


def test_sphere_beats_triangle_on_equal_t():
    """traceRay scans spheres first and replaces on strict '<' only (frag.glsl:269-285)."""
    sph = np.zeros(1, dtype=rt.SPHERE)
    sph[0]["center"], sph[0]["radius"] = (0, 0, 0), 1.0
    tri = np.zeros(1, dtype=rt.TRIANGLE)
    tri[0]["v0"]["position"], tri[0]["v1"]["position"], tri[0]["v2"]["position"] = (-4, -4, 1), (4, -4, 1), (0, 8, 1)
    h = oracle.trace(oracle.Scene(tri, sph), ray((0, 0, 3), (0, 0, -1)))[0]
    assert float(h["t"]) == 2.0 and int(h["prim"]) == rt.SPHERE_BIT
    h = oracle.trace(oracle.Scene(tri, None), ray((0, 0, 3), (0, 0, -1)))[0]
    assert float(h["t"]) == 2.0 and int(h["prim"]) == 0


def test_interval_and_miss(cornell):
    _, _, sc = cornell
    h = oracle.trace(sc, ray((0, 4, 1), (0, -1, 0), tmax=6.0))[0]          # hit at exactly tmax is not < tmax
    assert int(h["prim"]) == rt.MISS and float(h["t"]) == 6.0 and float(h["u"]) == 0.0
    h = oracle.trace(sc, ray((0, 4, 1), (0, -1, 0), tmin=6.5))[0]
    assert int(h["prim"]) == rt.MISS
    h = oracle.trace(oracle.Scene(None, None), ray((0, 0, 0), (0, 0, 1)))[0]   # empty scene
    assert int(h["prim"]) == rt.MISS and float(h["t"]) == np.float32(1e20)
    assert oracle.trace(sc, np.zeros(0, dtype=oracle.RAY)).size == 0


def test_degenerate_and_parallel_triangles_are_rejected():
    tri = np.zeros(2, dtype=rt.TRIANGLE)                                    # triangle 0: all-zero (the reference's
    tri[1]["v0"]["position"], tri[1]["v1"]["position"], tri[1]["v2"]["position"] = (0, 0, 0), (1, 0, 0), (0, 1, 0)
    sc = oracle.Scene(tri)                                                  # placeholder element, VkRenderer.cpp:262-270)
    assert int(oracle.trace(sc, ray((0.2, 0.2, 1), (1, 0, 0)))[0]["prim"]) == rt.MISS     # parallel: |a| < EPS
    assert int(oracle.trace(sc, ray((0.2, 0.2, 1), (0, 0, -1)))[0]["prim"]) == 1


def test_pixel_uv_and_default_camera_ray():
    """Default camera (VkRenderer.cpp:3-12): front (0,-1,0), right (1,0,0), up (0,0,-1); the centre of a
    2x2 image looks between pixels, pixel (0,0) is top-left => ray goes left (-x) and up (+z)."""
    cam = rt.default_camera(1.0)
    assert np.allclose(cam["front"][0], (0, -1, 0)) and np.allclose(cam["right"][0], (1, 0, 0))
    assert np.allclose(cam["up"][0], (0, 0, -1))
    r = oracle.primary_rays(cam, 2, 2)
    assert np.allclose(r[0]["o"], (0, 4, 1)) and r[0]["d"][0] < 0 and r[0]["d"][2] > 0 and r[0]["d"][1] < 0
    assert np.allclose(np.linalg.norm(r["d"], axis=1), 1.0, atol=1e-6)
    h = np.float32(np.tan(np.float32(np.float32(45.0) * np.float32(0.017453292519943295)) / np.float32(2)))
    assert abs(-r[0]["d"][0] / -r[0]["d"][1] - 0.5 * h) < 1e-6        # ndc = -0.5 at pixel 0 of 2


def test_golden_fixtures(cornell):
    tris, sph, sc = cornell
    meta = json.load(open(os.path.join(GOLDEN, "meta.json")))
    cam = rt.default_camera(1.0)
    assert sha(oracle.primary_rays(cam, 64, 64)) == meta["c1_cornell_rays_64x64_sha256"]
    assert sha(oracle.primary_rays(cam, 32, 32, spp=4, seed=1, jitter=True)) == meta["c1_cornell_rays_32x32x4_seed1_sha256"]
    g = np.load(os.path.join(GOLDEN, "c1_cornell_64x64_hits.npy"))
    assert oracle.trace_primary(sc, cam, 64, 64).tobytes() == g.tobytes()
    assert sha(oracle.trace_primary(sc, cam, 512, 512)) == meta["c1_cornell_512x512_sha256"]
    g = np.load(os.path.join(GOLDEN, "c1_cornell_61x37_hits.npy"))
    assert oracle.trace_primary(sc, rt.default_camera(), 61, 37).tobytes() == g.tobytes()
    t32, s32, _ = scenes.cornell32()
    g = np.load(os.path.join(GOLDEN, "c1_32_cornell_64x64_hits.npy"))
    assert oracle.trace_primary(oracle.Scene(t32, s32), cam, 64, 64).tobytes() == g.tobytes()
    st, rays = scenes.soup(2000, half_extent=0.05), scenes.random_rays(4096)
    assert sha(st) == meta["soup2k_tris_sha256"] and sha(rays) == meta["rays4k_sha256"]
    g = np.load(os.path.join(GOLDEN, "soup2k_rays4k_hits.npy"))
    assert oracle.trace(oracle.Scene(st), rays).tobytes() == g.tobytes()
    pr = oracle.primary_rays(cam, 64, 64)
    ph = oracle.trace(sc, pr)
    assert oracle.shadow_lit(sc, pr, ph, (0.0, 0.0, 1.9)).tobytes() == np.load(os.path.join(GOLDEN, "c1_cornell_64x64_lit.npy")).tobytes()
    assert oracle.bounce(sc, pr, ph, seed=7).tobytes() == np.load(os.path.join(GOLDEN, "c1_cornell_64x64_bounce_rays.npy")).tobytes()


def test_occluded_equals_closest_hit_found(cornell):
    _, _, sc = cornell
    rays = scenes.random_rays(2000)
    rays["o"] = rays["o"] * 3.0 - 1.5
    assert np.array_equal(oracle.occluded(sc, rays), (oracle.trace(sc, rays)["prim"] != rt.MISS).astype(np.uint8))


def test_hemisphere_polynomial_vs_libm():
    """orc_hemisphere folds acos/sin/cos analytically and uses a fixed sincos polynomial; the literal libm
    form of frag.glsl:190-202 must agree to 2e-6 per component (GLSL leaves their precision open)."""
    L = oracle.lib()
    rng = np.random.default_rng(3)
    worst = 0.0
    for _ in range(4000):
        n = rng.normal(size=3).astype(np.float32)
        n /= np.linalg.norm(n)
        if rng.random() < 0.1:
            n = np.array([0, 0, 1 if rng.random() < 0.5 else -1], dtype=np.float32)
        xi1, xi2 = np.float32(rng.integers(0, 1 << 24)) / np.float32(1 << 24), np.float32(rng.integers(0, 1 << 24)) / np.float32(1 << 24)
        a, b = np.zeros(3, dtype=np.float32), np.zeros(3, dtype=np.float32)
        nf = n.ctypes.data_as(C.POINTER(C.c_float))
        L.orc_hemisphere(nf, C.c_float(xi1), C.c_float(xi2), a.ctypes.data_as(C.POINTER(C.c_float)))
        L.orc_hemisphere_libm(nf, C.c_float(xi1), C.c_float(xi2), b.ctypes.data_as(C.POINTER(C.c_float)))
        worst = max(worst, float(np.abs(a - b).max()))
        assert np.dot(a, n) >= -1e-6
    assert worst < 2e-6, worst


def test_rand_is_uniform_and_seeded():
    L = oracle.lib()
    x = np.array([L.orc_rand(i, 2, 1) for i in range(20000)])
    assert 0.0 <= x.min() and x.max() < 1.0 and abs(x.mean() - 0.5) < 0.01 and abs(x.var() - 1 / 12) < 0.005
    assert L.orc_rand(5, 2, 1) != L.orc_rand(5, 3, 1) and L.orc_rand(5, 2, 1) != L.orc_rand(5, 2, 2)


def test_shadow_expression(cornell):
    """frag.glsl:321-330: a light inside the box lights the floor under it; a light far outside the
    closed back wall leaves points in front of that wall in shadow."""
    _, _, sc = cornell
    pr = ray((0.5, 1.0, 1.0), (0, 0, -1))          # KAT-2 ray, hits the floor at (0.5, 1, 0)
    ph = oracle.trace(sc, pr)
    assert oracle.shadow_lit(sc, pr, ph, (0.5, 1.0, 1.5))[0] == 1
    assert oracle.shadow_lit(sc, pr, ph, (0.0, -10.0, 1.0))[0] == 0
    miss = ray((0, 10, 1), (0, 1, 0))
    assert oracle.shadow_lit(sc, miss, oracle.trace(sc, miss), (0, 0, 1))[0] == 2


# ---- shading + path loop (frag.glsl:205-263, :303-360) ------------------------------------------------------
def test_brdf_known_values():
    """computeBRDF (frag.glsl:239-263) at N = V = L: NdotX = 1, VdotH = 1 => F = F0, D = 1/(pi a^2), G = 1/(1... )."""
    m = scenes.material12((0.8, 0.5, 0.2), roughness=0.5, metallic=0.0)
    got = oracle.brdf(m, (0, 0, 1), (0, 0, 1), (0, 0, 1)).astype(np.float64)
    a2 = 0.5 ** 4
    D = a2 / (np.pi * (a2 - 1.0 + 1.0) ** 2)
    k = (0.5 + 1.0) ** 2 / 8.0
    G = (1.0 / (1.0 * (1 - k) + k)) ** 2
    spec = D * 0.04 * G / (4.0 + 0.001)
    want = (1 - 0.04) * np.array([0.8, 0.5, 0.2]) / np.pi + spec
    assert np.allclose(got, want, rtol=2e-6)
    # metallic = 1: no diffuse lobe, F0 = albedo
    m = scenes.material12((1.0, 0.9, 0.0), roughness=0.1, metallic=1.0)
    got = oracle.brdf(m, (0, 0, 1), (0.6, 0, 0.8), (-0.6, 0, 0.8)).astype(np.float64)
    assert got[2] < got[1] < got[0] and got[2] >= 0


def test_shade_emission_miss_and_throughput(cornell):
    tris, sph, sc = cornell
    rays = np.concatenate([ray((0, 0, 0.5), (0, 0, 1)), ray((0, 10, 1), (0, 1, 0)), ray((0.5, 1, 1), (0, 0, -1))])
    hits = oracle.trace(sc, rays)
    assert int(hits[0]["prim"]) in (10, 11) and int(hits[1]["prim"]) == rt.MISS          # light panel / outside
    thr = np.ones((3, 4), dtype=np.float32); thr[:, 3] = 0
    rad = np.zeros((3, 4), dtype=np.float32)
    out = oracle.shade(sc, rays, hits, thr, rad, seed=3)
    assert np.allclose(rad[0, :3], 0.5) and np.all(rad[1] == 0) and np.all(rad[2] == 0)  # emission 1 x strength 0.5
    assert out[1]["tmax"] == 0 and np.all(thr[1, :3] == 1)                               # the path ended at the miss
    # diffuse white floor, cosine sampling: BRDF * cos / pdf = albedo * (1 - F) <= 1
    assert np.all(thr[2, :3] <= 1.0) and np.all(thr[2, :3] > 0.5)
    assert np.allclose(out[2]["o"], (0.5, 1.0, 0.001), atol=1e-6) and out[2]["d"][2] > 0


def test_render_golden_and_energy(cornell):
    tris, sph, sc = cornell
    cam = rt.default_camera(1.0)
    lin, rgba = oracle.render(sc, cam, 32, 32, 2, 3, 1)
    meta = json.load(open(os.path.join(GOLDEN, "meta.json")))
    assert sha(lin) == meta["c1_render_32x32_spp2_b3_seed1_linear_sha256"]
    assert not np.isnan(lin).any() and lin[..., :3].max() <= 1.0 and lin[..., :3].mean() > 0.02
    assert (rgba[..., 3] == 255).all()
    want = np.clip(np.power(lin[..., :3].astype(np.float64), 1 / 2.6), 0, 1) * 255 + 0.5
    assert np.abs(rgba[..., :3].astype(np.int64) - want.astype(np.int64)).max() <= 1      # frag.glsl:358
    # more bounces can only add light (every term is non-negative)
    lin5, _ = oracle.render(sc, cam, 32, 32, 2, 5, 1)
    assert (lin5[..., :3] >= lin[..., :3]).all()
