"""Scene setup for the BASELINE.json configs (host side; all geometry comes from the C ABI generators).

The reference compiles one scene into VkRenderer::createData (src/vulkan/VkRenderer.cpp:164-258) and
has no mesh loader; Bunny / Sponza / San Miguel are not in it and cannot be fetched, so C2/C3 use the
labelled procedural stand-ins of SURVEY.md 8(d), built from the reference's own tessellator
(src/math/Sphere.h:17-79).
"""
import ctypes as C
import os

import numpy as np

from . import CAMERA, LIGHT, RAY, SPHERE, TRIANGLE, _ptr, camera_look_at, lib

WHITE = (1.0, 1.0, 1.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0)   # albedo, emission, emissionStrength, roughness, metallic


def material12(albedo=(1, 1, 1), emission=(0, 0, 0), strength=0.0, roughness=0.0, metallic=0.0):
    """The 48-byte Material record (src/math/Material.h:6-11) as 12 floats."""
    m = np.zeros(12, dtype=np.float32)
    m[0:3] = albedo
    m[4:7] = emission
    m[7] = strength
    m[8] = roughness
    m[9] = metallic
    return m


def cat(parts):
    """Concatenate TRIANGLE arrays bytewise (np.concatenate would repack the padded struct dtype)."""
    raw = np.concatenate([np.ascontiguousarray(p).view(np.uint8).reshape(-1, TRIANGLE.itemsize) for p in parts])
    return raw.reshape(-1).view(TRIANGLE)


def cornell():
    """C1: 12 triangles + 3 spheres, 0 lights (src/vulkan/VkRenderer.cpp:171-258)."""
    L = lib()
    tris = np.zeros(L.rt_gen_cornell_triangles(None, 0), dtype=TRIANGLE)
    L.rt_gen_cornell_triangles(_ptr(tris), len(tris))
    sph = np.zeros(L.rt_gen_cornell_spheres(None, 0), dtype=SPHERE)
    L.rt_gen_cornell_spheres(_ptr(sph), len(sph))
    return tris, sph, np.zeros(0, dtype=LIGHT)


def sphere_geometry(center, radius, stacks, slices, mat=None):
    """Sphere::sphereGeometry(stacks, slices) (src/math/Sphere.h:17-79): 2*stacks*slices triangles."""
    L = lib()
    mat = material12() if mat is None else np.asarray(mat, dtype=np.float32)
    c = (C.c_float * 3)(*center)
    m = mat.ctypes.data_as(C.POINTER(C.c_float))
    n = L.rt_gen_sphere_geometry(c, C.c_float(radius), stacks, slices, m, None, 0)
    tris = np.zeros(n, dtype=TRIANGLE)
    L.rt_gen_sphere_geometry(c, C.c_float(radius), stacks, slices, m, _ptr(tris), n)
    return tris


def cornell32():
    """C1-32: the 12 Cornell triangles + sphere #1 tessellated with sphereGeometry(2,5) = 32 triangles
    (the call the reference has commented out at VkRenderer.cpp:246-247), other two spheres analytic."""
    tris, sph, lights = cornell()
    gold = np.frombuffer(sph[0:1].tobytes(), dtype=np.float32)[4:16]
    geo = sphere_geometry(sph[0]["center"], float(sph[0]["radius"]), 2, 5, gold)
    return cat([tris, geo]), sph[1:].copy(), lights


def _quad(p0, p1, p2, p3, normal, mat):
    """Two triangles (p0,p1,p2), (p0,p2,p3) with a shared normal."""
    t = np.zeros(2, dtype=TRIANGLE)
    for k, (a, b, c) in enumerate(((p0, p1, p2), (p0, p2, p3))):
        for name, p in (("v0", a), ("v1", b), ("v2", c)):
            t[k][name]["position"] = p
            t[k][name]["normal"] = normal
        raw = t[k:k + 1].view(np.uint8).reshape(-1)
        raw[96:144] = np.asarray(mat, dtype=np.float32).view(np.uint8)
    return t


def box(lo, hi, mat=None):
    """Axis-aligned room, 12 triangles, normals pointing inwards."""
    mat = material12() if mat is None else mat
    x0, y0, z0 = lo
    x1, y1, z1 = hi
    q = [
        _quad((x0, y0, z0), (x1, y0, z0), (x1, y1, z0), (x0, y1, z0), (0, 0, 1), mat),    # floor
        _quad((x0, y0, z1), (x0, y1, z1), (x1, y1, z1), (x1, y0, z1), (0, 0, -1), mat),   # ceiling
        _quad((x0, y0, z0), (x0, y0, z1), (x1, y0, z1), (x1, y0, z0), (0, 1, 0), mat),    # y = y0 wall
        _quad((x0, y1, z0), (x1, y1, z0), (x1, y1, z1), (x0, y1, z1), (0, -1, 0), mat),   # y = y1 wall
        _quad((x0, y0, z0), (x0, y1, z0), (x0, y1, z1), (x0, y0, z1), (1, 0, 0), mat),    # x = x0 wall
        _quad((x1, y0, z0), (x1, y0, z1), (x1, y1, z1), (x1, y1, z0), (-1, 0, 0), mat),   # x = x1 wall
    ]
    return cat(q)


def real_mesh(env_var, aspect, fov_deg=45.0):
    """a real mesh (Bunny / Sponza / San Miguel) is used when `env_var` names a readable OBJ file --
    none ships with the reference or this repository.  Returns (tris, camera framing the mesh's bounding box, light
    above it) or None.  Every triangle gets the default grey material of rt_load_obj."""
    path = os.environ.get(env_var, "")
    if not path or not os.path.isfile(path):
        return None
    from . import load_obj
    tris = load_obj(path)
    if len(tris) == 0:
        return None
    p = np.concatenate([tris[v]["position"] for v in ("v0", "v1", "v2")])
    lo, hi = p.min(axis=0), p.max(axis=0)
    ctr, ext = (lo + hi) / 2, float(np.max(hi - lo))
    eye = ctr + np.array([0.35, -1.0, 0.45], dtype=np.float32) * ext * 1.6
    cam = camera_look_at(tuple(float(x) for x in eye), tuple(float(x) for x in ctr), fov_deg, aspect)
    lights = np.zeros(1, dtype=LIGHT)
    lights[0]["position"] = tuple(float(x) for x in ctr + np.array([0.5, -0.5, 1.5], dtype=np.float32) * ext)
    lights[0]["color"] = (1, 1, 1)
    lights[0]["intensity"] = 5.0 * ext * ext
    return tris, cam, lights


def bunny_class():
    """C2 stand-in ("bunny-class"): sphereGeometry(186,186) = 69,192 triangles above a 2-triangle ground
    quad (so that the shadow ray has something to fall on); one point light.  NOT the Stanford Bunny."""
    real = real_mesh("RT_BUNNY_OBJ", 1920.0 / 1080.0)
    if real is not None:                                   # the real mesh, if the caller has one
        return real[0], np.zeros(0, dtype=SPHERE), real[2], real[1]
    body = sphere_geometry((0.0, 0.0, 1.0), 1.0, 186, 186, material12((0.8, 0.8, 0.7)))
    ground = _quad((-6, -6, 0), (6, -6, 0), (6, 6, 0), (-6, 6, 0), (0, 0, 1), material12())
    tris = cat([body, ground])
    lights = np.zeros(1, dtype=LIGHT)
    lights[0]["position"] = (3.0, 3.0, 5.0)
    lights[0]["color"] = (1, 1, 1)
    lights[0]["intensity"] = 5.0
    cam = camera_look_at((0.0, 4.0, 1.5), (0.0, 0.0, 0.9), 45.0, 1920.0 / 1080.0)
    return tris, np.zeros(0, dtype=SPHERE), lights, cam


def sponza_class(tess=45):
    """C3 stand-in ("sponza-class"): 8 x 8 grid of sphereGeometry(45,45) (4,050 triangles each = 259,200)
    inside a closed 12-triangle room = 259,212 triangles; camera inside the room.  NOT Sponza.
    `tess` other than 45 gives the same scene at another tessellation (scaling probes only)."""
    real = real_mesh("RT_SPONZA_OBJ", 3840.0 / 2160.0, 60.0) if tess == 45 else None
    if real is not None:                                   # the real mesh, if the caller has one
        return real[0], np.zeros(0, dtype=SPHERE), np.zeros(0, dtype=LIGHT), real[1]
    parts = [box((-8.5, -8.5, 0.0), (8.5, 8.5, 5.0))]
    for i in range(8):
        for j in range(8):
            z = 0.8 + 0.35 * ((i * 3 + j * 5) % 7)
            alb = (0.3 + 0.1 * (i % 5), 0.3 + 0.1 * (j % 5), 0.6)
            parts.append(sphere_geometry((-7.0 + 2.0 * i, -7.0 + 2.0 * j, z), 0.8, tess, tess, material12(alb)))
    tris = cat(parts)
    cam = camera_look_at((0.3, 8.0, 2.6), (0.0, 0.0, 1.2), 60.0, 3840.0 / 2160.0)
    return tris, np.zeros(0, dtype=SPHERE), np.zeros(0, dtype=LIGHT), cam


def soup(n=10_000_000, seed=0x5EED0001, half_extent=0.005):
    """C4: n random triangles in [0,1)^3."""
    tris = np.zeros(n, dtype=TRIANGLE)
    lib().rt_gen_soup(seed, n, C.c_float(half_extent), _ptr(tris), n)
    return tris


def random_rays(n, first=0, seed=0x5EED0002):
    """C4 rays [first, first+n): origin uniform in the unit cube, direction uniform on the sphere."""
    rays = np.zeros(n, dtype=RAY)
    lib().rt_gen_random_rays(seed, first, n, _ptr(rays))
    return rays


# ---- adversarial geometry for the tie / epsilon / far-origin parity tests (scripts/fuzz_ties.py, tests/) ----------
def tri_array(p0, p1, p2):
    """TRIANGLE records from three (n, 3) vertex arrays, normal +z, grey."""
    t = np.zeros(len(p0), dtype=TRIANGLE)
    for name, p in (("v0", p0), ("v1", p1), ("v2", p2)):
        t[name]["position"] = np.asarray(p, dtype=np.float32)
        t[name]["normal"] = (0, 0, 1)
    t["material"]["albedo"] = 0.7
    return t


def grid_scene(n, jitter_z, rng):
    """n x n quads over the unit square in the plane z = f(x, y) (jitter_z = 0: exactly coplanar), two triangles each,
    shared edges and vertices.  Returns (triangles, vertices)."""
    xs = np.linspace(0, 1, n + 1, dtype=np.float32)
    X, Y = np.meshgrid(xs, xs, indexing="ij")
    Z = (rng.standard_normal(X.shape) * jitter_z).astype(np.float32)
    P = np.stack([X, Y, Z], -1)
    a, b, c, d = P[:-1, :-1], P[1:, :-1], P[1:, 1:], P[:-1, 1:]
    t1 = tri_array(a.reshape(-1, 3), b.reshape(-1, 3), c.reshape(-1, 3))
    t2 = tri_array(a.reshape(-1, 3), c.reshape(-1, 3), d.reshape(-1, 3))
    return cat([t1, t2]), P.reshape(-1, 3)


def rays_at(points, origins, tmin=0.0, tmax=1e20):
    """Rays from `origins` through `points` (direction normalised in float64, then rounded to fp32)."""
    points, origins = np.asarray(points, dtype=np.float64), np.asarray(origins, dtype=np.float64)
    d = points - origins
    d = d / np.linalg.norm(d, axis=1, keepdims=True)
    r = np.zeros(len(points), dtype=RAY)
    r["o"], r["d"], r["tmin"], r["tmax"] = origins.astype(np.float32), d.astype(np.float32), tmin, tmax
    return r


def translated(tris, offset):
    """Copy of `tris` with every position moved by `offset` (rounded to fp32: the scene IS the rounded numbers)."""
    t = tris.copy()
    off = np.asarray(offset, dtype=np.float32)
    for v in ("v0", "v1", "v2"):
        t[v]["position"] = (t[v]["position"].astype(np.float32) + off).astype(np.float32)
    return t
