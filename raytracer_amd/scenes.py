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
    room = box((-8.5, -8.5, 0.0), (8.5, 8.5, 5.0))
    # the ceiling (second quad of box()) glows: the scene has no Light records (no shadow rays: SURVEY 8d's C3 is
    # primary + one diffuse bounce), so without an emitter every path would carry radiance 0 and the rendered frame --
    # and with it bench.py's frame_sha256 -- would be black whatever the kernels returned.  Geometry is untouched.
    room[2:4]["material"]["emission"] = (1.0, 0.95, 0.85)
    room[2:4]["material"]["emissionStrength"] = 4.0
    parts = [room]
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


# ---- a harder "Sponza-class" stand-in: a colonnade hall -----------------------------------------------------------
def _mesh(p0, p1, p2, albedo=(0.7, 0.7, 0.7), emission=None, strength=0.0, roughness=0.6, metallic=0.0):
    """TRIANGLE records with flat (geometric) normals from three (n, 3) float arrays."""
    p0, p1, p2 = (np.asarray(a, dtype=np.float32).reshape(-1, 3) for a in (p0, p1, p2))
    n = np.cross((p1 - p0).astype(np.float64), (p2 - p0).astype(np.float64))
    ln = np.linalg.norm(n, axis=1, keepdims=True)
    n = np.where(ln > 0, n / np.maximum(ln, 1e-300), (0.0, 0.0, 1.0)).astype(np.float32)
    t = np.zeros(len(p0), dtype=TRIANGLE)
    for name, q in (("v0", p0), ("v1", p1), ("v2", p2)):
        t[name]["position"] = q
        t[name]["normal"] = n
    m = t["material"]
    m["albedo"], m["roughness"], m["metallic"] = albedo, roughness, metallic
    if emission is not None:
        m["emission"], m["emissionStrength"] = emission, strength
    return t


def _grid(P, **mat):
    """Two triangles per cell of a (nu+1, nv+1, 3) grid of points."""
    P = np.asarray(P, dtype=np.float32)
    a, b, c, d = P[:-1, :-1], P[1:, :-1], P[1:, 1:], P[:-1, 1:]
    r = lambda x: x.reshape(-1, 3)
    return cat([_mesh(r(a), r(b), r(c), **mat), _mesh(r(a), r(c), r(d), **mat)])


def _box(lo, hi, **mat):
    """Closed axis-aligned box, 12 triangles, outward faces."""
    (x0, y0, z0), (x1, y1, z1) = lo, hi
    c = np.array([(x0, y0, z0), (x1, y0, z0), (x1, y1, z0), (x0, y1, z0), (x0, y0, z1), (x1, y0, z1), (x1, y1, z1), (x0, y1, z1)], dtype=np.float32)
    f = [(0, 2, 1), (0, 3, 2), (4, 5, 6), (4, 6, 7), (0, 1, 5), (0, 5, 4), (2, 3, 7), (2, 7, 6), (1, 2, 6), (1, 6, 5), (3, 0, 4), (3, 4, 7)]
    f = np.array(f)
    return _mesh(c[f[:, 0]], c[f[:, 1]], c[f[:, 2]], **mat)


def colonnade(detail=1.0):
    """A second, harder C3-class stand-in ("colonnade", ~262 k triangles at detail 1): a 36 x 14 x 10 hall with what the
    sphere grid of sponza_class() lacks -- long thin floor / wall / beam triangles beside a finely tiled mosaic
    (triangle edge lengths from 36 down to 0.03: more than 100:1), two rows of fluted columns with capitals, arches
    between them, a barrel vault, rows of thin balusters, and wavy hanging drapes (thin, overlapping, open surfaces).
    Emissive window panels light it.  NOT Sponza; same interface as sponza_class(): (tris, spheres, lights, camera)."""
    k = lambda n: max(2, int(round(n * detail)))
    parts = []
    stone, dark, cloth = (0.72, 0.68, 0.6), (0.35, 0.33, 0.3), (0.6, 0.15, 0.12)
    L, Wd, Ht = 36.0, 14.0, 10.0
    # floor: 24 strips 36 m long and 0.58 m wide (two triangles each: aspect ratio 60:1) ...
    ys = np.linspace(-Wd / 2, Wd / 2, 25, dtype=np.float32)
    P = np.zeros((2, 25, 3), dtype=np.float32); P[0, :, 0], P[1, :, 0] = -L / 2, L / 2; P[:, :, 1] = ys
    parts.append(_grid(P, albedo=stone))
    # ... and a finely tiled mosaic lying 2 mm above it (0.03 m tiles)
    n = k(150)
    u = np.linspace(-2.25, 2.25, n + 1, dtype=np.float32)
    X, Y = np.meshgrid(u + 4.0, u, indexing="ij")
    parts.append(_grid(np.stack([X, Y, np.full_like(X, 0.002)], -1), albedo=(0.2, 0.3, 0.6), roughness=0.2))
    # walls: long thin horizontal courses (36 m x 0.5 m), end walls in two triangles each
    zs = np.linspace(0.0, Ht, 21, dtype=np.float32)
    for y, flip in ((-Wd / 2, False), (Wd / 2, True)):
        P = np.zeros((2, 21, 3), dtype=np.float32); P[0, :, 0], P[1, :, 0] = (-L / 2, L / 2) if not flip else (L / 2, -L / 2)
        P[:, :, 1] = y; P[:, :, 2] = zs
        parts.append(_grid(P, albedo=stone))
    for x, flip in ((-L / 2, True), (L / 2, False)):
        P = np.zeros((2, 2, 3), dtype=np.float32); P[0, :, 1], P[1, :, 1] = (-Wd / 2, Wd / 2) if not flip else (Wd / 2, -Wd / 2)
        P[:, :, 0] = x; P[:, 0, 2], P[:, 1, 2] = 0.0, Ht
        parts.append(_grid(P, albedo=stone))
    # emissive window panels high on the long walls
    for i in range(6):
        x0 = -15.0 + 6.0 * i
        parts.append(_box((x0, -Wd / 2 + 0.01, 6.0), (x0 + 2.0, -Wd / 2 + 0.05, 8.5), albedo=(1, 1, 1), emission=(1.0, 0.95, 0.85), strength=6.0))
        parts.append(_box((x0, Wd / 2 - 0.05, 6.0), (x0 + 2.0, Wd / 2 - 0.01, 8.5), albedo=(1, 1, 1), emission=(0.85, 0.9, 1.0), strength=4.0))
    # two rows of 12 fluted columns (radius 0.45 with 16 flutes), base and capital boxes
    seg, rings = k(64), k(20)
    th = np.linspace(0.0, 2 * np.pi, seg + 1)
    zc = np.linspace(0.4, 5.6, rings + 1)
    col_x = np.linspace(-15.0, 15.0, 12)
    for cy in (-4.0, 4.0):
        for cx in col_x:
            r = 0.45 + 0.03 * np.cos(16 * th)
            P = np.stack([np.broadcast_to(cx + r * np.cos(th), (rings + 1, seg + 1)), np.broadcast_to(cy + r * np.sin(th), (rings + 1, seg + 1)),
                          np.broadcast_to(zc[:, None], (rings + 1, seg + 1))], -1)
            parts.append(_grid(P, albedo=stone, roughness=0.4))
            parts.append(_box((cx - 0.6, cy - 0.6, 0.0), (cx + 0.6, cy + 0.6, 0.4), albedo=dark))
            parts.append(_box((cx - 0.6, cy - 0.6, 5.6), (cx + 0.6, cy + 0.6, 6.0), albedo=dark))
    # arches between neighbouring columns of a row: a square profile swept along a half circle
    aseg = k(64)
    phi = np.linspace(0.0, np.pi, aseg + 1)
    span = (col_x[1] - col_x[0]) / 2
    for cy in (-4.0, 4.0):
        for i in range(len(col_x) - 1):
            mx = (col_x[i] + col_x[i + 1]) / 2
            prof = np.array([(-0.25, -0.2), (0.25, -0.2), (0.25, 0.2), (-0.25, 0.2), (-0.25, -0.2)])      # (dy, dr)
            rr = np.broadcast_to((span - 0.25) + prof[:, 1][None, :], (aseg + 1, len(prof)))
            P = np.stack([mx - rr * np.cos(phi)[:, None], np.broadcast_to(cy + prof[:, 0][None, :], rr.shape), 6.0 + rr * np.sin(phi)[:, None]], -1)
            parts.append(_grid(P, albedo=stone))
    # barrel vault over the hall
    nv_, nl = k(56), k(220)
    a = np.linspace(0.0, np.pi, nv_ + 1)
    xs = np.linspace(-L / 2, L / 2, nl + 1)
    P = np.stack([np.broadcast_to(xs[:, None], (nl + 1, nv_ + 1)), np.broadcast_to((Wd / 2) * np.cos(a)[None, :], (nl + 1, nv_ + 1)),
                  np.broadcast_to(Ht - 3.0 + 3.0 * np.sin(a)[None, :], (nl + 1, nv_ + 1))], -1)
    parts.append(_grid(P, albedo=(0.8, 0.78, 0.7)))
    # roof beams: 30 boxes 14 m long and 0.15 m thick
    for x in np.linspace(-17.0, 17.0, 30):
        parts.append(_box((x - 0.075, -Wd / 2, 6.6), (x + 0.075, Wd / 2, 6.8), albedo=dark))
    # balustrades along both rows: 2 x 300 thin balusters (0.05 x 0.05 x 0.9) under a 30 m rail
    for cy in (-5.2, 5.2):
        for x in np.linspace(-15.0, 15.0, 300):
            parts.append(_box((x - 0.025, cy - 0.025, 0.0), (x + 0.025, cy + 0.025, 0.9), albedo=dark))
        parts.append(_box((-15.0, cy - 0.05, 0.9), (15.0, cy + 0.05, 1.0), albedo=dark))
    # hanging drapes between the columns: wavy open sheets, long thin cells (0.02 x 0.12)
    du, dv = k(96), k(48)
    uu, vv = np.meshgrid(np.linspace(0.0, 1.0, du + 1), np.linspace(0.0, 1.0, dv + 1), indexing="ij")
    for j, cy in enumerate((-4.0, 4.0)):
        for i in range(0, len(col_x) - 1, 2):
            x0 = col_x[i] + 0.6
            X = x0 + uu * (col_x[i + 1] - col_x[i] - 1.2)
            Yd = cy + 0.18 * np.sin(uu * 9 * np.pi + 0.7 * i) * (1.0 - 0.5 * vv) + 0.03 * np.sin(vv * 11.0 + i + j)
            Z = 5.4 - vv * (3.6 + 0.4 * np.sin(uu * 3.0 + i))
            parts.append(_grid(np.stack([X, Yd, Z], -1), albedo=cloth, roughness=0.9))
    tris = cat(parts)
    cam = camera_look_at((-16.5, -1.2, 1.7), (6.0, 1.5, 3.2), 65.0, 3840.0 / 2160.0)
    return tris, np.zeros(0, dtype=SPHERE), np.zeros(0, dtype=LIGHT), cam
