"""ctypes wrapper of oracle/liboracle.so -- TEST INFRASTRUCTURE (see oracle/oracle.h).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.  It is
the checker for the HIP path, never part of it: nothing under raytracer_amd/ imports it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liboracle.so")

RAY = np.dtype([("o", "3f4"), ("tmin", "f4"), ("d", "3f4"), ("tmax", "f4")])
HIT = np.dtype([("t", "f4"), ("u", "f4"), ("v", "f4"), ("prim", "u4")])
MISS = 0xFFFFFFFF
SPHERE_BIT = 0x80000000


class orc_scene(C.Structure):
    _fields_ = [("tris", C.c_void_p), ("n_tris", C.c_uint32), ("tri_stride", C.c_uint32),
                ("spheres", C.c_void_p), ("n_spheres", C.c_uint32), ("sphere_stride", C.c_uint32),
                ("lights", C.c_void_p), ("n_lights", C.c_uint32), ("light_stride", C.c_uint32)]


_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = C.CDLL(LIB_PATH)
        P, U32, U64, F = C.c_void_p, C.c_uint32, C.c_uint64, C.c_float
        FP = C.POINTER(C.c_float)
        sig = {
            "orc_ray_triangle": (C.c_int, [P, FP, FP, FP, FP, FP, FP]),
            "orc_ray_sphere": (C.c_int, [P, FP, F, FP]),
            "orc_trace_one": (None, [P, P, P]),
            "orc_pixel_uv": (None, [U32, U32, U32, U32, FP]),
            "orc_camera_ray": (None, [P, U32, U32, U32, U32, U32, U32, C.c_int, P]),
            "orc_primary_rays": (None, [P, U32, U32, U32, U32, C.c_int, P]),
            "orc_trace": (None, [P, P, U64, P]),
            "orc_occluded": (None, [P, P, U64, P]),
            "orc_trace_bvh2": (None, [P, P, U32, P, P, U64, P]),
            "orc_hit_point": (C.c_int, [P, P, P, FP, FP]),
            "orc_shadow_lit": (None, [P, P, P, U64, FP, P, P]),
            "orc_bounce": (None, [P, P, P, U64, U32, P]),
            "orc_hemisphere": (None, [FP, F, F, FP]),
            "orc_hemisphere_libm": (None, [FP, F, F, FP]),
            "orc_rand": (F, [U32, U32, U32]),
            "orc_num_threads": (C.c_int, []),
            "orc_brdf": (None, [FP, FP, FP, FP, FP]),
            "orc_shade": (None, [P, P, P, U64, U32, U32, P, P, P]),
            "orc_resolve": (None, [P, U64, U32, P, P]),
            "orc_render": (None, [P, P, U32, U32, U32, U32, U32, P, P]),
        }
        for name, (res, args) in sig.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


def _p(a):
    return C.c_void_p(a.ctypes.data) if a is not None and len(a) else None


class Scene:
    """View of the reference-layout arrays (144-B triangles, 64-B spheres, 32-B lights)."""

    def __init__(self, tris=None, spheres=None, lights=None):
        self.tris = np.ascontiguousarray(tris) if tris is not None else np.zeros(0, dtype=np.uint8)
        self.spheres = np.ascontiguousarray(spheres) if spheres is not None else np.zeros(0, dtype=np.uint8)
        self.lights = np.ascontiguousarray(lights) if lights is not None else np.zeros(0, dtype=np.uint8)
        s = orc_scene()
        s.tris, s.n_tris, s.tri_stride = _p(self.tris), len(self.tris), 144
        s.spheres, s.n_spheres, s.sphere_stride = _p(self.spheres), len(self.spheres), 64
        s.lights, s.n_lights, s.light_stride = _p(self.lights), len(self.lights), 32
        self.c = s

    def ref(self):
        return C.byref(self.c)


def trace(scene, rays):
    rays = np.ascontiguousarray(rays, dtype=RAY)
    out = np.empty(len(rays), dtype=HIT)
    if len(rays):
        lib().orc_trace(scene.ref(), _p(rays), len(rays), _p(out))
    return out


def trace_bvh2(scene, nodes, tri_order, rays):
    """SURVEY 8(d) 'build CPU BVH, not reference': closest hit over a flattened BVH2 (uint32 [n_nodes, 16] as emitted by
    the product's rt_bvh_build_host, passed in as data) with the oracle's own primitive tests."""
    rays = np.ascontiguousarray(rays, dtype=RAY)
    nodes = np.ascontiguousarray(nodes, dtype=np.uint32)
    order = np.ascontiguousarray(tri_order, dtype=np.uint32)
    out = np.empty(len(rays), dtype=HIT)
    n_nodes = 0 if len(scene.tris) == 0 else len(nodes)
    if len(rays):
        lib().orc_trace_bvh2(scene.ref(), _p(nodes), n_nodes, _p(order), _p(rays), len(rays), _p(out))
    return out


def occluded(scene, rays):
    rays = np.ascontiguousarray(rays, dtype=RAY)
    out = np.empty(len(rays), dtype=np.uint8)
    if len(rays):
        lib().orc_occluded(scene.ref(), _p(rays), len(rays), _p(out))
    return out


def primary_rays(camera_ubo, w, h, spp=1, seed=0, jitter=False):
    cam = np.ascontiguousarray(camera_ubo)
    out = np.empty(w * h * spp, dtype=RAY)
    lib().orc_primary_rays(C.c_void_p(cam.ctypes.data), w, h, spp, seed, int(bool(jitter)), _p(out))
    return out


def trace_primary(scene, camera_ubo, w, h):
    return trace(scene, primary_rays(camera_ubo, w, h)).reshape(h, w)


def shadow_lit(scene, rays, hits, light_pos, want_rays=False):
    rays = np.ascontiguousarray(rays, dtype=RAY)
    hits = np.ascontiguousarray(hits, dtype=HIT)
    lit = np.empty(len(rays), dtype=np.uint8)
    sr = np.empty(len(rays), dtype=RAY) if want_rays else None
    lp = (C.c_float * 3)(*light_pos)
    if len(rays):
        lib().orc_shadow_lit(scene.ref(), _p(rays), _p(hits), len(rays), lp, _p(sr) if want_rays else None, _p(lit))
    return (lit, sr) if want_rays else lit


def bounce(scene, rays, hits, seed=0):
    rays = np.ascontiguousarray(rays, dtype=RAY)
    hits = np.ascontiguousarray(hits, dtype=HIT)
    out = np.empty(len(rays), dtype=RAY)
    if len(rays):
        lib().orc_bounce(scene.ref(), _p(rays), _p(hits), len(rays), seed, _p(out))
    return out


def shade(scene, rays, hits, throughput, radiance, key0=0, seed=0):
    """One bounce of frag.glsl:310-354; throughput/radiance ([n,4] float32) are updated in place."""
    rays = np.ascontiguousarray(rays, dtype=RAY)
    hits = np.ascontiguousarray(hits, dtype=HIT)
    assert throughput.dtype == np.float32 and radiance.dtype == np.float32 and throughput.flags.c_contiguous
    out = np.empty(len(rays), dtype=RAY)
    if len(rays):
        lib().orc_shade(scene.ref(), _p(rays), _p(hits), len(rays), key0, seed, _p(throughput), _p(radiance), _p(out))
    return out


def resolve(radiance, n_pixels, spp):
    lin = np.empty((n_pixels, 4), dtype=np.float32)
    rgba = np.empty((n_pixels, 4), dtype=np.uint8)
    lib().orc_resolve(_p(radiance), n_pixels, spp, _p(lin), _p(rgba))
    return lin, rgba


def render(scene, camera_ubo, w, h, spp, bounces, seed):
    cam = np.ascontiguousarray(camera_ubo)
    lin = np.empty((h, w, 4), dtype=np.float32)
    rgba = np.empty((h, w, 4), dtype=np.uint8)
    lib().orc_render(scene.ref(), C.c_void_p(cam.ctypes.data), w, h, spp, bounces, seed, _p(lin.reshape(-1)), _p(rgba.reshape(-1)))
    return lin, rgba


def brdf(material12, n, v, l):
    f = lambda a: np.ascontiguousarray(a, dtype=np.float32).ctypes.data_as(C.POINTER(C.c_float))
    out = np.zeros(3, dtype=np.float32)
    m, nn, vv, ll = (np.ascontiguousarray(x, dtype=np.float32) for x in (material12, n, v, l))
    lib().orc_brdf(m.ctypes.data_as(C.POINTER(C.c_float)), nn.ctypes.data_as(C.POINTER(C.c_float)),
                   vv.ctypes.data_as(C.POINTER(C.c_float)), ll.ctypes.data_as(C.POINTER(C.c_float)),
                   out.ctypes.data_as(C.POINTER(C.c_float)))
    return out


def num_threads():
    return lib().orc_num_threads()
