"""raytracer_amd -- MI355X-native ray-intersection engine for the PaulThythy/raytracer hot path.

Host-side mirror of the reference's renderer seams (src/vulkan/VkRenderer.cpp: createData :164-327,
updateUniformBuffer :855-862, the draw at :1123) over the C ABI in include/raytracer_amd.h.  The
compute lives in raytracer_amd/csrc (hand-written HIP for gfx950); this package is plumbing.
Errors follow the reference's convention (std::runtime_error caught in main, src/main.cpp:8-14):
any non-zero ABI return raises RuntimeError(rt_last_error()).
"""
import ctypes as C
import os

import numpy as np

from . import _abi
from ._abi import rt_counters, rt_scene_stats

__all__ = ["Renderer", "TRIANGLE", "SPHERE", "LIGHT", "CAMERA", "RAY", "HIT", "MATERIAL", "VERTEX",
           "MISS", "SPHERE_BIT", "camera_look_at", "camera_move", "camera_rotate", "default_camera", "lib"]

MISS = 0xFFFFFFFF
SPHERE_BIT = 0x80000000

# numpy twins of the reference's host structs (byte layouts: raytracer_amd/host/rt_types.hpp)
VERTEX = np.dtype({"names": ["position", "normal"], "formats": ["3f4", "3f4"], "offsets": [0, 16], "itemsize": 32})
MATERIAL = np.dtype({"names": ["albedo", "emission", "emissionStrength", "roughness", "metallic"],
                     "formats": ["3f4", "3f4", "f4", "f4", "f4"], "offsets": [0, 16, 28, 32, 36], "itemsize": 48})
TRIANGLE = np.dtype({"names": ["v0", "v1", "v2", "material"], "formats": [VERTEX, VERTEX, VERTEX, MATERIAL],
                     "offsets": [0, 32, 64, 96], "itemsize": 144})
SPHERE = np.dtype({"names": ["center", "radius", "material"], "formats": ["3f4", "f4", MATERIAL],
                   "offsets": [0, 12, 16], "itemsize": 64})
LIGHT = np.dtype({"names": ["position", "color", "intensity"], "formats": ["3f4", "3f4", "f4"],
                  "offsets": [0, 16, 28], "itemsize": 32})
CAMERA = np.dtype({"names": ["position", "lookAt", "front", "up", "right", "worldUp", "fov", "aspectRatio",
                             "nearPlane", "farPlane"],
                   "formats": ["3f4"] * 6 + ["f4"] * 4, "offsets": [0, 16, 32, 48, 64, 80, 92, 96, 100, 104],
                   "itemsize": 112})
RAY = np.dtype([("o", "3f4"), ("tmin", "f4"), ("d", "3f4"), ("tmax", "f4")])
HIT = np.dtype([("t", "f4"), ("u", "f4"), ("v", "f4"), ("prim", "u4")])
assert RAY.itemsize == 32 and HIT.itemsize == 16

OPT_TRAVERSAL, OPT_BVH_THREADS, OPT_COUNTERS, OPT_PAD_LOG2, OPT_MAX_LEAF, OPT_STRIP, OPT_REFILL, OPT_RAY_BINNING, OPT_LDS_STACK, OPT_PRIMARY, OPT_RAY_PAD = 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11
OPT_SAH_BINS, OPT_SAH_TRAV = 12, 13
OPT_RAY_PACKETS = 15
OPT_FRAME_EXCHANGE = 16
OPT_FUSED_PATHS = 18
OPT_RAY_QUEUE = 19
NOALLOC_MAX_RAYS = 1 << 22
TRAVERSAL_AUTO, TRAVERSAL_BRUTE, TRAVERSAL_BVH2, TRAVERSAL_BVH4Q = 0, 1, 2, 3
TRAVERSAL_BVH = TRAVERSAL_BVH4Q


def lib():
    return _abi.load()


def _ptr(x):
    """Raw address of a numpy array, a torch tensor (data_ptr) or an int; None -> NULL."""
    if x is None:
        return None
    if isinstance(x, int):
        return C.c_void_p(x)
    if isinstance(x, np.ndarray):
        return C.c_void_p(x.ctypes.data)
    if hasattr(x, "data_ptr"):
        return C.c_void_p(x.data_ptr())
    raise TypeError(f"cannot take the address of {type(x)}")


def _stream_ptr(stream):
    if stream is None:
        return None
    if isinstance(stream, int):
        return C.c_void_p(stream)
    return C.c_void_p(stream.cuda_stream)      # torch.cuda.Stream


def camera_look_at(pos, look_at, fov_deg, aspect):
    """Camera::Camera + updateCameraVectors (src/application/Camera.h:24-41) -> 112-byte UBO record."""
    ubo = np.zeros(1, dtype=CAMERA)
    p = (C.c_float * 3)(*pos)
    l = (C.c_float * 3)(*look_at)
    lib().rt_camera_look_at(p, l, C.c_float(fov_deg), C.c_float(aspect), _ptr(ubo))
    return ubo


def camera_move(ubo, forward=0.0, right=0.0, up=0.0):
    """Camera::moveForward / moveRight / moveUp (src/application/Camera.h:56-75) on a CAMERA record; returns a new record."""
    out = np.ascontiguousarray(ubo).copy()
    lib().rt_camera_move(_ptr(out), C.c_float(forward), C.c_float(right), C.c_float(up))
    return out


def camera_rotate(ubo, around_up_deg=0.0, around_right_deg=0.0):
    """Camera::rotateAroundUp / rotateAroundRight (src/application/Camera.h:77-91), degrees; returns a new record."""
    out = np.ascontiguousarray(ubo).copy()
    lib().rt_camera_rotate(_ptr(out), C.c_float(around_up_deg), C.c_float(around_right_deg))
    return out


def load_obj(path, material=None):
    """rt_load_obj: Wavefront OBJ -> TRIANGLE records with one material (a MATERIAL-dtype record or None)."""
    n = C.c_uint64(0)
    m = None if material is None else np.ascontiguousarray(material)
    L = lib()
    rc = L.rt_load_obj(os.fsencode(path), _ptr(m), None, 0, C.byref(n))
    if rc != 0:
        raise RuntimeError(f"rt_load_obj({path!r}) failed ({rc})")
    tris = np.zeros(n.value, dtype=TRIANGLE)
    if n.value:
        rc = L.rt_load_obj(os.fsencode(path), _ptr(m), _ptr(tris), n.value, C.byref(n))
        if rc != 0:
            raise RuntimeError(f"rt_load_obj({path!r}) failed ({rc})")
    return tris


def default_camera(aspect=1200.0 / 1000.0):
    """VkRenderer::VkRenderer (src/vulkan/VkRenderer.cpp:3-12)."""
    return camera_look_at((0.0, 4.0, 1.0), (0.0, 0.0, 1.0), 45.0, aspect)


def bvh_build_host(triangles, layout=TRAVERSAL_BVH2, max_leaf=4, pad_log2=15, threads=0, want_nodes=True):
    """Host-only BVH build + flatten (rt_bvh_build_host): (stats dict, nodes uint32[n_nodes,16], tri_order)."""
    t = np.ascontiguousarray(triangles)
    st = rt_scene_stats()
    L = lib()
    rc = L.rt_bvh_build_host(_ptr(t) if len(t) else None, len(t), TRIANGLE.itemsize, layout, max_leaf, pad_log2, threads,
                             C.byref(st), None, 0, None)
    if rc != 0:
        raise RuntimeError(f"rt_bvh_build_host failed ({rc})")
    stats = {f: getattr(st, f) for f, _ in st._fields_}
    if not want_nodes:
        return stats, None, None
    nodes = np.zeros((max(st.n_nodes, 1), 16), dtype=np.uint32)
    order = np.zeros(max(len(t), 1), dtype=np.uint32)
    rc = L.rt_bvh_build_host(_ptr(t) if len(t) else None, len(t), TRIANGLE.itemsize, layout, max_leaf, pad_log2, threads,
                             C.byref(st), _ptr(nodes), nodes.size, _ptr(order))
    if rc != 0:
        raise RuntimeError(f"rt_bvh_build_host failed ({rc})")
    return stats, nodes[: st.n_nodes], order[: len(t)]


class Renderer:
    """Owner of an rt_ctx: the compute-side stand-in for VkRenderer's scene/camera/draw seams."""

    def __init__(self, device_ids=(0,)):
        self._lib = lib()
        self._ctx = C.c_void_p()
        ids = (C.c_int * len(device_ids))(*device_ids)
        rc = self._lib.rt_create(C.byref(self._ctx), ids, len(device_ids))
        if rc != 0:
            self._ctx = C.c_void_p()
            raise RuntimeError(f"rt_create failed ({rc}): no usable HIP device or bad device id")
        self._keep = None

    # -- plumbing ---------------------------------------------------------------------------------
    def _check(self, rc):
        if rc != 0:
            raise RuntimeError(f"{self._lib.rt_last_error(self._ctx).decode()} (code {rc})")

    def close(self):
        if getattr(self, "_ctx", None) and self._ctx.value:
            self._lib.rt_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    @property
    def num_slots(self):
        return self._lib.rt_num_slots(self._ctx)

    @property
    def frame_exchange_used(self):
        """1 = the last multi-slot frame was gathered over RCCL, 2 = by peer copies, 0 = none yet."""
        return self._lib.rt_frame_exchange_used(self._ctx)

    def set_option(self, key, value):
        self._check(self._lib.rt_set_option(self._ctx, key, int(value)))

    # -- createData (src/vulkan/VkRenderer.cpp:164-327) --------------------------------------------
    def create_data(self, triangles=None, spheres=None, lights=None):
        def prep(a, dt):
            if a is None:
                return np.zeros(0, dtype=dt)
            a = np.ascontiguousarray(a)
            if a.dtype != dt:
                raise RuntimeError(f"expected dtype with itemsize {dt.itemsize}, got {a.dtype}")
            return a
        t, s, l = prep(triangles, TRIANGLE), prep(spheres, SPHERE), prep(lights, LIGHT)
        self._check(self._lib.rt_scene_upload(self._ctx, _ptr(t) if len(t) else None, len(t), TRIANGLE.itemsize,
                                              _ptr(s) if len(s) else None, len(s), SPHERE.itemsize,
                                              _ptr(l) if len(l) else None, len(l), LIGHT.itemsize))
        return self.scene_stats()

    def scene_stats(self):
        st = rt_scene_stats()
        self._check(self._lib.rt_scene_stats_get(self._ctx, C.byref(st)))
        return {f: getattr(st, f) for f, _ in st._fields_}

    # -- updateUniformBuffer (src/vulkan/VkRenderer.cpp:855-862) -----------------------------------
    def update_camera(self, ubo):
        ubo = np.ascontiguousarray(ubo)
        if ubo.nbytes != 112:
            raise RuntimeError("camera UBO must be 112 bytes")
        self._check(self._lib.rt_set_camera(self._ctx, _ptr(ubo)))

    def host_alloc(self, n, dtype):
        """rt_host_alloc: a pinned numpy array of n records; trace / occluded DMA from / to it without staging copies.
        Valid until host_free(array) or close()."""
        dt = np.dtype(dtype)
        p = C.c_void_p()
        self._check(self._lib.rt_host_alloc(self._ctx, max(n * dt.itemsize, 1), C.byref(p)))
        buf = (C.c_char * (n * dt.itemsize)).from_address(p.value)
        return np.frombuffer(buf, dtype=dt, count=n)

    def host_free(self, array):
        self._check(self._lib.rt_host_free(self._ctx, _ptr(array)))

    # -- the draw (src/vulkan/VkRenderer.cpp:1123), host buffers ------------------------------------
    def trace_primary(self, w, h):
        out = np.empty(w * h, dtype=HIT)
        self._check(self._lib.rt_trace_primary(self._ctx, w, h, _ptr(out)))
        return out.reshape(h, w)

    def trace(self, rays, out=None):
        rays = np.ascontiguousarray(rays)
        if rays.dtype != RAY:
            raise RuntimeError("rays must have dtype RAY (32 B)")
        if out is None:
            out = np.empty(len(rays), dtype=HIT)
        elif out.dtype != HIT or len(out) != len(rays) or not out.flags.c_contiguous:
            raise RuntimeError("out must be a contiguous HIT array as long as rays")
        self._check(self._lib.rt_trace(self._ctx, _ptr(rays) if len(rays) else None, len(rays),
                                       _ptr(out) if len(rays) else None))
        return out

    def occluded(self, rays):
        rays = np.ascontiguousarray(rays)
        if rays.dtype != RAY:
            raise RuntimeError("rays must have dtype RAY (32 B)")
        out = np.empty(len(rays), dtype=np.uint8)
        self._check(self._lib.rt_occluded(self._ctx, _ptr(rays) if len(rays) else None, len(rays),
                                          _ptr(out) if len(rays) else None))
        return out

    # -- device-resident stages (pointers: torch tensors on the slot's device, or raw ints) ---------
    def trace_primary_dev(self, w, h, hits, rays=None, y0=0, rows=None, spp=1, seed=0, jitter=False, slot=0,
                          stream=None):
        rows = h - y0 if rows is None else rows
        self._check(self._lib.rt_trace_primary_dev(self._ctx, slot, w, h, y0, rows, spp, seed, int(bool(jitter)),
                                                   _ptr(rays), _ptr(hits), _stream_ptr(stream)))

    def trace_primary_stripes_dev(self, w, h, y0, rows, stripe_h, stripe_step, hits, rays=None, spp=1, seed=0,
                                  jitter=False, slot=0, stream=None):
        self._check(self._lib.rt_trace_primary_stripes_dev(self._ctx, slot, w, h, y0, rows, stripe_h, stripe_step,
                                                           spp, seed, int(bool(jitter)), _ptr(rays), _ptr(hits),
                                                           _stream_ptr(stream)))

    def trace_dev(self, rays, n, hits, slot=0, stream=None):
        self._check(self._lib.rt_trace_dev(self._ctx, slot, _ptr(rays), n, _ptr(hits), _stream_ptr(stream)))

    def trace_dev_noalloc(self, rays, n, hits, slot=0, stream=None):
        """rt_trace_dev_noalloc: same records as trace_dev, no device allocation (n <= NOALLOC_MAX_RAYS)."""
        self._check(self._lib.rt_trace_dev_noalloc(self._ctx, slot, _ptr(rays), n, _ptr(hits), _stream_ptr(stream)))

    def occluded_dev(self, rays, n, out, slot=0, stream=None):
        self._check(self._lib.rt_occluded_dev(self._ctx, slot, _ptr(rays), n, _ptr(out), _stream_ptr(stream)))

    def shadow_dev(self, rays, hits, n, light_pos, lit, shadow_rays=None, slot=0, stream=None):
        lp = (C.c_float * 3)(*light_pos)
        self._check(self._lib.rt_shadow_dev(self._ctx, slot, _ptr(rays), _ptr(hits), n, lp, _ptr(shadow_rays),
                                            _ptr(lit), _stream_ptr(stream)))

    def bounce_dev(self, rays, hits, n, out_rays, key0=0, seed=0, slot=0, stream=None):
        self._check(self._lib.rt_bounce_dev(self._ctx, slot, _ptr(rays), _ptr(hits), n, key0, seed, _ptr(out_rays),
                                            _stream_ptr(stream)))

    def path_init_dev(self, n, throughput, radiance, slot=0, stream=None):
        self._check(self._lib.rt_path_init_dev(self._ctx, slot, n, _ptr(throughput), _ptr(radiance), _stream_ptr(stream)))

    def shade_dev(self, rays, hits, n, throughput, radiance, out_rays, key0=0, seed=0, slot=0, stream=None):
        self._check(self._lib.rt_shade_dev(self._ctx, slot, _ptr(rays), _ptr(hits), n, key0, seed, _ptr(throughput),
                                           _ptr(radiance), _ptr(out_rays), _stream_ptr(stream)))

    def shade_stripes_dev(self, rays, hits, n, throughput, radiance, out_rays, key0, row_len, stripe_h, stripe_step,
                          seed=0, slot=0, stream=None):
        self._check(self._lib.rt_shade_stripes_dev(self._ctx, slot, _ptr(rays), _ptr(hits), n, key0, row_len, stripe_h,
                                                   stripe_step, seed, _ptr(throughput), _ptr(radiance), _ptr(out_rays),
                                                   _stream_ptr(stream)))

    def primary_shade_stripes_dev(self, w, h, y0, rows, stripe_h, stripe_step, throughput, radiance, next_rays, spp=1,
                                  seed=0, jitter=False, shade_seed=0, rays=None, hits=None, slot=0, stream=None):
        """Segment 0 of the path loop in one launch (rt_primary_shade_stripes_dev)."""
        self._check(self._lib.rt_primary_shade_stripes_dev(self._ctx, slot, w, h, y0, rows, stripe_h, stripe_step, spp, seed,
                                                           int(bool(jitter)), shade_seed, _ptr(rays), _ptr(hits),
                                                           _ptr(throughput), _ptr(radiance), _ptr(next_rays),
                                                           _stream_ptr(stream)))

    def shade_resolve_dev(self, rays, hits, n, spp, throughput, radiance, linear=None, rgba8=None, slot=0, stream=None):
        """Last path segment + resolve in one launch (rt_shade_resolve_dev); `radiance` is read, not updated."""
        self._check(self._lib.rt_shade_resolve_dev(self._ctx, slot, _ptr(rays), _ptr(hits), n, spp, _ptr(throughput),
                                                   _ptr(radiance), _ptr(linear), _ptr(rgba8), _stream_ptr(stream)))

    def resolve_dev(self, radiance, n_pixels, spp, linear=None, rgba8=None, slot=0, stream=None):
        self._check(self._lib.rt_resolve_dev(self._ctx, slot, _ptr(radiance), n_pixels, spp, _ptr(linear), _ptr(rgba8),
                                             _stream_ptr(stream)))

    def render(self, w, h, spp=10, bounces=8, seed=1):
        """frag.glsl:main for a whole frame (defaults: the shader's SAMPLES / BOUNCES) -> (h, w, 4) uint8."""
        out = np.empty((h, w, 4), dtype=np.uint8)
        self._check(self._lib.rt_render(self._ctx, w, h, spp, bounces, seed, _ptr(out)))
        return out

    def counters(self, slot=0, reset=True):
        c = rt_counters()
        self._check(self._lib.rt_counters_get(self._ctx, slot, C.byref(c), int(reset)))
        return {f: getattr(c, f) for f, _ in c._fields_}

    def sync(self, slot=0):
        self._check(self._lib.rt_sync(self._ctx, slot))
