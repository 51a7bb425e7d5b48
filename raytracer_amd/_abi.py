"""ctypes binding of include/raytracer_amd.h (libraytracer_amd.so).

The library is the product; this file only declares its symbols.  There is no fallback: if the
shared object is missing, import fails loudly (build it with `make -C raytracer_amd/csrc` or
`python -c "import __graft_entry__ as g; g.build()"`).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# RT_AMD_LIB: development hook -- another build of the same library (A/B of kernel variants on one GPU box,
# scripts/ab_variants.sh).  Unset in every test, smoke() and the driver's bench run.
LIB_PATH = os.environ.get("RT_AMD_LIB") or os.path.join(_HERE, "csrc", "libraytracer_amd.so")


class rt_scene_stats(C.Structure):
    _fields_ = [("n_tris", C.c_uint32), ("n_spheres", C.c_uint32), ("n_lights", C.c_uint32),
                ("n_nodes", C.c_uint32), ("n_leaves", C.c_uint32), ("max_depth", C.c_uint32),
                ("node_bytes", C.c_uint32), ("tri_bytes", C.c_uint32),
                ("sah_cost", C.c_float), ("build_ms", C.c_float), ("upload_ms", C.c_float),
                ("device_bytes", C.c_uint64), ("stack_entries", C.c_uint32), ("reserved", C.c_uint32)]


class rt_counters(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("box_tests", C.c_uint64), ("nodes", C.c_uint64),
                ("tri_tests", C.c_uint64), ("sphere_tests", C.c_uint64),
                ("wave_max_nodes", C.c_uint64), ("wave_max_tris", C.c_uint64), ("waves", C.c_uint64),
                ("idle_lane_rounds", C.c_uint64), ("blocked_lane_rounds", C.c_uint64), ("drained_lane_rounds", C.c_uint64),
                ("tri_rounds", C.c_uint64)]


# every symbol include/raytracer_amd.h declares: (name, restype, argtypes)
_P = C.c_void_p
_U32, _U64, _I = C.c_uint32, C.c_uint64, C.c_int
SYMBOLS = [
    ("rt_create", _I, [C.POINTER(_P), C.POINTER(_I), _I]),
    ("rt_destroy", None, [_P]),
    ("rt_last_error", C.c_char_p, [_P]),
    ("rt_set_option", _I, [_P, _I, C.c_int64]),
    ("rt_num_slots", _I, [_P]),
    ("rt_frame_exchange_used", _I, [_P]),
    ("rt_scene_upload", _I, [_P, _P, _U32, _U32, _P, _U32, _U32, _P, _U32, _U32]),
    ("rt_scene_stats_get", _I, [_P, C.POINTER(rt_scene_stats)]),
    ("rt_bvh_build_host", _I, [_P, _U32, _U32, _I, _I, _I, _I, C.POINTER(rt_scene_stats), _P, _U64, _P]),
    ("rt_set_camera", _I, [_P, _P]),
    ("rt_host_alloc", _I, [_P, C.c_size_t, C.POINTER(_P)]),
    ("rt_host_free", _I, [_P, _P]),
    ("rt_trace_primary", _I, [_P, _U32, _U32, _P]),
    ("rt_trace", _I, [_P, _P, _U64, _P]),
    ("rt_occluded", _I, [_P, _P, _U64, _P]),
    ("rt_trace_primary_dev", _I, [_P, _I, _U32, _U32, _U32, _U32, _U32, _U32, _I, _P, _P, _P]),
    ("rt_trace_primary_stripes_dev", _I, [_P, _I, _U32, _U32, _U32, _U32, _U32, _U32, _U32, _U32, _I, _P, _P, _P]),
    ("rt_trace_dev", _I, [_P, _I, _P, _U64, _P, _P]),
    ("rt_trace_dev_noalloc", _I, [_P, _I, _P, _U64, _P, _P]),
    ("rt_occluded_dev", _I, [_P, _I, _P, _U64, _P, _P]),
    ("rt_shadow_dev", _I, [_P, _I, _P, _P, _U64, C.POINTER(C.c_float), _P, _P, _P]),
    ("rt_bounce_dev", _I, [_P, _I, _P, _P, _U64, _U32, _U32, _P, _P]),
    ("rt_path_init_dev", _I, [_P, _I, _U64, _P, _P, _P]),
    ("rt_shade_dev", _I, [_P, _I, _P, _P, _U64, _U32, _U32, _P, _P, _P, _P]),
    ("rt_shade_stripes_dev", _I, [_P, _I, _P, _P, _U64, _U32, _U32, _U32, _U32, _U32, _P, _P, _P, _P]),
    ("rt_primary_shade_stripes_dev", _I, [_P, _I, _U32, _U32, _U32, _U32, _U32, _U32, _U32, _U32, _I, _U32, _P, _P, _P, _P, _P, _P]),
    ("rt_shade_resolve_dev", _I, [_P, _I, _P, _P, _U64, _U32, _P, _P, _P, _P, _P]),
    ("rt_resolve_dev", _I, [_P, _I, _P, _U64, _U32, _P, _P, _P]),
    ("rt_render", _I, [_P, _U32, _U32, _U32, _U32, _U32, _P]),
    ("rt_counters_get", _I, [_P, _I, C.POINTER(rt_counters), _I]),
    ("rt_sync", _I, [_P, _I]),
    ("rt_gen_cornell_triangles", _U32, [_P, _U32]),
    ("rt_gen_cornell_spheres", _U32, [_P, _U32]),
    ("rt_gen_sphere_geometry", _U32, [C.POINTER(C.c_float), C.c_float, _U32, _U32, C.POINTER(C.c_float), _P, _U32]),
    ("rt_gen_soup", _U32, [_U64, _U32, C.c_float, _P, _U32]),
    ("rt_gen_random_rays", None, [_U64, _U64, _U64, _P]),
    ("rt_camera_look_at", None, [C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_float, C.c_float, _P]),
    ("rt_camera_move", None, [_P, C.c_float, C.c_float, C.c_float]),
    ("rt_camera_rotate", None, [_P, C.c_float, C.c_float]),
    ("rt_load_obj", _I, [C.c_char_p, _P, _P, _U64, C.POINTER(_U64)]),
]

_lib = None


def load():
    """Load libraytracer_amd.so and bind every declared symbol.  Raises if the library is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: the HIP extension is not built. "
            "Run `make -C raytracer_amd/csrc` (hipcc cross-compiles gfx950 without a GPU). "
            "There is no CPU fallback.")
    # One HIP runtime per process: PyTorch-ROCm wheels bundle their own libamdhip64.so (soname
    # libamdhip64.so.7, the one this library needs too) but link it as "libamdhip64.so".  If this library were
    # loaded first it would pull /opt/rocm's copy, torch would then load a second runtime, and whichever
    # initialises second cannot open the GPU.  Loading torch first makes the dynamic linker reuse its copy.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    for name, res, args in SYMBOLS:
        fn = getattr(lib, name)          # AttributeError if the export is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
