"""ctypes loader for libdr_mi355x.so (the C ABI in include/dr_mi355x.h).

There is no CPU fallback: if the HIP library is missing the import of any operator fails loudly.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# DR_MI355X_LIB: load another build of the same C ABI (used for within-box A/B timing of kernel variants)
LIB_PATH = os.environ.get("DR_MI355X_LIB") or os.path.join(_HERE, "libdr_mi355x.so")

DR_OK = 0
ERR_NAMES = {1: "DR_ERR_ARG", 2: "DR_ERR_PROTOCOL", 3: "DR_ERR_DEVICE", 4: "DR_ERR_IO", 5: "DR_ERR_CAPACITY",
             6: "DR_ERR_UNSUPPORTED"}


class DrError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("%s: %s" % (ERR_NAMES.get(code, code), msg))
        self.code = code


class FusionOptions(C.Structure):
    """struct DrFusionOptions, dr_fusion.h:18-36 (same field order as drf_options_t)."""
    _fields_ = [("voxel_size", C.c_float), ("num_buckets", C.c_int), ("bucket_size", C.c_int),
                ("num_blocks", C.c_int), ("block_size", C.c_int), ("max_sdf_weight", C.c_int),
                ("truncation_distance", C.c_float), ("max_sensor_depth", C.c_float),
                ("min_sensor_depth", C.c_float), ("num_render_streams", C.c_int),
                ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float),
                ("height", C.c_int), ("width", C.c_int)]


_lib = None
u8p, f32p, vp = C.POINTER(C.c_uint8), C.POINTER(C.c_float), C.c_void_p
f64p = C.POINTER(C.c_double)

SIGNATURES = {
    "dr_last_error": (C.c_char_p, []),
    "dr_version": (C.c_char_p, []),
    "drm_create": (C.c_int, [C.c_char_p, C.c_int, C.POINTER(vp)]),
    "drm_destroy": (None, [vp]),
    "drm_call_async": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(u8p), f32p, C.POINTER(f32p),
                                 C.c_float, C.c_float, C.c_float]),
    "drm_ready": (C.c_int, [vp]),
    "drm_wait": (C.c_int, [vp]),
    "drm_get_result": (C.c_int, [vp, f32p, f32p, f32p, f32p]),
    "drm_get_result_view": (C.c_int, [vp, C.POINTER(f32p), C.POINTER(f32p), C.POINTER(f32p), C.POINTER(f32p)]),
    "drm_host_alloc": (vp, [C.c_size_t]),
    "drm_host_free": (None, [vp]),
    "drm_upload": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(u8p), f32p, C.POINTER(f32p),
                             C.c_float, C.c_float, C.c_float]),
    "drm_forward": (C.c_int, [vp, C.c_int, f32p]),
    "drm_download": (C.c_int, [vp, f32p, f32p, f32p, f32p]),
    "drm_get_stage_output": (C.c_int, [vp, C.c_int, f32p, f32p]),
    "drm_get_tensor": (C.c_int, [vp, C.c_char_p, f32p, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_int)]),
    "drm_profile": (C.c_int, [vp, C.c_char_p, C.c_size_t, f32p, C.c_int, C.POINTER(C.c_int)]),
    "drm_work": (C.c_int, [vp, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "drm_debug_conv": (C.c_int, [C.c_int, f32p, C.c_int, C.c_int, C.c_int, C.c_int, f32p, C.c_int, C.c_int, C.c_int,
                                 C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, f32p, f32p, C.c_int, f32p, C.c_int,
                                 f32p, C.POINTER(C.c_int)]),
    "drm_debug_tail": (C.c_int, [C.c_int, f32p, f32p, f32p, f32p, f32p, f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, f32p]),
    "drm_autotune": (C.c_int, [vp, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "drm_set_view_shard": (C.c_int, [vp, C.c_int]),
    "drm_forward_phase": (C.c_int, [vp, C.c_int]),
    "drm_comm_available": (C.c_int, []),
    "drm_comm_unique_id": (C.c_int, [u8p]),
    "drm_comm_init": (C.c_int, [vp, C.c_int, C.c_int, u8p]),
    "drm_comm_destroy": (C.c_int, [vp]),
    "drm_comm_count": (C.c_int, [vp, C.POINTER(C.c_int)]),
    "drm_device_tensor": (C.c_int, [vp, C.c_char_p, C.POINTER(vp), C.POINTER(C.c_size_t)]),
    "drf_create": (C.c_int, [C.POINTER(FusionOptions), C.c_int, C.POINTER(vp)]),
    "drf_destroy": (None, [vp]),
    "drf_integrate_scan_async": (C.c_int, [vp, u8p, f32p, f32p]),
    "drf_render_async": (C.c_int, [vp, C.POINTER(f32p), C.c_int]),
    "drf_get_render_result": (C.c_int, [vp, C.POINTER(u8p), C.POINTER(f32p), C.c_int]),
    "drf_extract_mesh_async": (C.c_int, [vp, f32p, f32p]),
    "drf_get_mesh_sync": (C.c_int, [vp, C.c_size_t, C.POINTER(C.c_size_t), f32p, f32p]),
    "drf_mesh_num_triangles": (C.c_int, [vp, C.POINTER(C.c_size_t)]),
    "drf_save_mesh": (C.c_int, [vp, C.c_char_p, f32p, f32p]),
    "drf_get_render_device": (C.c_int, [vp, C.c_int, C.POINTER(vp), C.POINTER(vp)]),
    "drf_synchronize": (C.c_int, [vp]),
    "drf_stats": (C.c_int, [vp, C.POINTER(C.c_uint64)]),
    "drf_export_blocks": (C.c_int, [vp, C.c_int, C.POINTER(C.c_int32), u8p, C.POINTER(C.c_int)]),
    "drf_fast_div_status": (C.c_int, [vp, C.POINTER(C.c_int), C.POINTER(C.c_uint64)]),
    "drf_test_combine": (C.c_int, [vp, C.c_size_t, u8p, u8p, C.c_int, u8p]),
    "drf_integrate_device": (C.c_int, [vp, vp, vp, f32p]),
    "drt_create": (C.c_int, [C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, C.POINTER(vp)]),
    "drt_destroy": (None, [vp]),
    "drt_set_k": (C.c_int, [vp, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float]),
    "drt_init": (C.c_int, [vp, C.c_int]),
    "drt_set_reference": (C.c_int, [vp, C.c_int, f32p, f32p, f32p, f32p, C.c_float, f64p]),
    "drt_set_new": (C.c_int, [vp, f32p]),
    "drt_calc_res": (C.c_int, [vp, f64p, C.c_float, f64p, C.c_float, f64p, f64p]),
    "drt_calc_g": (C.c_int, [vp, f64p, f64p, C.c_float, f64p, f64p]),
    "drt_append_dense_reference": (C.c_int, [vp, vp, f32p, f32p, C.c_int, C.c_int, vp, vp, C.c_int, C.POINTER(C.c_int)]),
    "drt_synchronize": (C.c_int, [vp]),
    "drt_start_timing": (C.c_int, [vp]),
    "drt_end_timing_ms": (C.c_int, [vp, C.POINTER(C.c_float)]),
    "drt_get_points": (C.c_int, [vp, f32p, f32p, f32p, f32p, C.c_int, C.POINTER(C.c_int)]),
    "drt_get_warped": (C.c_int, [vp, C.c_int, f32p, C.c_int]),
    "drt_get_zbuffer": (C.c_int, [vp, f32p]),
    "dr_device_alloc": (C.c_int, [C.c_int, C.c_size_t, C.POINTER(vp)]),
    "dr_device_free": (C.c_int, [vp]),
    "dr_memcpy_d2d": (C.c_int, [vp, vp, C.c_size_t]),
    "dr_memcpy_h2d": (C.c_int, [vp, vp, C.c_size_t]),
    "dr_memcpy_d2h": (C.c_int, [vp, vp, C.c_size_t]),
    "drm_set_feature_cache": (C.c_int, [vp, C.c_int]),
    "drm_feature_cache_stats": (C.c_int, [vp, C.POINTER(C.c_uint64)]),
    "drf_bench_sequence": (C.c_int, [vp, vp, vp, f32p, C.c_int, C.c_int, f32p]),
    "drf_visited_blocks": (C.c_int, [vp, C.POINTER(C.c_uint64)]),
    "drf_bench_render_host": (C.c_int, [vp, C.c_int, C.c_int, C.POINTER(vp), C.POINTER(vp)]),
    "drf_bench_integrate": (C.c_int, [vp, vp, vp, f32p, C.c_int, f32p, f32p]),
}


HOOKS_LIB_PATH = os.path.join(_HERE, "libdr_mi355x_hooks.so")  # the PARITY build (-DDR_PARITY_HOOKS): superseded kernel generations selectable
_loaded = {}


def switch(path=None):
    """Make `path` (default: the product library) the library every operator object created from now on talks to.  Test
    infrastructure: the -m gpu cases that compare kernel generations switch to HOOKS_LIB_PATH for their duration (tests/conftest.py::
    parity_hooks).  Objects created before a switch must be closed first: a handle belongs to the library that made it."""
    global _lib, LIB_PATH
    LIB_PATH = path or os.environ.get("DR_MI355X_LIB") or os.path.join(_HERE, "libdr_mi355x.so")
    _lib = _loaded.get(LIB_PATH)


def lib():
    global _lib
    if _lib is None:
        if not os.path.isfile(LIB_PATH):
            raise ImportError("tandem_amd: %s not built -- run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)
        L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError = ABI/header mismatch: fail loudly
            fn.restype, fn.argtypes = res, args
        _lib = _loaded[LIB_PATH] = L
    return _lib


def check(code):
    if code != DR_OK:
        raise DrError(code, lib().dr_last_error().decode(errors="replace"))


def fptr(a):
    return a.ctypes.data_as(f32p)
