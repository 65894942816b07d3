"""ctypes wrapper of oracle/_ref/libcoarse_tracker_ref.so: the REFERENCE's own calcResKernelNew / calcGKernel
(ref:tandem/libdr/cuda_coarse_tracker/src/cuda_coarse_tracker_private.cu) compiled for the host (TEST INFRASTRUCTURE)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_ref", "libcoarse_tracker_ref.so")
_REF = os.environ.get("TANDEM_REFERENCE", "/root/reference")
_lib = None


def build():
    if os.path.isdir(os.path.join(_REF, "tandem", "libdr", "cuda_coarse_tracker", "src")):
        subprocess.check_call(["make", "-s", "-f", "oracle/Makefile.ref", "REF=" + _REF, "oracle/_ref/libcoarse_tracker_ref.so"],
                              cwd=os.path.dirname(_HERE))
    return _SO if os.path.isfile(_SO) else None


def available():
    try:
        return build() is not None
    except Exception:
        return False


def lib():
    global _lib
    if _lib is None:
        so = build()
        if so is None:
            raise RuntimeError("oracle/_ref/libcoarse_tracker_ref.so is missing and /root/reference is not present")
        L = C.CDLL(so)
        f, i, vp = C.c_float, C.c_int, C.c_void_p
        L.reftrk_calc_res.argtypes = [f, i, i, f, f, f, f, vp, vp, f, f, f, f, i, vp, vp, vp, vp, vp, vp, vp]
        L.reftrk_calc_g_float.argtypes = [f, f, f, f, f, i, i, vp, vp, vp]
        L.reftrk_calc_g_double.argtypes = [f, f, f, f, f, i, i, vp, vp, vp]
        _lib = L
    return _lib


def _ptrs(arrs):
    return (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])


def calc_res(huber, w, h, fx, fy, cx, cy, r2n16, Ki9, aff2, maxEnergy, cutoffTH, pc_u, pc_v, pc_idepth, pc_color, dInew):
    """-> (warped [7 arrays of n f32], outputs f32[7]) from the reference kernel."""
    n = len(pc_u)
    warped = [np.full(max(n, 1), np.nan, np.float32) for _ in range(7)]
    out = np.zeros(7, np.float32)
    a = [np.ascontiguousarray(x, np.float32) for x in (r2n16, Ki9, pc_u, pc_v, pc_idepth, pc_color, dInew)]
    lib().reftrk_calc_res(huber, w, h, fx, fy, cx, cy, a[0].ctypes.data, a[1].ctypes.data, float(aff2[0]), float(aff2[1]),
                          maxEnergy, cutoffTH, n, a[2].ctypes.data, a[3].ctypes.data, a[4].ctypes.data, a[5].ctypes.data,
                          a[6].ctypes.data, _ptrs(warped), out.ctypes.data)
    return [x[:n] for x in warped], out


def calc_g(fx, fy, aff2, ref_aff_b, pc_color, warped, loops=16, double=True):
    n = len(pc_color)
    w = [np.ascontiguousarray(x, np.float32) for x in warped]
    col = np.ascontiguousarray(pc_color, np.float32)
    out = np.zeros(45, np.float64 if double else np.float32)
    fn = lib().reftrk_calc_g_double if double else lib().reftrk_calc_g_float
    fn(fx, fy, float(aff2[0]), float(aff2[1]), ref_aff_b, n, loops, col.ctypes.data, _ptrs(w), out.ctypes.data)
    return out
