"""Host-side mirror of the reference's dense coarse tracker operator (class CudaCoarseTracker,
tandem/libdr/cuda_coarse_tracker/include/public/cuda_coarse_tracker.h:9-35) over the C ABI (drt_* in
include/dr_mi355x.h): same member names and argument meaning; Eigen arguments become numpy arrays.  Plus
appendDenseReference = the dense-depth hand-off loop of CoarseTracker::setCoarseTrackingRef
(src/FullSystem/CoarseTracker.cpp:655-725) run on the device.  No CPU fallback."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check

f32p, f64p = C.POINTER(C.c_float), C.POINTER(C.c_double)


def _f32(a):
    return np.ascontiguousarray(a, np.float32)


def _p32(a):
    return a.ctypes.data_as(f32p)


def _p64(a):
    return a.ctypes.data_as(f64p)


class DrCoarseTracker:
    def __init__(self, w, h, setting_huberTH, setting_coarseCutoffTH, device=0):
        self.w, self.h = w, h
        self._h = C.c_void_p()
        self._L = _lib.lib()  # the library this handle belongs to (tests may switch the process default, _lib.switch)
        check(self._L.drt_create(w, h, setting_huberTH, setting_coarseCutoffTH, device, C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None):
            self._L.drt_destroy(self._h)
            self._h = None

    __del__ = close

    def setK(self, w, h, fx, fy, cx, cy):
        check(self._L.drt_set_k(self._h, w, h, fx, fy, cx, cy))

    def init(self, n_max_in=0):
        check(self._L.drt_init(self._h, n_max_in))

    def setReference(self, pc_u, pc_v, pc_idepth, pc_color, ref_exposure, ref_aff_g2l):
        a = [_f32(x) for x in (pc_u, pc_v, pc_idepth, pc_color)]
        aff = np.ascontiguousarray(ref_aff_g2l, np.float64)
        check(self._L.drt_set_reference(self._h, len(a[0]), _p32(a[0]), _p32(a[1]), _p32(a[2]), _p32(a[3]), ref_exposure, _p64(aff)))

    def setNew(self, dInew):
        d = _f32(dInew)
        if d.size != 3 * self.w * self.h:
            raise ValueError("dInew must hold 3*w*h floats")
        check(self._L.drt_set_new(self._h, _p32(d)))

    def calcRes(self, refToNew, new_exposure, aff_g2l, cutoffTH, return_sums=False):
        T = np.ascontiguousarray(refToNew, np.float64).reshape(16)
        aff = np.ascontiguousarray(aff_g2l, np.float64)
        out, sums = np.zeros(6), np.zeros(7)
        check(self._L.drt_calc_res(self._h, _p64(T), new_exposure, _p64(aff), cutoffTH, _p64(out), _p64(sums)))
        return (out, sums) if return_sums else out

    def calcG(self, new_exposure, aff_g2l, return_raw=False):
        aff = np.ascontiguousarray(aff_g2l, np.float64)
        H, b, raw = np.zeros((8, 8)), np.zeros(8), np.zeros(45)
        check(self._L.drt_calc_g(self._h, _p64(H), _p64(b), new_exposure, _p64(aff), _p64(raw)))
        return (H, b, raw) if return_raw else (H, b)

    def appendDenseReference(self, depth, KRKi, Kt, step, dense_only, idepth0, dIp0, device_pointers=False):
        """depth / idepth0 / dIp0: numpy arrays, or integer device pointers when device_pointers=True."""
        K9, t3 = _f32(KRKi).reshape(9), _f32(Kt)
        n = C.c_int()
        if device_pointers:
            d, i0, c0 = (C.c_void_p(x) if x else None for x in (depth, idepth0, dIp0))
            keep = None
        else:
            keep = [_f32(depth), _f32(idepth0) if idepth0 is not None else None, _f32(dIp0)]
            d, i0, c0 = (C.c_void_p(x.ctypes.data) if x is not None else None for x in keep)
        check(self._L.drt_append_dense_reference(self._h, d, _p32(K9), _p32(t3), step, int(dense_only), i0, c0, int(device_pointers), C.byref(n)))
        return n.value

    def synchronize(self):
        check(self._L.drt_synchronize(self._h))

    def startTiming(self):
        check(self._L.drt_start_timing(self._h))

    def endTimingMilliseconds(self):
        ms = C.c_float()
        check(self._L.drt_end_timing_ms(self._h, C.byref(ms)))
        return ms.value

    # ---- introspection hooks (no reference counterpart) ----
    def points(self):
        n = C.c_int()
        check(self._L.drt_get_points(self._h, None, None, None, None, 1 << 30, C.byref(n)))
        a = [np.empty(max(n.value, 1), np.float32) for _ in range(4)]
        check(self._L.drt_get_points(self._h, _p32(a[0]), _p32(a[1]), _p32(a[2]), _p32(a[3]), len(a[0]), C.byref(n)))
        return [x[:n.value] for x in a]

    def warped(self):
        n = len(self.points()[0])
        out = []
        for k in range(7):
            a = np.empty(max(n, 1), np.float32)
            check(self._L.drt_get_warped(self._h, k, _p32(a), len(a)))
            out.append(a[:n])
        return out

    def zbuffer(self):
        a = np.empty((self.h, self.w), np.float32)
        check(self._L.drt_get_zbuffer(self._h, _p32(a)))
        return a
