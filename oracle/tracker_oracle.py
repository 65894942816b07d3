"""ctypes wrapper of oracle/libtracker_oracle.so (CPU ORACLE of the dense coarse tracker -- TEST INFRASTRUCTURE).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libtracker_oracle.so")
_SO_LEFT = os.path.join(_HERE, "libtracker_oracle_left.so")  # -DTRK_SUM_LEFT: the hand-off's 3-term products as (a0 + a1) + a2 (tests/test_ref_handoff.py)
_libs = {}


def build(sum_left=False):
    src = os.path.join(_HERE, "tracker_oracle.c")
    so = _SO_LEFT if sum_left else _SO
    if not os.path.isfile(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O2", "-std=c99", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math"] + (["-DTRK_SUM_LEFT"] if sum_left else []) +
                              [src, "-o", so, "-lm"])
    return so


def lib(sum_left=False):
    if sum_left not in _libs:
        L = C.CDLL(build(sum_left))
        vp = C.c_void_p
        L.trk_create.restype = vp
        L.trk_create.argtypes = [C.c_int, C.c_int, C.c_float, C.c_float, C.c_int]
        L.trk_destroy.argtypes = [vp]
        L.trk_set_k.argtypes = [vp, C.c_float, C.c_float, C.c_float, C.c_float]
        L.trk_set_reference.argtypes = [vp, C.c_int, vp, vp, vp, vp, C.c_float, vp]
        L.trk_set_new.argtypes = [vp, vp]
        L.trk_calc_res.argtypes = [vp, vp, C.c_float, vp, C.c_float, vp, vp]
        L.trk_calc_g.argtypes = [vp, vp, vp, C.c_float, vp, vp]
        L.trk_n.argtypes = [vp]
        L.trk_get_points.argtypes = [vp, vp, vp, vp, vp]
        L.trk_get_warped.argtypes = [vp, C.c_int, vp]
        L.trk_kernel_inputs.argtypes = [vp, vp, C.c_float, vp, C.c_float, vp, vp, vp, vp, vp]
        L.trk_append_dense.argtypes = [vp, vp, vp, vp, C.c_int, C.c_int, vp, vp, vp]
        _libs[sum_left] = L
    return _libs[sum_left]


def _f32(a):
    return np.ascontiguousarray(a, np.float32)


def _f64(a):
    return np.ascontiguousarray(a, np.float64)


class TrackerOracle:
    def __init__(self, w, h, huber, coarse_cutoff, n_max=0, sum_left=False):
        self.w, self.h = w, h
        self._L = lib(sum_left)
        self._h = self._L.trk_create(w, h, huber, coarse_cutoff, n_max)

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.trk_destroy(self._h)
            self._h = None

    def setK(self, fx, fy, cx, cy):
        self._L.trk_set_k(self._h, fx, fy, cx, cy)

    def setReference(self, u, v, idepth, color, ref_exposure, ref_aff_g2l):
        u, v, idepth, color, aff = _f32(u), _f32(v), _f32(idepth), _f32(color), _f64(ref_aff_g2l)
        assert self._L.trk_set_reference(self._h, len(u), u.ctypes.data, v.ctypes.data, idepth.ctypes.data, color.ctypes.data,
                                       ref_exposure, aff.ctypes.data) == 0

    def setNew(self, dInew):
        d = _f32(dInew)
        assert d.size == 3 * self.w * self.h
        self._L.trk_set_new(self._h, d.ctypes.data)

    def calcRes(self, refToNew, new_exposure, aff_g2l, cutoffTH):
        T, aff = _f64(refToNew).reshape(16), _f64(aff_g2l)
        out, sums = np.zeros(6), np.zeros(7)
        self._L.trk_calc_res(self._h, T.ctypes.data, new_exposure, aff.ctypes.data, cutoffTH, out.ctypes.data, sums.ctypes.data)
        return out, sums

    def calcG(self, new_exposure, aff_g2l):
        aff = _f64(aff_g2l)
        H, b, raw = np.zeros((8, 8)), np.zeros(8), np.zeros(45)
        self._L.trk_calc_g(self._h, H.ctypes.data, b.ctypes.data, new_exposure, aff.ctypes.data, raw.ctypes.data)
        return H, b, raw

    def kernel_inputs(self, refToNew, new_exposure, aff_g2l, cutoffTH):
        """The float inputs calcRes / calcG hand to their kernels: (r2n16, Ki9, affLL2, maxEnergy, ref_aff_b)."""
        T, aff = _f64(refToNew).reshape(16), _f64(aff_g2l)
        r2n, Ki, a2, me, rb = (np.zeros(k, np.float32) for k in (16, 9, 2, 1, 1))
        self._L.trk_kernel_inputs(self._h, T.ctypes.data, new_exposure, aff.ctypes.data, cutoffTH, r2n.ctypes.data, Ki.ctypes.data,
                                a2.ctypes.data, me.ctypes.data, rb.ctypes.data)
        return r2n, Ki, a2, float(me[0]), float(rb[0])

    def n(self):
        return self._L.trk_n(self._h)

    def points(self):
        n = self.n()
        a = [np.empty(max(n, 1), np.float32) for _ in range(4)]
        self._L.trk_get_points(self._h, *[x.ctypes.data for x in a])
        return [x[:n] for x in a]

    def warped(self):
        n = self.n()
        out = []
        for k in range(7):
            a = np.empty(max(n, 1), np.float32)
            self._L.trk_get_warped(self._h, k, a.ctypes.data)
            out.append(a[:n])
        return out

    def appendDenseReference(self, depth, KRKi, Kt, step, dense_only, idepth0, dIp0):
        depth, KRKi, Kt, dIp0 = _f32(depth), _f32(KRKi).reshape(9), _f32(Kt), _f32(dIp0)
        id0 = _f32(idepth0) if idepth0 is not None else None
        proj = np.empty((self.h, self.w), np.float32)
        n = self._L.trk_append_dense(self._h, depth.ctypes.data, KRKi.ctypes.data, Kt.ctypes.data, step, int(dense_only),
                                   id0.ctypes.data if id0 is not None else None, dIp0.ctypes.data, proj.ctypes.data)
        return n, proj
