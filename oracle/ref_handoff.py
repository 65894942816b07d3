"""ctypes wrapper of oracle/_ref/libdense_handoff_ref.so -- the REFERENCE's own dense-depth hand-off (CoarseTracker::setCoarseTrackingRef,
tandem/src/FullSystem/CoarseTracker.cpp:654-723) compiled for the host by oracle/Makefile.ref.  TEST INFRASTRUCTURE: only tests/ may import this."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_libs = {}


def path(sum_left=False):
    return os.path.join(_HERE, "_ref", "libdense_handoff_ref_left.so" if sum_left else "libdense_handoff_ref.so")


def available():
    return os.path.isfile(path(False)) and os.path.isfile(path(True))


def _lib(sum_left):
    if sum_left not in _libs:
        L = C.CDLL(path(sum_left))
        vp = C.c_void_p
        L.ref_dense_handoff.restype = C.c_int
        L.ref_dense_handoff.argtypes = [C.c_int, C.c_int, vp, vp, vp, vp, C.c_int, C.c_int, vp, vp, C.c_int, vp, vp, vp, vp, vp, vp, vp]
        _libs[sum_left] = L
    return _libs[sum_left]


def dense_handoff(depth, c2w_dense, c2w_last, K, step, dense_only, idepth0, dIp0, sparse=None, sum_left=False):
    """Runs the reference block.  sparse = (u, v, idepth, color) of the n0 points already in the list.  Returns dict(pc_n, u, v, idepth, color
    (the raw arrays, slot 0 .. pc_n INCLUSIVE: the block pre-increments), KRKi (9,), Kt (3,)): the block's own float products."""
    H, W = depth.shape
    f32 = lambda a: np.ascontiguousarray(a, np.float32)  # noqa: E731
    depth, idepth0, dIp0 = f32(depth), f32(idepth0), f32(dIp0)
    assert idepth0.shape == (H, W) and dIp0.shape == (H, W, 3)
    n0 = 0 if sparse is None else len(sparse[0])
    cap = W * H + n0 + 1
    arr = [np.full(cap, np.nan, np.float32) for _ in range(4)]
    if sparse is not None:
        for a, src in zip(arr, sparse):
            a[:n0] = src
    c2wd, c2wl, K9 = f32(c2w_dense).reshape(16), np.ascontiguousarray(c2w_last, np.float64).reshape(16), f32(K).reshape(9)
    KRKi, Kt, Ki = np.zeros(9, np.float32), np.zeros(3, np.float32), np.zeros(9, np.float32)
    n = _lib(sum_left).ref_dense_handoff(W, H, depth.ctypes.data, c2wd.ctypes.data, c2wl.ctypes.data, K9.ctypes.data, step, int(dense_only), idepth0.ctypes.data,
                                          dIp0.ctypes.data, n0, *[a.ctypes.data for a in arr], KRKi.ctypes.data, Kt.ctypes.data, Ki.ctypes.data)
    return dict(pc_n=n, n0=n0, u=arr[0], v=arr[1], idepth=arr[2], color=arr[3], KRKi=KRKi, Kt=Kt, Ki=Ki)
