"""Python mirror of TANDEM's `DrMvsnet` operator (tandem/libdr/dr_mvsnet/src/dr_mvsnet/dr_mvsnet.h:12-68)
on top of the C ABI of libdr_mi355x.so.  Same method names, argument meaning and error behaviour
(errors raise DrError instead of exit(EXIT_FAILURE)); results come from the HIP engine only."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check, fptr, u8p, f32p


class DrMvsnetOutput:
    """dr_mvsnet.h:12-34: four H*W float arrays."""

    def __init__(self, height, width):
        self.height, self.width = height, width
        self.depth = np.empty((height, width), np.float32)
        self.confidence = np.empty((height, width), np.float32)
        self.depth_dense = np.empty((height, width), np.float32)
        self.confidence_dense = np.empty((height, width), np.float32)


def _marshal(bgrs, intrinsic_matrix, cam_to_worlds, height, width):
    """The C ABI copies height*width*3 bytes per view, 16 floats per pose and 9 for K from raw pointers: sizes are
    checked here so that a wrongly shaped array is an error, not an out-of-bounds host read."""
    bgrs = [np.ascontiguousarray(b, dtype=np.uint8) for b in bgrs]
    for b in bgrs:
        if b.size != height * width * 3:
            raise ValueError("DrMvsnet: image of %d bytes, expected %d x %d x 3" % (b.size, height, width))
    c2ws = [np.ascontiguousarray(c, dtype=np.float32) for c in cam_to_worlds]
    if any(c.size != 16 for c in c2ws):
        raise ValueError("DrMvsnet: every cam_to_world must hold 16 floats")
    c2ws = [c.reshape(16) for c in c2ws]
    K = np.ascontiguousarray(intrinsic_matrix, dtype=np.float32)
    if K.size != 9:
        raise ValueError("DrMvsnet: intrinsic_matrix must hold 9 floats")
    K = K.reshape(9)
    V = len(bgrs)
    pb = (u8p * V)(*[b.ctypes.data_as(u8p) for b in bgrs])
    pc = (f32p * V)(*[fptr(c) for c in c2ws])
    return bgrs, c2ws, K, pb, pc


class _Owned:
    """Owner of something the C library must release exactly once (an engine handle, a drm_host_alloc block).  numpy views of
    library-owned memory keep their owner alive (they are built on a ctypes buffer that carries a reference to it), so the
    release happens when the last VIEW is gone, not when DrMvsnet.close() is called: a view can never dangle."""

    def __init__(self, release, what):
        self._release, self.what = release, what

    def __del__(self):
        self.release_now()

    def release_now(self):
        rel, self._release = self._release, None
        if rel is not None:
            rel(self.what)


def _view(ptr, nbytes, owner):
    """ctypes byte buffer over library-owned memory that keeps `owner` alive; np.frombuffer(...) views inherit that."""
    buf = (C.c_uint8 * nbytes).from_address(ptr if isinstance(ptr, int) else C.cast(ptr, C.c_void_p).value)
    buf._owner = owner
    return buf


class DrMvsnet:
    def __init__(self, filename, device=0):
        """dr_mvsnet.h:38 `explicit DrMvsnet(char const* filename)`; filename = TDMW weight blob."""
        self._h = C.c_void_p()
        self._L = _lib.lib()  # the library this handle belongs to (tests may switch the process default, _lib.switch)
        check(self._L.drm_create(str(filename).encode(), int(device), C.byref(self._h)))
        L = self._L
        self._engine = _Owned(lambda h: L.drm_destroy(h), C.c_void_p(self._h.value))  # result views hold a reference to it
        self._hw = None

    def close(self, force=False):
        """Drops the engine.  It is destroyed now unless result views (GetResultView) are still alive -- then when the last of them
        goes (the garbage collector decides when: device memory and streams of a closed engine can outlive this call); page-locked image
        blocks (alloc_images) likewise live as long as their arrays.  force=True destroys the engine NOW: views handed out earlier then point
        into freed page-locked memory and must not be touched again."""
        eng = getattr(self, "_engine", None)
        if force and eng is not None:
            eng.release_now()
        self._h = C.c_void_p()
        self._engine = None
        self._next_out = None

    __del__ = close

    def CallAsync(self, height, width, view_num, ref_index, bgrs, intrinsic_matrix, cam_to_worlds, depth_min,
                  depth_max, discard_percentage, debug_print=False):
        """dr_mvsnet.h:43-53.  Blocking for the last input, non-blocking for this one."""
        assert len(bgrs) == view_num and len(cam_to_worlds) == view_num
        keep = _marshal(bgrs, intrinsic_matrix, cam_to_worlds, height, width)
        check(self._L.drm_call_async(self._h, height, width, view_num, ref_index, keep[3], fptr(keep[2]), keep[4],
                                        depth_min, depth_max, discard_percentage))
        self._hw = (height, width)
        # the object GetResult() will return: allocated and its pages touched NOW, while the device works on the window (as the C++ shim does:
        # the first-touch page faults of 4.9 MB of fresh result memory otherwise sit inside GetResult, on the caller's critical path)
        # (one object per call: GetResult() hands the previous one over to the caller, who keeps it -- TandemBackend never deletes its outputs)
        nxt = getattr(self, "_next_out", None)
        if nxt is None or (nxt.height, nxt.width) != (height, width):
            nxt = DrMvsnetOutput(height, width)
            for a in (nxt.depth, nxt.confidence, nxt.depth_dense, nxt.confidence_dense):
                a.fill(0.0)
            self._next_out = nxt

    def Ready(self):
        return bool(self._L.drm_ready(self._h))

    def Wait(self):
        check(self._L.drm_wait(self._h))

    def GetResult(self):
        """dr_mvsnet.h:56 -- blocking; a second call without a new CallAsync is a protocol error."""
        if self._hw is None:
            raise _lib.DrError(2, "GetResult before CallAsync")
        out = getattr(self, "_next_out", None)
        self._next_out = None
        if out is None or (out.height, out.width) != tuple(self._hw):
            out = DrMvsnetOutput(*self._hw)
        check(self._L.drm_get_result(self._h, fptr(out.depth), fptr(out.confidence), fptr(out.depth_dense),
                                        fptr(out.confidence_dense)))
        return out

    def GetResultView(self):
        """GetResult without the host copy (include/dr_mi355x.h drm_get_result_view): the four maps are numpy VIEWS of the page-locked block
        the device wrote them to -- valid while the next call is processed, overwritten by the one after it."""
        if self._hw is None:
            raise _lib.DrError(2, "GetResult before CallAsync")
        p = [f32p() for _ in range(4)]
        check(self._L.drm_get_result_view(self._h, *[C.byref(q) for q in p]))
        out = DrMvsnetOutput.__new__(DrMvsnetOutput)
        out.height, out.width = self._hw
        n = self._hw[0] * self._hw[1] * 4
        out.depth, out.confidence, out.depth_dense, out.confidence_dense = [
            np.frombuffer(_view(q, n, self._engine), np.float32).reshape(self._hw) for q in p]
        return out

    def set_feature_cache(self, capacity):
        """Extension (drm_set_feature_cache): keep FeatureNet's outputs of the last `capacity` key-frame images; a window of which at most one image is
        new runs FeatureNet on that one view.  0 = off (the default).  Bit-identical results either way."""
        check(self._L.drm_set_feature_cache(self._h, int(capacity)))

    def feature_cache_stats(self):
        out = (C.c_uint64 * 6)()
        check(self._L.drm_feature_cache_stats(self._h, out))
        return dict(views_from_cache=int(out[0]), views_computed=int(out[1]), batch_windows=int(out[2]), key_collisions=int(out[3]),
                    single_view_plan=bool(out[4]), entries=int(out[5]))

    def alloc_images(self, view_num, height, width):
        """`view_num` (H, W, 3) u8 arrays in page-locked memory (drm_host_alloc): CallAsync uploads such images in place, without the
        gather into the engine's staging block.  The block is freed when the last of the arrays is gone."""
        n = view_num * height * width * 3
        ptr = self._L.drm_host_alloc(n)
        if not ptr:
            raise MemoryError("drm_host_alloc(%d)" % n)
        L = self._L
        flat = np.frombuffer(_view(ptr, n, _Owned(lambda q: L.drm_host_free(q), ptr)), np.uint8)
        return [flat[v * height * width * 3:(v + 1) * height * width * 3].reshape(height, width, 3) for v in range(view_num)]

    # ---- device-resident / introspection hooks (no reference counterpart) ----
    def upload(self, height, width, view_num, ref_index, bgrs, intrinsic_matrix, cam_to_worlds, depth_min, depth_max,
               discard_percentage):
        keep = _marshal(bgrs, intrinsic_matrix, cam_to_worlds, height, width)
        check(self._L.drm_upload(self._h, height, width, view_num, ref_index, keep[3], fptr(keep[2]), keep[4],
                                    depth_min, depth_max, discard_percentage))
        self._hw = (height, width)

    def forward(self, iters=1):
        ms = C.c_float()
        check(self._L.drm_forward(self._h, iters, C.byref(ms)))
        return ms.value

    def autotune(self, max_candidates=8):
        """Time the planner's top candidates per convolution layer on the device and keep the fastest (opt-in; results
        move at the 1e-7 level).  Returns (summed layer ms before, after)."""
        a, b = C.c_float(), C.c_float()
        check(self._L.drm_autotune(self._h, max_candidates, C.byref(a), C.byref(b)))
        return a.value, b.value

    # ---- view sharding hooks (include/dr_mi355x.h "view sharding"; host protocol in tandem_amd/view_shard.py) ----
    def set_view_shard(self, nsrc_total):
        check(self._L.drm_set_view_shard(self._h, int(nsrc_total)))

    def forward_phase(self, phase):
        check(self._L.drm_forward_phase(self._h, int(phase)))

    @staticmethod
    def comm_available():
        """True when this process can bind RCCL (librccl.so.1 or $DR_RCCL_LIB) for the in-engine collective."""
        return _lib.lib().drm_comm_available() == 0

    @staticmethod
    def comm_unique_id():
        """128-byte RCCL id drawn by one rank; hand it to every rank's comm_init."""
        buf = (C.c_uint8 * 128)()
        check(_lib.lib().drm_comm_unique_id(buf))
        return bytes(buf)

    def comm_init(self, rank, world, unique_id):
        """In-engine view-shard collective: afterwards a sharded window's cost volumes are reduced to rank 0 and the stage depth
        maps broadcast back on the engine's stream (include/dr_mi355x.h)."""
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        check(self._L.drm_comm_init(self._h, int(rank), int(world), buf))

    def comm_destroy(self):
        check(self._L.drm_comm_destroy(self._h))

    def comm_count(self):
        """Ranks RCCL reports for the engine's communicator (0: none, -1: the bound library has no ncclCommCount)."""
        n = C.c_int()
        check(self._L.drm_comm_count(self._h, C.byref(n)))
        return n.value

    def device_tensor(self, name):
        """(device pointer, float count) of a named internal tensor, e.g. "volume2"."""
        ptr, n = C.c_void_p(), C.c_size_t()
        check(self._L.drm_device_tensor(self._h, name.encode(), C.byref(ptr), C.byref(n)))
        return ptr.value, int(n.value)

    def download(self):
        out = DrMvsnetOutput(*self._hw)
        check(self._L.drm_download(self._h, fptr(out.depth), fptr(out.confidence), fptr(out.depth_dense),
                                      fptr(out.confidence_dense)))
        return out

    def stage_output(self, stage):
        sc = 2 ** (3 - stage)
        h, w = self._hw[0] // sc, self._hw[1] // sc
        d, c = np.empty((h, w), np.float32), np.empty((h, w), np.float32)
        check(self._L.drm_get_stage_output(self._h, stage, fptr(d), fptr(c)))
        return d, c

    def tensor(self, name):
        n, dims = C.c_size_t(), (C.c_int * 4)()
        check(self._L.drm_get_tensor(self._h, name.encode(), None, 0, C.byref(n), dims))
        out = np.empty(tuple(dims), np.float32)
        check(self._L.drm_get_tensor(self._h, name.encode(), fptr(out), n.value, C.byref(n), dims))
        return out

    def profile(self):
        names = C.create_string_buffer(1 << 16)
        ms = (C.c_float * 512)()
        cnt = C.c_int()
        check(self._L.drm_profile(self._h, names, len(names), ms, 512, C.byref(cnt)))
        rows = []
        for line, t in zip(names.value.decode().strip().split("\n"), list(ms)[:cnt.value]):
            op, kern, fl, by = line.split("\t")
            rows.append(dict(op=op, kernel=kern, flops=float(fl), bytes=float(by), ms=t))
        return rows

    def work(self):
        f, b = C.c_double(), C.c_double()
        check(self._L.drm_work(self._h, C.byref(f), C.byref(b)))
        return f.value, b.value


def debug_conv(x, weight, stride=(1, 1, 1), transposed=False, scale=None, bias=None, relu=False, add=None,
               add_up2=False, device=0):
    """Kernel unit-test hook: x (D,H,W,Cin) channels-last, weight in torch layout; returns (Do,Ho,Wo,Cout).
    transposed: False / True, or "up2": a 3x3 stride-1 layer over the nearest x2 upsampling (H, W) of x (ConvLayer::up2)."""
    x = np.ascontiguousarray(x, np.float32)
    w = np.ascontiguousarray(weight, np.float32)
    D, H, W, Cin = x.shape
    up2 = transposed == "up2"
    transposed = 2 if up2 else int(bool(transposed))
    Cout = w.shape[1] if transposed == 1 else w.shape[0]
    kd, kh, kw = w.shape[2:]
    sd, sh, sw = stride
    oD, oH, oW = ((D * sd, H * sh, W * sw) if transposed == 1 else ((D, 2 * H, 2 * W) if up2 else
                  ((D + 2 * (kd // 2) - kd) // sd + 1, (H + 2 * (kh // 2) - kh) // sh + 1, (W + 2 * (kw // 2) - kw) // sw + 1)))
    out = np.empty((oD, oH, oW, Cout), np.float32)
    dims = (C.c_int * 3)()
    opt = lambda a: fptr(np.ascontiguousarray(a, np.float32)) if a is not None else None
    keep = [np.ascontiguousarray(a, np.float32) if a is not None else None for a in (scale, bias, add)]
    check(_lib.lib().drm_debug_conv(device, fptr(x), D, H, W, Cin, fptr(w), Cout, kd, kh, kw, sd, sh, sw,
                                    int(transposed), fptr(keep[0]) if keep[0] is not None else None,
                                    fptr(keep[1]) if keep[1] is not None else None, int(relu),
                                    fptr(keep[2]) if keep[2] is not None else None, int(add_up2), fptr(out), dims))
    assert tuple(dims) == (oD, oH, oW)
    return out


def debug_tail(x, skip, w_deconv, scale, bias, w_prob, qy=0, zchunk=0, form=1, device=0):
    """Kernel unit-test hook for k_tail_m (form=1: matrix pipe) / k_tail (form=0: vector pipe): x (D/2, h/2, w/2, 16), skip (D, h, w, 8) channels-last; torch-layout weights; returns (D, h, w) logits."""
    x, skip = np.ascontiguousarray(x, np.float32), np.ascontiguousarray(skip, np.float32)
    D, h, w, _ = skip.shape
    assert x.shape == (D // 2, h // 2, w // 2, 16) and skip.shape[3] == 8
    wd, wp = np.ascontiguousarray(w_deconv, np.float32), np.ascontiguousarray(w_prob, np.float32)
    assert wd.shape == (16, 8, 3, 3, 3) and wp.shape == (1, 8, 3, 3, 3)
    sc, bi = np.ascontiguousarray(scale, np.float32), np.ascontiguousarray(bias, np.float32)
    out = np.empty((D, h, w), np.float32)
    check(_lib.lib().drm_debug_tail(device, fptr(x), fptr(skip), fptr(wd), fptr(sc), fptr(bi), fptr(wp), D, h, w, int(qy), int(zchunk), int(form), fptr(out)))
    return out
