"""ctypes wrapper of oracle/_ref/libdr_fusion_ref.so: the REFERENCE's own DrFusion (ref:tandem/libdr/dr_fusion/src/**)
compiled for the host by oracle/Makefile.ref (TEST INFRASTRUCTURE).

Same Python surface as oracle/tsdf_oracle.py::TsdfOracle, so tests can run the restatement and the reference side by
side.  `available()` is False when neither the prebuilt library nor /root/reference exists (then tests skip)."""
import ctypes as C
import os
import subprocess

import numpy as np

from .tsdf_oracle import Options

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_ref", "libdr_fusion_ref.so")
_SO_FMA = os.path.join(_HERE, "_ref", "libdr_fusion_ref_fma.so")  # the same sources with -ffp-contract=fast -mfma (Makefile.ref)
_REF = os.environ.get("TANDEM_REFERENCE", "/root/reference")


def build(fma=False):
    """(Re)build from the reference checkout when it exists; otherwise use the prebuilt library as it is."""
    so = _SO_FMA if fma else _SO
    if os.path.isdir(os.path.join(_REF, "tandem", "libdr", "dr_fusion", "src")):
        subprocess.check_call(["make", "-s", "-f", "oracle/Makefile.ref", "REF=" + _REF, os.path.relpath(so, os.path.dirname(_HERE))],
                              cwd=os.path.dirname(_HERE))
    return so if os.path.isfile(so) else None


def available(fma=False):
    try:
        return build(fma) is not None
    except Exception:
        return False


_libs = {}


def lib(fma=False):
    if fma not in _libs:
        so = build(fma)
        if so is None:
            raise RuntimeError("%s is missing and /root/reference is not present" % (_SO_FMA if fma else _SO))
        L = C.CDLL(so)
        L.refdrf_create.restype = C.c_void_p
        L.refdrf_create.argtypes = [C.POINTER(Options)]
        L.refdrf_destroy.argtypes = [C.c_void_p]
        L.refdrf_integrate.argtypes = [C.c_void_p] * 4
        L.refdrf_render.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.refdrf_num_blocks.argtypes = [C.c_void_p]
        L.refdrf_num_allocated_counter.argtypes = [C.c_void_p]
        L.refdrf_export_blocks.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.refdrf_get_mesh.restype = C.c_long
        L.refdrf_get_mesh.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_long, C.c_void_p, C.c_void_p]
        L.ref_combine.argtypes = [C.c_float, C.c_void_p, C.c_ubyte, C.c_float, C.c_void_p, C.c_ubyte, C.c_ubyte,
                                  C.c_void_p, C.c_void_p, C.c_void_p]
        L.ref_combine_colour_table.argtypes = [C.c_ubyte, C.c_void_p]
        L.ref_point3d.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p]
        L.ref_project.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.ref_norm.restype = C.c_float
        L.ref_norm.argtypes = [C.c_void_p]
        L.ref_inverse4.argtypes = [C.c_void_p, C.c_void_p]
        L.ref_xform.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.ref_world_maps.argtypes = [C.c_void_p] * 6
        L.ref_hash.argtypes = [C.c_void_p, C.c_void_p]
        _libs[fma] = L
    return _libs[fma]


class RefFusion:
    """The reference DrFusion on the host (serial kernels).  The reference's mandatory call order is kept:
    integrate -> render (exactly num_render_streams poses) -> ... (tsdf_volume.cu:520-524,635-653)."""

    def __init__(self, fma=False, **opts):
        self.o = Options(**opts)
        self._fma = fma  # True: the contracted twin (libdr_fusion_ref_fma.so)
        self._h = lib(fma).refdrf_create(C.byref(self.o))

    def close(self):
        if getattr(self, "_h", None):
            lib(self._fma).refdrf_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def integrate(self, bgr, depth, pose):
        bgr = np.ascontiguousarray(bgr, np.uint8)
        depth = np.ascontiguousarray(depth, np.float32)
        pose = np.ascontiguousarray(pose, np.float32).reshape(16)
        lib(self._fma).refdrf_integrate(self._h, bgr.ctypes.data, depth.ctypes.data, pose.ctypes.data)
        return 0

    def render(self, poses):
        """poses: list of exactly num_render_streams 4x4 -> list of (bgr, depth)."""
        n = len(poses)
        assert n == self.o.num_render_streams
        H, W = self.o.height, self.o.width
        ps = [np.ascontiguousarray(p, np.float32).reshape(16) for p in poses]
        bs = [np.empty((H, W, 3), np.uint8) for _ in range(n)]
        ds = [np.empty((H, W), np.float32) for _ in range(n)]
        pp = (C.c_void_p * n)(*[p.ctypes.data for p in ps])
        bp = (C.c_void_p * n)(*[b.ctypes.data for b in bs])
        dp = (C.c_void_p * n)(*[d.ctypes.data for d in ds])
        lib(self._fma).refdrf_render(self._h, pp, n, bp, dp)
        return list(zip(bs, ds))

    def num_blocks(self):
        return lib(self._fma).refdrf_num_blocks(self._h)

    def export_blocks(self):
        n = self.num_blocks()
        nv = self.o.block_size ** 3
        coords = np.empty((max(n, 1), 3), np.int32)
        vox = np.empty((max(n, 1), nv * 8), np.uint8)
        got = lib(self._fma).refdrf_export_blocks(self._h, n, coords.ctypes.data, vox.ctypes.data)
        return {tuple(int(v) for v in coords[i]): vox[i] for i in range(got)}

    def extract_mesh(self, lower, upper, max_tri=2_000_000):
        lo = np.ascontiguousarray(lower, np.float32)
        up = np.ascontiguousarray(upper, np.float32)
        vert, cols = np.empty((max_tri * 3, 3), np.float32), np.empty((max_tri * 3, 3), np.float32)
        n = lib(self._fma).refdrf_get_mesh(self._h, lo.ctypes.data, up.ctypes.data, max_tri, vert.ctypes.data, cols.ctypes.data)
        if n > max_tri:
            raise RuntimeError("reference mesh has %d triangles > max_tri=%d" % (n, max_tri))
        return vert[:3 * n].copy(), cols[:3 * n].copy()

    def world_maps(self, p):
        p = np.ascontiguousarray(p, np.float32)
        g, b, l = (np.empty(3, np.int32) for _ in range(3))
        w = np.empty(3, np.float32)
        lib(self._fma).ref_world_maps(self._h, p.ctypes.data, g.ctypes.data, b.ctypes.data, l.ctypes.data, w.ctypes.data)
        return g, b, l, w

    def hash(self, p):
        p = np.ascontiguousarray(p, np.int32)
        return lib(self._fma).ref_hash(self._h, p.ctypes.data)
