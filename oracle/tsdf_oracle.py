"""ctypes wrapper of oracle/libtsdf_oracle.so (CPU ORACLE of the DrFusion path -- TEST INFRASTRUCTURE).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libtsdf_oracle.so")


class Options(C.Structure):
    _fields_ = [("voxel_size", C.c_float), ("num_buckets", C.c_int), ("bucket_size", C.c_int),
                ("num_blocks", C.c_int), ("block_size", C.c_int), ("max_sdf_weight", C.c_int),
                ("truncation_distance", C.c_float), ("max_sensor_depth", C.c_float),
                ("min_sensor_depth", C.c_float), ("num_render_streams", C.c_int),
                ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float),
                ("height", C.c_int), ("width", C.c_int)]


def build(omp=False):
    """omp=True: the same source with -fopenmp (integration parallel over blocks) -- bench.py's multi-core CPU baseline."""
    src = os.path.join(_HERE, "tsdf_oracle.c")
    tab = os.path.join(_HERE, "..", "tandem_amd", "csrc", "mc_tables.h")
    so = _SO.replace(".so", "_omp.so") if omp else _SO
    if not os.path.isfile(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(tab)):
        subprocess.check_call(["gcc", "-O2", "-std=c99", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math"] + (["-fopenmp"] if omp else []) +
                              [src, "-o", so, "-lm"])
    return so


_lib = None
_libs = {}


def lib(omp=False):
    global _lib
    if omp not in _libs:
        L = C.CDLL(build(omp))
        L.tsdf_create.restype = C.c_void_p
        L.tsdf_create.argtypes = [C.POINTER(Options)]
        L.tsdf_destroy.argtypes = [C.c_void_p]
        L.tsdf_integrate.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.tsdf_render.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.tsdf_num_blocks.argtypes = [C.c_void_p]
        L.tsdf_stats.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
        L.tsdf_export_blocks.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.tsdf_inverse4.argtypes = [C.c_void_p, C.c_void_p]
        L.tsdf_extract_mesh.restype = C.c_long
        L.tsdf_extract_mesh.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_long, C.c_void_p, C.c_void_p]
        _libs[omp] = L
    return _libs[omp]


class TsdfOracle:
    def __init__(self, omp=False, **opts):
        self.o = Options(**opts)
        self._L = lib(omp)
        self._h = self._L.tsdf_create(C.byref(self.o))

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.tsdf_destroy(self._h)
            self._h = None

    def integrate(self, bgr, depth, pose):
        bgr = np.ascontiguousarray(bgr, np.uint8)
        depth = np.ascontiguousarray(depth, np.float32)
        pose = np.ascontiguousarray(pose, np.float32).reshape(16)
        return self._L.tsdf_integrate(self._h, bgr.ctypes.data, depth.ctypes.data, pose.ctypes.data)

    def render(self, pose):
        H, W = self.o.height, self.o.width
        pose = np.ascontiguousarray(pose, np.float32).reshape(16)
        bgr, depth = np.empty((H, W, 3), np.uint8), np.empty((H, W), np.float32)
        self._L.tsdf_render(self._h, pose.ctypes.data, bgr.ctypes.data, depth.ctypes.data)
        return bgr, depth

    def stats(self):
        out = (C.c_ulonglong * 4)()
        self._L.tsdf_stats(self._h, out)
        return dict(blocks=int(out[0]), updated_last=int(out[1]), updated_total=int(out[2]), mismatches=int(out[3]))

    def export_blocks(self):
        n = self._L.tsdf_num_blocks(self._h)
        coords = np.empty((max(n, 1), 3), np.int32)
        vox = np.empty((max(n, 1), 4096), np.uint8)
        got = self._L.tsdf_export_blocks(self._h, n, coords.ctypes.data, vox.ctypes.data)
        return {tuple(int(v) for v in coords[i]): vox[i] for i in range(got)}

    def extract_mesh(self, lower, upper, max_tri=2_000_000):
        """(vert, cols): float32 arrays of shape (3 * ntri, 3) in the GetMeshSync layout (cols are RGB in [0, 1])."""
        lo = np.ascontiguousarray(lower, np.float32)
        up = np.ascontiguousarray(upper, np.float32)
        vert, cols = np.empty((max_tri * 3, 3), np.float32), np.empty((max_tri * 3, 3), np.float32)
        n = self._L.tsdf_extract_mesh(self._h, lo.ctypes.data, up.ctypes.data, max_tri, vert.ctypes.data, cols.ctypes.data)
        if n > max_tri:
            raise RuntimeError("oracle mesh has %d triangles > max_tri=%d" % (n, max_tri))
        return vert[:3 * n].copy(), cols[:3 * n].copy()
