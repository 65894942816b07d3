"""Python mirror of TANDEM's `DrFusion` operator (tandem/libdr/dr_fusion/src/dr_fusion/dr_fusion.h:18-73)
on top of the C ABI of libdr_mi355x.so: same method names, call order contract and argument meaning.
Protocol violations raise DrError (the reference prints and exit()s)."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import FusionOptions, check, fptr, u8p, f32p


def DrFusionOptions(**kw):
    """struct DrFusionOptions (dr_fusion.h:18-36).  Defaults = what TANDEM runs (FullSystem.cpp:259-276)."""
    d = dict(voxel_size=0.01, num_buckets=1000000, bucket_size=10, num_blocks=1000000, block_size=8,
             max_sdf_weight=64, truncation_distance=0.04, max_sensor_depth=10.0, min_sensor_depth=0.1,
             num_render_streams=1, fx=500.0, fy=500.0, cx=319.5, cy=239.5, height=480, width=640)
    d.update(kw)
    return FusionOptions(**d)


class DrFusion:
    def __init__(self, options, device=0):
        self.options = options
        self._h = C.c_void_p()
        self._L = _lib.lib()  # the library this handle belongs to (tests may switch the process default, _lib.switch)
        check(self._L.drf_create(C.byref(options), int(device), C.byref(self._h)))
        self._hw = (options.height, options.width)

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._L.drf_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close

    def IntegrateScanAsync(self, bgr, depth, pose):
        """dr_fusion.h:50: bgr H*W*3 u8, depth H*W f32 metres (0 invalid), pose 16 f32 row-major cam-to-world."""
        bgr = np.ascontiguousarray(bgr, np.uint8)
        depth = np.ascontiguousarray(depth, np.float32)
        pose = np.ascontiguousarray(pose, np.float32).reshape(16)
        assert bgr.size == self._hw[0] * self._hw[1] * 3 and depth.size == self._hw[0] * self._hw[1]
        check(self._L.drf_integrate_scan_async(self._h, bgr.ctypes.data_as(u8p), fptr(depth), fptr(pose)))

    def RenderAsync(self, camera_poses):
        """dr_fusion.h:52: exactly num_render_streams poses."""
        poses = [np.ascontiguousarray(p, np.float32).reshape(16) for p in camera_poses]
        arr = (f32p * max(len(poses), 1))(*[fptr(p) for p in poses])
        check(self._L.drf_render_async(self._h, arr, len(poses)))

    def GetRenderResult(self, copy=True):
        """dr_fusion.h:54: returns (bgr list, depth list).  copy=True: copies of the library-owned pinned buffers; copy=False: what the C++ member hands
        out -- views of those buffers, valid until the NEXT GetRenderResult (tsdf_volume.cu:846-872, the "blocked" / "free" sets)."""
        n = self.options.num_render_streams
        pb, pd = (u8p * max(n, 1))(), (f32p * max(n, 1))()
        check(self._L.drf_get_render_result(self._h, pb, pd, n))
        H, W = self._hw
        bgrs = [np.ctypeslib.as_array(pb[i], shape=(H, W, 3)) for i in range(n)]
        depths = [np.ctypeslib.as_array(pd[i], shape=(H, W)) for i in range(n)]
        if copy:
            bgrs, depths = [b.copy() for b in bgrs], [d.copy() for d in depths]
        return bgrs, depths

    def ExtractMeshAsync(self, lower_corner, upper_corner):
        """dr_fusion.h:60: marching cubes over the lattice lower + g * voxel_size, legal after GetRenderResult."""
        lo, up = (np.ascontiguousarray(a, np.float32) for a in (lower_corner, upper_corner))
        check(self._L.drf_extract_mesh_async(self._h, fptr(lo), fptr(up)))

    def GetMeshSync(self):
        """dr_fusion.h:61: fills the public members dr_mesh_num (vertices = 3 * triangles), dr_mesh_vert, dr_mesh_cols
        ((num, 3) float32: positions, RGB colours in [0, 1]) and returns (vert, cols)."""
        ntri = C.c_size_t()
        check(self._L.drf_mesh_num_triangles(self._h, C.byref(ntri)))
        nv = 3 * ntri.value
        vert, cols = np.empty((max(nv, 1), 3), np.float32), np.empty((max(nv, 1), 3), np.float32)
        num = C.c_size_t()
        check(self._L.drf_get_mesh_sync(self._h, max(nv, 1), C.byref(num), fptr(vert), fptr(cols)))
        self.dr_mesh_num, self.dr_mesh_vert, self.dr_mesh_cols = int(num.value), vert[:nv], cols[:nv]
        return self.dr_mesh_vert, self.dr_mesh_cols

    def mesh_num_triangles(self):
        """Size of the pending mesh (waits for the extraction, does not consume it)."""
        ntri = C.c_size_t()
        check(self._L.drf_mesh_num_triangles(self._h, C.byref(ntri)))
        return int(ntri.value)

    def GetMesh(self, lower_corner, upper_corner):
        """dr_fusion.h:58 (DrMesh): synchronous extraction."""
        self.ExtractMeshAsync(lower_corner, upper_corner)
        return self.GetMeshSync()

    def SaveMeshToFile(self, filename, lower_corner, upper_corner):
        lo, up = (np.ascontiguousarray(a, np.float32) for a in (lower_corner, upper_corner))
        check(self._L.drf_save_mesh(self._h, str(filename).encode(), fptr(lo), fptr(up)))

    def render_device_pointers(self, stream=0):
        """(d_bgr, d_depth) device pointers of a render stream's result, valid until the next RenderAsync."""
        b, d = C.c_void_p(), C.c_void_p()
        check(self._L.drf_get_render_device(self._h, stream, C.byref(b), C.byref(d)))
        return b.value, d.value

    def Synchronize(self):
        check(self._L.drf_synchronize(self._h))

    # ---- introspection / measurement hooks (no reference counterpart) ----
    def stats(self):
        out = (C.c_uint64 * 4)()
        check(self._L.drf_stats(self._h, out))
        return dict(blocks=int(out[0]), updated_last=int(out[1]), updated_total=int(out[2]), mismatches=int(out[3]))

    def visited_blocks(self):
        """Blocks k_integrate has read so far (4 KB each): with stats()["updated_total"] the kernel's exact HBM bytes."""
        v = C.c_uint64()
        check(self._L.drf_visited_blocks(self._h, C.byref(v)))
        return int(v.value)

    def export_blocks(self):
        """Canonical dump: dict {(bx,by,bz): uint8[4096]} (512 voxels x {f32 sdf, u8 b,g,r, u8 weight})."""
        n = self.stats()["blocks"]
        coords = np.empty((max(n, 1), 3), np.int32)
        vox = np.empty((max(n, 1), 4096), np.uint8)
        got = C.c_int()
        check(self._L.drf_export_blocks(self._h, n, coords.ctypes.data_as(C.POINTER(C.c_int32)),
                                           vox.ctypes.data_as(u8p), C.byref(got)))
        return {tuple(int(v) for v in coords[i]): vox[i] for i in range(got.value)}

    def fast_div_status(self):
        """(enabled, mismatches) of the exact fast division self-check run at construction."""
        en, mm = C.c_int(), C.c_uint64()
        check(self._L.drf_fast_div_status(self._h, C.byref(en), C.byref(mm)))
        return bool(en.value), int(mm.value)

    def test_combine(self, a, b, max_weight):
        """Test hook: Combine(a[i], b[i]) by the integration kernel's device function; a, b: (n, 8) uint8 voxels."""
        a, b = np.ascontiguousarray(a, np.uint8), np.ascontiguousarray(b, np.uint8)
        assert a.shape == b.shape and a.shape[1] == 8
        out = np.empty_like(a)
        check(self._L.drf_test_combine(self._h, a.shape[0], a.ctypes.data_as(u8p), b.ctypes.data_as(u8p), int(max_weight),
                                          out.ctypes.data_as(u8p)))
        return out

    def bench_sequence(self, d_bgr, d_depth, poses, render=True):
        """BASELINE configs[3] loop over frames resident in HBM (device pointers; poses (n, 16) float32): dict of
        milliseconds -- total (hipEvents), allocate / integrate / raycast / d2h sums, host wall clock."""
        ps = np.ascontiguousarray(poses, np.float32).reshape(-1, 16)
        ms = (C.c_float * 6)()
        check(self._L.drf_bench_sequence(self._h, C.c_void_p(d_bgr), C.c_void_p(d_depth), fptr(ps), ps.shape[0], int(bool(render)), ms))
        return dict(total=ms[0], allocate=ms[1], integrate=ms[2], raycast=ms[3], d2h=ms[4], wall=ms[5])

    def bench_last_render(self, back=0, stream=0):
        """(bgr, depth) COPIES of the last (back=0) / second-to-last (back=1) ray-cast bench_sequence wrote for `stream` (test hook)."""
        b, d = C.c_void_p(), C.c_void_p()
        check(self._L.drf_bench_render_host(self._h, stream, back, C.byref(b), C.byref(d)))
        H, W = self._hw
        bgr = np.ctypeslib.as_array(C.cast(b, u8p), shape=(H, W, 3)).copy()
        depth = np.ctypeslib.as_array(C.cast(d, C.POINTER(C.c_float)), shape=(H, W)).copy()
        return bgr, depth

    def bench_integrate(self, bgrs, depths, poses):
        """Uploads the scans once, then times back-to-back allocate+integrate of all of them (HBM-resident)."""
        L = self._L
        n = len(bgrs)
        bg = np.ascontiguousarray(np.stack(bgrs), np.uint8)
        dp = np.ascontiguousarray(np.stack(depths), np.float32)
        ps = np.ascontiguousarray(np.stack([np.asarray(p, np.float32).reshape(16) for p in poses]), np.float32)
        d_b, d_d = C.c_void_p(), C.c_void_p()
        check(L.dr_device_alloc(0, bg.nbytes, C.byref(d_b)))
        check(L.dr_device_alloc(0, dp.nbytes, C.byref(d_d)))
        try:
            check(L.dr_memcpy_h2d(d_b, bg.ctypes.data_as(C.c_void_p), bg.nbytes))
            check(L.dr_memcpy_h2d(d_d, dp.ctypes.data_as(C.c_void_p), dp.nbytes))
            ms, kms = C.c_float(), C.c_float()
            check(L.drf_bench_integrate(self._h, d_b, d_d, fptr(ps), n, C.byref(ms), C.byref(kms)))
        finally:
            L.dr_device_free(d_b)
            L.dr_device_free(d_d)
        return ms.value, kms.value
