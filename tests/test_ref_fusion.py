"""CPU: oracle/tsdf_oracle.c (the restatement every -m gpu TSDF test checks the HIP path against) PINNED to the
REFERENCE ITSELF -- the reference's own dr_fusion sources compiled for the host by oracle/Makefile.ref
(oracle/_ref/libdr_fusion_ref.so: kernels run serially, one legal schedule of the CUDA program).

* helper by helper: Voxel::Combine (voxel.h:21-50; colour blend exhaustive over c, vc in 0..255 and w in 0..64),
  GetPoint3d / Project / norm (utils.h:44-108), float4x4 ctor, *float3, getInverse (matrix_utils.h:821-826,914-922,
  958-1083), World->GlobalVoxel/Block/LocalVoxel (tsdf_volume.cu:103-145);
* end to end: DrFusion::IntegrateScanAsync / RenderAsync / GetRenderResult / GetMesh (dr_fusion.cpp) on the same scans:
  allocated set, every voxel (sdf bits, BGR, weight), ray-cast depth + colour: bit-exact; mesh: equal triangle sets;
* the reference's (0,0,0)-alias defect (tsdf_volume.cu:451-455 walks free hash entries, which all carry position
  (0,0,0)) is demonstrated and confined: with block (0,0,0) inside the frustum everything BUT that block is identical.

Skipped when neither oracle/_ref/libdr_fusion_ref.so nor /root/reference exists."""
import ctypes as C

import numpy as np
import pytest

from oracle import ref_fusion
from synth import scene
from oracle import tsdf_oracle

pytestmark = pytest.mark.skipif(not ref_fusion.available(), reason="reference build (oracle/_ref) not available")


def _olib():
    L = tsdf_oracle.lib()
    L.tsdf_pin_combine.argtypes = [C.c_float, C.c_void_p, C.c_ubyte, C.c_float, C.c_void_p, C.c_ubyte, C.c_ubyte,
                                   C.c_void_p, C.c_void_p, C.c_void_p]
    L.tsdf_pin_combine_colour_table.argtypes = [C.c_ubyte, C.c_void_p]
    L.tsdf_pin_point3d.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p]
    L.tsdf_pin_project.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    L.tsdf_pin_norm.restype = C.c_float
    L.tsdf_pin_norm.argtypes = [C.c_void_p]
    L.tsdf_pin_xform.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.tsdf_pin_world_maps.argtypes = [C.c_void_p] * 6
    return L


def options(sc, H, W, vs, **kw):
    d = dict(voxel_size=vs, num_buckets=40000, bucket_size=10, num_blocks=40000, block_size=8, max_sdf_weight=64,
             truncation_distance=4 * vs, max_sensor_depth=10.0, min_sensor_depth=0.1, num_render_streams=1,
             fx=sc["fx"], fy=sc["fy"], cx=sc["cx"], cy=sc["cy"], height=H, width=W)
    d.update(kw)
    return d


def test_combine_colour_exhaustive():
    """every (c, vc, w): 256 x 256 x 65 colour blends of Voxel::Combine, reference vs restatement."""
    R, O = ref_fusion.lib(), _olib()
    a, b = np.empty(65536, np.uint8), np.empty(65536, np.uint8)
    for w in range(0, 65):
        R.ref_combine_colour_table(w, a.ctypes.data)
        O.tsdf_pin_combine_colour_table(w, b.ctypes.data)
        if w == 0:
            # 0/0 weights never occur (vw = 1) but w = 0 does: (c*0 + vc*1)/(0+1) = vc
            assert np.array_equal(a.reshape(256, 256)[0], np.arange(256, dtype=np.uint8))
        assert np.array_equal(a, b), f"w={w}: {np.count_nonzero(a != b)} colour blends differ"


def test_combine_sdf_weight_and_cap():
    R, O = ref_fusion.lib(), _olib()
    rng = np.random.RandomState(0)
    c, vc = np.zeros(3, np.uint8), np.zeros(3, np.uint8)
    for it in range(20000):
        sdf = np.float32(rng.uniform(-0.05, 0.05)) if it % 7 else np.float32(rng.choice([0.0, 0.02, -0.02, 1e-30, 3e38]))
        vsdf = np.float32(rng.uniform(-0.05, 0.05)) if it % 5 else np.float32(0.02)
        w, vw, mw = int(rng.randint(0, 256)), int(rng.randint(1, 3)), int(rng.choice([64, 255, 1, 8]))
        c[:] = rng.randint(0, 256, 3)
        vc[:] = rng.randint(0, 256, 3)
        outs = []
        for fn in (R.ref_combine, O.tsdf_pin_combine):
            s, co, wo = np.zeros(1, np.float32), np.zeros(3, np.uint8), np.zeros(1, np.uint8)
            fn(sdf, c.ctypes.data, w, vsdf, vc.ctypes.data, vw, mw, s.ctypes.data, co.ctypes.data, wo.ctypes.data)
            outs.append((s.view(np.uint32)[0], tuple(co), wo[0]))
        assert outs[0] == outs[1], (it, sdf, w, vsdf, vw, mw, outs)


def test_geometry_helpers():
    R, O = ref_fusion.lib(), _olib()
    rng = np.random.RandomState(1)
    rows, cols = 480, 640
    k4 = np.array([500.0, 498.5, 319.5, 239.5], np.float32)
    a3, b3 = np.zeros(3, np.float32), np.zeros(3, np.float32)
    a2, b2 = np.zeros(2, np.int32), np.zeros(2, np.int32)
    for it in range(20000):
        i = int(rng.randint(0, rows * cols))
        d = np.float32(rng.uniform(0.05, 12.0))
        R.ref_point3d(k4.ctypes.data, rows, cols, i, d, a3.ctypes.data)
        O.tsdf_pin_point3d(k4.ctypes.data, rows, cols, i, d, b3.ctypes.data)
        assert np.array_equal(a3.view(np.uint32), b3.view(np.uint32))
        # Project: points in front of / behind the camera, on half-pixel ties, far off-image (in int range)
        p = rng.uniform(-3, 3, 3).astype(np.float32)
        if it % 4 == 0:
            z = np.float32(rng.uniform(0.2, 4.0))
            u = np.float32(rng.randint(-50, 700) + 0.5)   # round() ties: half away from zero
            p = np.array([(u - k4[2]) * z / k4[0], (np.float32(rng.randint(-50, 500)) + 0.5 - k4[3]) * z / k4[1], z], np.float32)
        if it % 97 == 0:
            p[2] = np.float32(rng.choice([1e-3, -1e-3, 1e-4]))
        R.ref_project(k4.ctypes.data, rows, cols, p.ctypes.data, a2.ctypes.data)
        O.tsdf_pin_project(k4.ctypes.data, rows, cols, p.ctypes.data, b2.ctypes.data)
        assert np.array_equal(a2, b2), (p, a2, b2)
        assert np.float32(R.ref_norm(p.ctypes.data)).view(np.uint32) == np.float32(O.tsdf_pin_norm(p.ctypes.data)).view(np.uint32)


def test_pose_algebra():
    R, O = ref_fusion.lib(), _olib()
    rng = np.random.RandomState(2)
    a16, b16 = np.zeros(16, np.float32), np.zeros(16, np.float32)
    a3, b3 = np.zeros(3, np.float32), np.zeros(3, np.float32)
    for it in range(3000):
        T = scene._pose(*rng.uniform(-3.1, 3.1, 3), rng.uniform(-5, 5, 3)).astype(np.float32)
        if it % 10 == 0:
            T = rng.uniform(-2, 2, (4, 4)).astype(np.float32)  # general matrices too: same cofactor expansion
        m = np.ascontiguousarray(T).reshape(16)
        R.ref_inverse4(m.ctypes.data, a16.ctypes.data)
        O.tsdf_inverse4(m.ctypes.data, b16.ctypes.data)
        assert np.array_equal(a16.view(np.uint32), b16.view(np.uint32)), it
        p = rng.uniform(-4, 4, 3).astype(np.float32)
        R.ref_xform(m.ctypes.data, p.ctypes.data, a3.ctypes.data)
        O.tsdf_pin_xform(m.ctypes.data, p.ctypes.data, b3.ctypes.data)
        assert np.array_equal(a3.view(np.uint32), b3.view(np.uint32)), it


@pytest.mark.parametrize("vs,bs", [(0.01, 8), (0.005, 8), (0.04, 4)])
def test_coordinate_maps(vs, bs):
    sc = scene.make_scans(1, 8, 8)
    opt = options(sc, 8, 8, vs, block_size=bs, num_buckets=64, num_blocks=64)
    r, o = ref_fusion.RefFusion(**opt), tsdf_oracle.TsdfOracle(**opt)
    O = _olib()
    rng = np.random.RandomState(3)
    g, b, l = (np.zeros(3, np.int32) for _ in range(3))
    w = np.zeros(3, np.float32)
    for it in range(20000):
        p = rng.uniform(-6, 6, 3).astype(np.float32)
        if it % 3 == 0:   # exactly on voxel borders / half voxels / zero / negative zero
            p = (rng.randint(-900, 900, 3) * np.float32(0.5) * np.float32(vs)).astype(np.float32)
        if it % 101 == 0:
            p[rng.randint(3)] = np.float32(-0.0)
        rg, rb, rl, rw = r.world_maps(p)
        O.tsdf_pin_world_maps(o._h, p.ctypes.data, g.ctypes.data, b.ctypes.data, l.ctypes.data, w.ctypes.data)
        assert np.array_equal(rg, g) and np.array_equal(rb, b) and np.array_equal(rl, l), (p, rg, g, rb, b, rl, l)
        assert np.array_equal(rw.view(np.uint32), w.view(np.uint32))
    r.close()


def _run_both(sc, opt, shift=None, n_render=1):
    r, o = ref_fusion.RefFusion(**opt), tsdf_oracle.TsdfOracle(**opt)
    S = np.eye(4, dtype=np.float32)
    if shift is not None:
        S[:3, 3] = shift
    n = len(sc["scans"])
    for i, (bgr, depth, pose) in enumerate(sc["scans"]):
        pose = (S @ pose).astype(np.float32)
        r.integrate(bgr, depth, pose)
        assert o.integrate(bgr, depth, pose) == 0
        views = [(S @ sc["scans"][(i + 1 + k) % n][2]).astype(np.float32) for k in range(n_render)]
        rr = r.render(views)
        for v, (rb, rd) in zip(views, rr):
            ob, od = o.render(v)
            assert np.array_equal(rd.view(np.uint32), od.view(np.uint32)), f"scan {i}: ray-cast depth differs at {(rd != od).sum()} px"
            assert np.array_equal(rb, ob), f"scan {i}: ray-cast colour differs"
            assert (od > 0).sum() > 0.3 * od.size
    return r, o


@pytest.mark.parametrize("H,W,vs,n", [(96, 128, 0.02, 4), (120, 160, 0.01, 3), (64, 64, 0.04, 6), (60, 80, 0.005, 2)])
def test_reference_drfusion_equals_restatement(H, W, vs, n):
    sc = scene.make_scans(n, H, W, seed=H + n)
    opt = options(sc, H, W, vs, num_buckets=400000, num_blocks=400000) if vs < 0.01 else options(sc, H, W, vs)
    r, o = _run_both(sc, opt)
    a, b = r.export_blocks(), o.export_blocks()
    assert a.keys() == b.keys(), f"allocated sets differ: {len(a)} vs {len(b)}"
    bad = [k for k in a if not np.array_equal(a[k], b[k])]
    assert not bad, f"{len(bad)} of {len(a)} blocks differ, e.g. {bad[:3]}"
    assert lib_counter(r) == len(a)
    r.close()


def lib_counter(r):
    return ref_fusion.lib().refdrf_num_allocated_counter(r._h)


def test_two_render_streams_and_weight_cap():
    """num_render_streams = 2 and max_sdf_weight = 3 (the cap is hit after three scans)."""
    H, W = 64, 96
    sc = scene.make_scans(6, H, W, seed=5)
    opt = options(sc, H, W, 0.02, num_render_streams=2, max_sdf_weight=3)
    r, o = _run_both(sc, opt, n_render=2)
    a, b = r.export_blocks(), o.export_blocks()
    assert a.keys() == b.keys()
    assert all(np.array_equal(a[k], b[k]) for k in a)
    assert max(v.reshape(-1, 8)[:, 7].max() for v in a.values()) == 3
    r.close()


def test_origin_block_alias_is_the_only_deviation():
    """The reference integrates every FREE hash entry as block (0,0,0) (tsdf_volume.cu:451-455: no `pointer != kFreeEntry`
    test and InitEntriesKernel sets position (0,0,0), hash_table.cu:12-19).  With block (0,0,0) in front of the camera its
    voxels are therefore combined once per free entry per scan; the restatement (and the HIP path) integrate it once.
    Everything else must still be identical."""
    H, W, vs = 64, 96, 0.02
    sc = scene.make_scans(3, H, W, seed=11)
    opt = options(sc, H, W, vs, num_buckets=2000, num_blocks=4000)
    r, o = ref_fusion.RefFusion(**opt), tsdf_oracle.TsdfOracle(**opt)
    S = np.eye(4, dtype=np.float32)
    S[:3, 3] = (-0.07, -0.07, -0.9)     # world origin 0.9 m in front of the first camera: block (0,0,0) is carved free space
    for bgr, depth, pose in sc["scans"]:
        pose = (S @ pose).astype(np.float32)
        r.integrate(bgr, depth, pose)
        o.integrate(bgr, depth, pose)
        r.render([pose])
    a, b = r.export_blocks(), o.export_blocks()
    assert a.keys() == b.keys() and (0, 0, 0) in a
    bad = [k for k in a if not np.array_equal(a[k], b[k])]
    assert bad == [(0, 0, 0)], bad
    wr, wo = a[(0, 0, 0)].reshape(-1, 8)[:, 7], b[(0, 0, 0)].reshape(-1, 8)[:, 7]
    assert wo.max() <= 3 and wr.max() == 64     # three scans vs thousands of aliased updates, capped at max_sdf_weight
    r.close()


def _tri_set(vert, cols):
    t = np.concatenate([vert.reshape(-1, 9), cols.reshape(-1, 9)], axis=1).view(np.uint32)
    return t[np.lexsort(t.T[::-1])]


@pytest.mark.parametrize("lo,hi", [((-1.0, -1.0, 0.5), (1.0, 1.0, 3.0)), ((-0.503, -0.4, 1.0), (0.31, 0.4, 2.6))])
def test_reference_mesh_equals_restatement(lo, hi):
    """DrFusion::GetMesh (dense-lattice ExtractMeshKernel, mesh_extractor.cu:244-265) vs the restatement: same triangle SET
    (the reference appends with atomicAdd, so order is not part of the result)."""
    H, W, vs = 64, 96, 0.04
    sc = scene.make_scans(3, H, W, seed=2)
    opt = options(sc, H, W, vs)
    r, o = _run_both(sc, opt)
    rv, rc = r.extract_mesh(lo, hi)
    ov, oc = o.extract_mesh(lo, hi)
    assert len(rv) == len(ov) and len(rv) > 300
    assert np.array_equal(_tri_set(rv, rc), _tri_set(ov, oc))
    r.close()
