"""-m gpu: the HIP dense coarse tracker through the C ABI (drt_*; mirror of class CudaCoarseTracker,
cuda_coarse_tracker.h:9-35, plus the dense-depth hand-off of CoarseTracker.cpp:655-725) against the CPU oracle:
every per-point quantity BIT-EXACT (warped u/v/dx/dy/idepth/residual/weight, projected z-buffer, appended points);
the 7 + 45 reductions to SUM_RTOL (both sides accumulate fp32 terms in double, only the order differs -- the
reference itself accumulates in float with atomics, ~1e-6 noise)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
SUM_RTOL = 1e-11


def pair(H, W, seed, frac):
    from synth import scene
    return scene.make_tracking_pair(H, W, seed=seed, sparse_fraction=frac)


def both(p, n_max=0):
    from oracle.tracker_oracle import TrackerOracle
    from tandem_amd.dr_tracker import DrCoarseTracker
    W, H = p["width"], p["height"]
    g, o = DrCoarseTracker(W, H, 9.0, 20.0), TrackerOracle(W, H, 9.0, 20.0, n_max)
    g.setK(W, H, p["fx"], p["fy"], p["cx"], p["cy"])
    g.init(n_max)
    o.setK(p["fx"], p["fy"], p["cx"], p["cy"])
    return g, o


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.mark.parametrize("H,W,frac", [(120, 160, 0.05), (480, 640, 0.02), (480, 640, 1.0), (96, 128, 0.0)])
def test_calc_res_and_calc_g(H, W, frac):
    p = pair(H, W, H + int(100 * frac), frac)
    g, o = both(p)
    aff_ref, aff_new = [0.02, 1.5], [-0.01, -0.7]
    for t in (g, o):
        t.setReference(p["pc_u"], p["pc_v"], p["pc_idepth"], p["pc_color"], 1.3, aff_ref)
        t.setNew(p["dI_new"])
    for T, cutoff in ((p["refToNew"], 20.0), (np.eye(4), 20.0), (p["refToNew"], 40.0)):
        out_g, sums_g = g.calcRes(T, 0.9, aff_new, cutoff, return_sums=True)
        out_o, sums_o = o.calcRes(T, 0.9, aff_new, cutoff)
        for k, (a, b) in enumerate(zip(g.warped(), o.warped())):
            assert np.array_equal(bits(a), bits(b)), f"warped[{k}] differs at {(bits(a) != bits(b)).sum()} of {len(a)} points"
        assert np.allclose(sums_g, sums_o, rtol=SUM_RTOL, atol=0), (sums_g, sums_o)
        assert sums_g[1] == sums_o[1] and sums_g[2] == sums_o[2] and sums_g[3] == sums_o[3] and sums_g[6] == sums_o[6]
        if len(p["pc_u"]):
            assert np.allclose(out_g, out_o, rtol=SUM_RTOL, atol=0)
        Hg, bg, rg = g.calcG(0.9, aff_new, return_raw=True)
        Ho, bo, ro = o.calcG(0.9, aff_new)
        if sums_o[2] > 0:
            assert np.allclose(rg, ro, rtol=SUM_RTOL, atol=1e-9 * np.abs(ro).max())
            assert np.allclose(Hg, Ho, rtol=1e-9, atol=1e-9 * np.abs(Ho).max()) and np.allclose(bg, bo, rtol=1e-9, atol=1e-9 * np.abs(bo).max())
            assert np.array_equal(Hg, Hg.T)
    # repeatable bit for bit (fixed reduction order)
    a = g.calcRes(p["refToNew"], 0.9, aff_new, 20.0, return_sums=True)[1]
    b = g.calcRes(p["refToNew"], 0.9, aff_new, 20.0, return_sums=True)[1]
    assert np.array_equal(a, b)
    g.close()


@pytest.mark.parametrize("H,W,step,dense_only", [(120, 160, 1, True), (480, 640, 1, False), (480, 640, 2, False), (96, 128, 3, True)])
def test_append_dense_reference_bit_exact(H, W, step, dense_only):
    """The hand-off as TANDEM does it: the depth map rendered for the NEW keyframe pose is warped into the tracker's
    reference frame with KRKi = K R K^-1, Kt = K t (float products, CoarseTracker.cpp:670-673)."""
    p = pair(H, W, 11 + step, 0.03)
    g, o = both(p, n_max=H * W)
    K = np.array([[p["fx"], 0, p["cx"]], [0, p["fy"], p["cy"]], [0, 0, 1]], np.float32)
    Ki = np.linalg.inv(K.astype(np.float64)).astype(np.float32)
    T = np.linalg.inv(p["c2w_ref"]) @ p["c2w_new"]  # dense depth (new frame) -> reference frame
    KRKi = (K @ T[:3, :3].astype(np.float32)) @ Ki
    Kt = K @ T[:3, 3].astype(np.float32)
    depth = p["depth_new"].copy()
    depth[::7, ::5] = 0.0
    id0 = None if dense_only else p["idepth0"]
    for t in (g, o):
        t.setReference(p["pc_u"], p["pc_v"], p["pc_idepth"], p["pc_color"], 1.0, [0, 0])
    n_g = g.appendDenseReference(depth, KRKi, Kt, step, dense_only, id0, p["dI_ref"])
    n_o, proj_o = o.appendDenseReference(depth, KRKi, Kt, step, dense_only, id0, p["dI_ref"])
    assert n_g == n_o and n_o > len(p["pc_u"]) + 1000
    assert np.array_equal(bits(g.zbuffer()), bits(proj_o))
    for k, (a, b) in enumerate(zip(g.points(), o.points())):
        assert np.array_equal(bits(a), bits(b)), f"point array {k} differs"
    # the appended list is usable: a calcRes over sparse + dense points agrees with the oracle
    for t in (g, o):
        t.setNew(p["dI_new"])
    sg = g.calcRes(p["refToNew"], 1.0, [0, 0], 20.0, return_sums=True)[1]
    so = o.calcRes(p["refToNew"], 1.0, [0, 0], 20.0)[1]
    assert np.allclose(sg, so, rtol=SUM_RTOL, atol=0) and so[1] > 0.5 * n_o
    g.close()


def test_device_pointers_protocol_and_capacity():
    from tandem_amd import _lib
    from tandem_amd.dr_tracker import DrCoarseTracker
    p = pair(96, 128, 5, 0.05)
    H, W = 96, 128
    g = DrCoarseTracker(W, H, 9.0, 20.0)
    with pytest.raises(_lib.DrError, match="init has not been called"):
        g.setNew(p["dI_new"])
    with pytest.raises(_lib.DrError, match="wrong h,w"):
        g.setK(W + 1, H, 1, 1, 1, 1)
    g.setK(W, H, p["fx"], p["fy"], p["cx"], p["cy"])
    g.init(5000)
    with pytest.raises(_lib.DrError, match="more than once"):
        g.init(5000)
    with pytest.raises(_lib.DrError, match="n > n_max"):
        g.setReference(np.zeros(6000), np.zeros(6000), np.ones(6000), np.zeros(6000), 1.0, [0, 0])
    g.setReference(p["pc_u"], p["pc_v"], p["pc_idepth"], p["pc_color"], 1.0, [0, 0])
    with pytest.raises(_lib.DrError) as e:  # 96*128 dense points do not fit n_max = 5000
        g.appendDenseReference(p["depth_ref"], np.eye(3), np.zeros(3), 1, True, None, p["dI_ref"])
    assert e.value.code == 5
    assert len(g.points()[0]) == len(p["pc_u"])  # nothing was appended
    g.close()
    # device-pointer inputs (a render result left in HBM) give the same list as host inputs
    g = DrCoarseTracker(W, H, 9.0, 20.0)
    g.setK(W, H, p["fx"], p["fy"], p["cx"], p["cy"])
    g.init()
    g.setReference([], [], [], [], 1.0, [0, 0])
    n_host = g.appendDenseReference(p["depth_ref"], np.eye(3), np.zeros(3), 1, True, None, p["dI_ref"])
    host_pts = [a.copy() for a in g.points()]
    ptrs = []
    for arr in (p["depth_ref"], p["dI_ref"]):
        a = np.ascontiguousarray(arr, np.float32)
        d = C.c_void_p()
        _lib.check(_lib.lib().dr_device_alloc(0, a.nbytes, C.byref(d)))
        _lib.check(_lib.lib().dr_memcpy_h2d(d, a.ctypes.data_as(C.c_void_p), a.nbytes))
        ptrs.append(d)
    g.setReference([], [], [], [], 1.0, [0, 0])
    n_dev = g.appendDenseReference(ptrs[0].value, np.eye(3), np.zeros(3), 1, True, None, ptrs[1].value, device_pointers=True)
    assert n_dev == n_host
    for a, b in zip(g.points(), host_pts):
        assert np.array_equal(bits(a), bits(b))
    for d in ptrs:
        _lib.check(_lib.lib().dr_device_free(d))
    # timing hooks (cuda_coarse_tracker.cpp:374-396)
    g.startTiming()
    g.setNew(p["dI_new"])
    g.calcRes(np.eye(4), 1.0, [0, 0], 20.0)
    assert g.endTimingMilliseconds() > 0
    with pytest.raises(_lib.DrError, match="Did not start before"):
        g.endTimingMilliseconds()
    g.close()


def test_render_to_tracker_handoff_stays_on_the_device():
    """SURVEY 8(f) row 3 end to end: DrFusion renders the fused map for the keyframe pose, the tracker appends the
    dense reference points straight from the render's device buffer -- same list as through the host copy."""
    from synth import scene
    from tandem_amd.dr_fusion import DrFusion, DrFusionOptions
    from tandem_amd.dr_tracker import DrCoarseTracker
    H, W = 96, 128
    sc = scene.make_scans(3, H, W, seed=4)
    f = DrFusion(DrFusionOptions(voxel_size=0.02, num_buckets=40000, bucket_size=10, num_blocks=40000, block_size=8, max_sdf_weight=64,
                                 truncation_distance=0.08, max_sensor_depth=10.0, min_sensor_depth=0.1, num_render_streams=1,
                                 fx=sc["fx"], fy=sc["fy"], cx=sc["cx"], cy=sc["cy"], height=H, width=W))
    for bgr, depth, pose in sc["scans"]:
        f.IntegrateScanAsync(bgr, depth, pose)
        f.RenderAsync([pose])
        rb, rd = f.GetRenderResult()
    _, d_depth = f.render_device_pointers(0)
    dI = np.zeros((H, W, 3), np.float32)
    dI[..., 0] = rb[0].mean(axis=2)
    K = np.array([[sc["fx"], 0, sc["cx"]], [0, sc["fy"], sc["cy"]], [0, 0, 1]], np.float32)
    T = np.linalg.inv(sc["scans"][1][2].astype(np.float64)) @ sc["scans"][2][2].astype(np.float64)  # render frame -> tracker frame
    KRKi = (K @ T[:3, :3].astype(np.float32)) @ np.linalg.inv(K.astype(np.float64)).astype(np.float32)
    Kt = K @ T[:3, 3].astype(np.float32)
    lists = []
    for use_device in (False, True):
        g = DrCoarseTracker(W, H, 9.0, 20.0)
        g.setK(W, H, sc["fx"], sc["fy"], sc["cx"], sc["cy"])
        g.init()
        g.setReference([], [], [], [], 1.0, [0, 0])
        if use_device:
            import ctypes as C
            from tandem_amd import _lib
            a = np.ascontiguousarray(dI)
            d_dip = C.c_void_p()
            _lib.check(_lib.lib().dr_device_alloc(0, a.nbytes, C.byref(d_dip)))
            _lib.check(_lib.lib().dr_memcpy_h2d(d_dip, a.ctypes.data_as(C.c_void_p), a.nbytes))
            n = g.appendDenseReference(d_depth, KRKi, Kt, 1, True, None, d_dip.value, device_pointers=True)
            _lib.check(_lib.lib().dr_device_free(d_dip))
        else:
            n = g.appendDenseReference(rd[0], KRKi, Kt, 1, True, None, dI)
        assert n > 2000
        lists.append([a.copy() for a in g.points()])
        g.close()
    for a, b in zip(*lists):
        assert np.array_equal(bits(a), bits(b))
    f.close()


@pytest.mark.parametrize("H,W,frac", [(120, 160, 0.2), (240, 320, 0.5)])
def test_hip_kernels_equal_the_reference_build(H, W, frac):
    """The HIP tracker against the REFERENCE's own calcResKernelNew / calcGKernel compiled for the host
    (oracle/_ref/libcoarse_tracker_ref.so, oracle/Makefile.ref): warped buffers bit-exact; the reference's float
    block sums + atomicAdd against the HIP path's double sums to fp32 accumulation noise."""
    from oracle import ref_tracker
    if not ref_tracker.available():
        pytest.skip("oracle/_ref/libcoarse_tracker_ref.so not present")
    p = pair(H, W, 5, frac)
    g, o = both(p)
    aff_ref, aff_new, cutoff, expo = [0.02, 1.5], [0.03, 1.2], 12.0, 1.0
    for t in (g, o):
        t.setReference(p["pc_u"], p["pc_v"], p["pc_idepth"], p["pc_color"], 1.0, aff_ref)
        t.setNew(p["dI_new"])
    out_g, sums_g = g.calcRes(p["refToNew"], expo, aff_new, cutoff, return_sums=True)
    r2n, Ki, a2, maxE, rb = o.kernel_inputs(p["refToNew"], expo, aff_new, cutoff)   # host-side input preparation only
    rw, rout = ref_tracker.calc_res(9.0, W, H, p["fx"], p["fy"], p["cx"], p["cy"], r2n, Ki, a2, maxE, cutoff,
                                    p["pc_u"], p["pc_v"], p["pc_idepth"], p["pc_color"], p["dI_new"])
    for k, (a, b) in enumerate(zip(g.warped(), rw)):
        assert np.array_equal(bits(a), bits(b)), f"warped[{k}] differs from the reference kernel at {(bits(a) != bits(b)).sum()} points"
    for k in (1, 2, 3, 6):
        assert sums_g[k] == rout[k]
    assert rout[3] > 0 and rout[2] > 1000
    np.testing.assert_allclose(rout[[0, 4, 5]], np.asarray(sums_g)[[0, 4, 5]], rtol=2e-5)
    Hg, bg, rg = g.calcG(expo, aff_new, return_raw=True)
    rd1 = ref_tracker.calc_g(p["fx"], p["fy"], a2, rb, p["pc_color"], rw, loops=1, double=True)
    np.testing.assert_allclose(rg, rd1, rtol=1e-11, atol=1e-11 * np.abs(rd1).max())
    g.close()
