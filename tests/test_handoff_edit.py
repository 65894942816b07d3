"""The integration edit of CoarseTracker::setCoarseTrackingRef as CODE: tandem_amd/libdr/patches/CoarseTracker_dense_handoff.inc
is what replaces CoarseTracker.cpp:654-723 in a TANDEM build that tracks on the MI355X.  It is compiled here (tests/cpp/handoff_edit.cpp) against the
header-compatible shim, in the same frame of member names as the reference's own block, run on the GPU and compared with that block
(oracle/_ref/libdense_handoff_ref.so): same appended points, bit for bit, minus the reference's pre-increment defect."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build(tmp_path):
    so = str(tmp_path / "libhandoff_edit.so")
    subprocess.check_call(["g++", "-std=c++14", "-O2", "-Wall", "-Werror", "-fPIC", "-shared", "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"),
                           "-I" + os.path.join(ROOT, "tandem_amd", "libdr"), "-I" + os.path.join(ROOT, "tandem_amd", "libdr", "patches"),
                           "-I" + os.path.join(ROOT, "oracle", "ref_stub_eigen"), os.path.join(ROOT, "tests", "cpp", "handoff_edit.cpp"), "-o", so,
                           "-L" + os.path.join(ROOT, "tandem_amd"), "-ldr_mi355x", "-Wl,-rpath," + os.path.join(ROOT, "tandem_amd")])
    return so


def test_the_edit_compiles_against_the_shim(tmp_path):
    assert os.path.isfile(build(tmp_path))


@pytest.mark.gpu
@pytest.mark.parametrize("H,W,step,dense_only", [(480, 640, 1, True), (96, 128, 2, False)])
def test_the_edit_equals_the_reference_block(tmp_path, H, W, step, dense_only):
    from oracle import ref_handoff
    if not ref_handoff.available():
        pytest.skip("oracle/_ref/libdense_handoff_ref.so not built")
    from synth import scene
    p = scene.make_tracking_pair(H, W, seed=1, sparse_fraction=0.03)
    K = np.array([[p["fx"], 0, p["cx"]], [0, p["fy"], p["cy"]], [0, 0, 1]], np.float32)
    sparse = (p["pc_u"], p["pc_v"], p["pc_idepth"], p["pc_color"])
    ref = ref_handoff.dense_handoff(p["depth_new"], p["c2w_new"], p["c2w_ref"], K, step, dense_only, p["idepth0"], p["dI_ref"], sparse)
    L = C.CDLL(build(tmp_path))
    vp = C.c_void_p
    L.edit_dense_handoff.restype = C.c_int
    L.edit_dense_handoff.argtypes = [C.c_int, C.c_int, vp, vp, vp, vp, vp, C.c_int, C.c_int, vp, vp, C.c_int, vp, vp, vp, vp, C.c_int, vp, vp]
    n0, cap = len(sparse[0]), W * H + len(sparse[0]) + 1
    arr = [np.zeros(cap, np.float32) for _ in range(4)]
    for a, s in zip(arr, sparse):
        a[:n0] = s
    f32 = lambda a: np.ascontiguousarray(a, np.float32)  # noqa: E731
    depth, c2wd, c2wl = f32(p["depth_new"]), f32(p["c2w_new"]).reshape(16), np.ascontiguousarray(p["c2w_ref"], np.float64).reshape(16)
    id0, dI = f32(p["idepth0"]), f32(p["dI_ref"])
    KRKi, Kt = np.zeros(9, np.float32), np.zeros(3, np.float32)
    n = L.edit_dense_handoff(W, H, depth.ctypes.data, c2wd.ctypes.data, c2wl.ctypes.data, K.ctypes.data, ref["Ki"].ctypes.data, step, int(dense_only),
                             id0.ctypes.data, dI.ctypes.data, n0, *[a.ctypes.data for a in arr], cap, KRKi.ctypes.data, Kt.ctypes.data)
    assert n == ref["pc_n"], (n, ref["pc_n"])
    assert np.array_equal(KRKi.view(np.uint32), ref["KRKi"].view(np.uint32)) and np.array_equal(Kt.view(np.uint32), ref["Kt"].view(np.uint32))
    for name, mine, theirs in zip("u v idepth color".split(), arr, (ref["u"], ref["v"], ref["idepth"], ref["color"])):
        assert np.array_equal(mine[:n0].view(np.uint32), theirs[:n0].view(np.uint32)), name
        assert np.array_equal(mine[n0:n].view(np.uint32), theirs[n0 + 1:n + 1].view(np.uint32)), "%s: appended points differ from the reference block's" % name
    assert n - n0 > 0.2 * W * H / (step * step)


@pytest.mark.gpu
def test_without_a_dense_depth_the_edit_still_sets_the_sparse_reference(tmp_path):
    """the .inc replaces the reference's trailing `setReference` (:732) too, so the branch without a rendered depth map must make that
    call itself -- the tracker then holds exactly the sparse points."""
    from synth import scene
    H, W = 96, 128
    p = scene.make_tracking_pair(H, W, seed=3, sparse_fraction=0.05)
    K = np.array([[p["fx"], 0, p["cx"]], [0, p["fy"], p["cy"]], [0, 0, 1]], np.float32)
    Ki = np.linalg.inv(K.astype(np.float64)).astype(np.float32)
    sparse = [np.ascontiguousarray(a, np.float32) for a in (p["pc_u"], p["pc_v"], p["pc_idepth"], p["pc_color"])]
    L = C.CDLL(build(tmp_path))
    vp = C.c_void_p
    L.edit_dense_handoff.restype = C.c_int
    L.edit_dense_handoff.argtypes = [C.c_int, C.c_int, vp, vp, vp, vp, vp, C.c_int, C.c_int, vp, vp, C.c_int, vp, vp, vp, vp, C.c_int, vp, vp]
    n0, cap = len(sparse[0]), W * H + len(sparse[0]) + 1
    arr = [np.full(cap, np.nan, np.float32) for _ in range(4)]
    for a, s in zip(arr, sparse):
        a[:n0] = s
    f32 = lambda a: np.ascontiguousarray(a, np.float32)  # noqa: E731
    c2wd, c2wl = f32(p["c2w_new"]).reshape(16), np.ascontiguousarray(p["c2w_ref"], np.float64).reshape(16)
    id0, dI = f32(p["idepth0"]), f32(p["dI_ref"])
    KRKi, Kt = np.zeros(9, np.float32), np.zeros(3, np.float32)
    n = L.edit_dense_handoff(W, H, None, c2wd.ctypes.data, c2wl.ctypes.data, K.ctypes.data, Ki.ctypes.data, 1, 1, id0.ctypes.data, dI.ctypes.data, n0,
                             *[a.ctypes.data for a in arr], cap, KRKi.ctypes.data, Kt.ctypes.data)
    assert n == n0 > 0
    for mine, s in zip(arr, sparse):
        assert np.array_equal(mine[:n0].view(np.uint32), s.view(np.uint32))
