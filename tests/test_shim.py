"""The header-compatible C++ shim (tandem_amd/libdr/*.h) compiled with plain g++ exactly as a TANDEM translation
unit would include it, linked against the C ABI; run end-to-end on the GPU (test_dr_mvsnet semantics,
dr_mvsnet.cpp:376-556, then one DrFusion integrate/render round)."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build(tmp_path):
    import __graft_entry__ as g
    if not os.path.isfile(os.path.join(ROOT, "tandem_amd", "libdr_mi355x.so")):
        g.build()
    exe = str(tmp_path / "shim_smoke")
    subprocess.check_call(["g++", "-std=c++14", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           "-I" + os.path.join(ROOT, "tandem_amd", "libdr"), os.path.join(ROOT, "tests/cpp/shim_smoke.cpp"),
                           "-o", exe, "-L" + os.path.join(ROOT, "tandem_amd"), "-ldr_mi355x",
                           "-Wl,-rpath," + os.path.join(ROOT, "tandem_amd")])
    return exe


def test_shim_compiles_and_links_with_gcc(tmp_path):
    exe = build(tmp_path)
    assert os.path.isfile(exe)


@pytest.mark.gpu
def test_shim_end_to_end(tmp_path, trained_blob):
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from export_fixture import write_tdms
    exe = build(tmp_path)
    g = np.load(os.path.join(ROOT, "tests/golden/mvsnet_v7_64x96.npz"))
    sample = str(tmp_path / "sample.tdms")
    write_tdms(sample, g["bgrs"], g["K"], g["c2ws"], g["ref_index"], g["depth_min"], g["depth_max"], g["discard"],
               g["ref_s3_depth"], g["ref_s3_confidence"])
    r = subprocess.run([exe, trained_blob, sample], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "All looks good!" in r.stdout and "fusion: rendered" in r.stdout


def build_tracker(tmp_path):
    import __graft_entry__ as g
    if not os.path.isfile(os.path.join(ROOT, "tandem_amd", "libdr_mi355x.so")):
        g.build()
    exe = str(tmp_path / "tracker_smoke")
    subprocess.check_call(["g++", "-std=c++14", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           "-I" + os.path.join(ROOT, "tandem_amd", "libdr"), os.path.join(ROOT, "tests/cpp/tracker_smoke.cpp"),
                           "-o", exe, "-L" + os.path.join(ROOT, "tandem_amd"), "-ldr_mi355x",
                           "-Wl,-rpath," + os.path.join(ROOT, "tandem_amd")])
    return exe


def test_tracker_shim_compiles_and_links_with_gcc(tmp_path):
    assert os.path.isfile(build_tracker(tmp_path))


@pytest.mark.gpu
def test_tracker_shim_end_to_end(tmp_path):
    """class CudaCoarseTracker (cuda_coarse_tracker.h:9-35) through the shim: closed-form residual on a ramp image."""
    r = subprocess.run([build_tracker(tmp_path)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "tracker: E=" in r.stdout


REF_TEST = os.path.join(ROOT, "oracle", "_ref", "dr_mvsnet_test")
REF_SRC = "/root/reference/tandem/libdr/dr_mvsnet/src/dr_mvsnet_test.cpp"


@pytest.mark.skipif(not os.path.isfile(REF_SRC), reason="reference checkout not present")
def test_reference_test_program_compiles_unchanged_against_the_shim(tmp_path):
    """ref:tandem/libdr/dr_mvsnet/src/dr_mvsnet_test.cpp, as it is, with tandem_amd/libdr/dr_mvsnet.h in place of the
    reference header (oracle/Makefile.ref builds the copy that travels to the GPU box)."""
    exe = str(tmp_path / "dr_mvsnet_test")
    subprocess.check_call(["g++", "-std=c++14", "-Wall", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "tandem_amd", "libdr"),
                           REF_SRC, "-o", exe, "-L" + os.path.join(ROOT, "tandem_amd"), "-ldr_mi355x",
                           "-Wl,-rpath," + os.path.join(ROOT, "tandem_amd")])
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode != 0 and "usage: ./dr_mvsnet_test" in r.stderr


@pytest.mark.gpu
def test_reference_test_program_runs(tmp_path, trained_blob):
    """The reference's own dr_mvsnet_test (built here from the reference source, unchanged) on the GPU: model load,
    5 warm-up + 3 timed CallAsync / Ready / GetResult rounds against a stored window at the shape TANDEM ships
    (320 x 512, 7 views, planes (48,4,4)), the reference's < 1e-2 criterion, and the out_folder dump."""
    import sys
    if not os.path.isfile(REF_TEST):
        pytest.skip("oracle/_ref/dr_mvsnet_test not built (needs the reference checkout at build time)")
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from export_fixture import write_tdms
    from tandem_amd import weights as Wt
    g = np.load(os.path.join(ROOT, "tests/golden/mvsnet_v7_320x512_shipped.npz"))
    _, tens = Wt.read_blob(trained_blob)
    blob = str(tmp_path / "shipped.tdmw")
    Wt.write_blob(blob, tens, depth_num=tuple(int(v) for v in g["planes"]))
    sample = str(tmp_path / "sample.tdms")
    write_tdms(sample, g["bgrs"], g["K"], g["c2ws"], g["ref_index"], g["depth_min"], g["depth_max"], g["discard"],
               g["ref_s3_depth"], g["ref_s3_confidence"])
    out_dir = str(tmp_path) + "/"
    r = subprocess.run([REF_TEST, blob, sample, "3", out_dir], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "All looks good!" in r.stdout and "Loading Model" in r.stdout and "GetResult" in r.stdout
    pred = np.load(out_dir + "pred_outputs.npy")
    assert pred.shape == g["ref_s3_depth"].shape
    same = (pred == 0) == (g["ref_s3_depth"] == 0)
    assert same.mean() > 0.998 and np.abs(pred - g["ref_s3_depth"])[same].mean() < 1e-4


BACKEND_RUN = os.path.join(ROOT, "oracle", "_ref", "tandem_backend_run")
BACKEND_SRC = "/root/reference/tandem/src/tandem/tandem_backend.cpp"


@pytest.mark.skipif(not os.path.isfile(BACKEND_SRC), reason="reference checkout not present")
def test_reference_tandem_backend_compiles_unchanged_against_the_shims(tmp_path):
    """ref:tandem/src/tandem/tandem_backend.cpp -- TANDEM's own caller of DrMvsnet AND DrFusion -- compiled as it is against
    tandem_amd/libdr/{dr_mvsnet,dr_fusion}.h; cv::Mat / boost::thread / Output3DWrapper come from oracle/ref_stub_backend/
    (container, <thread>, three virtuals), util/Timer.h is the reference's own.  The product library resolves every symbol."""
    exe = str(tmp_path / "tandem_backend_run")
    subprocess.check_call(["g++", "-std=c++14", "-w", "-I" + os.path.join(ROOT, "oracle", "ref_stub_backend"), "-I/root/reference/tandem/src",
                           "-I/root/reference/tandem/src/tandem", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "tandem_amd", "libdr"),
                           os.path.join(ROOT, "tools", "tandem_backend_main.cpp"), BACKEND_SRC, "-o", exe, "-L" + os.path.join(ROOT, "tandem_amd"),
                           "-ldr_mi355x", "-lpthread", "-Wl,-rpath," + os.path.join(ROOT, "tandem_amd")])
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 2 and "usage:" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("mesh_freq", [0, 3])
def test_reference_tandem_backend_drives_both_operators(tmp_path, trained_blob, mesh_freq):
    """The reference's TandemBackend (unchanged) as the integration driver: 8 keyframes of a 96 x 128 x 5-view window through
    CallAsync / GetResult / IntegrateScanAsync / RenderAsync / GetRenderResult (+ ExtractMeshAsync / GetMeshSync every 3rd call) in the
    reference's own order; the output wrapper receives a depth map per keyframe, the tracker's depth map becomes valid and is
    mostly covered, the mesh is non-empty."""
    import json
    import sys
    if not os.path.isfile(BACKEND_RUN):
        pytest.skip("oracle/_ref/tandem_backend_run not built (needs the reference checkout at build time)")
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from export_fixture import write_tdms
    from synth import scene
    H, W, V = 96, 128, 5
    win = scene.make_window(H, W, V, seed=3)
    sample = str(tmp_path / "w.tdms")
    z = np.zeros((H, W), np.float32)
    write_tdms(sample, np.stack(win["bgrs"]), win["K"], win["c2ws"], win["ref_index"], win["depth_min"], win["depth_max"], 10.0, z, z)
    r = subprocess.run([BACKEND_RUN, trained_blob, sample, "8", "0.02", str(mesh_freq), "1"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["driver"].startswith("reference tandem_backend.cpp") and d["keyframes"] == 8
    # 3 warm-up + 8 timed calls: every call after the first pushes the previous keyframe's image and depth map
    assert d["pushed"]["depth_maps"] == 10 and d["pushed"]["images"] == 10
    assert d["tracking_maps_valid"] >= 6 and d["tracked_sample"] > 0.5 * (H * W / 97)
    assert d["last_depth_sample_sum"] > 0
    if mesh_freq:
        assert d["pushed"]["meshes"] >= 2 and d["pushed"]["last_mesh_vertices"] > 1000
    else:
        assert d["pushed"]["meshes"] == 0
