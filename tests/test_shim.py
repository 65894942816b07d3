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
