"""CPU: the marching convolution kernel's planner, index arithmetic and ring protocol (tandem_amd/csrc/conv_march.h,
march_plan.h) executed on the host by tests/cpp/march_emul.hip -- every k_conv_m plan candidate (3-D march, 2-D tiles, row march) of thirteen layer shapes
(CostRegNet conv0 / conv2 module.py:546-552, FeatureNet's 3x3 layers module.py:461-494) against a direct convolution.
No device code runs here; the GPU side is tests/test_conv_gpu.py."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.fixture(scope="module")
def march_emul(tmp_path_factory):
    if not (os.path.exists(HIPCC) or shutil.which("hipcc")):
        pytest.skip("needs hipcc to compile the host emulation")
    exe = tmp_path_factory.mktemp("march_emul") / "march_emul"
    subprocess.check_call([HIPCC if os.path.exists(HIPCC) else "hipcc", "--offload-arch=gfx950", "-O2", "-std=c++17", "-Wno-unused-function",
                           "-Wno-pass-failed", "-Wno-unused-result", os.path.join(ROOT, "tests", "cpp", "march_emul.hip"), "-o", str(exe)])
    return str(exe)


def test_march_emulation_matches_direct_convolution(march_emul):
    exe = march_emul
    out = subprocess.run([str(exe), "80"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-4000:] + out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if " plan " in l]
    assert len(lines) >= 30 and all(" ok " in l for l in lines), out.stdout[-4000:]
    # every instance family and both pass structures were exercised
    text = out.stdout
    for needle in ("ci=8 nup=6", "ci=16 nup=12", "ci=16 nup=9", "ct=2", "pt=4", "pt=1", "w=12", "NPI=2", "NPO=2", "rows w=10", "rows w=8", "nup=2", "nup=3", "nup=4"):
        assert needle in text, needle


def test_winograd_march_emulation_matches_direct_convolution(march_emul):
    """march_consumer_w (the y axis of the 3-D layers in Winograd F(2,3) form on the marching kernel): the planner's row-pair geometry, the raw
    kernel rows [chunk of x taps][row k] it packs (g1 halved), u1 / u2 derived per lane with the kernel's own helper, the four rows of a
    pair's window through the swizzled ring image, the output transform, both outer-pass structures -- and the ring protocol unchanged."""
    exe = march_emul
    out = subprocess.run([str(exe), "5", "wino"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-4000:] + out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if " plan " in l]
    w = [l for l in lines if " plan wino " in l]
    assert all(" ok " in l for l in lines) and len(w) >= 6, out.stdout[-4000:]
    for needle in ("xpair3d_c16", "xpair3d_c32", "xpair3d_c8", "conv3d_16_16", "NPO=2", "nup=6", "nup=9", "nup=12"):
        assert any(needle in l for l in w), needle
    assert max(float(l.split("max rel err ")[1].rstrip(")")) for l in w) < 6e-6
