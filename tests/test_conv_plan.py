"""CPU: the convolution planner (tandem_amd/csrc/conv_mfma.h plan_conv: tap tables, packed weights incl. the XPAIR / X8 shifts, the
three parity forms of the transposed layers, the summed kernel entries of ConvLayer::up2, classes and row groups) through a host
emulation of the generic kernel's data flow (tests/cpp/conv_emul.hip) against direct evaluations of the layers' definitions -- every
layer type of the depth pipeline (FeatureNet module.py:461-531, CostRegNet module.py:546-600), several plan candidates each.  No device
code runs here; the persistent kernels have their own emulation (tests/test_march_plan.py), the GPU side is tests/test_conv_gpu.py."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.fixture(scope="module")
def conv_emul(tmp_path_factory):
    if not (os.path.exists(HIPCC) or shutil.which("hipcc")):
        pytest.skip("needs hipcc to compile the host emulation")
    exe = tmp_path_factory.mktemp("conv_emul") / "conv_emul"
    subprocess.check_call([HIPCC if os.path.exists(HIPCC) else "hipcc", "--offload-arch=gfx950", "-O2", "-std=c++17", "-Wno-unused-function",
                           "-Wno-pass-failed", "-Wno-unused-result", "-DDR_PARITY_HOOKS",  # (the transposed layers' other forms and the bf16 x 3 plans exist in the parity build only)
                           os.path.join(ROOT, "tests", "cpp", "conv_emul.hip"), "-o", str(exe)])
    return str(exe)


@pytest.mark.parametrize("form", ["default", "0", "1", "2"])
def test_generic_kernel_emulation_matches_the_layer_definitions(conv_emul, form):
    env = dict(os.environ)
    env.pop("DR_DECONV_FORM", None)
    env.pop("DR_CONV_BF16X3", None)
    if form != "default":
        env["DR_DECONV_FORM"] = form
    out = subprocess.run([conv_emul, "12"], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stdout[-4000:] + out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if " plan rank " in l]
    assert len(lines) >= 60 and all(": ok " in l for l in lines), out.stdout[-4000:]
    text = out.stdout
    for needle in ("conv2d_5x5_s2_8_16", "xpair2d_4_8", "x8_prob_8_1", "deconv_16_8_skip", "deconv_64_32_s122", "up2_32_8_inplace_add", "classes 1 rows 32"):
        assert needle in text, needle
    if form in ("1", "2"):  # split parities really became classes
        assert ("classes 4" in text) or ("classes 8" in text)
    if form == "0":
        assert "classes 4" not in text and "classes 8" not in text


def test_bf16x3_kernel_emulation_matches_the_layer_definitions(conv_emul):
    """The opt-in precision mode (DR_CONV_BF16X3=1, tandem_amd/csrc/conv_bf3.h): every layer with Cin % 8 == 0 is planned for k_conv_b
    -- 32-wide K chunks, weights packed as hi / lo bf16 fragments -- and emulated with the kernel's byte layout of the staged tile
    ([CI hi | CI lo | pad] records), its operand gathers and three bf16 products per term pair; the error against the layer's
    definition is the three-term split's (~1e-5 of the value range), an order above fp32 reassociation and three below plain bf16.
    The first layer (Cin = 4) stays on the fp32 kernel."""
    env = dict(os.environ)
    env.pop("DR_DECONV_FORM", None)
    env["DR_CONV_BF16X3"] = "1"
    out = subprocess.run([conv_emul, "12"], capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stdout[-4000:] + out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if " plan rank " in l]
    assert len(lines) >= 60 and all(": ok " in l for l in lines), out.stdout[-4000:]
    b = [l for l in lines if " k_conv_b " in l]
    assert len(b) >= 50 and all("xpair2d_4_8" in l for l in lines if " k_conv " in l), out.stdout[-4000:]
    errs = [float(l.split("max rel err ")[1].rstrip(")")) for l in b]
    assert 1e-6 < max(errs) < 1e-4, max(errs)  # really the split arithmetic, and no worse than it should be
    for needle in ("conv2d_5x5_s2_8_16", "deconv_64_32_s122", "up2_32_8_inplace_add", "xpair3d_16_8", "x8_prob_8_1", "ci=8 "):
        assert any(needle in l for l in b), needle


def test_persistent_kernel_emulation_matches_the_layer_definitions(conv_emul):
    """The same layers with k_conv_a's plans ranked first (DR_CONV_ASYNC=1 inside the program): the swizzled LDS image every DMA
    piece produces (conv_a_slot / conv_a_unit), zeros for staged elements outside the tensor, the workgroups' tile lists (XCD
    ranges, round-robin inside: every tile exactly once), 8 waves x PT position tiles; layers no persistent plan fits (stride-2
    tiles, several classes) fall back to k_conv and are emulated as such."""
    out = subprocess.run([conv_emul, "12", "async"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-4000:] + out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if " plan rank " in l]
    assert all(": ok " in l for l in lines), out.stdout[-4000:]
    a = [l for l in lines if " k_conv_a " in l]
    assert len(a) >= 25, len(a)
    for needle in ("xpair2d_4_8", "conv2d_1x1_16_32_up2add", "conv3d_64_64", "up2_32_8_inplace_add"):  # CI = 4, an upsample-add epilogue, 4 passes, the phase layers
        assert any(needle in l for l in a), needle


def test_winograd_kernel_emulation_matches_the_layer_definitions(conv_emul):
    """The same layers with k_conv_w's plans ranked first (DR_CONV_WINO=2 inside the program; csrc/conv_wino.h): the y axis re-described as
    row pairs (stride 2, 4-row window, tap table without y), the weights transformed per Winograd point ([chunk][point][row tile][lane],
    XPAIR shifts included), the four rows every lane reads per chunk, the fp32 input / output transforms and the two output rows per
    position.  The error against the layer's definition must stay at the direct kernel's level (fp32 reassociation), not the 1e-4 of a
    reduced-precision form; layers the form does not apply to (strided, transposed, 5x5, odd heights, X8) fall back to k_conv."""
    out = subprocess.run([conv_emul, "12", "wino"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-4000:] + out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if " plan rank " in l]
    assert all(": ok " in l for l in lines), out.stdout[-4000:]
    w = [l for l in lines if " k_conv_w " in l]
    assert len(w) >= 30, len(w)
    for needle in ("xpair2d_8_8", "conv2d_3x3_32_32_up2add", "conv3d_16_16_skip", "xpair3d_32_8", "conv3d_32_32", "ct=2 ", "pt=2 ", "ci=8 "):
        assert any(needle in l for l in w), needle
    assert not any(n in l for l in w for n in ("conv2d_5x5", "deconv_", "up2_32_8", "x8_prob", "conv3d_s2", "xpair2d_4_8 "))  # (xpair2d_4_8: H = 9 is odd)
    assert max(float(l.split("max rel err ")[1].rstrip(")")) for l in w) < 6e-6


def test_every_tuned_row_names_a_plan_the_planner_can_build(tmp_path):
    """conv_tuned.h rows are matched by layer signature and then by plan parameters; a row whose plan no longer exists (an instance
    removed, a tile rule changed) is silently ignored and the layer falls back to the cost model.  tests/cpp/tuned_rows.hip plans
    every row's layer on the host and checks that the row's plan is the one that comes out (and that duplicated signatures agree)."""
    if not (os.path.exists(HIPCC) or shutil.which("hipcc")):
        pytest.skip("needs hipcc to compile the host check")
    exe = tmp_path / "tuned_rows"
    subprocess.check_call([HIPCC if os.path.exists(HIPCC) else "hipcc", "--offload-arch=gfx950", "-O2", "-std=c++17", "-Wno-unused-function",
                           "-Wno-pass-failed", "-Wno-unused-result", os.path.join(ROOT, "tests", "cpp", "tuned_rows.hip"), "-o", str(exe)])
    env = {k: v for k, v in os.environ.items() if not k.startswith("DR_")}
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0 and " 0 stale" in out.stdout, out.stdout[-3000:] + out.stderr[-1000:]


def test_fused_front_kernel_emulation_matches_the_two_layers(tmp_path):
    """k_fn_front (csrc/fn_front.h: u8 -> float, conv0.0 and conv0.1 of FeatureNet in one launch, module.py:461-470) through a host emulation
    that uses the kernel's own geometry helpers and weight packing (tests/cpp/front_emul.hip): XPAIR packing of both layers, every LDS
    index, tile origins and the masks at ragged image edges, against a direct evaluation of the two layers in double."""
    if not (os.path.exists(HIPCC) or shutil.which("hipcc")):
        pytest.skip("needs hipcc to compile the host emulation")
    exe = tmp_path / "front_emul"
    subprocess.check_call([HIPCC if os.path.exists(HIPCC) else "hipcc", "--offload-arch=gfx950", "-O2", "-std=c++17", "-Wno-unused-function",
                           "-Wno-pass-failed", "-Wno-unused-result", os.path.join(ROOT, "tests", "cpp", "front_emul.hip"), "-o", str(exe)])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("front ")]
    assert len(lines) == 4 and all(l.endswith(" ok") for l in lines), out.stdout


def test_fused_head3_kernel_emulation_matches_the_definition(tmp_path):
    """k_fn_head3 (csrc/fn_head3.h: FeatureNet's folded stage-3 head in one launch, module.py:480-485,524-529) through a host emulation that
    uses the kernel's own geometry helpers and weight packing (tests/cpp/head3_emul.hip): the XPAIR packing of the composed layer, the
    summed kernel rows / columns of the upsampled axes per parity, every LDS index, tile origins, ragged edges, the padded output, against
    conv3x3(conv0) + conv3x3(nearest_up2(inter2)) + the bias shares of the taps inside the image, in double."""
    if not (os.path.exists(HIPCC) or shutil.which("hipcc")):
        pytest.skip("needs hipcc to compile the host emulation")
    exe = tmp_path / "head3_emul"
    subprocess.check_call([HIPCC if os.path.exists(HIPCC) else "hipcc", "--offload-arch=gfx950", "-O2", "-std=c++17", "-Wno-unused-function",
                           "-Wno-pass-failed", "-Wno-unused-result", os.path.join(ROOT, "tests", "cpp", "head3_emul.hip"), "-o", str(exe)])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("head3 ")]
    assert len(lines) == 4 and all(l.endswith(" ok") for l in lines), out.stdout
