"""-m gpu: the MFMA tap-table convolution kernel (tandem_amd/csrc/conv_mfma.h) through the C ABI
(drm_debug_conv) against torch fp32 CPU convolutions of the same op -- every layer shape class the
depth pipeline uses (FeatureNet module.py:461-494, CostRegNet module.py:546-575)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

CASES = [
    # name, (D,H,W), Cin, Cout, (kd,kh,kw), stride, transposed, relu, add ("none"|"same"|"up2")
    ("fn.conv0.0 rgb0->8 xpair", (2, 32, 64), 4, 8, (1, 3, 3), (1, 1, 1), False, True, "none"),
    ("fn.conv0.1 8->8 xpair", (2, 32, 64), 8, 8, (1, 3, 3), (1, 1, 1), False, True, "none"),
    ("fn.conv1.0 5x5s2 8->16", (2, 32, 64), 8, 16, (1, 5, 5), (1, 2, 2), False, True, "none"),
    ("fn.conv2.0 5x5s2 16->32", (2, 32, 32), 16, 32, (1, 5, 5), (1, 2, 2), False, True, "none"),
    ("fn.conv2.1 32->32", (2, 16, 24), 32, 32, (1, 3, 3), (1, 1, 1), False, True, "none"),
    ("fn.out1 1x1 32->32", (3, 8, 24), 32, 32, (1, 1, 1), (1, 1, 1), False, False, "none"),
    ("fn.skip2 1x1 16->32 +up2", (2, 16, 48), 16, 32, (1, 1, 1), (1, 1, 1), False, False, "up2"),
    ("fn.skip3 1x1 8->32 +up2", (2, 16, 48), 8, 32, (1, 1, 1), (1, 1, 1), False, False, "up2"),
    ("fn.out2 32->16", (2, 16, 24), 32, 16, (1, 3, 3), (1, 1, 1), False, False, "none"),
    ("fn.out3 32->8 xpair", (2, 16, 32), 32, 8, (1, 3, 3), (1, 1, 1), False, False, "none"),
    ("cr.conv0 32->8 xpair", (12, 16, 24), 32, 8, (3, 3, 3), (1, 1, 1), False, True, "none"),
    ("cr.conv0 16->8 xpair", (8, 12, 24), 16, 8, (3, 3, 3), (1, 1, 1), False, True, "none"),
    ("cr.conv1 8->16 s2", (12, 16, 24), 8, 16, (3, 3, 3), (2, 2, 2), False, True, "none"),
    ("cr.conv2 16->16", (6, 8, 12), 16, 16, (3, 3, 3), (1, 1, 1), False, True, "none"),
    ("cr.conv3 16->32 s2", (6, 8, 12), 16, 32, (3, 3, 3), (2, 2, 2), False, True, "none"),
    ("cr.conv5 32->64 s2 odd", (6, 15, 20), 32, 64, (3, 3, 3), (2, 2, 2), False, True, "none"),
    ("cr.conv5 32->64 s(1,2,2)", (1, 8, 12), 32, 64, (3, 3, 3), (1, 2, 2), False, True, "none"),
    ("cr.conv6 64->64", (3, 4, 6), 64, 64, (3, 3, 3), (1, 1, 1), False, True, "none"),
    ("cr.conv7 deconv 64->32 +skip", (3, 4, 6), 64, 32, (3, 3, 3), (2, 2, 2), True, True, "same"),
    ("cr.conv7 deconv s(1,2,2)", (1, 4, 6), 64, 32, (3, 3, 3), (1, 2, 2), True, True, "same"),
    ("cr.conv9 deconv 32->16 +skip", (3, 5, 10), 32, 16, (3, 3, 3), (2, 2, 2), True, True, "same"),
    ("cr.conv11 deconv 16->8 +skip", (6, 8, 12), 16, 8, (3, 3, 3), (2, 2, 2), True, True, "same"),
    ("cr.prob 8->1 x8", (12, 16, 24), 8, 1, (3, 3, 3), (1, 1, 1), False, False, "none"),
    ("cr.prob 8->1 x8 D=4", (4, 16, 16), 8, 1, (3, 3, 3), (1, 1, 1), False, False, "none"),
]


def torch_ref(x, w, stride, transposed, scale, bias, relu, add, add_mode):
    xt = torch.from_numpy(x).permute(3, 0, 1, 2)[None]  # (1,C,D,H,W)
    wt = torch.from_numpy(w)
    pad = tuple(k // 2 for k in w.shape[2:])
    if transposed:
        y = F.conv_transpose3d(xt, wt, None, stride, pad, tuple(s - 1 for s in stride))
    else:
        y = F.conv3d(xt, wt, None, stride, pad)
    y = y * torch.from_numpy(scale).view(1, -1, 1, 1, 1) + torch.from_numpy(bias).view(1, -1, 1, 1, 1)
    if relu:
        y = F.relu(y)
    if add_mode == "same":
        y = y + torch.from_numpy(add).permute(3, 0, 1, 2)[None]
    elif add_mode == "up2":
        a = torch.from_numpy(add).permute(0, 3, 1, 2)  # (D,C,h,w) nearest x2 in (h,w)
        y = y + F.interpolate(a, scale_factor=2, mode="nearest").permute(1, 0, 2, 3)[None]
    return y[0].permute(1, 2, 3, 0).contiguous().numpy()


_REF_CACHE = {}


def run_case(case, rel_tol=2e-5):
    from tandem_amd.dr_mvsnet import debug_conv
    name, dims, cin, cout, k, stride, transposed, relu, add_mode = case
    if name in _REF_CACHE:  # plan sweeps re-run one case many times: inputs and the torch reference are computed once
        x, w, scale, bias, add, ref = _REF_CACHE[name]
        got = debug_conv(x, w, stride, transposed, scale, bias, relu, add, add_mode == "up2")
        err = np.abs(got - ref).max()
        tol = rel_tol * max(1.0, np.abs(ref).max())
        assert got.shape == ref.shape and err <= tol, f"{name}: max|err| {err:.3e} > {tol:.3e}"
        return
    rng = np.random.RandomState(abs(hash(name)) % (2 ** 31))
    x = rng.randn(*dims, cin).astype(np.float32)
    wshape = (cin, cout) + k if transposed else (cout, cin) + k
    w = (rng.randn(*wshape) / np.sqrt(cin * np.prod(k))).astype(np.float32)
    scale = (1.0 + 0.3 * rng.randn(cout)).astype(np.float32)
    bias = (0.2 * rng.randn(cout)).astype(np.float32)
    if transposed:
        od = tuple(d * s for d, s in zip(dims, stride))
    else:
        od = tuple((d + 2 * (kk // 2) - kk) // s + 1 for d, kk, s in zip(dims, k, stride))
    add = None
    if add_mode == "same":
        add = rng.randn(*od, cout).astype(np.float32)
    elif add_mode == "up2":
        add = rng.randn(od[0], od[1] // 2, od[2] // 2, cout).astype(np.float32)
    got = debug_conv(x, w, stride, transposed, scale, bias, relu, add, add_mode == "up2")
    ref = torch_ref(x, w, stride, transposed, scale, bias, relu, add, add_mode)
    _REF_CACHE[name] = (x, w, scale, bias, add, ref)
    assert got.shape == ref.shape
    err = np.abs(got - ref).max()
    tol = rel_tol * max(1.0, np.abs(ref).max())  # default: fp32 reassociation only (MFMA fp32 == fmaf chain)
    assert err <= tol, f"{name}: max|err| {err:.3e} > {tol:.3e}"
    return err / max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_conv_matches_torch(case):
    run_case(case)


SWEEP = [
    ("sweep 1x1 8->32 +up2, 7 views", (7, 16, 64), 8, 32, (1, 1, 1), (1, 1, 1), False, False, "up2"),
    ("sweep 3x3 32->16", (3, 16, 32), 32, 16, (1, 3, 3), (1, 1, 1), False, False, "none"),
    ("sweep 3x3x3 16->16", (10, 8, 32), 16, 16, (3, 3, 3), (1, 1, 1), False, True, "none"),
    ("sweep 3x3x3 s2 8->16", (12, 16, 32), 8, 16, (3, 3, 3), (2, 2, 2), False, True, "none"),
    ("sweep deconv 32->16 +skip", (5, 6, 16), 32, 16, (3, 3, 3), (2, 2, 2), True, True, "same"),
]


@pytest.mark.parametrize("case", SWEEP, ids=[c[0] for c in SWEEP])
def test_every_plan_candidate_is_correct(case, monkeypatch):
    """The planner ranks ~100 (channel pass, row tiles, position tiles, tile shape) candidates per layer and the
    autotuner / conv_tuned.h may pick any of them: each must compute the same convolution (this sweep is what caught
    the divisor-1 case of the halo decode: tiles with one input row but several planes)."""
    for rank in range(0, 400, 1):
        monkeypatch.setenv("DR_CONV_RANK", str(rank))
        run_case(case)
        if rank > 160:  # beyond the candidate count the last one repeats
            break


# ---- k_conv_m: the marching producer/consumer kernel (conv_march.h).  DR_CONV_MARCH=2 puts its candidates first in the
# planner's ranking; the shapes are large enough for several steps per workgroup (ring wrap-around, column changes in
# the middle of a range, both channel-pass structures).
MARCH = [
    ("march xpair 3x3x3 16->8", (20, 96, 160), 16, 8, (3, 3, 3), (1, 1, 1), False, True, "none"),
    ("march xpair 3x3x3 32->8 (two outer passes)", (12, 64, 128), 32, 8, (3, 3, 3), (1, 1, 1), False, True, "none"),
    ("march xpair 3x3x3 8->8", (8, 128, 256), 8, 8, (3, 3, 3), (1, 1, 1), False, True, "none"),
    ("march 3x3x3 16->16", (16, 64, 96), 16, 16, (3, 3, 3), (1, 1, 1), False, True, "none"),
    ("march xpair 3x3 8->8, 7 views", (7, 96, 192), 8, 8, (1, 3, 3), (1, 1, 1), False, True, "none"),
    ("march 3x3 16->16 +skip, 7 views", (7, 80, 112), 16, 16, (1, 3, 3), (1, 1, 1), False, True, "same"),
    ("march 3x3 32->32 (inner passes), 7 views", (7, 48, 80), 32, 32, (1, 3, 3), (1, 1, 1), False, True, "none"),
    ("march 3x3 32->16, 7 views", (7, 64, 96), 32, 16, (1, 3, 3), (1, 1, 1), False, False, "none"),
    ("march odd sizes 3x3x3 16->8", (5, 35, 70), 16, 8, (3, 3, 3), (1, 1, 1), False, True, "none"),
]


@pytest.mark.parametrize("case", MARCH, ids=[c[0] for c in MARCH])
def test_marching_kernel_candidates(case, monkeypatch):
    monkeypatch.setenv("DR_CONV_MARCH", "2")
    for rank in range(8):  # the marching candidates (tile shapes, PT, CT) lead the ranking; later ranks repeat other families
        monkeypatch.setenv("DR_CONV_RANK", str(rank))
        run_case(case)


# ---- the row march: 2-D 3x3 layers on k_conv_m, marching down the rows of each image (DR_CONV_ROWMARCH=2 ranks its candidates
# first).  Widths as in the pipeline (640 / 320 / 160 columns = 20 / 20 / 10 position tiles, 512-wide = 16) and ragged ones.
ROWMARCH = [
    ("rows xpair 3x3 8->8, 7 views", (7, 40, 640), 8, 8, (1, 3, 3), (1, 1, 1), False, True, "none"),
    ("rows 3x3 16->16, 3 views", (3, 50, 320), 16, 16, (1, 3, 3), (1, 1, 1), False, True, "none"),
    ("rows 3x3 32->32, 7 views", (7, 30, 160), 32, 32, (1, 3, 3), (1, 1, 1), False, True, "none"),
    ("rows 3x3 32->16", (2, 37, 320), 32, 16, (1, 3, 3), (1, 1, 1), False, False, "none"),
    ("rows xpair 3x3 32->8", (2, 33, 640), 32, 8, (1, 3, 3), (1, 1, 1), False, False, "none"),
    ("rows ragged 3x3 16->16", (3, 21, 300), 16, 16, (1, 3, 3), (1, 1, 1), False, True, "none"),
    ("rows 3x3 16->16, 256 wide", (2, 20, 256), 16, 16, (1, 3, 3), (1, 1, 1), False, True, "none"),
    ("rows one-row image 3x3 16->16", (4, 1, 320), 16, 16, (1, 3, 3), (1, 1, 1), False, True, "none"),
]


@pytest.mark.parametrize("case", ROWMARCH, ids=[c[0] for c in ROWMARCH])
def test_row_march_candidates(case, monkeypatch, capfd):
    monkeypatch.setenv("DR_CONV_ROWMARCH", "2")
    monkeypatch.setenv("DR_CONV_PRINT", "1")
    kinds = []
    for rank in range(10):
        monkeypatch.setenv("DR_CONV_RANK", str(rank))
        run_case(case)
        kinds += [l.split()[1].split("<")[0] for l in capfd.readouterr().err.splitlines() if l.startswith("debug_conv:")]
    assert kinds and kinds[0] == "rowmarch" and kinds.count("rowmarch") >= 2, kinds


# ---- k_conv_w (conv_wino.h): the y axis of the stride-1 3-tap layers in Winograd F(2,3) form.  DR_CONV_WINO=2 ranks its candidates first;
# the reference and the bound are the direct kernels' (torch fp32, 2e-5 of the value range): the transform may not cost accuracy.
WINO = [
    ("wino xpair 3x3 8->8, 7 views", (7, 64, 192), 8, 8, (1, 3, 3), (1, 1, 1), False, True, "none"),
    ("wino 3x3 16->16 +skip, 7 views", (7, 48, 112), 16, 16, (1, 3, 3), (1, 1, 1), False, True, "same"),
    ("wino 3x3 32->32 (two passes)", (3, 40, 80), 32, 32, (1, 3, 3), (1, 1, 1), False, True, "none"),
    ("wino 3x3 32->16 +up2", (2, 32, 96), 32, 16, (1, 3, 3), (1, 1, 1), False, False, "up2"),
    ("wino xpair 3x3 32->8", (2, 34, 128), 32, 8, (1, 3, 3), (1, 1, 1), False, False, "none"),
    ("wino 3x3x3 16->16 +skip", (10, 24, 48), 16, 16, (3, 3, 3), (1, 1, 1), False, True, "same"),
    ("wino xpair 3x3x3 16->8", (8, 36, 70), 16, 8, (3, 3, 3), (1, 1, 1), False, True, "none"),
    ("wino xpair 3x3x3 32->8", (6, 20, 64), 32, 8, (3, 3, 3), (1, 1, 1), False, True, "none"),
    ("wino xpair 3x3x3 8->8", (4, 30, 96), 8, 8, (3, 3, 3), (1, 1, 1), False, True, "none"),
    ("wino 3x3x3 32->32", (4, 12, 40), 32, 32, (3, 3, 3), (1, 1, 1), False, True, "none"),
    ("wino 3x3x3 64->64", (3, 8, 20), 64, 64, (3, 3, 3), (1, 1, 1), False, True, "none"),
    ("wino ragged 3x3 16->16", (3, 22, 300), 16, 16, (1, 3, 3), (1, 1, 1), False, True, "none"),
    ("wino two-row image 3x3 16->16", (4, 2, 64), 16, 16, (1, 3, 3), (1, 1, 1), False, True, "none"),
]


@pytest.mark.parametrize("case", WINO, ids=[c[0] for c in WINO])
def test_winograd_kernel_candidates(case, monkeypatch, capfd):
    monkeypatch.setenv("DR_CONV_WINO", "2")
    monkeypatch.setenv("DR_CONV_PRINT", "1")
    kinds = []
    for rank in range(0, 36):
        monkeypatch.setenv("DR_CONV_RANK", str(rank))
        run_case(case)
        kinds += [l.split()[1].split("<")[0] for l in capfd.readouterr().err.splitlines() if l.startswith("debug_conv:")]
    assert kinds and kinds[0] in ("wino", "winomarch") and kinds.count("wino") >= 4, kinds  # (3-D layers: the marching form's candidates lead)


# ---- the same form on the marching kernel (conv_march.h march_consumer_w: raw kernel rows in LDS, u1 / u2 derived per lane): 3-D layers,
# shapes with several steps per workgroup, two outer channel passes (raw partial sums), odd depth, ragged width
WINO_MARCH = [
    ("winomarch xpair 3x3x3 16->8", (20, 96, 160), 16, 8, (3, 3, 3), (1, 1, 1), False, True, "none"),
    ("winomarch xpair 3x3x3 32->8 (two outer passes)", (12, 64, 128), 32, 8, (3, 3, 3), (1, 1, 1), False, True, "none"),
    ("winomarch xpair 3x3x3 8->8", (8, 128, 256), 8, 8, (3, 3, 3), (1, 1, 1), False, True, "none"),
    ("winomarch 3x3x3 16->16 +skip", (16, 64, 96), 16, 16, (3, 3, 3), (1, 1, 1), False, True, "same"),
    ("winomarch ragged 3x3x3 16->8", (5, 36, 70), 16, 8, (3, 3, 3), (1, 1, 1), False, True, "none"),
    ("winomarch two planes 3x3x3 8->8", (2, 60, 96), 8, 8, (3, 3, 3), (1, 1, 1), False, True, "none"),
]


@pytest.mark.parametrize("case", WINO_MARCH, ids=[c[0] for c in WINO_MARCH])
def test_winograd_marching_candidates(case, monkeypatch, capfd):
    monkeypatch.setenv("DR_CONV_WINO", "2")
    monkeypatch.setenv("DR_CONV_PRINT", "1")
    kinds = []
    for rank in range(6):
        monkeypatch.setenv("DR_CONV_RANK", str(rank))
        run_case(case)
        kinds += [l.split()[1].split("<")[0] for l in capfd.readouterr().err.splitlines() if l.startswith("debug_conv:")]
    assert kinds and kinds[0] == "winomarch", kinds


def test_winograd_form_is_not_planned_where_it_does_not_apply(monkeypatch, capfd):
    """Odd output height, strided, transposed and 5x5 layers stay on the direct kernels even when the Winograd form is preferred."""
    monkeypatch.setenv("DR_CONV_WINO", "2")
    monkeypatch.setenv("DR_CONV_PRINT", "1")
    monkeypatch.setenv("DR_CONV_RANK", "0")
    for name in ("cr.conv5 32->64 s2 odd", "fn.conv1.0 5x5s2 8->16", "cr.conv9 deconv 32->16 +skip", "cr.prob 8->1 x8"):
        run_case([c for c in CASES if c[0] == name][0])
    run_case(("odd rows 3x3 16->16", (2, 15, 48), 16, 16, (1, 3, 3), (1, 1, 1), False, True, "none"))
    kinds = [l.split()[1].split("<")[0] for l in capfd.readouterr().err.splitlines() if l.startswith("debug_conv:")]
    assert len(kinds) == 5 and "wino" not in kinds, kinds


# ---- transposed stride-2 layers: the three parity forms (conv_mfma.h axis_classes: dense rows / x dense + (z, y) classes / one class per
# parity) compute the same layer; every plan candidate of each form.
DECONV = [c for c in CASES if c[6]] + [
    ("deconv 16->8 +skip, wide", (5, 20, 48), 16, 8, (3, 3, 3), (2, 2, 2), True, True, "same"),
    ("deconv 32->16 no skip", (3, 9, 20), 32, 16, (3, 3, 3), (2, 2, 2), True, False, "none"),
]


@pytest.mark.parametrize("form", [0, 1, 2])
@pytest.mark.parametrize("case", DECONV, ids=[c[0] for c in DECONV])
def test_transposed_parity_forms(case, form, monkeypatch, parity_hooks):
    monkeypatch.setenv("DR_DECONV_FORM", str(form))
    for rank in range(0, 60, 3):
        monkeypatch.setenv("DR_CONV_RANK", str(rank))
        run_case(case)


# ---- k_conv's class loop (single-pass transposed layers: one workgroup walks the parity classes on one staged tile, the next class's weights in flight)
# against the class-per-workgroup launch of the same plan: same bits, every plan candidate.
CLASS_LOOP = [
    ("class loop 16->8 +skip (conv11's shape)", (6, 20, 48), 16, 8, (3, 3, 3), (2, 2, 2), True, True, "same"),
    ("class loop 16->8 ragged", (3, 7, 21), 16, 8, (3, 3, 3), (2, 2, 2), True, True, "same"),
    ("class loop 8->8 no skip", (4, 9, 33), 8, 8, (3, 3, 3), (2, 2, 2), True, False, "none"),
    ("class loop 16->16 one plane", (1, 12, 40), 16, 16, (3, 3, 3), (2, 2, 2), True, True, "same"),
]


@pytest.mark.parametrize("case", CLASS_LOOP, ids=[c[0] for c in CLASS_LOOP])
def test_class_loop_is_bit_identical(case, monkeypatch, parity_hooks, capfd):
    from tandem_amd.dr_mvsnet import debug_conv
    name, dims, cin, cout, k, stride, transposed, relu, add_mode = case
    rng = np.random.RandomState(abs(hash(name)) % (2 ** 31))
    x = rng.randn(*dims, cin).astype(np.float32)
    w = (rng.randn(cin, cout, *k) / np.sqrt(cin * np.prod(k))).astype(np.float32)
    scale = (1.0 + 0.3 * rng.randn(cout)).astype(np.float32)
    bias = (0.2 * rng.randn(cout)).astype(np.float32)
    add = rng.randn(*(d * 2 for d in dims), cout).astype(np.float32) if add_mode == "same" else None
    ref = torch_ref(x, w, stride, transposed, scale, bias, relu, add, add_mode)
    looped = 0
    for rank in range(0, 40, 2):
        monkeypatch.setenv("DR_CONV_RANK", str(rank))
        monkeypatch.setenv("DR_CONV_PRINT", "1")
        monkeypatch.delenv("DR_CONV_NO_CLASS_LOOP", raising=False)
        a = debug_conv(x, w, stride, transposed, scale, bias, relu, add, False)
        looped += "class loop" in capfd.readouterr().err
        monkeypatch.setenv("DR_CONV_NO_CLASS_LOOP", "1")
        b = debug_conv(x, w, stride, transposed, scale, bias, relu, add, False)
        assert "class loop" not in capfd.readouterr().err
        assert np.array_equal(a, b), f"{name}, candidate {rank}: class loop differs from class per workgroup (max {np.abs(a - b).max():.3e})"
        assert np.abs(a - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max())
    assert looped > 0, "no candidate took the class loop"


# ---- ConvLayer::up2: a 3x3 layer over the nearest x2 upsampling of its (half-resolution) input, as 2 x 2-tap phase convolutions
# with summed kernel entries (conv_mfma.h axis_classes_up2) -- the second half of the folded out.stage3 (FeatureNet).
UP2 = [
    ("up2 32->8", (2, 24, 40), 32, 8),
    ("up2 32->8, 7 views", (7, 30, 80), 32, 8),
    ("up2 16->16 ragged", (3, 11, 23), 16, 16),
    ("up2 one row", (2, 1, 64), 32, 8),
]


@pytest.mark.parametrize("case", UP2, ids=[c[0] for c in UP2])
def test_conv_over_upsampled_input(case, monkeypatch, parity_hooks):
    from tandem_amd.dr_mvsnet import debug_conv
    name, dims, cin, cout = case
    rng = np.random.RandomState(abs(hash(name)) % (2 ** 31))
    x = rng.randn(*dims, cin).astype(np.float32)
    w = (rng.randn(cout, cin, 1, 3, 3) / np.sqrt(cin * 9)).astype(np.float32)
    bias = (0.2 * rng.randn(cout)).astype(np.float32)
    add = rng.randn(dims[0], 2 * dims[1], 2 * dims[2], cout).astype(np.float32)
    xt = torch.from_numpy(x).permute(0, 3, 1, 2)  # (D, C, h, w)
    up = F.interpolate(xt, scale_factor=2, mode="nearest")
    ref = F.conv2d(up, torch.from_numpy(w[:, :, 0]), torch.from_numpy(bias), 1, 1).permute(0, 2, 3, 1).numpy() + add
    for rank in range(0, 40, 3):
        monkeypatch.setenv("DR_CONV_RANK", str(rank))
        got = debug_conv(x, w, (1, 1, 1), "up2", None, bias, False, add, False)
        err = np.abs(got - ref).max()
        assert got.shape == ref.shape and err <= 2e-5 * max(1.0, np.abs(ref).max()), f"{name} rank {rank}: max|err| {err:.3e}"


# ---- k_conv_b (conv_bf3.h), the opt-in bf16 x 3 precision mode (DR_CONV_BF16X3=1).  First run on an MI355X in round 4
# (profiles/r04_first_ab.txt): all cases green, so they are part of the suite.  The mode is never the headline (operands carry 16
# mantissa bits, not 24): bench.py reports it as its own object.
@pytest.mark.parametrize("case", [c for c in CASES if c[2] % 8 == 0], ids=[c[0] for c in CASES if c[2] % 8 == 0])
def test_bf16x3_conv_matches_torch(case, monkeypatch, capfd, parity_hooks):
    """Same layers, same torch fp32 reference; the bound is the three-term split's (tools/study_split_bf16.py: ~2^-16 per product,
    1e-5 of the value range measured on the emulation), and the error must be ABOVE fp32 reassociation -- or the fp32 kernel ran."""
    monkeypatch.setenv("DR_CONV_BF16X3", "1")
    monkeypatch.setenv("DR_CONV_PRINT", "1")
    _REF_CACHE.pop(case[0], None)
    rel = run_case(case, rel_tol=1e-4)
    assert "bf16x3" in capfd.readouterr().err
    assert rel is None or rel > 2e-7


@pytest.mark.parametrize("case", SWEEP, ids=[c[0] for c in SWEEP])
def test_bf16x3_every_plan_candidate(case, monkeypatch, parity_hooks):
    monkeypatch.setenv("DR_CONV_BF16X3", "1")
    _REF_CACHE.pop(case[0], None)
    for rank in range(0, 120, 1):
        monkeypatch.setenv("DR_CONV_RANK", str(rank))
        run_case(case, rel_tol=1e-4)
