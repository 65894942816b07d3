"""k_tail (tandem_amd/csrc/tail_kernels.h): CostRegNet's conv11 (ConvTranspose3d 16 -> 8 + BN + ReLU, + conv0) and prob (Conv3d 8 -> 1) in ONE launch,
against the same two layers in plain PyTorch fp32 (cva_mvsnet/models/module.py:571-575,598-599).  Every tile shape the planner can pick, depth chunks that
do and do not divide D, planes with and without a second input plane (odd / even, the last odd plane has none), image borders inside a tile.
Bound: fp32 reassociation -- 2e-5 of the logit range, the bound the MFMA convolution kernels are held to (tests/test_conv_gpu.py)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def reference(x, skip, wd, scale, bias, wp):
    import torch
    import torch.nn.functional as F
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))  # noqa: E731
    xi = t(x).permute(3, 0, 1, 2)[None]                       # (1, 16, D/2, h/2, w/2)
    y = F.conv_transpose3d(xi, t(wd), stride=2, padding=1, output_padding=1)
    y = y * t(scale).view(1, 8, 1, 1, 1) + t(bias).view(1, 8, 1, 1, 1)
    y = torch.relu(y) + t(skip).permute(3, 0, 1, 2)[None]
    return F.conv3d(y, t(wp), padding=1)[0, 0].numpy()


@pytest.mark.parametrize("D,h,w,qy,zchunk", [
    (8, 24, 40, 0, 0), (8, 24, 40, 4, 4), (8, 24, 40, 8, 3), (8, 24, 40, 16, 8), (8, 24, 40, 32, 5),
    (4, 12, 28, 0, 0), (4, 12, 28, 8, 4),          # has_four_depths: D = 4
    (16, 60, 124, 0, 0), (16, 60, 124, 4, 6), (16, 60, 124, 16, 16), (6, 120, 60, 32, 0),
    (2, 2, 2, 8, 0),                               # the smallest volume: one input position
])
@pytest.mark.parametrize("form", [1, 0])
def test_tail_matches_torch(D, h, w, qy, zchunk, form, parity_hooks):
    if form == 1 and qy == 32:
        pytest.skip("k_tail_m works on groups of 16 cells along x: no 32 x 8 tile")
    from tandem_amd.dr_mvsnet import debug_tail
    rng = np.random.RandomState(D * 1000 + h * 10 + w + qy)
    x = rng.standard_normal((D // 2, h // 2, w // 2, 16)).astype(np.float32)
    skip = rng.standard_normal((D, h, w, 8)).astype(np.float32)
    wd = (rng.standard_normal((16, 8, 3, 3, 3)) * 0.15).astype(np.float32)
    wp = (rng.standard_normal((1, 8, 3, 3, 3)) * 0.2).astype(np.float32)
    scale = (0.5 + rng.rand(8)).astype(np.float32)
    bias = (rng.standard_normal(8) * 0.3).astype(np.float32)
    ref = reference(x, skip, wd, scale, bias, wp)
    out = debug_tail(x, skip, wd, scale, bias, wp, qy=qy, zchunk=zchunk, form=form)
    err = np.abs(out - ref).max()
    assert err <= 2e-5 * np.abs(ref).max(), f"max err {err} of range {np.abs(ref).max()} (form {form}, qy {qy}, zchunk {zchunk})"


def test_tail_fused_and_two_kernel_paths_agree(trained_blob, monkeypatch, parity_hooks):
    """The engine with k_tail (DR_TAIL_FUSED=1, opt-in) against the default two-kernel path (transposed convolution on the MFMA kernel, then k_prob2):
    every stage's depth map within fp32 reassociation of each other, on a fixture-sized window."""
    from synth import scene
    from tandem_amd.dr_mvsnet import DrMvsnet
    win = scene.make_window(96, 160, 4, seed=11)
    outs = []
    for off in (False, True, False):
        if off:
            monkeypatch.delenv("DR_TAIL_FUSED", raising=False)
        else:
            monkeypatch.setenv("DR_TAIL_FUSED", "1" if not outs else "2")  # first k_tail_m, last k_tail
        m = DrMvsnet(trained_blob)
        m.upload(96, 160, 4, win["ref_index"], win["bgrs"], win["K"], list(win["c2ws"]), win["depth_min"], win["depth_max"], 2.5)
        m.forward(1)
        ops = [r["op"] for r in m.profile()]
        assert any(o.endswith(".tail") for o in ops) != off and any(o.endswith(".conv11") for o in ops) == off, ops
        outs.append([m.stage_output(s) for s in (1, 2, 3)])
        m.close()
    for fused in (0, 2):
        for s in range(3):
            d = np.abs(outs[fused][s][0] - outs[1][s][0])
            assert d.mean() < 2e-5 and d.max() < 2e-3, (fused, s + 1, d.mean(), d.max())
