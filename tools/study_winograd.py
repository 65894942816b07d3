"""CPU study: would Winograd F(2, 3) convolutions keep the depth pipeline inside its fp32 parity bounds?

Every stride-1 3x3 (FeatureNet) and 3x3x3 (CostRegNet) convolution of the oracle is replaced by its Winograd form in fp32 --
input tiles of 4 (stride 2) per spatial axis transformed with B^T, weights with G, a channel contraction per transform point,
outputs with A^T -- which needs 2.25x (2-D) / 3.375x (3-D) fewer multiplications than the direct form and is still fp32 arithmetic,
i.e. admissible for the headline path if (and only if) the depth maps stay inside the bounds of tests/test_mvsnet_gpu.py::compare.
Strided, transposed, 1x1 and 5x5 layers stay direct.  Compared with the fp32 oracle on the trained-weight fixtures.

    python tools/study_winograd.py [fixture.npz ...]
Test infrastructure only: nothing here is on the product path.
"""
import glob
import os
import sys

import numpy as np
import torch
import torch.nn.functional as TF

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from oracle import mvsnet_oracle as O  # noqa: E402
from tandem_amd import weights as Wt  # noqa: E402
from study_split_bf16 import report  # noqa: E402

BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float32)
G = torch.tensor([[1, 0, 0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0, 0, 1]], dtype=torch.float32)
AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float32)


def along(t, m, dim):  # apply matrix m (r x c) to axis `dim` (length c) of t
    return torch.movedim(torch.tensordot(t, m, dims=([dim], [1])), -1, dim)


def winograd(x, w, nd):
    """x (1, C, *S), w (Co, Ci, 3[,3[,3]]) over the last `nd` axes, stride 1, padding 1, fp32."""
    S = x.shape[-nd:]
    T = [(n + 1) // 2 for n in S]
    pad = []
    for n, t in zip(reversed(S), reversed(T)):
        pad += [1, 1 + 2 * t - n]
    xp = TF.pad(x[0], pad)  # (C, *S + halo)
    for k in range(nd):  # tiles of 4, stride 2, per axis: (C, T0.., 4..)
        xp = xp.unfold(1 + k, 4, 2)
    U = w
    for k in range(nd):
        xp = along(xp, BT, 1 + nd + k)
        U = along(U, G, 2 + k)
    # contraction over input channels per transform point
    letters = "abc"[:nd]
    tl = "xyz"[:nd]
    M = torch.einsum("i%s%s,oi%s->o%s%s" % (tl, letters, letters, tl, letters), xp, U)
    for k in range(nd):
        M = along(M, AT, 1 + nd + k)
    # (Co, T.., 2..) -> (Co, 2T..)
    perm = [0]
    for k in range(nd):
        perm += [1 + k, 1 + nd + k]
    y = M.permute(perm).reshape([M.shape[0]] + [2 * t for t in T])
    idx = (slice(None),) + tuple(slice(0, n) for n in S)
    return y[idx][None]


class WinoF:
    """Stand-in for torch.nn.functional inside the oracle module."""

    def __init__(self):
        self.n = 0

    def __getattr__(self, name):
        return getattr(TF, name)

    def conv2d(self, x, w, bias=None, stride=1, padding=0):
        if tuple(w.shape[-2:]) == (3, 3) and stride in (1, (1, 1)) and padding in (1, (1, 1)):
            self.n += 1
            y = torch.cat([winograd(x[i:i + 1], w, 2) for i in range(x.shape[0])])
            return y if bias is None else y + bias.view(1, -1, 1, 1)
        return TF.conv2d(x, w, bias, stride, padding)

    def conv3d(self, x, w, bias=None, stride=1, padding=0):
        if tuple(w.shape[-3:]) == (3, 3, 3) and stride in (1, (1, 1, 1)) and padding in (1, (1, 1, 1)) and x.shape[0] == 1:
            self.n += 1
            y = winograd(x, w, 3)
            return y if bias is None else y + bias.view(1, -1, 1, 1, 1)
        return TF.conv3d(x, w, bias, stride, padding)


def run(g, w, wino):
    f = WinoF() if wino else TF
    O.F = f
    try:
        out = O.forward(w, g["bgrs"], g["K"], list(g["c2ws"]), int(g["ref_index"]), float(g["depth_min"]), float(g["depth_max"]), float(g["discard"]))
    finally:
        O.F = TF
    return out, (f.n if wino else 0)


def main():
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    x = torch.randn(1, 5, 7, 9, 10)
    wt = torch.randn(6, 5, 3, 3, 3)
    assert (winograd(x, wt, 3) - TF.conv3d(x, wt, None, 1, 1)).abs().max() < 1e-3  # the transform itself
    assert (winograd(x[:, :, 0], wt[:, :, 0], 2) - TF.conv2d(x[:, :, 0], wt[:, :, 0], None, 1, 1)).abs().max() < 1e-3
    files = sys.argv[1:] or [f for f in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "mvsnet_*.npz"))) if str(np.load(f)["weights"]) == "trained"]
    for f in files:
        g = np.load(f)
        planes = tuple(int(v) for v in g["planes"])
        meta, tens = Wt.read_blob(os.path.join(ROOT, "weights", "tandem_va.tdmw"))
        w = O.Weights(dict(meta, depth_num=planes), tens)
        ref, _ = run(g, w, False)
        out, n = run(g, w, True)
        print("%s  %s planes %s: %d convolutions in Winograd form\n   F(2,3) fp32: %s" % (os.path.basename(f), g["bgrs"].shape, planes, n, report(out, ref)), flush=True)


if __name__ == "__main__":
    main()
