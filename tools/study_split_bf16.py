"""CPU study for an opt-in reduced-precision mode of the convolution kernels: what the depth maps
lose when every convolution's operands are split into bf16 terms and multiplied on the bf16 MFMA path (products exact, fp32
accumulation -- which torch's fp32 convolution of bf16-representable values reproduces up to summation order).

    x = xh + xl,  w = wh + wl  (xh = bf16(x), xl = bf16(x - xh))
    1 term : xh*wh                         (plain bf16)
    2 terms: xh*wh + xl*wh                 (activations split, weights rounded)
    3 terms: xh*wh + xl*wh + xh*wl         (the classic bf16x3; the dropped xl*wl term is ~2^-16 relative)

The oracle (oracle/mvsnet_oracle.py) runs unchanged except for its conv2d / conv3d / conv_transpose3d calls; everything else
(BatchNorm affine, warp, gates' sigmoid-free arithmetic, soft-argmin, filter) stays fp32 as in the kernels' epilogues and
vector kernels.  Compared with the fp32 oracle through the bounds of tests/test_mvsnet_gpu.py::compare.

    python tools/study_split_bf16.py [fixture.npz ...]      (default: the trained-weight fixtures under tests/golden)
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
from oracle import mvsnet_oracle as O  # noqa: E402
from tandem_amd import weights as Wt  # noqa: E402


def split(t):
    hi = t.to(torch.bfloat16).to(torch.float32)
    lo = (t - hi).to(torch.bfloat16).to(torch.float32)
    return hi, lo


class SplitF:
    """Stand-in for torch.nn.functional inside the oracle module: the three convolutions in `terms` bf16 products."""

    def __init__(self, terms, only=None):
        self.terms, self.only = terms, only

    def __getattr__(self, name):
        return getattr(TF, name)

    def _conv(self, fn, x, w, bias, *a):
        if self.terms == 0:
            return fn(x, w, bias, *a)
        xh, xl = split(x)
        wh, wl = split(w)
        y = fn(xh, wh, None, *a)
        if self.terms >= 2:
            y = y + fn(xl, wh, None, *a)
        if self.terms >= 3:
            y = y + fn(xh, wl, None, *a)
        if bias is not None:
            y = y + bias.view(1, -1, *([1] * (y.dim() - 2)))
        return y

    def conv2d(self, x, w, bias=None, stride=1, padding=0):
        return self._conv(TF.conv2d, x, w, bias, stride, padding)

    def conv3d(self, x, w, bias=None, stride=1, padding=0):
        return self._conv(TF.conv3d, x, w, bias, stride, padding)

    def conv_transpose3d(self, x, w, bias=None, stride=1, padding=0, output_padding=0):
        return self._conv(TF.conv_transpose3d, x, w, bias, stride, padding, output_padding)


def run(g, w, terms):
    O.F = SplitF(terms)
    try:
        return O.forward(w, g["bgrs"], g["K"], list(g["c2ws"]), int(g["ref_index"]), float(g["depth_min"]), float(g["depth_max"]), float(g["discard"]))
    finally:
        O.F = TF


def report(out, ref):
    d = np.abs(out["depth_dense"] - ref["depth_dense"])
    c = np.abs(out["confidence_dense"] - ref["confidence_dense"])
    flips = ((out["depth"] == 0) != (ref["depth"] == 0)).mean()
    ok = d.mean() < 1e-4 and (d < 2e-3).mean() > 0.999 and d.max() < 5e-2 and c.mean() < 1e-4 and flips < 2e-3
    return "depth mean %.2e  max %.2e  >2mm %.4f%%  conf mean %.2e  flips %.4f%%  -> %s" % (
        d.mean(), d.max(), 100 * (d >= 2e-3).mean(), c.mean(), 100 * flips, "within the fp32 bounds" if ok else "OUTSIDE the fp32 bounds")


def main():
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    files = sys.argv[1:] or [f for f in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "mvsnet_*.npz"))) if str(np.load(f)["weights"]) == "trained"]
    for f in files:
        g = np.load(f)
        planes = tuple(int(v) for v in g["planes"])
        if str(g["weights"]) == "trained":
            meta, tens = Wt.read_blob(os.path.join(ROOT, "weights", "tandem_va.tdmw"))
        else:
            meta, tens = None, Wt.random_state(planes, seed=7)
        va = bool(g["view_aggregation"]) if "view_aggregation" in g else True
        meta = dict(depth_num=planes, interval_ratio=(1.0, 0.5, 0.25) if meta is None else meta["interval_ratio"], view_aggregation=va, base_channels=8)
        w = O.Weights(meta, tens)
        ref = run(g, w, 0)
        gold = np.abs(ref["depth_dense"] - g["ref_s3_depth_dense"]).max()
        print("%s  %s planes %s (fp32 oracle vs the committed fixture: max %.1e)" % (os.path.basename(f), g["bgrs"].shape, planes, gold))
        for terms in (1, 2, 3):
            print("   %d bf16 term%s: %s" % (terms, " " if terms == 1 else "s", report(run(g, w, terms), ref)), flush=True)


if __name__ == "__main__":
    main()
