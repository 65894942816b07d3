"""GPU diagnostic: compares every intermediate tensor of the HIP DrMvsnet engine with the CPU oracle.
Usage (on a GPU box): python tools/check_mvsnet.py [H W V]   -- prints one line per tensor, never stops early."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import mvsnet_oracle as O  # noqa: E402
from synth import scene  # noqa: E402
from tandem_amd import weights as Wt  # noqa: E402
from tandem_amd.dr_mvsnet import DrMvsnet  # noqa: E402


def cl(t):  # (C,D,h,w) or (V,C,h,w) torch -> channels-last numpy
    t = t.numpy()
    return np.moveaxis(t, 0 if t.ndim == 4 and False else 0, -1)


def report(name, got, ref):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    if got.shape != ref.shape:
        print(f"{name:22s} SHAPE MISMATCH got {got.shape} ref {ref.shape}")
        return
    err = np.abs(got - ref)
    scale = np.abs(ref).max() + 1e-30
    print(f"{name:22s} shape {str(got.shape):22s} max|err| {err.max():.3e} (rel {err.max() / scale:.2e}) mean {err.mean():.3e} "
          f"nan {int(np.isnan(got).sum())} frac>1e-3rel {(err > 1e-3 * scale).mean():.4f}")


def main():
    H, W, V = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (64, 96, 3)
    blob = os.path.join(ROOT, "weights/tandem_va.tdmw")
    meta, tens = Wt.read_blob(blob)
    w = O.Weights(meta, tens)
    win = scene.make_window(H, W, V, seed=3)
    t0 = time.time()
    ref = O.forward(w, win["bgrs"], win["K"], win["c2ws"], win["ref_index"], win["depth_min"], win["depth_max"], 10.0,
                    return_debug=True)
    print("oracle %.2fs" % (time.time() - t0))
    m = DrMvsnet(blob)
    m.upload(H, W, V, win["ref_index"], win["bgrs"], win["K"], win["c2ws"], win["depth_min"], win["depth_max"], 10.0)
    ms = m.forward(1)
    print("hip forward %.3f ms" % ms)
    image, _, _ = O.preprocess(win["bgrs"], win["K"], win["c2ws"], win["ref_index"])
    if os.environ.get("DR_FN_FRONT") == "0":  # (k_fn_front, the default, keeps the float image in LDS: the tensor exists in the three-launch form only)
        report("image", m.tensor("image")[..., :3], image.permute(0, 2, 3, 1).numpy())
    for s in (1, 2, 3):
        report(f"feat{s}", m.tensor(f"feat{s}"), ref["debug"]["features"][s - 1].permute(0, 2, 3, 1).numpy())
    for s in (1, 2, 3):
        d = ref["debug"][s]
        report(f"volume{s}", m.tensor(f"volume{s}"), d["volume"].permute(1, 2, 3, 0).numpy())
        # feed the oracle's cost-reg with the oracle volume but also check layer by layer
        logits, allr = O.cost_reg(d["volume"], w, s, return_all=True)
        for k, nm in (("c0", "conv0"), ("c1", "conv1"), ("c2", "conv2"), ("c3", "conv3"), ("c4", "conv4"), ("c5", "conv5"),
                      ("c6", "conv6"), ("x7", "conv7"), ("x9", "conv9"), ("x11", "conv11")):
            report(f"s{s}.{nm}", m.tensor(f"s{s}.{nm}"), allr[k].permute(1, 2, 3, 0).numpy())
        report(f"logits{s}", m.tensor(f"logits{s}")[..., 0], d["logits"].numpy())
        dep, conf = m.stage_output(s)
        report(f"depth{s}", dep, ref["stages"][s]["depth_dense"].numpy())
        report(f"conf{s}", conf, ref["stages"][s]["confidence_dense"].numpy())
    out = m.download()
    report("edge", m.tensor("edge")[0, :, :, 0], ref["stages"][3]["edge"].numpy())
    report("depth(filtered)", out.depth, ref["depth"])
    report("confidence(filt)", out.confidence, ref["confidence"])
    print("mask mismatch fraction", float(((out.depth == 0) != (ref["depth"] == 0)).mean()), "filtered frac",
          float((out.depth == 0).mean()), "ref", float((ref["depth"] == 0).mean()))
    print("reference criterion mean|depth-ref| =", float(np.abs(out.depth - ref["depth"]).mean()),
          " mean|conf-ref| =", float(np.abs(out.confidence - ref["confidence"]).mean()))
    for r in m.profile():
        print("  %-18s %8.3f ms  %-16s %8.2f GF %8.1f MB" % (r["op"], r["ms"], r["kernel"], r["flops"] / 1e9, r["bytes"] / 1e6))
    f, b = m.work()
    print("work: %.2f GFLOP  %.3f GB" % (f / 1e9, b / 1e9))
    ms = m.forward(5)
    print("5 forwards: %.3f ms each" % (ms / 5))


if __name__ == "__main__":
    main()
