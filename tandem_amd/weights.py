"""Flat weight blob ("TDMW") for the CVA-MVSNet depth pipeline.

The reference ships its network as a TorchScript archive that libtorch
interprets (tandem/libdr/dr_mvsnet/src/dr_mvsnet.cpp:24 `torch::jit::load`).
The MI355X engine has no TorchScript interpreter; `DrMvsnet(filename)` keeps its
signature (dr_mvsnet.h:38) but `filename` now names a TDMW blob: the raw
state-dict tensors (same key names as `CvaMVSNet.state_dict()`), which the C++
engine folds (BatchNorm -> scale/bias) and packs for its MFMA kernels at load.

Layout (little endian):
    char[8]  magic  = b"TDMW0001"
    int32    depth_num[3]
    float32  depth_interval_ratio[3]
    int32    view_aggregation (0/1)
    int32    feature_base_channels
    uint32   n_tensors
    n_tensors x { uint32 name_len; char name[name_len]; uint32 ndim;
                  uint32 dims[ndim]; float32 data[prod(dims)] }
"""
import struct
from collections import OrderedDict

import numpy as np

MAGIC = b"TDMW0001"


def write_blob(path, tensors, depth_num=(48, 32, 8), interval_ratio=(1.0, 0.5, 0.25),
               view_aggregation=True, base_channels=8):
    """tensors: ordered mapping name -> array-like float32 (non-float tensors are skipped)."""
    items = []
    for name, t in tensors.items():
        a = np.asarray(t.detach().cpu().numpy() if hasattr(t, "detach") else t)
        if a.dtype.kind != "f":
            continue  # num_batches_tracked etc.
        items.append((name, np.ascontiguousarray(a, dtype=np.float32)))
    with open(path, "wb") as f:
        f.write(MAGIC)
        f.write(struct.pack("<3i", *depth_num))
        f.write(struct.pack("<3f", *interval_ratio))
        f.write(struct.pack("<ii", int(bool(view_aggregation)), int(base_channels)))
        f.write(struct.pack("<I", len(items)))
        for name, a in items:
            nb = name.encode()
            f.write(struct.pack("<I", len(nb)))
            f.write(nb)
            f.write(struct.pack("<I", a.ndim))
            f.write(struct.pack("<%dI" % a.ndim, *a.shape))
            f.write(a.tobytes())


def read_blob(path):
    """Returns (meta dict, OrderedDict name -> np.float32 array)."""
    with open(path, "rb") as f:
        buf = f.read()
    if buf[:8] != MAGIC:
        raise ValueError("%s: not a TDMW blob" % path)
    off = 8
    depth_num = struct.unpack_from("<3i", buf, off); off += 12
    ratio = struct.unpack_from("<3f", buf, off); off += 12
    va, base = struct.unpack_from("<ii", buf, off); off += 8
    (n,) = struct.unpack_from("<I", buf, off); off += 4
    out = OrderedDict()
    for _ in range(n):
        (ln,) = struct.unpack_from("<I", buf, off); off += 4
        name = buf[off:off + ln].decode(); off += ln
        (nd,) = struct.unpack_from("<I", buf, off); off += 4
        dims = struct.unpack_from("<%dI" % nd, buf, off); off += 4 * nd
        cnt = int(np.prod(dims)) if nd else 1
        out[name] = np.frombuffer(buf, dtype="<f4", count=cnt, offset=off).reshape(dims).copy()
        off += 4 * cnt
    meta = dict(depth_num=tuple(depth_num), interval_ratio=tuple(ratio),
                view_aggregation=bool(va), base_channels=base)
    return meta, out


def random_state(depth_num=(48, 32, 8), base=8, seed=0):
    """Seeded random weights of the CvaMVSNet(view_aggregation=True) architecture
    (same key names / shapes as the reference state_dict, cva_mvsnet.py:58-83,
    module.py:461-494,546-575). BatchNorm statistics are randomised too so the
    folding path is exercised."""
    rng = np.random.RandomState(seed)
    sd = OrderedDict()

    def conv(name, co, ci, *k, bias=False):
        fan = ci * int(np.prod(k))
        sd[name + ".weight"] = (rng.randn(co, ci, *k) * np.sqrt(2.0 / fan)).astype(np.float32)
        if bias:
            sd[name + ".bias"] = (rng.randn(co) * 0.1).astype(np.float32)

    def bn(name, c):
        sd[name + ".weight"] = (1.0 + 0.2 * rng.randn(c)).astype(np.float32)
        sd[name + ".bias"] = (0.1 * rng.randn(c)).astype(np.float32)
        sd[name + ".running_mean"] = (0.1 * rng.randn(c)).astype(np.float32)
        sd[name + ".running_var"] = (0.5 + rng.rand(c)).astype(np.float32)

    def cbr2(name, co, ci, k):
        conv(name + ".conv", co, ci, k, k); bn(name + ".bn", co)

    b = base
    cbr2("feature_net.conv0.0", b, 3, 3); cbr2("feature_net.conv0.1", b, b, 3)
    cbr2("feature_net.conv1.0", 2 * b, b, 5); cbr2("feature_net.conv1.1", 2 * b, 2 * b, 3)
    cbr2("feature_net.conv1.2", 2 * b, 2 * b, 3)
    cbr2("feature_net.conv2.0", 4 * b, 2 * b, 5); cbr2("feature_net.conv2.1", 4 * b, 4 * b, 3)
    cbr2("feature_net.conv2.2", 4 * b, 4 * b, 3)
    conv("feature_net.out.stage1", 4 * b, 4 * b, 1, 1)
    conv("feature_net.out.stage2", 2 * b, 4 * b, 3, 3)
    conv("feature_net.out.stage3", b, 4 * b, 3, 3)
    conv("feature_net.skip.stage2", 4 * b, 2 * b, 1, 1, bias=True)
    conv("feature_net.skip.stage3", 4 * b, b, 1, 1, bias=True)
    cin = {1: 4 * b, 2: 2 * b, 3: b}
    for s in (1, 2, 3):
        p = "cost_regularization_net.stage%d." % s

        def cbr3(n, co, ci):
            conv(p + n + ".conv", co, ci, 3, 3, 3); bn(p + n + ".bn", co)

        def dbr3(n, ci, co):  # ConvTranspose3d weight is (Cin, Cout, 3,3,3)
            fan = ci * 27 / 8.0
            sd[p + n + ".conv.weight"] = (rng.randn(ci, co, 3, 3, 3) * np.sqrt(2.0 / fan)).astype(np.float32)
            bn(p + n + ".bn", co)

        cbr3("conv0", 8, cin[s]); cbr3("conv1", 16, 8); cbr3("conv2", 16, 16)
        cbr3("conv3", 32, 16); cbr3("conv4", 32, 32); cbr3("conv5", 64, 32); cbr3("conv6", 64, 64)
        dbr3("conv7", 64, 32); dbr3("conv9", 32, 16); dbr3("conv11", 16, 8)
        conv(p + "prob", 1, 8, 3, 3, 3)
    for s in (1, 2, 3):
        p = "volume_gates.stage%d." % s
        conv(p + "0", 1, cin[s], 1, 1, 1, bias=True)
        sd[p + "0.weight"] = np.abs(sd[p + "0.weight"])  # keep gates alive (ReLU)
        bn(p + "1", 1)
        conv(p + "3", 1, 1, 1, 1, 1, bias=True); bn(p + "4", 1)
    return sd


# ---- converter: the reference's training / export artefacts -> TDMW ----
_PREFIXES = ("cva_mvsnet.", "model.cva_mvsnet.", "model.", "module.")


def _strip(name):
    for p in _PREFIXES:
        if name.startswith(p):
            return name[len(p):]
    return name


def load_reference_weights(path, allow_pickle=False):
    """Reads what the reference can hand over for a trained CVA-MVSNet and returns (state dict name -> float32 array,
    hparams dict or None):
      * a PyTorch-Lightning checkpoint written by cva_mvsnet/train.py (dict with 'state_dict' whose keys carry the
        `cva_mvsnet.` prefix of models/tandem.py:16, and the hyper-parameters under 'hparams' / 'hyper_parameters');
      * a TorchScript archive written by cva_mvsnet/export_model.py:197-209 (model.pt, unfrozen: parameters intact) --
        key names are those of CvaMVSNet.state_dict() already, hyper-parameters are not stored in it;
      * a plain state_dict saved with torch.save."""
    import torch
    hparams = None
    try:
        sd = torch.jit.load(path, map_location="cpu").state_dict()
    except Exception:
        try:  # tensors and plain containers only: no code from the file is executed
            obj = torch.load(path, map_location="cpu", weights_only=True)
        except Exception as e:
            if not allow_pickle:
                raise ValueError("%s cannot be read with weights_only=True (%s); a Lightning checkpoint that pickles arbitrary objects needs "
                                 "--allow-pickle, which EXECUTES code stored in the file -- only for checkpoints you trust" % (path, str(e).splitlines()[0][:120]))
            obj = torch.load(path, map_location="cpu", weights_only=False)
        if isinstance(obj, dict) and "state_dict" in obj:
            hparams = obj.get("hparams") or obj.get("hyper_parameters")
            sd = obj["state_dict"]
        elif isinstance(obj, dict):
            sd = obj
        else:
            sd = obj.state_dict()
    out = OrderedDict()
    for k, v in sd.items():
        a = v.detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v)
        if a.dtype.kind == "f":
            out[_strip(k)] = np.ascontiguousarray(a, np.float32)
    if not any(k.startswith("feature_net.") for k in out) or not any(k.startswith("cost_regularization_net.") for k in out):
        raise ValueError("%s: no CvaMVSNet parameters found (feature_net.* / cost_regularization_net.*)" % path)
    return out, (dict(hparams) if hparams else None)


def convert(path, out_path, depth_num=None, interval_ratio=None, view_aggregation=None, allow_pickle=False):
    """`python -m tandem_amd.weights <ckpt | model.pt> out.tdmw`: header fields come from the checkpoint's hyper-parameters
    (MODEL.DEPTH_NUM, MODEL.DEPTH_INTERVAL_RATIO, MODEL.VIEW_AGGREGATION, MODEL.FEATURE_NET_BASE_CHANNELS; config.py) when
    it has them, else from the arguments; view aggregation is recognised by the volume_gates.* parameters."""
    sd, hp = load_reference_weights(path, allow_pickle)
    hp = hp or {}
    # The hypothesis counts are NOT in the parameters: a TorchScript model.pt (and a bare state_dict) does not carry them,
    # and the shipped tandem_512x320/model.pt is a (48,4,4) model -- a silent (48,32,8) default would write a blob that
    # loads, runs and is wrong.
    if not depth_num and "MODEL.DEPTH_NUM" not in hp:
        raise ValueError("%s stores no hyper-parameters: pass --depth-num (e.g. 48,32,8; 48,4,4 for the shipped tandem_512x320 model)" % path)
    if int(hp.get("MODEL.COST_VOLUME_BASE_CHANNELS", 8)) != 8:
        raise ValueError("MODEL.COST_VOLUME_BASE_CHANNELS=%s: only 8 (the architecture TANDEM ships) is supported" % hp["MODEL.COST_VOLUME_BASE_CHANNELS"])
    if (hp.get("MODEL.CONV2D_NORMALIZATION", "batchnorm") != "batchnorm" or hp.get("MODEL.CONV3D_NORMALIZATION", "batchnorm") != "batchnorm"
            or hp.get("MODEL.CONV2D_USE_BN_SKIP", False)):
        raise ValueError("only the batchnorm / no-BN-skip architecture TANDEM ships is supported")
    dn = tuple(depth_num or hp.get("MODEL.DEPTH_NUM") or (48, 32, 8))
    ratio = tuple(interval_ratio or hp.get("MODEL.DEPTH_INTERVAL_RATIO") or (1.0, 0.5, 0.25))
    has_gates = any(k.startswith("volume_gates.") for k in sd)
    va = has_gates if view_aggregation is None else bool(view_aggregation)
    if "MODEL.VIEW_AGGREGATION" in hp and bool(hp["MODEL.VIEW_AGGREGATION"]) != va:
        raise ValueError("MODEL.VIEW_AGGREGATION=%s contradicts the parameters (volume_gates.* %s)" % (hp["MODEL.VIEW_AGGREGATION"], "present" if has_gates else "absent"))
    if va and not has_gates:
        raise ValueError("view aggregation requested but the checkpoint has no volume_gates.* parameters")
    base = int(hp.get("MODEL.FEATURE_NET_BASE_CHANNELS", sd["feature_net.conv0.0.conv.weight"].shape[0]))
    if len(dn) != 3 or len(ratio) != 3:
        raise ValueError("depth_num / interval_ratio need three stages")
    write_blob(out_path, sd, depth_num=dn, interval_ratio=ratio, view_aggregation=va, base_channels=base)
    return dict(tensors=len(sd), depth_num=dn, interval_ratio=ratio, view_aggregation=va, base_channels=base)


def _main(argv=None):
    import argparse
    ap = argparse.ArgumentParser(prog="python -m tandem_amd.weights",
                                 description="Convert a CVA-MVSNet Lightning .ckpt / exported model.pt / state_dict to the TDMW blob DrMvsnet loads")
    ap.add_argument("input")
    ap.add_argument("output")
    ap.add_argument("--depth-num", type=lambda s: tuple(int(v) for v in s.split(",")), default=None,
                    help="e.g. 48,4,4 (the shipped tandem_512x320 model) -- needed for model.pt, which does not store it")
    ap.add_argument("--interval-ratio", type=lambda s: tuple(float(v) for v in s.split(",")), default=None)
    ap.add_argument("--view-aggregation", type=int, choices=(0, 1), default=None)
    ap.add_argument("--allow-pickle", action="store_true", help="fall back to torch.load(weights_only=False) -- executes code stored in the checkpoint")
    a = ap.parse_args(argv)
    info = convert(a.input, a.output, a.depth_num, a.interval_ratio, a.view_aggregation, a.allow_pickle)
    print("wrote %s: %d tensors, depth_num %s, interval ratio %s, view aggregation %s, base channels %d"
          % (a.output, info["tensors"], info["depth_num"], info["interval_ratio"], info["view_aggregation"], info["base_channels"]))


if __name__ == "__main__":
    _main()
