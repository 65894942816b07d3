"""CPU ORACLE for the DrMvsnet hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this file.  The product path (tandem_amd/) never does and fails loudly when
its HIP library is missing.

What it is: a stand-alone restatement, on torch CPU fp32 tensor ops, of the
arithmetic the reference executes for one `DrMvsnet::CallAsync` ->
`GetResult` round trip:

  host pre-processing   tandem/libdr/dr_mvsnet/src/dr_mvsnet.cpp:184-256
  network               cva_mvsnet/models/cva_mvsnet.py:98-184 (+ module.py parts cited below)
  host post-processing  tandem/libdr/dr_mvsnet/src/dr_mvsnet.cpp:296-329

The reference network is itself a sequence of ATen ops run by libtorch
(dr_mvsnet.cpp:294), so the restatement uses the same op family
(conv2d/conv3d/conv_transpose3d/grid_sample/...) on explicit weight arrays read
from the TDMW blob -- no nn.Module, no TorchScript, nothing from /root/reference
at run time (that tree does not exist on the GPU box).

Pinning: the reference's own fixture `sample_inputs.pt` is a
missing LFS blob, so parity with the shipped artefact is unpinned by the
reference.  This oracle is pinned instead against the reference's Python model
imported in the build container (oracle/ref_model.py) -- see
oracle/gen_golden.py and tests/test_oracle_mvsnet.py: bit-identical stage
outputs on the committed fixtures in tests/golden/.
"""
import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-5  # nn.BatchNorm2d/3d default, module.py:85,186


class Weights:
    def __init__(self, meta, tensors):
        self.meta = meta
        self.t = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in tensors.items()}

    def __getitem__(self, k):
        return self.t[k]


def _bn(x, w, p):
    # eval-mode BatchNorm, module.py:104-110 / :213-219
    return F.batch_norm(x, w[p + ".running_mean"], w[p + ".running_var"], w[p + ".weight"], w[p + ".bias"],
                        False, 0.1, BN_EPS)


def _cbr2(x, w, p, stride, pad):
    # Conv2d(bias=False)+BN+ReLU, module.py:61-110
    return F.relu(_bn(F.conv2d(x, w[p + ".conv.weight"], None, stride, pad), w, p + ".bn"))


def preprocess(bgrs, K, c2ws, ref_index):
    """dr_mvsnet.cpp:184-256: view reorder [ref, others in order]; BGR u8 -> RGB f32 / 255.0
    (computed in double, stored as float); K pyramid rows 0-1 x 0.25 / 0.5 / 1 (the C++ rule, :226-247)."""
    V = len(bgrs)
    order = [ref_index] + [i for i in range(V) if i != ref_index]
    imgs = []
    for v in order:
        a = np.asarray(bgrs[v])[..., ::-1].astype(np.float32).astype(np.float64) / 255.0
        imgs.append(torch.from_numpy(np.ascontiguousarray(a.astype(np.float32))).permute(2, 0, 1))
    image = torch.stack(imgs)  # (V,3,H,W)
    K = np.asarray(K, np.float32).reshape(3, 3)
    Ks = []
    for s in (0.25, 0.5, 1.0):
        k = K.copy()
        if s != 1.0:
            k[:2, :] = (np.float64(s) * k[:2, :].astype(np.float64)).astype(np.float32)
        Ks.append(torch.from_numpy(k))
    c2w = torch.from_numpy(np.stack([np.asarray(c2ws[v], np.float32).reshape(4, 4) for v in order]))
    return image, Ks, c2w


def feature_net(image, w):
    """module.py:496-531 (FeatureNet.forward), layers :461-494.  image (V,3,H,W)."""
    p = "feature_net."
    c3 = _cbr2(_cbr2(image, w, p + "conv0.0", 1, 1), w, p + "conv0.1", 1, 1)
    c2 = _cbr2(_cbr2(_cbr2(c3, w, p + "conv1.0", 2, 2), w, p + "conv1.1", 1, 1), w, p + "conv1.2", 1, 1)
    c1 = _cbr2(_cbr2(_cbr2(c2, w, p + "conv2.0", 2, 2), w, p + "conv2.1", 1, 1), w, p + "conv2.2", 1, 1)
    f1 = F.conv2d(c1, w[p + "out.stage1.weight"])
    i2 = F.interpolate(c1, scale_factor=2, mode="nearest") + F.conv2d(c2, w[p + "skip.stage2.weight"],
                                                                      w[p + "skip.stage2.bias"])
    f2 = F.conv2d(i2, w[p + "out.stage2.weight"], None, 1, 1)
    i3 = F.interpolate(i2, scale_factor=2, mode="nearest") + F.conv2d(c3, w[p + "skip.stage3.weight"],
                                                                      w[p + "skip.stage3.bias"])
    f3 = F.conv2d(i3, w[p + "out.stage3.weight"], None, 1, 1)
    return [f1, f2, f3]


def uniform_planes(dmin, dmax, D, h, w_):
    """module.py:1480-1500."""
    dmin = torch.tensor([dmin], dtype=torch.float32)
    dmax = torch.tensor([dmax], dtype=torch.float32)
    interval = (dmax - dmin) / (D - 1)
    k = torch.arange(D).type_as(interval)
    d = dmin[:, None] + interval[:, None] * k[None, :]  # (1,D)
    return d.unsqueeze(-1).unsqueeze(-1).repeat(1, 1, h, w_)[0], interval


def adaptive_planes(prev_depth, D, interval, h, w_):
    """cva_mvsnet.py:143-152 + module.py:1503-1565: bilinear x2 (align_corners=False) of the previous
    stage's *unfiltered* depth, lo = clamp_min(d - D/2*Delta, 1e-3), d_k = lo + ((lo + D*Delta) - lo) * k/D."""
    cur = F.interpolate(prev_depth[None, None], (h, w_), mode="bilinear", align_corners=False).squeeze(1)
    lo = (cur - (D / 2) * interval[:, None, None]).clamp(min=0.001)  # (1,h,w)
    hi = lo + D * interval[:, None, None]
    lin = torch.linspace(0, 1, D + 1)[:-1].type_as(cur).reshape(1, -1, 1, 1)
    return (lo.unsqueeze(1) + (hi - lo).unsqueeze(1) * lin)[0]


def warp(src_feat, planes, K, c2w_ref, c2w_src):
    """module.py:764-908 homo_warping.  src_feat (C,h,w), planes (D,h,w) -> (C,D,h,w)."""
    D, h, w_ = planes.shape
    K = K[None]
    r_w2c = torch.inverse(c2w_ref[None])
    s_w2c = torch.inverse(c2w_src[None])
    r_w2p = torch.clone(r_w2c)
    r_w2p[:, :3, :4] = torch.matmul(K, r_w2c[:, :3, :4])
    r_p2w = torch.inverse(r_w2p)
    s_w2p = torch.clone(s_w2c)
    s_w2p[:, :3, :4] = torch.matmul(K, s_w2c[:, :3, :4])
    M = torch.matmul(s_w2p, r_p2w)[0]
    rot, trans = M[:3, :3], M[:3, 3:4]
    y, x = torch.meshgrid(torch.arange(0, h, dtype=torch.float32), torch.arange(0, w_, dtype=torch.float32),
                          indexing="ij")
    xyz = torch.stack((x.reshape(-1), y.reshape(-1), torch.ones(h * w_)))[None]  # (1,3,hw)
    rot_xyz = torch.matmul(rot[None], xyz)
    proj = rot_xyz.unsqueeze(2) * planes.reshape(1, 1, D, -1) + trans.view(1, 3, 1, 1)  # (1,3,D,hw)
    xy = proj[:, :2] / proj[:, 2:3]
    gx = xy[:, 0] / (0.5 * (w_ - 1)) - 1
    gy = xy[:, 1] / (0.5 * (h - 1)) - 1
    grid = torch.stack((gx, gy), dim=3)  # (1,D,hw,2)
    neg = proj[:, 2:3] < 0.001  # (1,1,D,hw)
    out = F.grid_sample(src_feat[None], grid.view(1, D * h, w_, 2), mode="bilinear", padding_mode="zeros",
                        align_corners=True)  # (1,C,D*h,w)
    out = out.permute(0, 2, 3, 1)
    out[neg.view(1, D * h, w_)] = 0
    out = out.permute(0, 3, 1, 2)
    out[torch.isnan(out)] = 0
    return out.view(1, -1, D, h, w_)[0]


def gate(x, w, p):
    """cva_mvsnet.py:73-83 volume gate: Conv3d(C,1,1)+BN+ReLU+Conv3d(1,1,1)+BN+ReLU.  x (1,C,D,h,w)."""
    z = F.relu(_bn(F.conv3d(x, w[p + "0.weight"], w[p + "0.bias"]), w, p + "1"))
    return F.relu(_bn(F.conv3d(z, w[p + "3.weight"], w[p + "3.bias"]), w, p + "4"))


def cost_volume(feats, planes, K, c2w, w, stage, view_aggregation=True):
    """module.py:1061-1110.  feats (V,C,h,w) -> (C,D,h,w)."""
    V = feats.shape[0]
    D = planes.shape[0]
    ref = feats[0].unsqueeze(1).expand(-1, D, -1, -1)[None]
    if view_aggregation:
        acc = 0.0
        for v in range(1, V):
            d2 = (warp(feats[v], planes, K, c2w[0], c2w[v])[None] - ref).pow_(2)
            g = gate(d2, w, "volume_gates.stage%d." % stage)
            acc += (g + 1) * d2
        return acc.div_(V - 1)[0]
    ref = ref[0]
    s, s2 = ref, ref ** 2
    for v in range(1, V):
        wv = warp(feats[v], planes, K, c2w[0], c2w[v])
        s, s2 = s + wv, s2 + wv ** 2
    return s2 / V - (s / V) ** 2


def _cbr3(x, w, p, stride):
    return F.relu(_bn(F.conv3d(x, w[p + ".conv.weight"], None, stride, 1), w, p + ".bn"))


def _dbr3(x, w, p, stride, opad):
    return F.relu(_bn(F.conv_transpose3d(x, w[p + ".conv.weight"], None, stride, 1, opad), w, p + ".bn"))


def cost_reg(vol, w, stage, return_all=False):
    """module.py:577-600 CostRegNet.forward (layers :546-575).  vol (C,D,h,w) -> logits (D,h,w)."""
    p = "cost_regularization_net.stage%d." % stage
    four = vol.shape[1] == 4
    s5 = (1, 2, 2) if four else 2
    op7 = (0, 1, 1) if four else 1
    x = vol[None]
    c0 = _cbr3(x, w, p + "conv0", 1)
    c1 = _cbr3(c0, w, p + "conv1", 2)
    c2 = _cbr3(c1, w, p + "conv2", 1)
    c3 = _cbr3(c2, w, p + "conv3", 2)
    c4 = _cbr3(c3, w, p + "conv4", 1)
    c5 = _cbr3(c4, w, p + "conv5", s5)
    c6 = _cbr3(c5, w, p + "conv6", 1)
    x7 = c4 + _dbr3(c6, w, p + "conv7", s5, op7)
    x9 = c2 + _dbr3(x7, w, p + "conv9", 2, 1)
    x11 = c0 + _dbr3(x9, w, p + "conv11", 2, 1)
    logits = F.conv3d(x11, w[p + "prob.weight"], None, 1, 1)[0, 0]
    if return_all:
        return logits, dict(c0=c0[0], c1=c1[0], c2=c2[0], c3=c3[0], c4=c4[0], c5=c5[0], c6=c6[0], x7=x7[0],
                            x9=x9[0], x11=x11[0])
    return logits


def regress(logits, planes):
    """module.py:1116-1133: softmax over D, expectation, 4-neighbour confidence at trunc(E[k])."""
    D = logits.shape[0]
    p = F.softmax(logits[None], dim=1)  # (1,D,h,w)
    depth = torch.sum(p * planes[None], dim=1)
    sum4 = 4 * F.avg_pool3d(F.pad(p.unsqueeze(1), pad=[0, 0, 0, 0, 1, 2]), kernel_size=(4, 1, 1), stride=1,
                            padding=0).squeeze(1)
    idx = torch.sum(p * torch.arange(D, dtype=torch.float32).view(1, D, 1, 1), dim=1).long().clamp(0, D - 1)
    conf = torch.gather(sum4, 1, idx.unsqueeze(1)).squeeze(1)
    return depth[0], conf[0]


def edge_measure(depth, window=5):
    """module.py:1320-1343: 15th smallest |d(nb)-d(c)| over the zero-padded 5x5 window."""
    h, w_ = depth.shape
    w2 = window // 2
    m = (window * window) // 2
    num = window * (w2 + 1)
    dw = F.unfold(depth[None, None], kernel_size=(window, window), padding=w2, stride=1)  # (1,25,hw)
    edge = torch.abs(dw - dw[:, m:m + 1, :])
    edge, _ = torch.kthvalue(edge, k=num, dim=1)
    return edge.reshape(h, w_)


def filter_edges(depth, discard_percentage):
    """module.py:1345-1361: threshold = sorted(edge)[clamp(long(HW*(100-p)/100))] with the index computed
    in float32; mask = edge > threshold; returns (filtered depth copy, mask)."""
    h, w_ = depth.shape
    edge = edge_measure(depth)
    srt, _ = torch.sort(edge.reshape(-1))
    p = torch.tensor([discard_percentage], dtype=torch.float32)
    cut = (h * w_ * (100 - p) / 100.0).to(torch.long).clamp(0, h * w_ - 1)
    thr = srt[cut[0]]
    mask = edge > thr
    out = depth.clone()
    out[mask] = 0
    return out, mask, edge, thr


def forward(w, bgrs, K, c2ws, ref_index, depth_min, depth_max, discard_percentage, return_debug=False):
    """One DrMvsnet call.  Returns dict with stage-3 `depth`, `confidence`, `depth_dense`,
    `confidence_dense` (the four arrays of DrMvsnetOutput, dr_mvsnet.h:12-34) plus all stages."""
    meta = w.meta
    with torch.no_grad():
        image, Ks, c2w = preprocess(bgrs, K, c2ws, ref_index)
        H, W = image.shape[-2:]
        feats = feature_net(image, w)
        out, dbg = {}, {"features": feats}
        prev, base_interval = None, None
        for s in (1, 2, 3):
            D = meta["depth_num"][s - 1]
            sc = 2 ** (3 - s)
            h, w_ = H // sc, W // sc
            if s == 1:
                planes, base_interval = uniform_planes(depth_min, depth_max, D, h, w_)
            else:
                planes = adaptive_planes(prev, D, meta["interval_ratio"][s - 1] * base_interval, h, w_)
            vol = cost_volume(feats[s - 1], planes, Ks[s - 1], c2w, w, s, meta["view_aggregation"])
            logits = cost_reg(vol, w, s)
            depth, conf = regress(logits, planes)
            prev = depth
            out[s] = dict(depth_dense=depth, confidence_dense=conf)
            if return_debug:
                dbg[s] = dict(planes=planes, volume=vol, logits=logits)
        for s in (1, 2, 3):  # cva_mvsnet.py:166-173 -- after all stages
            d, mask, edge, thr = filter_edges(out[s]["depth_dense"], discard_percentage)
            c = out[s]["confidence_dense"].clone()
            c[mask] = 0
            out[s].update(depth=d, confidence=c, edge=edge, threshold=thr)
    res = {k: out[3][k].numpy() for k in ("depth", "confidence", "depth_dense", "confidence_dense")}
    res["stages"] = out
    if return_debug:
        res["debug"] = dbg
    return res
