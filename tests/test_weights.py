"""CPU: the checkpoint converter `python -m tandem_amd.weights` (reference: the Lightning checkpoint
layout of cva_mvsnet/models/tandem.py:13-24 and the TorchScript export of cva_mvsnet/export_model.py:197-209)."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from tandem_amd import weights as Wt

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_lightning_checkpoint_prefix_and_hparams(tmp_path):
    sd = Wt.random_state((48, 4, 4), seed=3)
    ck = {"state_dict": {"cva_mvsnet." + k: torch.from_numpy(v) for k, v in sd.items()},
          "hparams": {"MODEL.DEPTH_NUM": (48, 4, 4), "MODEL.DEPTH_INTERVAL_RATIO": (1.0, 0.5, 0.25), "MODEL.VIEW_AGGREGATION": True,
                      "MODEL.FEATURE_NET_BASE_CHANNELS": 8, "MODEL.COST_VOLUME_BASE_CHANNELS": 8}}
    ck["state_dict"]["cva_mvsnet.feature_net.conv0.0.bn.num_batches_tracked"] = torch.tensor(7)
    src, dst = str(tmp_path / "m.ckpt"), str(tmp_path / "m.tdmw")
    torch.save(ck, src)
    r = subprocess.run([sys.executable, "-m", "tandem_amd.weights", src, dst], capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0, r.stderr
    meta, tens = Wt.read_blob(dst)
    assert meta == dict(depth_num=(48, 4, 4), interval_ratio=(1.0, 0.5, 0.25), view_aggregation=True, base_channels=8)
    assert list(tens) == list(sd) and all(np.array_equal(tens[k], sd[k]) for k in sd)


def test_plain_state_dict_without_gates_and_contradictions(tmp_path):
    sd = {k: torch.from_numpy(v) for k, v in Wt.random_state((48, 32, 8), seed=1).items() if not k.startswith("volume_gates.")}
    src, dst = str(tmp_path / "sd.pt"), str(tmp_path / "sd.tdmw")
    torch.save(sd, src)
    info = Wt.convert(src, dst, depth_num=(48, 32, 8))
    assert info["view_aggregation"] is False and Wt.read_blob(dst)[0]["view_aggregation"] is False
    with pytest.raises(ValueError):
        Wt.convert(src, dst, view_aggregation=True)
    torch.save({"state_dict": sd, "hparams": {"MODEL.VIEW_AGGREGATION": True}}, src)
    with pytest.raises(ValueError):
        Wt.convert(src, dst)
    torch.save({"something": torch.zeros(3)}, src)
    with pytest.raises(ValueError):
        Wt.convert(src, dst)
    # no hyper-parameters in the file and none on the command line: refuse (the hypothesis counts cannot be guessed --
    # the shipped model.pt is a (48,4,4) model), and refuse a cost-volume width the kernels do not implement
    torch.save(sd, src)
    with pytest.raises(ValueError, match="depth-num"):
        Wt.convert(src, dst)
    torch.save({"state_dict": sd, "hparams": {"MODEL.DEPTH_NUM": (48, 32, 8), "MODEL.COST_VOLUME_BASE_CHANNELS": 16}}, src)
    with pytest.raises(ValueError, match="COST_VOLUME_BASE_CHANNELS"):
        Wt.convert(src, dst)


EXPORTED = "/root/reference/tandem/exported/tandem_512x320/model.pt"


@pytest.mark.skipif(not os.path.isfile(EXPORTED), reason="reference checkout not present")
def test_shipped_torchscript_archive_gives_the_committed_blob(tmp_path, trained_blob):
    dst = str(tmp_path / "shipped.tdmw")
    info = Wt.convert(EXPORTED, dst, depth_num=(48, 32, 8))
    assert info["view_aggregation"] and info["tensors"] == 236   # 280 state-dict entries minus 44 integer num_batches_tracked
    a, b = Wt.read_blob(dst), Wt.read_blob(trained_blob)
    assert a[0] == b[0] and list(a[1]) == list(b[1]) and all(np.array_equal(a[1][k], b[1][k]) for k in a[1])


def test_sample_inputs_pt_converts_to_the_same_tdms(tmp_path):
    """The reference's own fixture format (`sample_inputs.pt`, a TorchScript container of named tensors written by
    cva_mvsnet/export_model.py:55-65,164-180, read by test_dr_mvsnet, dr_mvsnet.cpp:388-459): a container built the way
    export_model.py builds it converts (tools/export_fixture.py) to byte-for-byte the TDMS file the arrays give directly."""
    import sys
    import torch
    from torch import nn
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from export_fixture import read_sample_inputs_pt, write_tdms
    g = np.load(os.path.join(ROOT, "tests", "golden", "mvsnet_v3_64x96.npz"))
    bgrs = g["bgrs"]  # (V, H, W, 3) u8 BGR, ref view at index V - 2 as TANDEM passes it
    V = bgrs.shape[0]
    assert int(g["ref_index"]) == V - 2
    image = torch.from_numpy(bgrs[..., ::-1].transpose(0, 3, 1, 2).astype(np.float32) / 255.0)[None]  # (1, V, 3, H, W) RGB
    image = image + 0.2 / 255.0  # values between two u8 steps: the reader must truncate like `(unsigned char)(255.0 * x)` does
    K = torch.from_numpy(g["K"].astype(np.float32))[None]

    class TensorContainer(nn.Module):  # export_model.py:55-59
        def __init__(self, tensor_dict):
            super().__init__()
            for key, value in tensor_dict.items():
                setattr(self, key, value)

    td = {"image": image, "intrinsic_matrix.stage1": K * 0.25, "intrinsic_matrix.stage2": K * 0.5, "intrinsic_matrix.stage3": K,
          "cam_to_world": torch.from_numpy(g["c2ws"].astype(np.float32))[None],
          "depth_min": torch.tensor([float(g["depth_min"])]), "depth_max": torch.tensor([float(g["depth_max"])]),
          "discard_percentage": torch.tensor([float(g["discard"])]),
          "outputs.stage3.depth": torch.from_numpy(g["ref_s3_depth"])[None], "outputs.stage3.confidence": torch.from_numpy(g["ref_s3_confidence"])[None]}
    pt = str(tmp_path / "sample_inputs.pt")
    torch.jit.script(TensorContainer(td)).save(pt)  # export_model.py:62-65
    r = read_sample_inputs_pt(pt)
    assert np.array_equal(r["bgrs"], bgrs) and r["ref_index"] == V - 2
    a, b = str(tmp_path / "a.tdms"), str(tmp_path / "b.tdms")
    write_tdms(a, r["bgrs"], r["K"], r["c2ws"], r["ref_index"], r["depth_min"], r["depth_max"], r["discard"], r["ref_s3_depth"], r["ref_s3_confidence"])
    write_tdms(b, bgrs, g["K"], g["c2ws"], g["ref_index"], g["depth_min"], g["depth_max"], g["discard"], g["ref_s3_depth"], g["ref_s3_confidence"])
    assert open(a, "rb").read() == open(b, "rb").read()
