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


EXPORTED = "/root/reference/tandem/exported/tandem_512x320/model.pt"


@pytest.mark.skipif(not os.path.isfile(EXPORTED), reason="reference checkout not present")
def test_shipped_torchscript_archive_gives_the_committed_blob(tmp_path, trained_blob):
    dst = str(tmp_path / "shipped.tdmw")
    info = Wt.convert(EXPORTED, dst, depth_num=(48, 32, 8))
    assert info["view_aggregation"] and info["tensors"] == 236   # 280 state-dict entries minus 44 integer num_batches_tracked
    a, b = Wt.read_blob(dst), Wt.read_blob(trained_blob)
    assert a[0] == b[0] and list(a[1]) == list(b[1]) and all(np.array_equal(a[1][k], b[1][k]) for k in a[1])
