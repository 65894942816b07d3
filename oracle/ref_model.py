"""Loader for the REAL reference model (build container only) -- TEST INFRASTRUCTURE.

Imports cva_mvsnet/models/{module,cva_mvsnet}.py from /root/reference by file path under a
synthetic package (their package __init__ pulls in pytorch_lightning, which is absent) with an
empty `torchvision` stub (only an off-path class uses it, module.py:3).  Used by
oracle/gen_golden.py to pin oracle/mvsnet_oracle.py and to write tests/golden/*.npz.
/root/reference does not exist on the GPU box: nothing under tests -m gpu / bench / smoke calls this.
"""
import importlib.util
import os
import sys
import types

REF = os.environ.get("TANDEM_REFERENCE", "/root/reference")
EXPORTED = os.path.join(REF, "tandem/exported/tandem_512x320/model.pt")


def available():
    return os.path.isfile(os.path.join(REF, "cva_mvsnet/models/cva_mvsnet.py"))


def _load():
    if "refmodels.cva_mvsnet" in sys.modules:
        return sys.modules["refmodels.module"], sys.modules["refmodels.cva_mvsnet"]
    sys.modules.setdefault("torchvision", types.ModuleType("torchvision"))
    pkg = types.ModuleType("refmodels")
    pkg.__path__ = [os.path.join(REF, "cva_mvsnet/models")]
    sys.modules["refmodels"] = pkg
    mods = []
    for name in ("module", "cva_mvsnet"):
        spec = importlib.util.spec_from_file_location("refmodels." + name,
                                                      os.path.join(REF, "cva_mvsnet/models", name + ".py"))
        m = importlib.util.module_from_spec(spec)
        sys.modules["refmodels." + name] = m
        spec.loader.exec_module(m)
        mods.append(m)
    return tuple(mods)


def exported_state_dict():
    """The 280-tensor state dict recovered from the shipped (unfrozen) TorchScript archive."""
    import torch
    return torch.jit.load(EXPORTED, map_location="cpu").state_dict()


def build(depth_num=(48, 32, 8), state_dict=None, view_aggregation=True):
    import torch
    _, cva = _load()
    net = cva.CvaMVSNet(depth_num=tuple(depth_num), view_aggregation=view_aggregation).eval()
    sd = exported_state_dict() if state_dict is None else state_dict
    sd = {k: (v if isinstance(v, torch.Tensor) else torch.from_numpy(v)) for k, v in sd.items()}
    missing = net.load_state_dict(sd, strict=False)
    assert not [k for k in missing.missing_keys if "num_batches_tracked" not in k], missing
    return net, cva


def run(net, cva, image, Ks, c2w, depth_min, depth_max, discard_percentage):
    """image (V,3,H,W) f32; Ks [K1,K2,K3]; c2w (V,4,4).  Returns the reference's Outputs tuple."""
    import torch
    with torch.no_grad():
        return net(image[None], cva.StageTensor(*[k[None] for k in Ks]), c2w[None],
                   torch.tensor([depth_min], dtype=torch.float32), torch.tensor([depth_max], dtype=torch.float32),
                   torch.tensor([discard_percentage], dtype=torch.float32))
