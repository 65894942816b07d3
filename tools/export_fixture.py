"""Writes a TDMS sample file -- the MI355X equivalent of the reference's sample_inputs.pt
(cva_mvsnet/export_model.py:62-65,164-180) consumed by test_dr_mvsnet (tandem_amd/libdr/dr_mvsnet.h).
Usage: python tools/export_fixture.py tests/golden/mvsnet_v3_64x96.npz out.tdms
       python tools/export_fixture.py path/to/sample_inputs.pt out.tdms     (the reference's own fixture format)

A `sample_inputs.pt` is a TorchScript container of named tensors (export_model.py:55-65: one attribute per key, dotted
names included).  read_sample_inputs_pt() takes out of it exactly what the reference's test_dr_mvsnet takes
(dr_mvsnet.cpp:388-459) and the same way: image (1,V,3,H,W) float RGB -> u8 BGR by `(unsigned char)(255.0 * x)`,
ref_index = view_num - 2, K = intrinsic_matrix.stage3[0], cam_to_world[0], depth_min / depth_max / discard_percentage[0],
outputs.stage3.{depth,confidence} as the expected result."""
import struct
import sys

import numpy as np


def read_sample_inputs_pt(path):
    import torch
    m = torch.jit.load(path, map_location="cpu")

    def attr(name):  # dotted names are single attributes of the container, not sub-modules
        try:
            return m._c.getattr(name)
        except Exception:
            return getattr(m, name)
    image = attr("image").float().numpy()  # (1, V, 3, H, W), RGB in [0, 1]
    if image.ndim != 5 or image.shape[0] != 1 or image.shape[2] != 3:
        raise ValueError("sample_inputs.pt: image has shape %s, expected (1, V, 3, H, W)" % (image.shape,))
    V = image.shape[1]
    rgb = np.trunc(255.0 * image[0].astype(np.float64))  # dr_mvsnet.cpp:417-427: (unsigned char)(255.0 * x), RGB -> BGR
    bgrs = np.ascontiguousarray(np.clip(rgb, 0, 255).astype(np.uint8)[:, ::-1].transpose(0, 2, 3, 1))
    one = lambda name: float(attr(name).float().reshape(-1)[0])
    return dict(bgrs=bgrs, K=attr("intrinsic_matrix.stage3").float().numpy()[0], c2ws=attr("cam_to_world").float().numpy()[0],
                ref_index=V - 2, depth_min=one("depth_min"), depth_max=one("depth_max"), discard=one("discard_percentage"),
                ref_s3_depth=attr("outputs.stage3.depth").float().numpy().reshape(image.shape[3], image.shape[4]),
                ref_s3_confidence=attr("outputs.stage3.confidence").float().numpy().reshape(image.shape[3], image.shape[4]))


def write_tdms(path, bgrs, K, c2ws, ref_index, depth_min, depth_max, discard, depth_ref, conf_ref):
    bgrs = np.ascontiguousarray(bgrs, np.uint8)
    V, H, W, _ = bgrs.shape
    with open(path, "wb") as f:
        f.write(b"TDMS0001")
        f.write(struct.pack("<4i", V, H, W, int(ref_index)))
        f.write(struct.pack("<3f", float(depth_min), float(depth_max), float(discard)))
        f.write(np.ascontiguousarray(K, "<f4").tobytes())
        f.write(np.ascontiguousarray(c2ws, "<f4").tobytes())
        f.write(bgrs.tobytes())
        f.write(np.ascontiguousarray(depth_ref, "<f4").tobytes())
        f.write(np.ascontiguousarray(conf_ref, "<f4").tobytes())


if __name__ == "__main__":
    g = read_sample_inputs_pt(sys.argv[1]) if sys.argv[1].endswith(".pt") else np.load(sys.argv[1])
    write_tdms(sys.argv[2], g["bgrs"], g["K"], g["c2ws"], g["ref_index"], g["depth_min"], g["depth_max"], g["discard"],
               g["ref_s3_depth"], g["ref_s3_confidence"])
