"""Writes a TDMS sample file -- the MI355X equivalent of the reference's sample_inputs.pt
(cva_mvsnet/export_model.py:62-65,164-180) consumed by test_dr_mvsnet (tandem_amd/libdr/dr_mvsnet.h).
Usage: python tools/export_fixture.py tests/golden/mvsnet_v3_64x96.npz out.tdms"""
import struct
import sys

import numpy as np


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
    g = np.load(sys.argv[1])
    write_tdms(sys.argv[2], g["bgrs"], g["K"], g["c2ws"], g["ref_index"], g["depth_min"], g["depth_max"], g["discard"],
               g["ref_s3_depth"], g["ref_s3_confidence"])
