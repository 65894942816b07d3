"""Writes tests/golden/fusion_tracker_small.npz: outputs of the CPU oracles (oracle/tsdf_oracle.c, tracker_oracle.c) on
small seeded inputs.  The reference ships no fixtures for dr_fusion / cuda_coarse_tracker (parity unpinned by the
reference), so these vectors pin OUR canonical results across rounds: tests check both the oracle and the HIP path
against them (bit patterns; reductions as float64).  Regenerate only when a documented deviation changes:
    python oracle/gen_golden_fusion.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from synth import scene  # noqa: E402
from oracle.tracker_oracle import TrackerOracle  # noqa: E402
from oracle.tsdf_oracle import TsdfOracle  # noqa: E402

H, W, VS = 48, 64, 0.04
MESH_LO, MESH_HI = (-2.0, -2.0, 0.0), (2.0, 2.0, 4.0)


def fusion_options(sc):
    return dict(voxel_size=VS, num_buckets=8000, bucket_size=10, num_blocks=8000, block_size=8, max_sdf_weight=64,
                truncation_distance=4 * VS, max_sensor_depth=10.0, min_sensor_depth=0.1, num_render_streams=1,
                fx=sc["fx"], fy=sc["fy"], cx=sc["cx"], cy=sc["cy"], height=H, width=W)


def canon_blocks(blocks):
    keys = sorted(blocks)
    return np.array(keys, np.int32).reshape(-1, 3), np.stack([blocks[k] for k in keys]).astype(np.uint8)


def canon_mesh(vert, cols):
    t = np.concatenate([vert.reshape(-1, 9), cols.reshape(-1, 9)], axis=1).view(np.uint32)
    return t[np.lexsort(t.T[::-1])]


def tracker_case():
    p = scene.make_tracking_pair(H, W, seed=6, sparse_fraction=0.25)
    return p, dict(huber=9.0, cutoff=20.0, ref_exposure=1.2, ref_aff=[0.03, 2.0], new_exposure=0.8, new_aff=[-0.02, -1.0])


def main():
    sc = scene.make_scans(3, H, W, seed=12)
    o = TsdfOracle(**fusion_options(sc))
    for bgr, depth, pose in sc["scans"]:
        assert o.integrate(bgr, depth, pose) == 0
    rb, rd = o.render(sc["scans"][0][2])
    coords, vox = canon_blocks(o.export_blocks())
    mv, mc = o.extract_mesh(MESH_LO, MESH_HI)
    p, c = tracker_case()
    t = TrackerOracle(W, H, c["huber"], c["cutoff"])
    t.setK(p["fx"], p["fy"], p["cx"], p["cy"])
    t.setReference(p["pc_u"], p["pc_v"], p["pc_idepth"], p["pc_color"], c["ref_exposure"], c["ref_aff"])
    t.setNew(p["dI_new"])
    res, sums = t.calcRes(p["refToNew"], c["new_exposure"], c["new_aff"], c["cutoff"])
    warped = np.stack(t.warped())
    Hm, b, raw = t.calcG(c["new_exposure"], c["new_aff"])
    out = os.path.join(ROOT, "tests", "golden", "fusion_tracker_small.npz")
    np.savez_compressed(out, block_coords=coords, block_voxels=vox, render_bgr=rb, render_depth=rd, mesh=canon_mesh(mv, mc),
                        stats=np.array([o.stats()[k] for k in ("blocks", "updated_last", "updated_total", "mismatches")], np.int64),
                        trk_res=res, trk_sums=sums, trk_warped=warped, trk_H=Hm, trk_b=b, trk_raw=raw)
    print("wrote", out, os.path.getsize(out), "bytes;", len(coords), "blocks,", len(mv) // 3, "triangles,", len(p["pc_u"]), "points")


if __name__ == "__main__":
    main()
