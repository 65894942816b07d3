"""GPU probe: per-kernel split of the BASELINE configs[3] loop, first lap (map growing) vs second lap (map built)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from synth import room
from tandem_amd.dr_fusion import DrFusion, DrFusionOptions
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
H, W = 480, 640
poses = room.loop_poses(1000, seed=7)[:n]
fr = room.render_frames(poses, H, W, device="cuda:0", seed=0)
opt = dict(voxel_size=0.005, num_buckets=500000, bucket_size=10, num_blocks=2500000, block_size=8, max_sdf_weight=64,
           truncation_distance=0.02, max_sensor_depth=10.0, min_sensor_depth=0.1, num_render_streams=1,
           fx=fr["fx"], fy=fr["fy"], cx=fr["cx"], cy=fr["cy"], height=H, width=W)
torch.cuda.synchronize()
f = DrFusion(DrFusionOptions(**opt))
for lap in range(3):
    ms = f.bench_sequence(fr["bgr"].data_ptr(), fr["depth"].data_ptr(), poses, render=True)
    st = f.stats()
    print("lap", lap, {k: round(v / n, 4) for k, v in ms.items()}, "blocks", st["blocks"], "upd/frame", st["updated_last"], flush=True)
f.close()
