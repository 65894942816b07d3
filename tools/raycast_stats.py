"""Iteration statistics of the ray-caster over the first K frames of bench.py's TSDF loop.  Run by bench.py as a subprocess with
DR_MI355X_LIB=tandem_amd/libdr_mi355x_hooks.so DR_RAYCAST_STATS=1: in the PARITY build that switch makes every render a synchronous, counting
launch of k_raycast2's own loop, which prints one "raycast stats:" line to stderr per frame.  The parent averages `iterations/lane` -- the
`steps` of SURVEY 8(d)'s lower bound for the ray-cast, 7 B + steps x 8 corners x 8 B per output pixel.
    python tools/raycast_stats.py K [voxel_size truncation]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
K = int(sys.argv[1]) if len(sys.argv) > 1 else 100
vs, tr = (float(sys.argv[2]), float(sys.argv[3])) if len(sys.argv) > 3 else (0.005, 0.02)
import torch  # noqa: E402
from synth import room  # noqa: E402
from tandem_amd.dr_fusion import DrFusion, DrFusionOptions  # noqa: E402

H, W = 480, 640
poses = room.loop_poses(K, seed=7)
fr = room.render_frames(poses, H, W, device="cuda:0", seed=0)
f = DrFusion(DrFusionOptions(voxel_size=vs, num_buckets=500000, bucket_size=10, num_blocks=600000, block_size=8, max_sdf_weight=64, truncation_distance=tr,
                             max_sensor_depth=10.0, min_sensor_depth=0.1, num_render_streams=1, fx=fr["fx"], fy=fr["fy"], cx=fr["cx"], cy=fr["cy"], height=H, width=W))
torch.cuda.synchronize()
f.bench_sequence(fr["bgr"].data_ptr(), fr["depth"].data_ptr(), poses, render=True)
f.close()
