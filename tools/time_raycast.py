"""Times DrFusion::RenderAsync -> GetRenderResult (ray-cast kernel + D2H) on the bench's fused map, integration drained first."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from synth import scene
from tandem_amd.dr_fusion import DrFusion, DrFusionOptions
H, W = 480, 640
sc = scene.make_scans(10, H, W, seed=100, texture_terms=3)
f = DrFusion(DrFusionOptions(voxel_size=0.005, num_buckets=400000, bucket_size=10, num_blocks=2000000, block_size=8, max_sdf_weight=64,
                             truncation_distance=0.02, max_sensor_depth=10.0, min_sensor_depth=0.1, num_render_streams=1,
                             fx=sc["fx"], fy=sc["fy"], cx=sc["cx"], cy=sc["cy"], height=H, width=W))
z = np.zeros((H, W), np.float32); zb = np.zeros((H, W, 3), np.uint8)
for bgr, depth, pose in sc["scans"]:
    f.IntegrateScanAsync(bgr, depth, pose); f.RenderAsync([pose]); f.GetRenderResult()
ts = []
for i in range(8):
    pose = sc["scans"][i % 10][2]
    f.IntegrateScanAsync(zb, z, pose)  # integrates nothing; keeps the call protocol
    f.Synchronize()
    t0 = time.perf_counter(); f.RenderAsync([pose]); b, d = f.GetRenderResult(); ts.append(time.perf_counter() - t0)
print("raycast + D2H: min %.3f ms, median %.3f ms; hit fraction %.3f" % (1e3 * min(ts), 1e3 * sorted(ts)[len(ts) // 2], float((d[0] > 0).mean())))
f.close()
