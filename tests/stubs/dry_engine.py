"""Stand-in for tandem_amd.dr_mvsnet.DrMvsnet used ONLY by `DR_BENCH_DRY_RUN=1 python bench.py ...` (tests/test_bench_launch.py):
no GPU, no arithmetic -- it records what the launcher asked each rank's engine to do, so that bench.py's own spawning path
(`--gpus 8` -> torch.distributed.run -> 8 ranks), its rank / world plumbing, the replicas reduction, the view-shard leg's
participant logic (shard_world) and the JSON assembly can be exercised on a CPU box before the one real 8-GPU run.
TEST INFRASTRUCTURE: a dry run reports "value": null and "dry_run": true, never a measurement."""
import json
import os
import time

import numpy as np


class _Out:
    def __init__(self, h, w):
        self.depth_dense = np.full((h, w), 1.25, np.float32)  # every rank "computes" the same map


class DryMvsnet:
    def __init__(self, filename, device=0):
        self.log = dict(rank=int(os.environ.get("RANK", "0")), uploads=[], forwards=0, comm_init=None, set_view_shard=None, closed=False)
        self._hw = (0, 0)

    def upload(self, height, width, view_num, ref_index, bgrs, K, c2ws, depth_min, depth_max, discard):
        self._hw = (height, width)
        self.log["uploads"].append(dict(views=view_num, ref_index=ref_index))

    def forward(self, iters=1):
        time.sleep(0.0005 * iters)
        self.log["forwards"] += iters
        return 2.0 * iters

    def work(self):
        return 1e9, 1e9

    def profile(self):
        return [dict(op="dry", kernel="dry", flops=1e9, bytes=1e9, ms=1.0)]

    def set_view_shard(self, nsrc_total):
        self.log["set_view_shard"] = nsrc_total

    @staticmethod
    def comm_available():
        return True

    @staticmethod
    def comm_unique_id():
        return bytes(128)

    def comm_init(self, rank, world, unique_id):
        assert len(unique_id) == 128
        self.log["comm_init"] = [rank, world]

    def comm_count(self):
        return self.log["comm_init"][1] if self.log["comm_init"] else 0

    def device_tensor(self, name):
        return 0, 1000

    def download(self):
        return _Out(*self._hw)

    def close(self):
        if self.log["closed"]:
            return
        self.log["closed"] = True
        d = os.environ.get("DR_BENCH_DRY_DIR")
        if d:
            with open(os.path.join(d, "rank%d_%d.json" % (self.log["rank"], id(self))), "w") as f:
                json.dump(self.log, f)


def make_window(H, W, V, seed=0):
    """Window of the right SHAPES only (synth.scene.make_window renders 5 s of textures per 640x480x7 window)."""
    return dict(bgrs=[np.zeros((H, W, 3), np.uint8) for _ in range(V)], K=np.eye(3, dtype=np.float32),
                c2ws=[np.eye(4, dtype=np.float32) for _ in range(V)], ref_index=V - 2, depth_min=0.5, depth_max=5.0,
                gt_depth=np.ones((H, W), np.float32), height=H, width=W, view_num=V)
