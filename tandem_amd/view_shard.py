"""View-sharded keyframe-window inference (BASELINE configs[2]): the source views of ONE window are
spread over the GPUs and the partial cost volumes are sum-reduced (RCCL over xGMI) once per cascade stage.

Where the exchange is: the reference builds the stage volume as  sum_v (gate_v + 1) * (warp_v - ref)^2 / (V - 1)
(module.py:1097-1108) -- a plain sum over source views, the only cross-view coupling of the whole network (the
FeatureNet runs per view, everything after the volume has no view axis).  So rank r
  * runs FeatureNet on [reference view] + its own source views only,
  * builds the partial volume of its views with the WHOLE window's divisor (drm_set_view_shard),
  * in-engine collective (drm_comm_*, the form bench.py measures): the fp32 volume (118 / 157 / 79 MB for stages 1..3 at
    640x480, (48,32,8)) is REDUCED to rank 0, which alone runs the (view-free) regularisation and regression and
    BROADCASTS the stage's depth map (77 / 307 / 1229 KB; stage 3: depth + confidence) back -- the next stage's planes
    hang on it; every rank then runs the edge filter and ends with the same four maps;
  * host-driven protocol (drm_forward_phase + any all-reduce; the gloo / one-GPU test double): the volume is sum
    all-reduced in place and every rank regularises redundantly, so no broadcast is needed.
At most one rank per source view takes part (shard_world): further ranks would only add zero volumes to the reduce.

This is NOT the throughput configuration: by SURVEY 5.8/8e the reduce alone (>= 2.3 ms per depth map over xGMI)
exceeds the whole single-GPU pipeline (bench.py's replicas mode is the headline); it exists for windows whose
feature maps / volumes exceed one GPU, and is measured by bench.py's `view_sharded` object when N > 1.
The result equals the unsharded one up to fp32 summation order (tests/test_view_shard*.py state the tolerance).
"""
import numpy as np


def partition(view_num, ref_index, rank, world):
    """Original view indices of rank's sub-window: [ref] + its source views (round-robin over the source views in
    the model's order [ref, others in original order], dr_mvsnet.cpp:190-197)."""
    if not (0 <= ref_index < view_num):
        raise ValueError("ref_index out of range")
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    sources = [i for i in range(view_num) if i != ref_index]
    return [ref_index] + sources[rank::world]


class TorchAllReduce:
    """Sum-reduce a device buffer over the ranks of the default torch.distributed group (nccl == RCCL) through a
    torch staging tensor: D2D in, all_reduce, D2D out (two extra HBM passes of <= 157 MB, ~0.1 ms, against a
    multi-millisecond xGMI reduce)."""

    def __init__(self, device, max_floats):
        import torch
        self.torch = torch
        self.buf = torch.empty(max_floats, dtype=torch.float32, device=torch.device("cuda", device))
        self.bytes = 0

    def __call__(self, dptr, nfloats):
        import torch.distributed as dist
        from . import _lib
        t = self.buf[:nfloats]
        self.torch.cuda.synchronize(self.buf.device)
        _lib.check(_lib.lib().dr_memcpy_d2d(t.data_ptr(), dptr, nfloats * 4))
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        self.torch.cuda.synchronize(self.buf.device)
        _lib.check(_lib.lib().dr_memcpy_d2d(dptr, t.data_ptr(), nfloats * 4))
        self.bytes += nfloats * 4


def upload(model, window, rank, world):
    """Upload rank's sub-window of `window` (dict: bgrs, K, c2ws, ref_index, depth_min, depth_max, discard; H, W
    from the images) with the view-shard divisor set."""
    bgrs, c2ws = window["bgrs"], window["c2ws"]
    V, ref = len(bgrs), window["ref_index"]
    H, W = np.asarray(bgrs[0]).shape[:2]
    mine = partition(V, ref, rank, world)
    model.set_view_shard(V - 1)
    model.upload(H, W, len(mine), 0, [bgrs[i] for i in mine], window["K"], [c2ws[i] for i in mine],
                 window["depth_min"], window["depth_max"], window["discard"])
    return mine


def forward(model, allreduce):
    """One depth map of the uploaded (sharded) window; `allreduce(device_ptr, nfloats)` sums in place over ranks."""
    for p in range(3):
        model.forward_phase(p)
        ptr, n = model.device_tensor("volume%d" % (p + 1))
        allreduce(ptr, n)
    model.forward_phase(3)


def shard_world(view_num, world):
    """Ranks that take part in a sharded window: at most one per source view (a rank without source views would only add a
    zero volume to the reduce; with 8 GPUs and 6 source views two ranks stay out)."""
    return max(1, min(world, view_num - 1))


def init_engine_collective(model, rank, world, participants=None):
    """Give `model` an RCCL communicator of its own (drm_comm_init) over the first `participants` ranks (default: all):
    every rank probes whether it can bind RCCL and the ranks agree (MIN over the group) BEFORE anyone enters
    ncclCommInitRank -- a rank that cannot must not leave the others blocked in it.  Rank 0 draws the id, torch.distributed
    carries the 128 bytes.  Afterwards `model.forward(n)` / CallAsync run a sharded window without any host step.
    Returns False (on every rank alike) when some rank could not bind RCCL; the caller then uses the host-driven phases.
    Ranks >= participants return True without a communicator: they take no part in the window."""
    import sys
    import torch
    import torch.distributed as dist
    participants = world if participants is None else participants
    ok = 1 if type(model).comm_available() else 0
    if world > 1:
        dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
        t = torch.tensor([ok], dtype=torch.int32, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        ok = int(t.item())
    if not ok:
        if rank == 0:
            print("view_shard: engine collective unavailable on some rank; using torch.distributed", file=sys.stderr)
        return False
    uid = [type(model).comm_unique_id() if rank == 0 else None]
    if world > 1:
        dist.broadcast_object_list(uid, src=0)
    if rank < participants:
        model.comm_init(rank, participants, uid[0])
    return True


def run(model, window, rank, world, allreduce):
    upload(model, window, rank, world)
    forward(model, allreduce)
    return model.download()
