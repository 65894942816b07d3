"""Multi-GPU host logic for the DrMvsnet / DrFusion hot path: independent replicas, one process per GPU.

The reference has no inference-time collective; keyframe windows are independent units, so the
path shards as replicas with NO data-path collective: rank r processes windows r, r+N, r+2N, ...
torch.distributed is used only for the launch contract's barrier and the max-over-ranks clock
(backend nccl == RCCL on GPUs, gloo in the CPU tests).  TSDF fusion of ONE map is serial in the voxel state:
"replicas only" (each rank fuses its own map)."""
import os


def env_world():
    """(rank, local_rank, world_size) from the torch.distributed.run environment (defaults: single process)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def units_for_rank(total_units, rank, world):
    """Round-robin shard of independent work units (keyframe windows / maps)."""
    return list(range(rank, total_units, world))


def init(backend=None, device=None):
    import torch.distributed as dist
    rank, local_rank, world = env_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        kw = {}
        if backend == "nccl" and device is not None:
            import torch
            kw["device_id"] = torch.device("cuda", device)
        dist.init_process_group(backend or "gloo", rank=rank, world_size=world, **kw)
    return rank, local_rank, world


def barrier(device=None):
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        if device is not None and dist.get_backend() == "nccl":
            dist.barrier(device_ids=[device])
        else:
            dist.barrier()


def reduce_max_sum(seconds, units, device=None):
    """Returns (max over ranks of `seconds`, sum over ranks of `units`)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return float(seconds), float(units)
    dev = torch.device("cuda", device) if (device is not None and dist.get_backend() == "nccl") else torch.device("cpu")
    t = torch.tensor([float(seconds)], dtype=torch.float64, device=dev)
    u = torch.tensor([float(units)], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(u, op=dist.ReduceOp.SUM)
    return float(t.item()), float(u.item())
