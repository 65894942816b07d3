"""CPU rehearsal of the one run this repo cannot make itself: `python bench.py --gpus 8` on an 8-GPU node.

bench.py's OWN spawning path (respawn_under_torchrun -> torch.distributed.run -> 8 ranks) is launched with DR_BENCH_DRY_RUN=1: gloo,
stand-in engines (tests/stubs/dry_engine.py) that record what each rank was asked to do.  Checked: one JSON line with n_gpus 8 and all
K steps counted across the ranks; the view-shard leg runs with 6 of 8 ranks (one per source view, view_shard.shard_world), ranks 6-7
never touch comm_init (ncclCommInitRank) and never run a sharded forward, the communicator counts 6 ranks, the ranks agree."""
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_eight_rank_launch_through_bench_spawn_path(tmp_path):
    env = dict(os.environ, DR_BENCH_DRY_RUN="1", DR_BENCH_DRY_DIR=str(tmp_path), OMP_NUM_THREADS="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "12", "--warmup", "1"], capture_output=True, text=True,
                       timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]  # rank 0 prints ONE line
    out = json.loads(lines[0])
    assert out["dry_run"] is True and out["value"] is None  # a rehearsal never carries a number
    assert out["n_gpus"] == 8 and out["steps"] == 12 and out["scaling"] == "weak"
    assert out["units_counted"] == 8 * 12  # K steps per rank, summed over the ranks by the replicas reduction
    vs = out["view_sharded"]
    assert vs["n_gpus"] == 8 and vs["participants"] == 6 and vs["source_views_total"] == 6
    assert vs["rccl_comm_ranks"] == 6 and vs["ranks_in_communicator"] == 6
    assert vs["ranks_agree"] is True
    assert "6 of 8 ranks take part" in vs["collective"]
    # what every rank's engines were asked to do
    logs = {}
    for f in glob.glob(os.path.join(str(tmp_path), "rank*_*.json")):
        d = json.load(open(f))
        logs.setdefault(d["rank"], []).append(d)
    assert sorted(logs) == list(range(8))
    for rank, engines in logs.items():
        shard = [e for e in engines if e["set_view_shard"] is not None or e["comm_init"] is not None]
        plain = [e for e in engines if e not in shard]
        assert sum(e["forwards"] for e in plain) >= 12  # this rank's K replica steps (plus warm-up / latency probes)
        if rank < 6:
            assert len(shard) == 1 and shard[0]["comm_init"] == [rank, 6] and shard[0]["set_view_shard"] == 6
            assert shard[0]["uploads"][-1]["views"] == 2 and shard[0]["forwards"] > 0  # the reference view + ONE source view
        else:  # idle ranks: no communicator, no sharded window, no forward on the sharded engine
            assert all(e["comm_init"] is None and e["set_view_shard"] is None for e in engines)
            idle = [e for e in engines if e["forwards"] == 0]
            assert len(idle) == 1 and idle[0]["uploads"] == []
