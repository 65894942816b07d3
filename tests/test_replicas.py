"""CPU: the N>1 host path (tandem_amd/replicas.py) with world_size 2 over gloo: round-robin unit sharding,
barrier, max-over-ranks clock and unit sum -- exactly what bench.py does around its timed region."""
import os
import socket
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import json, sys, time
    sys.path.insert(0, %r)
    from tandem_amd import replicas
    rank, local_rank, world = replicas.init("gloo")
    mine = replicas.units_for_rank(11, rank, world)
    replicas.barrier()
    t0 = time.perf_counter()
    time.sleep(0.05 * (rank + 1))          # rank 1 is the slow one
    dt = time.perf_counter() - t0
    replicas.barrier()
    tmax, units = replicas.reduce_max_sum(dt, len(mine))
    print(json.dumps(dict(rank=rank, world=world, mine=mine, dt=dt, tmax=tmax, units=units)))
""") % ROOT


def test_two_rank_gloo_replicas(tmp_path):
    from conftest import launch_gloo_ranks
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    outs = launch_gloo_ranks(script, timeout=120)
    assert outs[0]["mine"] == [0, 2, 4, 6, 8, 10] and outs[1]["mine"] == [1, 3, 5, 7, 9]
    assert outs[0]["units"] == outs[1]["units"] == 11
    assert outs[0]["tmax"] == outs[1]["tmax"] == max(outs[0]["dt"], outs[1]["dt"])
    assert outs[0]["tmax"] >= 0.1


def test_single_process_defaults():
    from tandem_amd import replicas
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        os.environ.pop(k, None)
    assert replicas.env_world() == (0, 0, 1)
    assert replicas.units_for_rank(5, 0, 1) == [0, 1, 2, 3, 4]
    assert replicas.reduce_max_sum(1.5, 7) == (1.5, 7.0)
