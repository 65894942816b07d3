"""-m gpu: view sharding (BASELINE configs[2]) on the real engine.  One GPU box, so the ranks are emulated: `world`
DrMvsnet engines in one process, each uploaded with its rank's sub-window (tandem_amd/view_shard.upload), and an
all-reduce that sums the engines' partial volumes on the device's behalf (D2H, fp32 sum in rank order, H2D to every
engine) exactly where RCCL's all_reduce sits in the multi-process path.  The sharded result must equal the
unsharded engine's to fp32 summation order, every rank must end with the same depth map, and a rank without source
views must contribute nothing."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
DEPTH_TOL = 2e-4   # metres, max |depth_dense difference| sharded vs unsharded (summation order of <= 6 views)


def host_allreduce(models):
    from tandem_amd import _lib

    def reduce_all(name):
        parts = []
        for m in models:
            ptr, n = m.device_tensor(name)
            a = np.empty(n, np.float32)
            _lib.check(_lib.lib().dr_memcpy_d2h(a.ctypes.data_as(C.c_void_p), ptr, n * 4))
            parts.append(a)
        total = parts[0].copy()
        for a in parts[1:]:
            total += a
        for m in models:
            ptr, n = m.device_tensor(name)
            _lib.check(_lib.lib().dr_memcpy_h2d(ptr, total.ctypes.data_as(C.c_void_p), n * 4))
        return parts, total
    return reduce_all


def run_sharded(blob, window, world):
    from tandem_amd import view_shard
    from tandem_amd.dr_mvsnet import DrMvsnet
    models = [DrMvsnet(blob) for _ in range(world)]
    subs = [view_shard.upload(m, window, r, world) for r, m in enumerate(models)]
    reduce_all = host_allreduce(models)
    parts_log = {}
    # TWO forwards on the same engines: after the first one every volume holds the reduced sum of all ranks, so a rank whose
    # partial volume is the empty sum (reference view only) must actively zero it -- a fresh allocation hides that
    for _ in range(2):
        for p in range(3):
            for m in models:
                m.forward_phase(p)
            parts_log[p + 1] = reduce_all("volume%d" % (p + 1))
        for m in models:
            m.forward_phase(3)
    outs = [m.download() for m in models]
    for m in models:
        m.close()
    return outs, subs, parts_log


@pytest.mark.parametrize("H,W,V,world", [(64, 96, 7, 2), (64, 96, 7, 3), (96, 128, 4, 8), (64, 96, 7, 1)])
def test_sharded_equals_unsharded(H, W, V, world):
    import os
    from synth import scene
    from tandem_amd.dr_mvsnet import DrMvsnet
    blob = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "weights", "tandem_va.tdmw")
    win = scene.make_window(H, W, V, seed=V + world)
    window = dict(bgrs=win["bgrs"], K=win["K"], c2ws=list(win["c2ws"]), ref_index=win["ref_index"],
                  depth_min=0.5, depth_max=5.0, discard=2.5)
    full = DrMvsnet(blob)
    full.CallAsync(H, W, V, win["ref_index"], win["bgrs"], win["K"], list(win["c2ws"]), 0.5, 5.0, 2.5)
    ref = full.GetResult()
    full.close()
    outs, subs, parts = run_sharded(blob, window, world)
    # partition: every source view on exactly one rank, the reference on all
    assert all(s[0] == win["ref_index"] for s in subs)
    assert sorted(i for s in subs for i in s[1:]) == [i for i in range(V) if i != win["ref_index"]]
    for r, s in enumerate(subs):
        if len(s) == 1:  # a rank with no source views adds a zero volume
            assert all(not parts[k][0][r].any() for k in parts)
    for o in outs:
        assert np.abs(o.depth_dense - ref.depth_dense).max() < DEPTH_TOL
        assert np.abs(o.confidence_dense - ref.confidence_dense).mean() < 1e-4
        assert ((o.depth == 0) != (ref.depth == 0)).mean() < 2e-3
    for o in outs[1:]:  # all ranks regularise the same reduced volume: identical results, no broadcast needed
        assert np.array_equal(o.depth_dense.view(np.uint32), outs[0].depth_dense.view(np.uint32))
        assert np.array_equal(o.depth.view(np.uint32), outs[0].depth.view(np.uint32))


def test_shard_requires_view_aggregation_and_resets():
    import os
    from synth import scene
    from tandem_amd import _lib
    from tandem_amd.dr_mvsnet import DrMvsnet
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    win = scene.make_window(64, 96, 3, seed=1)
    m = DrMvsnet(os.path.join(root, "weights", "tandem_va.tdmw"))
    with pytest.raises(_lib.DrError):  # a one-view window is only legal on a shard rank
        m.upload(64, 96, 1, 0, win["bgrs"][:1], win["K"], list(win["c2ws"])[:1], 0.5, 5.0, 2.5)
    m.set_view_shard(2)
    m.upload(64, 96, 1, 0, win["bgrs"][:1], win["K"], list(win["c2ws"])[:1], 0.5, 5.0, 2.5)
    m.forward_phase(0)
    ptr, n = m.device_tensor("volume1")
    a = np.empty(n, np.float32)
    _lib.check(_lib.lib().dr_memcpy_d2h(a.ctypes.data_as(C.c_void_p), ptr, n * 4))
    assert n == 48 * 16 * 24 * 32 and not a.any()
    with pytest.raises(_lib.DrError):
        m.forward_phase(4)
    m.set_view_shard(0)  # back to the plain path
    m.CallAsync(64, 96, 3, win["ref_index"], win["bgrs"], win["K"], list(win["c2ws"]), 0.5, 5.0, 2.5)
    assert np.isfinite(m.GetResult().depth_dense).all()
    m.close()


def test_two_process_bench_path_on_one_gpu(tmp_path):
    """bench.py's N > 1 code path end to end -- torch.distributed.run, barrier, max-over-ranks clock, replicas leg,
    view-sharded leg with the torch all-reduce between phases -- as two processes sharing cuda:0 over gloo
    (DR_BENCH_ONE_DEVICE=1: RCCL refuses two ranks on one device; the driver's 8-GPU run uses nccl)."""
    import json
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for attempt in range(3):  # the rendezvous port is found by bind-and-release; retry if another process wins it
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-cpu", "--no-tsdf"]
        p = subprocess.run(cmd, env=dict(os.environ, DR_BENCH_ONE_DEVICE="1"), capture_output=True, text=True, timeout=600, cwd=root)
        if p.returncode == 0:
            break
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "rank 0 prints exactly one JSON line"
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["scaling"] == "weak" and d["value"] > 0
    vs = d["view_sharded"]
    assert vs["n_gpus"] == 2 and vs["source_views_total"] == 6 and vs["source_views_this_rank"] == 3
    assert vs["ranks_agree"] is True
    assert abs(vs["allreduce_mb_per_depth_map"] - 4e-6 * (48 * 120 * 160 * 32 + 32 * 240 * 320 * 16 + 8 * 480 * 640 * 8)) < 1e-3


def test_engine_collective_single_rank(trained_blob):
    """drm_comm_init / in-stream ncclAllReduce: with one rank that holds every source view the sharded forward (divisor
    = all source views, all-reduce over a 1-rank communicator after every cost volume, no host step) must reproduce the
    ordinary forward bit for bit; the communicator can be destroyed and re-created."""
    from synth import scene
    from tandem_amd.dr_mvsnet import DrMvsnet
    H, W, V = 64, 96, 5
    win = scene.make_window(H, W, V, seed=8)
    args = (H, W, V, win["ref_index"], win["bgrs"], win["K"], list(win["c2ws"]), win["depth_min"], win["depth_max"], 10.0)
    a = DrMvsnet(trained_blob)
    a.upload(*args)
    a.forward(1)
    ref = a.download()
    b = DrMvsnet(trained_blob)
    for _ in range(2):
        b.comm_init(0, 1, DrMvsnet.comm_unique_id())
        b.set_view_shard(V - 1)
        b.upload(*args)
        b.forward(2)
        out = b.download()
        for x, y in ((ref.depth, out.depth), (ref.confidence, out.confidence), (ref.depth_dense, out.depth_dense)):
            assert np.array_equal(x.view(np.uint32), y.view(np.uint32))
        b.comm_destroy()
    a.close(); b.close()


_RANK_SCRIPT = r'''
import os, sys, time
import numpy as np
rank, world, outdir, blob, H, W, V = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], int(sys.argv[5]), int(sys.argv[6]), int(sys.argv[7])
sys.path.insert(0, sys.argv[8])
from synth import scene
from tandem_amd import view_shard
from tandem_amd.dr_mvsnet import DrMvsnet
win = scene.make_window(H, W, V, seed=11)
window = dict(bgrs=win["bgrs"], K=win["K"], c2ws=list(win["c2ws"]), ref_index=win["ref_index"], depth_min=0.5, depth_max=5.0, discard=2.5)
assert DrMvsnet.comm_available()
m = DrMvsnet(blob)
uidf = os.path.join(outdir, "uid.bin")
if rank == 0:
    open(uidf + ".tmp", "wb").write(DrMvsnet.comm_unique_id())
    os.rename(uidf + ".tmp", uidf)
t0 = time.time()
while not os.path.exists(uidf):
    assert time.time() - t0 < 60
    time.sleep(0.05)
m.comm_init(rank, world, open(uidf, "rb").read())
mine = view_shard.upload(m, window, rank, world)
m.forward(2)  # two depth maps back to back: the collectives of consecutive forwards stay in step
o = m.download()
np.savez(os.path.join(outdir, "rank%d.npz" % rank), depth=o.depth, conf=o.confidence, dd=o.depth_dense, cd=o.confidence_dense, mine=np.array(mine))
m.comm_destroy()
m.close()
'''


@pytest.mark.parametrize("world,allreduce", [(2, False), (3, False), (2, True)])
def test_engine_collective_with_several_ranks_on_one_gpu(world, allreduce, tmp_path, trained_blob):
    """The engine's OWN multi-rank path -- drm_comm_init, then per stage ncclReduce of the cost volume to rank 0, CostRegNet
    and regression on rank 0 only, ncclBroadcast of the stage depth map (stage 3: depth + confidence) back; or, with
    DR_SHARD_ALLREDUCE=1, round 2's in-place all-reduce -- with world > 1.  RCCL refuses two ranks on one device, so the
    ranks are processes sharing cuda:0 and the engine binds tests/cpp/rccl_stub.cpp (shared-memory stand-in, prototypes
    from the real rccl.h) through DR_RCCL_LIB.  Checked: every rank ends with the same four maps (bit for bit) and they
    equal the unsharded engine's to fp32 summation order; two forwards in a row stay in step."""
    import os
    import subprocess
    import sys
    from synth import scene
    from tandem_amd.dr_mvsnet import DrMvsnet
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    stub = str(tmp_path / "librccl_stub.so")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-shared", "-fPIC", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                           os.path.join(root, "tests", "cpp", "rccl_stub.cpp"), "-L/opt/rocm/lib", "-lamdhip64", "-lrt", "-o", stub])
    H, W, V = 64, 96, 7
    script = str(tmp_path / "rank.py")
    open(script, "w").write(_RANK_SCRIPT)
    env = dict(os.environ, DR_RCCL_LIB=stub)
    if allreduce:  # round 2's form lives in the parity build: the rank processes load that library
        from tandem_amd import _lib
        if not os.path.isfile(_lib.HOOKS_LIB_PATH):
            pytest.skip("tandem_amd/libdr_mi355x_hooks.so not built")
        env["DR_SHARD_ALLREDUCE"] = "1"
        env["DR_MI355X_LIB"] = _lib.HOOKS_LIB_PATH
    procs = [subprocess.Popen([sys.executable, script, str(r), str(world), str(tmp_path), trained_blob, str(H), str(W), str(V), root], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(world)]
    for p in procs:
        try:
            _, err = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise AssertionError("a rank hung in the collective path")
        assert p.returncode == 0, err[-3000:]
    win = scene.make_window(H, W, V, seed=11)
    full = DrMvsnet(trained_blob)
    full.CallAsync(H, W, V, win["ref_index"], win["bgrs"], win["K"], list(win["c2ws"]), 0.5, 5.0, 2.5)
    ref = full.GetResult()
    full.close()
    outs = [np.load(str(tmp_path / ("rank%d.npz" % r))) for r in range(world)]
    assert sorted(i for o in outs for i in o["mine"][1:]) == [i for i in range(V) if i != win["ref_index"]]
    for o in outs:
        assert np.abs(o["dd"] - ref.depth_dense).max() < DEPTH_TOL
        assert np.abs(o["cd"] - ref.confidence_dense).mean() < 1e-4
        assert ((o["depth"] == 0) != (ref.depth == 0)).mean() < 2e-3
    for o in outs[1:]:
        for k in ("depth", "conf", "dd", "cd"):
            assert np.array_equal(o[k].view(np.uint32), outs[0][k].view(np.uint32)), k
