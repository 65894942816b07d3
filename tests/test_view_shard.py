"""CPU: the view-sharded N>1 host path (tandem_amd/view_shard.py) with world_size 2 over gloo.  No GPU here, so the
DrMvsnet engine is replaced by a stand-in with the SAME protocol surface (set_view_shard / upload / forward_phase /
device_tensor / download) whose arithmetic is the CPU oracle's; what is under test is the host logic -- partition,
phase order, one sum all-reduce per stage volume -- and the identity it rests on:
    sum over ranks of [ sum over the rank's views of (gate+1)*(warp-ref)^2 / (V-1) ]  ==  the unsharded volume
(module.py:1097-1108), to fp32 summation order.  tests/test_view_shard_gpu.py runs the real engine."""
import json
import os
import socket
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_partition_covers_every_source_view_once():
    from tandem_amd.view_shard import partition
    for V in range(2, 9):
        for ref in range(V):
            for world in (1, 2, 3, 4, 8):
                parts = [partition(V, ref, r, world) for r in range(world)]
                assert all(p[0] == ref for p in parts)
                src = sorted(i for p in parts for i in p[1:])
                assert src == [i for i in range(V) if i != ref]
                assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
    # model order is [ref, others in original order] (dr_mvsnet.cpp:190-197); round-robin over that order
    assert partition(7, 5, 0, 2) == [5, 0, 2, 4] and partition(7, 5, 1, 2) == [5, 1, 3, 6]
    assert partition(7, 5, 7, 8) == [5]  # more ranks than source views: reference only


WORKER = textwrap.dedent("""
    import json, sys
    sys.path.insert(0, %r)
    import numpy as np, torch
    import torch.distributed as dist
    torch.set_num_threads(4)
    from oracle import mvsnet_oracle as O
    from synth import scene
    from tandem_amd import replicas, view_shard, weights as Wt

    class OracleShardModel:  # protocol stand-in for tandem_amd.dr_mvsnet.DrMvsnet, arithmetic = oracle
        def __init__(self, w): self.w, self.nsrc, self.t, self.calls = w, 0, {}, []
        def set_view_shard(self, n): self.nsrc = n
        def upload(self, H, W, V, ref, bgrs, K, c2ws, dmin, dmax, disc):
            self.win = (bgrs, K, c2ws, ref, dmin, dmax, disc); self.H, self.W = H, W
        def _partial_volume(self, s):
            feats, planes = self.feats[s - 1], self.planes
            D = planes.shape[0]
            ref = feats[0].unsqueeze(1).expand(-1, D, -1, -1)[None]
            acc = torch.zeros_like(ref)
            for v in range(1, feats.shape[0]):
                d2 = (O.warp(feats[v], planes, self.Ks[s - 1], self.c2w[0], self.c2w[v])[None] - ref).pow_(2)
                acc += (O.gate(d2, self.w, "volume_gates.stage%%d." %% s) + 1) * d2
            self.t["volume%%d" %% s] = acc.div_(self.nsrc)[0].contiguous()
        def forward_phase(self, p):
            self.calls.append(p)
            meta = self.w.meta
            with torch.no_grad():
                if p == 0:
                    bgrs, K, c2ws, ref, dmin, dmax, disc = self.win
                    image, self.Ks, self.c2w = O.preprocess(bgrs, K, c2ws, ref)
                    self.feats = O.feature_net(image, self.w)
                    self.planes, self.base = O.uniform_planes(dmin, dmax, meta["depth_num"][0], self.H // 4, self.W // 4)
                    self._partial_volume(1)
                    return
                s = p
                logits = O.cost_reg(self.t["volume%%d" %% s], self.w, s)
                self.depth, self.conf = O.regress(logits, self.planes)
                if s < 3:
                    sc = 2 ** (3 - (s + 1))
                    self.planes = O.adaptive_planes(self.depth, meta["depth_num"][s], meta["interval_ratio"][s] * self.base,
                                                    self.H // sc, self.W // sc)
                    self._partial_volume(s + 1)
        def device_tensor(self, name): return name, self.t[name].numel()
        def download(self):
            d, mask, _, _ = O.filter_edges(self.depth, self.win[6])
            return dict(depth=d.numpy(), depth_dense=self.depth.numpy(), confidence_dense=self.conf.numpy())

    rank, local_rank, world = replicas.init("gloo")
    meta, tens = Wt.read_blob(%r)
    w = O.Weights(meta, tens)
    win = scene.make_window(64, 96, 5, seed=4)
    window = dict(bgrs=win["bgrs"], K=win["K"], c2ws=list(win["c2ws"]), ref_index=win["ref_index"],
                  depth_min=0.5, depth_max=5.0, discard=2.5)
    m = OracleShardModel(w)
    reduced = []
    def allreduce(name, n):
        assert m.t[name].numel() == n
        dist.all_reduce(m.t[name], op=dist.ReduceOp.SUM)
        reduced.append(name)
    out = view_shard.run(m, window, rank, world, allreduce)
    full = O.forward(w, win["bgrs"], win["K"], win["c2ws"], win["ref_index"], 0.5, 5.0, 2.5)
    err = float(np.abs(out["depth_dense"] - full["depth_dense"]).max())
    flips = float(((out["depth"] == 0) != (full["depth"] == 0)).mean())
    print(json.dumps(dict(rank=rank, mine=view_shard.partition(5, win["ref_index"], rank, world), calls=m.calls, reduced=reduced,
                          nsrc=m.nsrc, err=err, flips=flips, checksum=float(out["depth_dense"].astype(np.float64).sum()))))
""") % (ROOT, os.path.join(ROOT, "weights", "tandem_va.tdmw"))


def test_two_rank_gloo_view_shard(tmp_path):
    from conftest import launch_gloo_ranks
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    outs = launch_gloo_ranks(script, timeout=600)
    a, b = outs
    assert a["mine"][0] == b["mine"][0] and sorted(a["mine"][1:] + b["mine"][1:]) == [i for i in range(5) if i != a["mine"][0]]
    for o in outs:
        assert o["calls"] == [0, 1, 2, 3] and o["reduced"] == ["volume1", "volume2", "volume3"] and o["nsrc"] == 4
        # tolerance: fp32 summation order of 4 views split 2 + 2 -- same bar as the HIP path vs the oracle
        assert o["err"] < 1e-4 and o["flips"] < 1e-3, o  # measured: 3.6e-6 m, 0 flips
    assert a["checksum"] == b["checksum"]  # every rank ends with the same depth map (no broadcast needed)


def test_engine_collective_falls_back_when_rccl_cannot_be_bound():
    """init_engine_collective must not raise -- and must not enter ncclCommInitRank on any rank -- when some rank cannot bind
    RCCL: every rank probes (drm_comm_available), the ranks agree on the minimum, it reports False on every rank alike and
    bench.py's sharded leg then uses the host-driven phases."""
    from tandem_amd import view_shard

    class NoRccl:
        @staticmethod
        def comm_available():
            return False

        @staticmethod
        def comm_unique_id():
            raise AssertionError("must not be reached when RCCL cannot be bound")

        def comm_init(self, rank, world, uid):
            raise AssertionError("must not be reached without an id")

    class WithRccl:
        def __init__(self):
            self.got = None

        @staticmethod
        def comm_available():
            return True

        @staticmethod
        def comm_unique_id():
            return b"\x01" * 128

        def comm_init(self, rank, world, uid):
            self.got = (rank, world, uid)

    assert view_shard.init_engine_collective(NoRccl(), 0, 1) is False
    m = WithRccl()
    assert view_shard.init_engine_collective(m, 0, 1) is True and m.got == (0, 1, b"\x01" * 128)
    assert view_shard.shard_world(7, 8) == 6 and view_shard.shard_world(7, 2) == 2 and view_shard.shard_world(2, 8) == 1
