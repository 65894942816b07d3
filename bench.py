#!/usr/bin/env python
"""bench.py -- the hot path's headline metric on MI355X (BASELINE.json):
    "depth-maps/sec @ 640x480x7-view x 3-stage; TSDF voxels integrated/sec".

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one pass of the DrMvsnet hot path (pre-process .. edge filter) over one synthetic keyframe window
that is already resident in HBM when the timed region starts.  Workload at every N = BASELINE configs[1]:
640x480, ref + 6 src views, 3-stage cascade with (48,32,8) depth planes, fp32, view aggregation on.
Multi-GPU: independent replicas, one window stream per rank, no data-path collective ("scaling": "weak");
value = depth maps of all ranks / max-over-ranks time.  Rank 0 prints ONE JSON line.  Riding along in it:
  "boundary"  the same windows through the operator boundary the reference times (CallAsync / Ready / GetResult,
              dr_mvsnet.cpp:540-545): host u8 images in, four float maps out -- PCIe-inclusive, never `value`;
  "tsdf"      BASELINE configs[3]: 1000 distinct depth maps of a camera loop through an analytic room fused into a 5 mm
              hashed voxel grid, integrate + ray-cast per frame, map growing (dr_debug_example.cpp:78-162);
  "shipped_model"  the 320x512 (48,4,4) model TANDEM ships (published 4.96 FPS), latency and 3-engine throughput;
  "tandem_loop"  BASELINE configs[4] stand-in: TandemBackend's call order (depth network of keyframe k beside fusion +
              ray-cast of k-1) through the C++ shim on one GPU, keyframes/s;
  "tracker", "view_sharded" (N > 1).
`python bench.py --gpus N` without a torch.distributed.run environment spawns the N ranks itself.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

def source_stamp():
    """sha1 over the HIP sources the library is built from: the stamp a PMC profile carries (tools/pmc_to_json.py) and this
    run compares, so that counters of another tree are never reported as this run's."""
    import glob
    import hashlib
    h = hashlib.sha1()
    for f in sorted(glob.glob(os.path.join(ROOT, "tandem_amd", "csrc", "*.h")) + glob.glob(os.path.join(ROOT, "tandem_amd", "csrc", "*.hip"))):
        h.update(os.path.basename(f).encode()); h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic(kernel, fetch="corrected"):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 --pmc passes (profiles/r*_pmc_traffic.json, written by
    tools/gpu_r5.sh pmc + tools/pmc_to_json.py on the same workload; counters cannot be read from inside this process).
    FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for gfx950.  None when there is no profile, when the profile
    does not carry THIS tree's source stamp (a stale file is refused, not reported), or when it lacks the kernel."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
    if not files:
        return None
    prof = json.load(open(files[-1]))
    if prof.get("_meta", {}).get("source_stamp") != source_stamp():
        return None
    e = prof.get(kernel)
    if e is None:  # template instances are stored under their full name (k_raycast2<true,false,1>): a bare kernel name means its only instance
        hits = [k for k in prof if k.startswith(kernel + "<")]
        e = prof[hits[0]] if len(hits) == 1 else None
    if not e or "fetch_bytes_corrected" not in e or "write_bytes" not in e:
        return None
    # fetch="raw": kernels whose loads are 8 bytes per lane (k_integrate's voxel reads).  The x2 of MI355X_MICROARCH.md is calibrated on 16-byte-per-lane
    # streams ("other access widths ... uncalibrated"); k_integrate's own calibration is exact -- it reads 4 KB per visited block, and its RAW FETCH_SIZE
    # equals that count (tsdf.integrate_traffic_model)
    return e["fetch_bytes_" + fetch] + e["write_bytes"]


def pmc_profile_state():
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
    if not files:
        return "no PMC profile committed"
    st = json.load(open(files[-1])).get("_meta", {}).get("source_stamp")
    return "%s: %s" % (os.path.basename(files[-1]), "matches this tree's sources" if st == source_stamp() else "STALE (source stamp %s, tree %s): traffic reported as null" % (st, source_stamp()))


def host_cores():
    """(physical cores, logical CPUs) of this box, from /proc/cpuinfo."""
    phys, logical = set(), 0
    try:
        pid = cid = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("processor"):
                logical += 1
            elif line.startswith("physical id"):
                pid = line.split(":")[1].strip()
            elif line.startswith("core id"):
                cid = line.split(":")[1].strip()
                phys.add((pid, cid))
    except OSError:
        pass
    return (len(phys) or os.cpu_count() or 1), (logical or os.cpu_count() or 1)


PEAK_FP32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md: dense fp32 MFMA (== fp32 vector) peak
PEAK_HBM_GBPS = 8000.0          # HBM3E spec peak (6.3 TB/s measured achievable)
H, W, V = 480, 640, 7
PLANES = (48, 32, 8)
DISCARD = 10.0                  # TANDEM's mvsnet_discard_percentage default (settings.cpp:300)
# the headline window is run at the depth range eval.py / export_model.py pass (cva_mvsnet/eval.py:31-32);
# the scene's own range (synth/scene.py: 0.5 .. 5.0, what the parity tests use) rides along as `scene_depth_range`.
DEPTH_MIN, DEPTH_MAX = 0.01, 10.0


# DR_BENCH_DRY_RUN=1 (tests/test_bench_launch.py): the launcher, the rank plumbing, the replicas reduction, the view-shard leg's participant
# logic and the JSON assembly run on a CPU box over gloo with a stand-in engine that does nothing (tests/stubs/dry_engine.py).  A dry run
# prints "value": null and "dry_run": true -- it measures nothing and is never a result.
DRY = os.environ.get("DR_BENCH_DRY_RUN") == "1"


def engine_class():
    if DRY:
        sys.path.insert(0, os.path.join(ROOT, "tests", "stubs"))
        from dry_engine import DryMvsnet
        return DryMvsnet
    from tandem_amd.dr_mvsnet import DrMvsnet
    return DrMvsnet


def make_window(h, w, v, seed):
    if DRY:
        sys.path.insert(0, os.path.join(ROOT, "tests", "stubs"))
        import dry_engine
        return dry_engine.make_window(h, w, v, seed)
    from synth import scene  # synthetic input generator (test infrastructure, not the measured path)
    return scene.make_window(h, w, v, seed=seed)


def gpu_sync():
    if not DRY:
        import torch
        torch.cuda.synchronize()


def gpu_state():
    """sclk / mclk / power / temperature of GPU 0 as rocm-smi reports them right now (boxes of the pool differ by up to 20 %: the line says
    what state the card was in around the timed region).  None where rocm-smi is missing."""
    import subprocess
    try:
        r = subprocess.run(["rocm-smi", "-d", "0", "--showclocks", "--showpower", "--showtemp", "--json"], capture_output=True, text=True, timeout=20)
        card = next(iter(json.loads(r.stdout).values()))
        keep = {}
        for k, v in card.items():
            kl = k.lower()
            if "sclk" in kl or "mclk" in kl or "power" in kl or ("temperature" in kl and "edge" in kl) or "fclk" in kl:
                keep[k] = v
        return keep
    except Exception as e:  # noqa: BLE001 -- reporting only
        return dict(error=str(e)[:120])


_BLOB = {}


def model_blob():
    """Weights blob of the configured model: the committed trained weights, with the hypothesis counts of --config."""
    if PLANES not in _BLOB:
        blob = os.path.join(ROOT, "weights", "tandem_va.tdmw")
        if PLANES != (48, 32, 8):  # --config shipped: same weights, the shipped model's hypothesis counts
            import tempfile
            from tandem_amd import weights as Wt
            _, tens = Wt.read_blob(blob)
            blob = os.path.join(tempfile.mkdtemp(), "w_%d_%d_%d.tdmw" % PLANES)
            Wt.write_blob(blob, tens, depth_num=PLANES)
        _BLOB[PLANES] = blob
    return _BLOB[PLANES]


def mvsnet_leg(args, rank, dev, world):
    from tandem_amd import replicas
    import threading
    DrMvsnet = engine_class()
    blob = model_blob()
    # E independent DrMvsnet engines per GPU (each its own stream, worker and keyframe window): the launches of
    # different windows overlap, which fills the CUs during the many small kernels of the coarse UNet levels
    # (measured: 275 -> 338 -> 366 depth-maps/s for E = 1, 2, 3).  E = 1 is the single-window latency configuration.
    E = max(1, min(args.engines, args.steps))
    engines, wins = [], []
    for e in range(E):
        win = make_window(H, W, V, rank * 16 + e)
        m = DrMvsnet(blob, device=dev)
        m.upload(H, W, V, win["ref_index"], win["bgrs"], win["K"], list(win["c2ws"]), DEPTH_MIN, DEPTH_MAX, DISCARD)
        if args.warmup > 0:
            m.forward(args.warmup)
        engines.append(m); wins.append(win)
    m, win = engines[0], wins[0]
    share = [args.steps // E + (1 if e < args.steps % E else 0) for e in range(E)]  # exactly K steps in total
    ev = [0.0] * E

    def run(e):
        ev[e] = engines[e].forward(share[e])  # enqueues share[e] forwards on the engine's stream, then stream-synchronises

    # The timed region -- EXACTLY K steps between barrier + synchronize on both sides, max over ranks -- is repeated and the
    # MEDIAN repeat is reported (a 20-step region is 50 ms: one shot of it is at the mercy of a clock ramp or a stray
    # interrupt); min / max of the repeats ride along.  9 repeats at the driver's K = 20, 3 at the default K = 300.
    repeats = max(3, min(9, -(-400 // max(1, args.steps))))
    spans = []
    state_before = gpu_state() if rank == 0 and not DRY else None
    for _ in range(repeats):
        threads = [threading.Thread(target=run, args=(e,)) for e in range(E)]
        replicas.barrier(dev)
        gpu_sync()
        t0 = time.perf_counter()
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        gpu_sync()
        t1 = time.perf_counter()
        replicas.barrier(dev)
        tmax, units = replicas.reduce_max_sum(t1 - t0, args.steps, dev)
        spans.append((tmax, units, max(ev)))
    tmax, units, evmax = sorted(spans)[len(spans) // 2]
    state_after = gpu_state() if rank == 0 and not DRY else None
    res = dict(value=units / tmax, units=units, ms_per_step=1e3 * tmax / args.steps, event_ms_per_step=evmax / args.steps, engines_per_gpu=E,  # engines run concurrently: the slowest one's hipEvent span covers the job
               repeats=dict(n=repeats, statistic="median", ms_per_step_min=1e3 * min(s[0] for s in spans) / args.steps,
                            ms_per_step_max=1e3 * max(s[0] for s in spans) / args.steps))
    if state_before is not None:
        res["gpu_state"] = dict(before_timed_region=state_before, after_timed_region=state_after, source="rocm-smi -d 0 --showclocks --showpower --showtemp")
    if rank == 0:  # the single-window latency (ONE DrMvsnet object = TANDEM's usage) next to the throughput figure
        nlat = max(10, min(100, args.steps))
        lat = m.forward(nlat) / nlat
        res["single_engine"] = dict(ms_per_depth_map=lat, depth_maps_per_s=1e3 / lat, forwards=nlat)
    for extra in engines[1:]:
        extra.close()
    if rank == 0:
        flops, nbytes = m.work()
        prof = m.profile()  # hipEvents around every launch of one forward, on the engine's own stream
        by = {}
        for r in prof:
            k = by.setdefault(r["kernel"], dict(ms=0.0, flops=0.0, bytes=0.0, n=0))
            k["ms"] += r["ms"]; k["flops"] += r["flops"]; k["bytes"] += r["bytes"]; k["n"] += 1
        name, dom = max(by.items(), key=lambda kv: kv[1]["ms"])
        ach = dom["flops"] / (dom["ms"] * 1e-3) / 1e12
        res["roofline"] = dict(bound="mfma", kernel=name, launches_per_step=dom["n"],
                               avg_launch_ms=dom["ms"] / dom["n"], flops_per_launch=dom["flops"] / dom["n"],
                               achieved=ach, peak=PEAK_FP32_MFMA_TFLOPS, unit="TFLOP/s", frac=ach / PEAK_FP32_MFMA_TFLOPS,
                               traffic=pmc_traffic(name),
                               # the next instances by time in the same forward (same definitions), so that one line shows where the
                               # pipeline stands, not only its single largest entry
                               top=[dict(kernel=k, launches=v["n"], ms=round(v["ms"], 4), frac=round(v["flops"] / (v["ms"] * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4))
                                    for k, v in sorted(by.items(), key=lambda kv: -kv[1]["ms"])[:8] if v["flops"] > 0])
        step_s = res["ms_per_step"] * 1e-3
        res["pipeline"] = dict(gflop_per_depth_map=flops / 1e9, gb_per_depth_map=nbytes / 1e9,
                               tflops=flops / step_s / 1e12, frac_mfma=flops / step_s / 1e12 / PEAK_FP32_MFMA_TFLOPS,
                               gbps=nbytes / step_s / 1e9, frac_hbm=nbytes / step_s / 1e9 / PEAK_HBM_GBPS,
                               kernels={k: round(v["ms"], 4) for k, v in sorted(by.items(), key=lambda kv: -kv[1]["ms"])},
                               # every launch of one sequential forward against BOTH roofs: algorithmic flops / 157.3 TFLOP/s and
                               # algorithmic bytes (the engine's layer-boundary model: inputs + outputs + weights of the launch) / 8 TB/s over its hipEvent span
                               launches=[dict(op=r["op"], kernel=r["kernel"], ms=round(r["ms"], 4),
                                              mfma_frac=round(r["flops"] / (r["ms"] * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 3) if r["ms"] > 0 else None,
                                              hbm_frac=round(r["bytes"] / (r["ms"] * 1e-3) / 1e9 / PEAK_HBM_GBPS, 3) if r["ms"] > 0 else None) for r in prof])
        res["roofline"]["traffic_source"] = pmc_profile_state()
        # the same window at the scene's own depth range (synth/scene.py: 0.5 .. 5.0 m, what the parity tests run): the FLOPs do
        # not change with the range, the gather footprint of the cost-volume kernels does
        m.upload(H, W, V, win["ref_index"], win["bgrs"], win["K"], list(win["c2ws"]), win["depth_min"], win["depth_max"], DISCARD)
        m.forward(5)
        nalt = max(10, min(60, args.steps))
        alt = m.forward(nalt) / nalt
        res["scene_depth_range"] = dict(depth_min=win["depth_min"], depth_max=win["depth_max"], single_window_ms=alt, forwards=nalt)
        if world == 1 and not args.no_cpu:
            res["cpu_baseline"] = mvsnet_cpu_baseline(win, blob)
    m.close()
    return res


def golden_parity(dev):
    """Self-check printed in the line: the committed headline fixture -- the REFERENCE model's own outputs at 640x480x7,
    planes (48,32,8), depth range 0.01 .. 10 (tests/golden/mvsnet_v7_480x640_headline.npz, written by oracle/gen_golden.py from the imported
    reference) -- through CallAsync / GetResult of the library that is about to be timed, with the error figures tests/test_mvsnet_gpu.py::compare
    bounds (mean 1e-4 m, max 5e-3 m, mask flips 2e-3).  A data file is read; nothing under oracle/ runs."""
    from tandem_amd.dr_mvsnet import DrMvsnet
    path = os.path.join(ROOT, "tests", "golden", "mvsnet_v7_480x640_headline.npz")
    if not os.path.isfile(path) or PLANES != (48, 32, 8):
        return None
    g = np.load(path)
    bgrs = [np.ascontiguousarray(b) for b in g["bgrs"]]
    m = DrMvsnet(model_blob(), device=dev)
    m.CallAsync(bgrs[0].shape[0], bgrs[0].shape[1], len(bgrs), int(g["ref_index"]), bgrs, g["K"], list(g["c2ws"]), float(g["depth_min"]), float(g["depth_max"]), float(g["discard"]))
    out = m.GetResult()
    m.close()
    err = np.abs(out.depth_dense - g["ref_s3_depth_dense"])
    cerr = np.abs(out.confidence_dense - g["ref_s3_confidence_dense"])
    flips = float(((out.depth == 0) != (g["ref_s3_depth"] == 0)).mean())
    ok = bool(err.mean() < 1e-4 and err.max() < 5e-3 and flips < 2e-3 and cerr.mean() < 1e-4)
    return dict(fixture="tests/golden/mvsnet_v7_480x640_headline.npz (outputs of the reference's own CvaMVSNet, imported from /root/reference by oracle/gen_golden.py)",
                mean_abs_err=float(err.mean()), max_err=float(err.max()), unit="m", confidence_mean_abs_err=float(cerr.mean()), mask_flips=flips,
                bounds=dict(mean_abs_err=1e-4, max_err=5e-3, mask_flips=2e-3), within_bounds=ok)


def mvsnet_cpu_baseline(win, blob):
    """The reference's CPU path on the host cores of this box, same window, model forward only (eval.py's FPS also
    counts its DataLoader and metrics, which cannot run without the dataset).  kind "reference": the reference's own
    CvaMVSNet imported from /root/reference (oracle/ref_model.py) when that checkout exists (the build container);
    kind "port": oracle/mvsnet_oracle.py -- the same ATen CPU ops, bit-identical to the reference model on the committed
    fixtures -- where it does not (the GPU box)."""
    import torch
    from tandem_amd import weights as Wt
    meta, tens = Wt.read_blob(blob)
    phys, logical = host_cores()
    torch.set_num_threads(phys)
    kind, run = "port", None
    if os.path.isdir("/root/reference/cva_mvsnet"):
        try:
            from oracle import mvsnet_oracle as O, ref_model
            net, cva = ref_model.build(PLANES, tens, view_aggregation=True)
            image, Ks, c2w = O.preprocess(win["bgrs"], win["K"], win["c2ws"], win["ref_index"])
            run = lambda: ref_model.run(net, cva, image, Ks, c2w, DEPTH_MIN, DEPTH_MAX, DISCARD)
            kind = "reference"
        except Exception as e:  # fall back to the port, say why
            print("bench.py: reference model not usable (%s); timing the port" % e, file=sys.stderr)
    if run is None:
        from oracle import mvsnet_oracle as O
        w = O.Weights(meta, tens)
        run = lambda: O.forward(w, win["bgrs"], win["K"], win["c2ws"], win["ref_index"], DEPTH_MIN, DEPTH_MAX, DISCARD)
    pinned_by = ("tests/test_oracle_mvsnet.py::test_oracle_reproduces_reference_bit_exactly (the port is bit-identical to the imported reference model on the eight committed "
                 "fixtures of tests/golden/, this window's shape and depth range among them: mvsnet_v7_480x640_headline.npz)")
    run(); run()  # two warm-ups
    times = []
    while len(times) < 5 and sum(times) < 90.0:  # >= 5 timed forwards (about 11 s each on the GPU box's host)
        t0 = time.perf_counter(); run(); times.append(time.perf_counter() - t0)
    med = sorted(times)[len(times) // 2]
    return dict(value=1.0 / med, unit="depth-maps/s", cores=torch.get_num_threads(), physical_cores=phys, logical_cpus=logical, kind=kind, pinned_by=pinned_by,
                sample="%d timed forwards of the same %dx%dx7-view (%d,%d,%d) window (depth range %g .. %g) after 2 warm-ups, torch CPU fp32, %d threads; median %.2f s, best %.2f s"
                       % ((len(times), W, H) + PLANES + (DEPTH_MIN, DEPTH_MAX, torch.get_num_threads(), med, min(times))))


def shipped_leg(args, dev):
    """The model TANDEM actually ships and runs (tandem/exported/tandem_512x320: 320 x 512, 7 views, planes (48,4,4) =
    configs/abl04_fewer_depth_planes.yaml; published: 4.96 FPS on an unstated GPU incl. data loading, pretrained/ablation/
    abl04_fewer_depth_planes.txt:5), same weights, resident window: single-engine latency and 3-engine throughput."""
    import tempfile
    import threading
    from synth import scene
    from tandem_amd import weights as Wt
    from tandem_amd.dr_mvsnet import DrMvsnet
    h, w, planes = 320, 512, (48, 4, 4)
    _, tens = Wt.read_blob(os.path.join(ROOT, "weights", "tandem_va.tdmw"))
    with tempfile.TemporaryDirectory() as td:
        blob = os.path.join(td, "shipped.tdmw")
        Wt.write_blob(blob, tens, depth_num=planes)
        engines = []
        for e in range(3):
            win = scene.make_window(h, w, V, seed=60 + e)
            m = DrMvsnet(blob, device=dev)
            m.upload(h, w, V, win["ref_index"], win["bgrs"], win["K"], list(win["c2ws"]), DEPTH_MIN, DEPTH_MAX, DISCARD)
            m.forward(5)
            engines.append(m)
        n1 = max(20, min(200, args.steps))
        lat = engines[0].forward(n1) / n1
        per = max(20, min(100, args.steps // 3))
        threads = [threading.Thread(target=lambda m=m: m.forward(per)) for m in engines]
        t0 = time.perf_counter()
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        dt = time.perf_counter() - t0
        flops, nbytes = engines[0].work()
        for m in engines:
            m.close()
    return dict(workload="320x512 ref+6-src window, planes (48,4,4) (the shipped tandem_512x320 model), fp32, resident in HBM",
                single_engine=dict(ms_per_depth_map=lat, depth_maps_per_s=1e3 / lat, forwards=n1),
                engines_3=dict(depth_maps_per_s=3 * per / dt, windows=3 * per),
                gflop_per_depth_map=flops / 1e9, reference_published_fps=4.96)


def boundary_leg(args, dev):
    """The operator boundary as the reference times it (test_dr_mvsnet, dr_mvsnet.cpp:540-545): CallAsync(host u8 images,
    K, poses) -> Ready -> GetResult (four host float maps) per window, E engines = E independent DrMvsnet objects.
    PCIe-inclusive: 6.45 MB in, 4.9 MB out per depth map."""
    import threading
    from synth import scene
    from tandem_amd.dr_mvsnet import DrMvsnet
    blob = model_blob()
    out = {}
    per = max(10, min(60, args.steps // 3))
    for E, pinned in ((1, False), (3, False), (1, True)):
        engines = [DrMvsnet(blob, device=dev) for _ in range(E)]
        wins = [scene.make_window(H, W, V, seed=40 + e) for e in range(E)]
        call_ms = [0.0] * E
        imgs = [w["bgrs"] for w in wins]
        if pinned:  # the extensions of include/dr_mi355x.h: key-frame images in page-locked memory (uploaded in place), results as views
            imgs = []
            for m, w in zip(engines, wins):
                imgs.append(m.alloc_images(V, H, W))
                for dst, src in zip(imgs[-1], w["bgrs"]):
                    dst[...] = src

        def loop(e, n):
            m, w = engines[e], wins[e]
            for _ in range(n):
                t0 = time.perf_counter()
                m.CallAsync(H, W, V, w["ref_index"], imgs[e], w["K"], list(w["c2ws"]), DEPTH_MIN, DEPTH_MAX, DISCARD)
                call_ms[e] += 1e3 * (time.perf_counter() - t0)
                m.GetResultView() if pinned else m.GetResult()
        for e in range(E):
            loop(e, 3)  # warm-up (plans, pinned buffers)
        call_ms = [0.0] * E
        threads = [threading.Thread(target=loop, args=(e, per)) for e in range(E)]
        t0 = time.perf_counter()
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        dt = time.perf_counter() - t0
        out["engines_%d%s" % (E, "_pinned" if pinned else "")] = dict(depth_maps_per_s=E * per / dt, ms_per_depth_map_per_engine=1e3 * dt / per,
                                     call_async_ms=sum(call_ms) / (E * per), windows=E * per)
        for m in engines:
            m.close()
    out["note"] = ("CallAsync(host u8 BGR x7, K, poses) -> GetResult (4 float maps) per window; host reorder + H2D + forward + D2H.  "
                   "engines_1_pinned: the same through the extensions drm_host_alloc (images in page-locked memory, uploaded in place) and "
                   "drm_get_result_view (the maps as views of the engine's pinned block): no host copy on either side")
    return out


def raycast_steps(k):
    """Mean samples per ray over the first k frames of the TSDF loop, counted by the parity build's measuring launch of k_raycast2 (tools/raycast_stats.py,
    a subprocess: the product library has no counting kernel).  None where the parity build is absent."""
    import re
    import subprocess
    hooks = os.path.join(ROOT, "tandem_amd", "libdr_mi355x_hooks.so")
    if not os.path.isfile(hooks):
        return None
    try:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "raycast_stats.py"), str(k)], capture_output=True, text=True, timeout=300,
                           env=dict(os.environ, DR_MI355X_LIB=hooks, DR_RAYCAST_STATS="1"))
        it = [float(x) for x in re.findall(r"iterations/lane ([0-9.]+)", r.stderr)]
        wl = [float(x) for x in re.findall(r"mean wave-longest ([0-9.]+)", r.stderr)]
        return dict(frames=len(it), samples_per_ray=sum(it) / len(it), wave_longest_ray=sum(wl) / len(wl)) if it else None
    except Exception as e:  # noqa: BLE001 -- reporting only
        return dict(error=str(e)[:160])


def tsdf_boundary(fr, poses, opt, dev, hook_stats):
    """The same frames through the operator API the way TANDEM and the reference's own driver call it (dr_debug_example.cpp:121,147,151;
    tandem_backend.cpp:166-177): HOST buffers into IntegrateScanAsync, RenderAsync from the frame's pose, GetRenderResult (which host-synchronises
    before the next scan can be queued -- the allocation of scan k + 1 therefore cannot overlap the ray-cast of scan k as it does in the
    device-resident hook).  Wall clock over the whole loop, map starting empty; the final map must equal the hook run's (block and update counts)."""
    from tandem_amd.dr_fusion import DrFusion, DrFusionOptions
    n = len(poses)
    hb, hd = fr["bgr"].cpu().numpy(), fr["depth"].cpu().numpy()
    f = DrFusion(DrFusionOptions(**opt), device=dev)
    for i in range(2):  # pinned staging, streams
        f.IntegrateScanAsync(hb[i], hd[i], poses[i]); f.RenderAsync([poses[i]]); f.GetRenderResult(copy=False)
    f.close()
    f = DrFusion(DrFusionOptions(**opt), device=dev)
    t_int = t_ren = t_get = 0.0
    t0 = time.perf_counter()
    for i in range(n):
        a = time.perf_counter(); f.IntegrateScanAsync(hb[i], hd[i], poses[i])
        b = time.perf_counter(); f.RenderAsync([poses[i]])
        c = time.perf_counter(); f.GetRenderResult(copy=False)
        d = time.perf_counter()
        t_int += b - a; t_ren += c - b; t_get += d - c
    t1 = time.perf_counter()
    st = f.stats()
    f.close()
    return dict(ms_per_frame=1e3 * (t1 - t0) / n, frames_per_s=n / (t1 - t0), value=st["updated_total"] / (t1 - t0), unit="voxels/s", frames=n,
                host_ms_per_frame=dict(IntegrateScanAsync=1e3 * t_int / n, RenderAsync=1e3 * t_ren / n, GetRenderResult=1e3 * t_get / n),
                same_map_as_hook_run=bool(st["blocks"] == hook_stats["blocks"] and st["updated_total"] == hook_stats["updated_total"]),
                note="IntegrateScanAsync(host bgr, host depth, pose) -> RenderAsync({pose}) -> GetRenderResult per frame through the C ABI (ctypes mirror of "
                     "dr_fusion.h), PCIe-inclusive (2.15 MB up, 2.15 MB down per frame), wall clock; the device-resident figure beside it comes from the "
                     "measuring hook drf_bench_sequence, whose allocate(k+1) / ray-cast(k) overlap this call order cannot reach")


def tsdf_leg(args, rank, dev, world):
    """BASELINE configs[3] as dr_debug_example.cpp:78-162 runs it: `--tsdf-frames` DISTINCT depth maps from a camera loop
    through an analytic room (synth/room.py, generated straight into HBM), per frame allocate + integrate and one
    ray-cast from the frame's pose (incl. the D2H of the rendered images), into a map that starts empty and keeps
    growing.  value = voxels updated by the whole run / its duration (hipEvents, first allocate .. last copy)."""
    import torch
    from synth import room
    from tandem_amd import replicas
    from tandem_amd.dr_fusion import DrFusion, DrFusionOptions
    n = args.tsdf_frames
    poses = room.loop_poses(n, seed=7 + rank)
    fr = room.render_frames(poses, H, W, device="cuda:%d" % dev, seed=rank)
    opt = dict(voxel_size=0.005, num_buckets=500000, bucket_size=10, num_blocks=2500000, block_size=8, max_sdf_weight=64,
               truncation_distance=0.02, max_sensor_depth=10.0, min_sensor_depth=0.1, num_render_streams=1,
               fx=fr["fx"], fy=fr["fy"], cx=fr["cx"], cy=fr["cy"], height=H, width=W)
    if args.tsdf_blocks > 0:  # SURVEY 8(d)'s pool size (>= 16 M blocks = 64 GB of voxels) instead of the default 2.5 M
        opt.update(num_blocks=args.tsdf_blocks, num_buckets=max(500000, args.tsdf_blocks // 5))
    torch.cuda.synchronize()
    warm = DrFusion(DrFusionOptions(**dict(opt, num_blocks=400000)), device=dev)  # loads the kernels; its map is thrown away
    warm.bench_sequence(fr["bgr"].data_ptr(), fr["depth"].data_ptr(), poses[:2], render=True)
    warm.close()
    f = DrFusion(DrFusionOptions(**opt), device=dev)
    replicas.barrier(dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ms = f.bench_sequence(fr["bgr"].data_ptr(), fr["depth"].data_ptr(), poses, render=True)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    replicas.barrier(dev)
    st = f.stats()
    vox = st["updated_total"]
    tmax, units = replicas.reduce_max_sum(ms["total"] * 1e-3, vox, dev)
    res = dict(metric="TSDF voxels integrated/sec", value=units / tmax, unit="voxels/s", frames=n,
               ms_per_frame=1e3 * tmax / n, wall_ms_per_frame=1e3 * (t1 - t0) / n, frames_per_s=n * world / tmax,
               blocks=st["blocks"], voxels_per_frame=vox / n, mismatches=st["mismatches"],
               kernel_ms_per_frame=dict(allocate_commit_cull=ms["allocate"] / n, integrate=ms["integrate"] / n, raycast=ms["raycast"] / n,
                                        render_d2h=ms["d2h"] / n),  # hipEvents on the engine's streams; `integrate` brackets k_integrate alone
               integrate_only_voxels_per_s=vox / (ms["integrate"] * 1e-3),
               config=dict(workload="%d distinct synthetic 640x480 depth maps (camera loop through a 6x4x3 m room with a sphere, 2.5 %% invalid "
                                    "pixels) fused into an initially empty 5 mm hashed voxel grid (truncation 20 mm; num_blocks = %.1f M = %.0f GB of voxels; SURVEY 8d asks for >= 16 M: "
                                    "the loop allocates 378 k blocks and bump allocation makes the pool size irrelevant to every kernel -- shown once with --tsdf-blocks 16000000, "
                                    "profiles/r06_tsdf_16m_blocks.json): per frame allocate + integrate + one ray-cast from the frame's pose incl. D2H of the render; frames resident in HBM; "
                                    "THIS figure is produced by the measuring hook drf_bench_sequence (device pointers in, allocate(k+1) beside ray-cast(k)); the same frames through "
                                    "the public call order are `boundary` below" % (n, opt["num_blocks"] / 1e6, opt["num_blocks"] * 4096 / 1e9)))
    if rank == 0:
        visited = f.visited_blocks()
        # what k_integrate really moves: every visible block is READ whole (4 KB, updated or not), every updated voxel is written (8 B)
        res["integrate_traffic_model"] = dict(visited_blocks_per_frame=visited / n, read_bytes_per_frame=4096.0 * visited / n, write_bytes_per_frame=8.0 * vox / n,
                                              algorithmic_bytes_per_frame=16.0 * vox / n,
                                              note="read = 4 KB x visible blocks (k_cull's list), written = 8 B x updated voxels; the PMC FETCH_SIZE of this kernel is compared with "
                                                   "read_bytes_per_frame UNCORRECTED: its 8-byte-per-lane loads are not the 16-byte streaming reads the x2 rule of MI355X_MICROARCH.md was calibrated on")
        ach = 16.0 * vox / (ms["integrate"] * 1e-3) / 1e9
        res["roofline"] = dict(bound="hbm", kernel="k_integrate", avg_launch_ms=ms["integrate"] / n, bytes_per_launch=16.0 * vox / n,
                               achieved=ach, peak=PEAK_HBM_GBPS, unit="GB/s", frac=ach / PEAK_HBM_GBPS, traffic=pmc_traffic("k_integrate", fetch="raw"),
                               traffic_if_fetch_doubled=pmc_traffic("k_integrate"))
        # the ray-cast (74 % of the frame) against SURVEY 8(d)'s own lower bound: 7 B written + steps x 8 corners x 8 B read per output pixel, with the
        # measured step count; and against what the counters say actually left the caches
        rs = None if (args.no_tsdf_boundary or world > 1) else raycast_steps(min(n, 200))  # (a subprocess on this rank's GPU: single-GPU runs only)
        rc_ms = ms["raycast"] / n
        rc = dict(bound="hbm", kernel="k_raycast2", avg_launch_ms=rc_ms, steps=rs, peak=PEAK_HBM_GBPS, unit="GB/s",
                  traffic=pmc_traffic("k_raycast2", fetch="raw"), traffic_if_fetch_doubled=pmc_traffic("k_raycast2"))
        if rs and "samples_per_ray" in rs:
            lb = H * W * (7.0 + rs["samples_per_ray"] * 64.0)
            rc.update(bytes_per_launch=lb, achieved=lb / (rc_ms * 1e-3) / 1e9, frac=lb / (rc_ms * 1e-3) / 1e9 / PEAK_HBM_GBPS,
                      note="bytes_per_launch = pixels x (7 + samples_per_ray x 8 corners x 8 B): SURVEY 8(d)'s lower bound with the measured step count; these are "
                           "requests the caches serve (every sample of a ray re-reads 7 of the 8 corners' lines of its neighbour), `traffic` is what reached the memory side")
        if rc["traffic"]:
            rc["hbm_frac_of_counter_traffic"] = rc["traffic"] / (rc_ms * 1e-3) / 1e9 / PEAK_HBM_GBPS
        res["roofline_raycast"] = rc
        if world == 1 and not args.no_tsdf_boundary:
            res["boundary"] = tsdf_boundary(fr, poses, opt, dev, st)
        # marching cubes (DrFusion::ExtractMeshAsync + GetMeshSync) of the fused map over TANDEM's (-5..5 m)^3 box
        # (tandem_backend.cpp:80-81): device time = until the triangle count is known, total adds the D2H copy
        lo, hi = (-5.0, -5.0, -5.0), (5.0, 5.0, 5.0)
        f.ExtractMeshAsync(lo, hi); f.GetMeshSync()  # first call allocates the triangle buffers
        t0 = time.perf_counter(); f.ExtractMeshAsync(lo, hi); ntri = f.mesh_num_triangles(); t1 = time.perf_counter()
        f.GetMeshSync(); t2 = time.perf_counter()
        res["mesh"] = dict(triangles=ntri, blocks=st["blocks"], extract_ms=1e3 * (t1 - t0), get_ms_incl_d2h=1e3 * (t2 - t1),
                           lattice="2000^3 cells at 5 mm, visited per allocated block")
        if world == 1 and not args.no_cpu:
            k = min(n, 24)
            res["cpu_baseline"] = tsdf_cpu_baseline([(fr["bgr"][i].cpu().numpy(), fr["depth"][i].cpu().numpy(), poses[i]) for i in range(k)], opt, dev)
            res["parity"] = res["cpu_baseline"].pop("parity", None)
    f.close()
    if rank == 0 and not args.no_tsdf_native:  # the reference's native setting (FullSystem.cpp:260,266: 1 cm voxels, 4 cm truncation), same frames
        g = DrFusion(DrFusionOptions(**dict(opt, voxel_size=0.01, truncation_distance=0.04, num_blocks=600000)), device=dev)
        ms10 = g.bench_sequence(fr["bgr"].data_ptr(), fr["depth"].data_ptr(), poses, render=True)
        st10 = g.stats()
        res["native_10mm"] = dict(value=st10["updated_total"] / (ms10["total"] * 1e-3), unit="voxels/s", ms_per_frame=ms10["total"] / n,
                                  voxels_per_frame=st10["updated_total"] / n, blocks=st10["blocks"],
                                  kernel_ms_per_frame=dict(allocate_commit_cull=ms10["allocate"] / n, integrate=ms10["integrate"] / n,
                                                           raycast=ms10["raycast"] / n, render_d2h=ms10["d2h"] / n),
                                  note="1 cm voxels, 4 cm truncation: TANDEM's own DrFusionOptions; same 1000 frames and loop")
        g.close()
    del fr
    torch.cuda.empty_cache()
    return res


def tandem_loop_leg(args, dev):
    """BASELINE configs[4] stand-in (the DSO front-end cannot run here).  Driver: the REFERENCE's own TandemBackend
    (tandem/src/tandem/tandem_backend.cpp, compiled UNCHANGED against tandem_amd/libdr/*.h by oracle/Makefile.ref ->
    oracle/_ref/tandem_backend_run; tools/tandem_backend_main.cpp plays FullSystem::deliverDrFrame): GetResult(k-1), CallAsync(k),
    IntegrateScanAsync / RenderAsync / GetRenderResult of k-1, the tracker's depth-map hand-over, all in the reference's code and
    order, on this GPU, host buffers at the boundary.  What is measured is libdr_mi355x.so; the binary is the reference's caller,
    not a checker.  Where that binary was not built (no reference checkout at build time) the hand-written imitation of the same
    call order, tools/tandem_loop.cpp, is compiled and timed instead; `driver` in the result says which one ran."""
    import subprocess
    import tempfile
    from synth import scene
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from export_fixture import write_tdms
    out = {}
    with tempfile.TemporaryDirectory() as td:
        exe = os.path.join(ROOT, "oracle", "_ref", "tandem_backend_run")
        if not os.path.isfile(exe):
            exe = os.path.join(td, "tandem_loop")
            try:
                subprocess.check_call(["g++", "-std=c++14", "-O2", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "tandem_amd", "libdr"),
                                       os.path.join(ROOT, "tools", "tandem_loop.cpp"), "-o", exe, "-L" + os.path.join(ROOT, "tandem_amd"),
                                       "-ldr_mi355x", "-Wl,-rpath," + os.path.join(ROOT, "tandem_amd")])
            except (OSError, subprocess.CalledProcessError) as e:
                return dict(error="g++ build of tools/tandem_loop.cpp failed: %s" % e)
        blob = os.path.join(ROOT, "weights", "tandem_va.tdmw")
        for name, (h, w, vs) in (("640x480_5mm", (480, 640, "0.005")), ("640x480_10mm", (480, 640, "0.01"))):
            win = scene.make_window(h, w, V, seed=5)
            sample = os.path.join(td, name + ".tdms")
            z = np.zeros((h, w), np.float32)
            # depth range: TANDEM passes depth_min = 0.01 (FullSystem.h:388) and 3 x the 0.2-quantile of the tracker's sparse depths
            # (FullSystem.cpp:1175-1193): here that quantile is taken from the window's ground-truth depth
            dmax = 3.0 * float(np.quantile(win["gt_depth"], 0.2))
            write_tdms(sample, np.stack(win["bgrs"]), win["K"], win["c2ws"], win["ref_index"], 0.01, dmax, DISCARD, z, z)
            env = dict(os.environ, HIP_VISIBLE_DEVICES=str(dev)) if dev else dict(os.environ)
            r = subprocess.run([exe, blob, sample, str(args.loop_keyframes), vs, "0", "1"], capture_output=True, text=True, timeout=900, env=env)
            if r.returncode != 0:
                out[name] = dict(error=(r.stdout + r.stderr)[-500:])
                continue
            out[name] = json.loads(r.stdout.strip().splitlines()[-1])
            out[name]["depth_range"] = [0.01, dmax]
            if name == "640x480_10mm" and exe.endswith("tandem_backend_run"):
                # the window SLIDES as TANDEM's does (six of a window's seven images were in the previous one, one is new), first without, then with the
                # key-frame feature cache (drm_set_feature_cache: FeatureNet on the new image only).  Its own leg: the legs above re-send one window and
                # would hit the cache with every image -- they never enable it, and neither does anything that feeds `value` / `single_window_ms`.
                sl = {}
                for cache in (0, 16, 0, 16, 0, 16):
                    r = subprocess.run([exe, blob, sample, str(3 * args.loop_keyframes), vs, "0", "1", "1", str(cache)], capture_output=True, text=True, timeout=900, env=env)
                    if r.returncode != 0:
                        sl["error"] = (r.stdout + r.stderr)[-500:]
                        break
                    d = json.loads(r.stdout.strip().splitlines()[-1])
                    sl.setdefault("cache_%d" % cache, []).append(dict(keyframes_per_s=d["keyframes_per_s"], ms_per_keyframe=d["ms_per_keyframe"], mean_ms=d["mean_ms"]))
                if "error" not in sl:  # ... and with GetResult() handing out views of the page-locked result block on top (DrMvsnet::SetResultViews: no 4.9 MB copy per key frame)
                    for _ in range(3):
                        r = subprocess.run([exe, blob, sample, str(3 * args.loop_keyframes), vs, "0", "1", "1", "16", "1"], capture_output=True, text=True, timeout=900, env=env)
                        if r.returncode != 0:
                            sl["error"] = (r.stdout + r.stderr)[-500:]
                            break
                        d = json.loads(r.stdout.strip().splitlines()[-1])
                        sl.setdefault("cache_16_result_views", []).append(dict(keyframes_per_s=d["keyframes_per_s"], ms_per_keyframe=d["ms_per_keyframe"], mean_ms=d["mean_ms"]))
                if "error" not in sl:
                    sl["keyframes_per_s_cache_on_result_views"] = sorted(x["keyframes_per_s"] for x in sl["cache_16_result_views"])[1]
                    off, on = (sorted(x["keyframes_per_s"] for x in sl["cache_%d" % c])[1] for c in (0, 16))  # the median of three
                    sl.update(keyframes_per_s_cache_off=off, keyframes_per_s_cache_on=on, speedup=on / off,
                              note="1 cm voxels (TANDEM's setting), sliding synthetic sequence: each key frame drops the oldest image of the window and adds a new one; "
                                   "three alternating runs of %d key frames each, the median reported; feature cache of 16 key frames" % (3 * args.loop_keyframes))
                out["sliding_window"] = sl
    out["note"] = ("tandem_backend.cpp's own call order (the reference's file, unchanged, where oracle/_ref/tandem_backend_run exists); depth network of "
                   "keyframe k overlaps fusion + ray-cast of keyframe k-1; (48,32,8) planes, discard 10 %, dense tracking render on; TANDEM's own setting is 10 mm")
    return out


def tracker_leg(args, dev):
    """SURVEY 8(f) rows 3-4 (not a headline metric): the dense coarse tracker at its largest size -- every pixel of a
    640x480 keyframe as a reference point -- one calcRes + calcG (one Gauss-Newton iteration of
    CoarseTracker::trackNewestCoarse on level 0) and the dense-depth hand-off, hipEvent-timed on the tracker stream;
    the single-threaded C restatement beside it."""
    import numpy as np
    from synth import scene
    from tandem_amd.dr_tracker import DrCoarseTracker
    p = scene.make_tracking_pair(H, W, seed=1, sparse_fraction=1.0)
    g = DrCoarseTracker(W, H, 9.0, 20.0, device=dev)
    g.setK(W, H, p["fx"], p["fy"], p["cx"], p["cy"])
    g.init()
    g.setReference(p["pc_u"], p["pc_v"], p["pc_idepth"], p["pc_color"], 1.0, [0.0, 0.0])
    g.setNew(p["dI_new"])
    # warm-up covers everything the timed loops touch, the timing events included: the driver's fresh-lease runs of rounds 1
    # and 2 showed a one-off of ~55 ms inside whichever loop was timed first.  Every call is also timed on its own (it ends
    # with a stream synchronise), so a one-off shows up as `max_ms` instead of inflating the figure: the MEDIAN call is reported.
    # ... and it lasts long enough (>= 0.25 s of back-to-back calls) to take the card out of the idle power state the preceding CPU-baseline
    # legs leave it in: round 4's driver run showed ONE 60 ms call (calc_res_max_ms) beside a 0.037 ms median -- the first call after ~25 s
    # without GPU work.  `slowest_call_index` below says where in the timed loop the slowest call sat, should one appear again.
    g.startTiming()
    t_spin = time.perf_counter()
    while time.perf_counter() - t_spin < 0.25:
        g.calcRes(p["refToNew"], 1.0, [0.0, 0.0], 20.0); g.calcG(1.0, [0.0, 0.0])
    g.endTimingMilliseconds()
    reps = 50

    def per_call(fn):
        ts = []
        g.startTiming()
        for _ in range(reps):
            t0 = time.perf_counter(); fn(); ts.append(1e3 * (time.perf_counter() - t0))
        span = g.endTimingMilliseconds() / reps
        worst = max(range(reps), key=lambda i: ts[i])
        return sorted(ts)[len(ts) // 2], (ts[worst], worst), span
    t_res, max_res, span_res = per_call(lambda: g.calcRes(p["refToNew"], 1.0, [0.0, 0.0], 20.0))
    t_g, max_g, span_g = per_call(lambda: g.calcG(1.0, [0.0, 0.0]))
    K = np.array([[p["fx"], 0, p["cx"]], [0, p["fy"], p["cy"]], [0, 0, 1]], np.float32)
    T = np.linalg.inv(p["c2w_ref"]) @ p["c2w_new"]
    KRKi = (K @ T[:3, :3].astype(np.float32)) @ np.linalg.inv(K.astype(np.float64)).astype(np.float32)
    Kt = K @ T[:3, 3].astype(np.float32)
    g.setReference([], [], [], [], 1.0, [0.0, 0.0])
    g.appendDenseReference(p["depth_new"], KRKi, Kt, 1, True, None, p["dI_ref"])
    g.setReference([], [], [], [], 1.0, [0.0, 0.0])
    t0 = time.perf_counter(); n_dense = g.appendDenseReference(p["depth_new"], KRKi, Kt, 1, True, None, p["dI_ref"]); t1 = time.perf_counter()
    g.close()
    n = len(p["pc_u"])
    res = dict(points=n, calc_res_ms=t_res, calc_g_ms=t_g, gauss_newton_iterations_per_s=1e3 / (t_res + t_g),
               per_call=dict(statistic="median of %d host-timed calls after a 0.25 s spin-up" % reps, calc_res_max_ms=max_res[0], calc_g_max_ms=max_g[0],
                             slowest_call_index=dict(calc_res=max_res[1], calc_g=max_g[1]),
                             calc_res_event_span_ms=span_res, calc_g_event_span_ms=span_g),
               hbm_gbps=dict(calc_res=n * 4.0 * (4 + 7) / (t_res * 1e-3) / 1e9, calc_g=n * 4.0 * 8 / (t_g * 1e-3) / 1e9),
               dense_handoff_ms_incl_uploads=1e3 * (t1 - t0), dense_points=n_dense,
               note="each call ends with a stream synchronise + 7/45-double D2H, as the reference's does; launch/sync latency bound")
    if not args.no_cpu:
        from oracle.tracker_oracle import TrackerOracle
        o = TrackerOracle(W, H, 9.0, 20.0)
        o.setK(p["fx"], p["fy"], p["cx"], p["cy"])
        o.setReference(p["pc_u"], p["pc_v"], p["pc_idepth"], p["pc_color"], 1.0, [0.0, 0.0])
        o.setNew(p["dI_new"])
        t0 = time.perf_counter()
        for _ in range(5):
            o.calcRes(p["refToNew"], 1.0, [0.0, 0.0], 20.0); o.calcG(1.0, [0.0, 0.0])
        res["cpu_baseline"] = dict(value=5.0 / (time.perf_counter() - t0), unit="gauss-newton iterations/s", cores=1, kind="port",
                                   sample="5 x (calcRes + calcG) over the same 307200 points, single-threaded C restatement (oracle/tracker_oracle.c)")
    return res


def view_shard_leg(args, rank, dev, world):
    """BASELINE configs[2]: ONE 7-view window, its source views sharded over the ranks, one RCCL sum all-reduce of the
    fp32 cost volume per cascade stage (tandem_amd/view_shard.py).  Reported next to the replicas headline, never as it:
    by SURVEY 8e the reduce alone exceeds the single-GPU pipeline, so this configuration loses throughput by design."""
    from tandem_amd import replicas, view_shard
    DrMvsnet = engine_class()
    steps, warmup = min(args.steps, 20), 2
    win = make_window(H, W, V, 0)  # the SAME window on every rank
    window = dict(bgrs=win["bgrs"], K=win["K"], c2ws=list(win["c2ws"]), ref_index=win["ref_index"],
                  depth_min=DEPTH_MIN, depth_max=DEPTH_MAX, discard=DISCARD)
    m = DrMvsnet(model_blob(), device=dev)
    one_dev = os.environ.get("DR_BENCH_ONE_DEVICE") == "1"  # test scaffold: RCCL refuses two ranks on one device
    P = view_shard.shard_world(V, world)  # at most one rank per source view takes part (8 GPUs, 6 source views: two stay out)
    if not one_dev and view_shard.init_engine_collective(m, rank, world, participants=P):
        # the engine's own collective, stream-ordered, no host step per phase: ncclReduce of each cost volume to rank 0,
        # which alone regularises the stage, ncclBroadcast of the stage's depth map back
        active = rank < P
        mine = view_shard.upload(m, window, rank, P) if active else [win["ref_index"]]
        step = (lambda n: m.forward(n)) if active else (lambda n: None)
        mode = "in-engine RCCL: reduce of the fp32 cost volume to rank 0 after each stage's cost-volume kernel, broadcast of the stage depth map back; %d of %d ranks take part" % (P, world)
        nbytes = sum(48 * 120 * 160 * 32 * 4 if s == 1 else (32 * 240 * 320 * 16 * 4 if s == 2 else 8 * 480 * 640 * 8 * 4) for s in (1, 2, 3)) if PLANES == (48, 32, 8) and (H, W) == (480, 640) else 0
        if active:
            nbytes = sum(m.device_tensor("volume%d" % s)[1] for s in (1, 2, 3)) * 4
    else:
        active = True
        mine = view_shard.upload(m, window, rank, world)
        nmax = max(m.device_tensor("volume%d" % s)[1] for s in (1, 2, 3))
        ar = view_shard.TorchAllReduce(dev, nmax)

        def step(n):
            for _ in range(n):
                view_shard.forward(m, ar)
        mode = "host-driven phases with torch.distributed all-reduce (%s)" % ("one-device test scaffold" if one_dev else "engine could not bind RCCL")
        nbytes = sum(m.device_tensor("volume%d" % s)[1] for s in (1, 2, 3)) * 4
    step(warmup)
    replicas.barrier(dev)
    gpu_sync()
    t0 = time.perf_counter()
    step(steps)
    gpu_sync()
    t1 = time.perf_counter()
    replicas.barrier(dev)
    tmax, nsrc = replicas.reduce_max_sum(t1 - t0, len(mine) - 1, dev)
    # every participating rank holds the same depth map (idle ranks report the maximum's negative so that they never win it)
    mysum = float(m.download().depth_dense.astype("float64").sum()) if active else -1e300
    chk = replicas.reduce_max_sum(mysum, 0, dev)[0]
    same = (not active) or abs(chk - mysum) == 0.0
    same = replicas.reduce_max_sum(0.0 if same else 1.0, 0, dev)[0] == 0.0
    # how many ranks RCCL itself counts in the engine's communicator (ncclCommCount), the largest and the number of ranks holding one: a
    # driver record then shows that the collective really spanned `participants` GPUs and that the idle ranks never joined it
    cnt = m.comm_count() if hasattr(m, "comm_count") else -1
    cmax, joined = replicas.reduce_max_sum(float(cnt), 1.0 if cnt > 0 else 0.0, dev)
    m.close()
    return dict(depth_maps_per_s=steps / tmax, ms_per_depth_map=1e3 * tmax / steps, steps=steps, n_gpus=world, participants=P,
                rccl_comm_ranks=int(cmax), ranks_in_communicator=int(joined),
                source_views_total=int(nsrc), source_views_this_rank=len(mine) - 1,
                allreduce_mb_per_depth_map=nbytes / 1e6, ranks_agree=bool(same), collective=mode,
                note="one window sharded over the ranks; 3 fp32 volume reductions (RCCL) per depth map")


def tsdf_cpu_baseline(scans, opt, dev=0):
    """the C restatement on one core, and its OpenMP-over-blocks build (integration parallel over the
    allocated blocks, allocation serial) on the physical cores of this box -- same frames, bounded sample.  The state the single-core run
    ends with is not thrown away: the engine fuses the same frames through IntegrateScanAsync and the two maps are compared
    block for block, bit for bit (`parity`)."""
    from oracle.tsdf_oracle import TsdfOracle
    phys, _ = host_cores()
    out = {}
    parity = None
    mism_free = lambda st: st["mismatches"] == 0  # noqa: E731 -- (an OpenMP run whose blocks raced is void and is not a checker either)
    for omp in (False, True):
        if omp:
            os.environ["OMP_NUM_THREADS"] = str(phys)
        try:
            o = TsdfOracle(omp=omp, **dict(opt, num_blocks=600000))
        except Exception as e:  # no libgomp on this box: report the single-core figure only
            out["openmp_error"] = str(e)[:200]
            continue
        n, t = 0, 0.0
        for bgr, depth, pose in scans:
            t0 = time.perf_counter(); o.integrate(bgr, depth, pose); t += time.perf_counter() - t0
            n += 1
            if t > (10.0 if omp else 15.0):
                break
        st = o.stats()
        key = "openmp" if omp else "single"
        out[key] = dict(value=st["updated_total"] / t, frames=n, seconds=t, cores=phys if omp else 1, mismatches=st["mismatches"])
        if mism_free(st) and (parity is None or n > parity["tsdf_frames_equal"]):  # the run that got furthest in its time budget is the one compared
            from tandem_amd.dr_fusion import DrFusion, DrFusionOptions
            f = DrFusion(DrFusionOptions(**dict(opt, num_blocks=600000)), device=dev)
            for bgr, depth, pose in scans[:n]:
                f.IntegrateScanAsync(bgr, depth, pose); f.RenderAsync([pose]); f.GetRenderResult(copy=False)
            a, b = f.export_blocks(), o.export_blocks()
            f.close()
            equal = a.keys() == b.keys() and all(np.array_equal(a[k], b[k]) for k in a)
            parity = dict(tsdf_frames_equal=n if equal else 0, blocks=len(a), oracle_blocks=len(b), voxel_state="bit-exact" if equal else "DIFFERS",
                          checker="oracle/tsdf_oracle.c%s (pinned to the reference's own sources compiled for the host by tests/test_ref_fusion.py), the first %d frames of this "
                                  "leg's workload through IntegrateScanAsync, maps compared block for block" % (", OpenMP build" if omp else "", n))
    best = out.get("openmp", out["single"])
    if best is not out["single"] and best["mismatches"] != 0:  # blocks were NOT independent in the parallel run: its figure is void
        out["openmp"]["void"] = "round-trip mismatches != 0: the OpenMP integration raced; single-core figure reported"
        best = out["single"]
    return dict(value=best["value"], unit="voxels/s", cores=best["cores"], kind="port", single_core=out["single"], openmp=out.get("openmp"), parity=parity,
                sample="allocate + integrate of the first %d of the same frames, C restatement (oracle/tsdf_oracle.c, pinned to the reference build by "
                       "tests/test_ref_fusion.py): one core, and integration parallel over blocks with OpenMP on %d physical cores (allocation serial)"
                       % (best["frames"], phys))


def respawn_under_torchrun(args):
    """`python bench.py --gpus N` without a torch.distributed.run environment: launch the N ranks ourselves."""
    import socket
    import subprocess
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--tsdf-frames", type=int, default=1000, help="distinct synthetic depth maps fused by the TSDF leg (BASELINE configs[3])")
    ap.add_argument("--loop-keyframes", type=int, default=100, help="keyframes of the TandemBackend-shaped loop (tools/tandem_loop.cpp)")
    ap.add_argument("--no-boundary", action="store_true", help="skip the operator-boundary (CallAsync/GetResult) leg")
    ap.add_argument("--no-loop", action="store_true", help="skip the TandemBackend-shaped loop leg")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline legs")
    ap.add_argument("--engines", type=int, default=4, help="DrMvsnet engines (independent windows in flight) per GPU; 1 = latency configuration")
    ap.add_argument("--no-tsdf", action="store_true")
    ap.add_argument("--no-tsdf-native", action="store_true", help="skip the extra TSDF run at the reference's native 1 cm / 4 cm setting (profiling runs: keeps "
                                                                   "per-kernel averages to the 5 mm loop the roofline is quoted on)")
    ap.add_argument("--no-view-shard", action="store_true", help="N > 1 only: skip the view-sharded (configs[2]) leg")
    ap.add_argument("--no-tsdf-boundary", action="store_true", help="skip the TSDF leg's API-level run (host buffers through IntegrateScanAsync / RenderAsync / GetRenderResult) and the ray-cast step count")
    ap.add_argument("--tsdf-blocks", type=int, default=0, help="voxel-block pool of the TSDF leg (0: 2.5 M = 10 GB; SURVEY 8d's configuration is 16000000 = 64 GB)")
    ap.add_argument("--config", choices=["headline", "shipped"], default="headline",
                    help="headline: BASELINE.json's metric configuration, 640x480x7 views, planes (48,32,8).  shipped: the model TANDEM exports and runs "
                         "(tandem_512x320: 320x512x7 views, planes (48,4,4)); every leg of the line (value, roofline, cpu_baseline, boundary) is then on that shape")
    args = ap.parse_args()
    if args.config == "shipped":
        global H, W, PLANES
        H, W, PLANES = 320, 512, (48, 4, 4)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        respawn_under_torchrun(args)
    import torch
    from tandem_amd import replicas
    rank, local_rank, world = replicas.env_world()
    if world != args.gpus and rank == 0:
        print("bench.py: --gpus %d but WORLD_SIZE=%d (running %d replica%s)" % (args.gpus, world, world, "" if world == 1 else "s"), file=sys.stderr)
    if DRY:  # launcher / plumbing rehearsal on a CPU box: gloo, stand-in engines, only the legs every rank takes part in
        replicas.init("gloo", None)
        mv = mvsnet_leg(args, rank, None, world)
        vs = view_shard_leg(args, rank, None, world) if world > 1 and not args.no_view_shard else None
        if rank == 0:
            print(json.dumps({"metric": "depth-maps/sec @ 640\u00d7480\u00d77-view\u00d73-stage; TSDF voxels integrated/sec", "value": None, "unit": "depth-maps/s",
                              "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": None, "higher_is_better": True, "scaling": "weak",
                              "vs_baseline": None, "dtype": "f32", "data": "DRY RUN: stand-in engine, no GPU work -- launcher and rank plumbing only", "dry_run": True,
                              "config": {"workload": "none (dry run)"}, "engines_per_gpu": mv["engines_per_gpu"], "units_counted": mv["units"], "view_sharded": vs}), flush=True)
        import torch.distributed as dist
        if dist.is_initialized():
            dist.destroy_process_group()
        return
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback in the product path)")
    # test scaffolding for 1-GPU boxes: DR_BENCH_ONE_DEVICE=1 runs every rank on cuda:0 over gloo (RCCL refuses two ranks
    # on one device) so that the N > 1 code path can be exercised end to end; never set by the driver
    one_dev = os.environ.get("DR_BENCH_ONE_DEVICE") == "1"
    if one_dev:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    replicas.init("gloo" if one_dev else "nccl", local_rank)

    par = golden_parity(local_rank) if (rank == 0 and args.config == "headline") else None  # before anything is timed
    mv = mvsnet_leg(args, rank, local_rank, world)
    ts = None if args.no_tsdf else tsdf_leg(args, rank, local_rank, world)
    tr = tracker_leg(args, local_rank) if (rank == 0 and not args.no_tsdf) else None
    bd = boundary_leg(args, local_rank) if (rank == 0 and world == 1 and not args.no_boundary) else None
    lp = tandem_loop_leg(args, local_rank) if (rank == 0 and world == 1 and not args.no_loop) else None
    sh = shipped_leg(args, local_rank) if (rank == 0 and world == 1 and not args.no_boundary) else None
    vs, vs_hung = None, False
    if world > 1 and not args.no_view_shard:
        # The sharded leg is the only part of this program with a data-path collective.  It runs under a watchdog: if a
        # rank gets stuck in it (a collective that never completes cannot be cancelled from Python) the headline line
        # above is still printed and every rank leaves with os._exit instead of hanging the launcher.
        import threading
        box = {}

        def run_vs():
            try:
                box["r"] = view_shard_leg(args, rank, local_rank, world)
            except Exception as e:  # reported in the line, never fatal for the headline measurement
                box["r"] = dict(error="%s: %s" % (type(e).__name__, e))
        th = threading.Thread(target=run_vs, daemon=True)
        th.start()
        th.join(float(os.environ.get("DR_BENCH_VS_TIMEOUT", "240")))
        vs_hung = th.is_alive()
        vs = dict(error="view-sharded leg did not finish within the watchdog time") if vs_hung else box.get("r")
    if rank == 0:
        out = {
            # BASELINE.json's metric string, verbatim, for the configuration it is quoted on
            "metric": ("depth-maps/sec @ 640\u00d7480\u00d77-view\u00d73-stage; TSDF voxels integrated/sec" if args.config == "headline"
                       else "depth-maps/sec @ %d\u00d7%d\u00d77-view\u00d73-stage; TSDF voxels integrated/sec" % (W, H)),
            "value": mv["value"], "unit": "depth-maps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": mv["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": ("%dx%d ref+6-src keyframe window, 3-stage cascade (%d/%d/%d hypotheses), view aggregation, fp32; " % ((W, H) + PLANES)) +
                                   "independent windows, %d in flight per GPU (one DrMvsnet engine each) x %d GPU replica(s); trained weights recovered from the reference's exported "
                                   "tandem_512x320 model (same architecture), inputs resident in HBM" % (mv["engines_per_gpu"], world),
                       "height": H, "width": W, "views": V, "planes": list(PLANES), "discard_percentage": DISCARD, "name": args.config,
                       "depth_min": DEPTH_MIN, "depth_max": DEPTH_MAX,
                       "parallelism": "replicas x%d, %d engines per GPU" % (world, mv["engines_per_gpu"]),
                       "reference_published": "2.70 FPS (abl03, unstated GPU, incl. data loading) -- not the same clock, so vs_baseline is null"},
            "event_ms_per_step": mv["event_ms_per_step"],
        }
        for k in ("repeats", "gpu_state", "roofline", "cpu_baseline", "pipeline", "engines_per_gpu", "single_engine", "scene_depth_range"):
            if k in mv:
                out[k] = mv[k]
        # TANDEM's usage is ONE window in flight (tandem_backend.cpp:147): its latency and the operator boundary's rate as flat keys
        if "single_engine" in mv:
            out["single_window_ms"] = mv["single_engine"]["ms_per_depth_map"]
        if bd is not None and "engines_1" in bd:
            out["boundary_single_engine_ms"] = bd["engines_1"]["ms_per_depth_map_per_engine"]
            out["boundary_single_engine_depth_maps_per_s"] = bd["engines_1"]["depth_maps_per_s"]
            out["boundary_pinned_single_engine_ms"] = bd["engines_1_pinned"]["ms_per_depth_map_per_engine"]
        if par is not None:
            out["parity"] = par
        if ts is not None:
            out["tsdf"] = ts
        if bd is not None:
            out["boundary"] = bd
        if sh is not None:
            out["shipped_model"] = sh
        if lp is not None:
            out["tandem_loop"] = lp
        if tr is not None:
            out["tracker"] = tr
        if vs is not None:
            out["view_sharded"] = vs
        print(json.dumps(out), flush=True)
    if vs_hung:
        os._exit(0)
    import torch.distributed as dist
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
