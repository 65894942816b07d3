#!/bin/bash
# round 2, call 2: the fixed tracker reference test + the reworked bench line (all legs, reduced sizes)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_tracker_gpu.py -m gpu -q --no-header -p no:cacheprovider -k reference_build 2>&1 | tail -5
timeout 900 python bench.py --steps 150 --warmup 5 --tsdf-frames 300 --loop-keyframes 40 > gpurun_out/r2_bench_try.json 2> gpurun_out/r2_bench_try.err
echo "bench rc=$?"; tail -5 gpurun_out/r2_bench_try.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r2_bench_try.json").read().strip().splitlines()[-1])
for k in ("value", "ms_per_step", "single_engine", "boundary", "tandem_loop"):
    print(k, json.dumps(d.get(k)))
t = d.get("tsdf", {})
print({k: t.get(k) for k in ("value", "frames", "ms_per_frame", "blocks", "voxels_per_frame", "kernel_ms_per_frame", "integrate_only_voxels_per_s", "roofline", "cpu_baseline", "mesh")})
print(d.get("roofline")); print(d.get("cpu_baseline")); print(d.get("tracker"))
PY
