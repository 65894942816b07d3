#!/bin/bash
# mid-round check: whole -m gpu suite, smoke, the bench line (reduced sizes)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider --durations=8 > gpurun_out/r2_gpu_tests_full.log 2>&1; grep -E "passed|failed|^FAILED|^ERROR|s call" gpurun_out/r2_gpu_tests_full.log | tail -14
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 900 python bench.py --steps 150 --warmup 5 --tsdf-frames 400 --loop-keyframes 40 > gpurun_out/r2_bench_try.json 2> gpurun_out/r2_bench_try.err
echo "bench rc=$?"; tail -3 gpurun_out/r2_bench_try.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r2_bench_try.json").read().strip().splitlines()[-1])
for k in ("value", "ms_per_step", "single_engine", "shipped_model", "boundary"):
    print(k, json.dumps(d.get(k)))
print("loop", {k: (v.get("keyframes_per_s"), v.get("mean_ms")) for k, v in d.get("tandem_loop", {}).items() if isinstance(v, dict)})
t = d.get("tsdf", {})
print({k: t.get(k) for k in ("value", "frames", "ms_per_frame", "blocks", "voxels_per_frame", "kernel_ms_per_frame", "integrate_only_voxels_per_s", "roofline", "mesh")})
print(d.get("roofline"))
PY
