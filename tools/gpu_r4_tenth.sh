#!/bin/bash
# Round 4, tenth call: the whole -m gpu suite and the driver's bench command on the tree with both Winograd forms in the plan table.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider -x > gpurun_out/r4j_gpu_tests.log 2>&1
echo "== gpu suite: $(grep -E 'passed|failed' gpurun_out/r4j_gpu_tests.log | tail -1)"; grep -E "^FAILED|^ERROR" gpurun_out/r4j_gpu_tests.log | head -8
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4j_bench.json 2> gpurun_out/r4j_bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4j_bench.json").read().strip().splitlines()[-1])
print({k: d.get(k) for k in ("value", "ms_per_step", "single_window_ms", "boundary_single_engine_ms", "boundary_pinned_single_engine_ms")})
print({k: d["roofline"].get(k) for k in ("kernel", "frac", "achieved", "avg_launch_ms", "launches_per_step")})
print(d["roofline"].get("top"))
print(d.get("tsdf", {}).get("value"), d.get("tandem_loop", {}).get("keyframes_per_s"), d.get("shipped_model", {}).get("engines_3"))
PY
