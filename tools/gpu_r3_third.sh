#!/bin/bash
# Round 3, third GPU run: k_conv_m ablations, the fused-skip marching path, per-layer times with the new tuned table,
# headline bench (depth legs only).
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
bash tools/gpu_r3_ablate.sh 2>&1 | tail -14
timeout 600 python -m pytest tests/test_mvsnet_gpu.py -q --no-header -p no:cacheprovider --maxfail=8 -k "fused or golden or headline or full" > $O/r3c_mvs.log 2>&1; grep -E "passed|failed|^FAILED|^ERROR|Error" $O/r3c_mvs.log | tail -8
timeout 300 python tools/profile_ops.py . > $O/r3c_ops.txt 2>&1; tr ' ' '\n' < $O/r3c_ops.txt | grep -E "out3|conv0=|costvol|forward|ms" | tr '\n' ' '; echo
DR_FZ_NO_MARCH=1 timeout 300 python tools/profile_ops.py out3 2>&1 | tail -1
timeout 600 python bench.py --steps 120 --no-cpu --no-tsdf --no-loop --no-boundary > $O/r3c_bench.json 2> $O/r3c_bench.err; python - <<'PY'
import json
d = json.load(open("gpurun_out/r3c_bench.json"))
print("value", d["value"], "ms/step", d["ms_per_step"], "single", d.get("single_engine"), "roofline", d.get("roofline"))
PY
