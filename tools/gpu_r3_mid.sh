#!/bin/bash
# Round 3, mid-round check: the whole -m gpu suite, smoke, the driver's bench command.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
timeout 1800 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider --durations=12 > $O/r3m_gpu_tests.log 2>&1; grep -E "passed|failed|^FAILED|^ERROR" $O/r3m_gpu_tests.log | tail -12
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 1200 python bench.py --steps 20 --warmup 5 > $O/r3m_bench.json 2> $O/r3m_bench.err; echo "bench rc=$?"; python - <<'PY'
import json
d = json.load(open("gpurun_out/r3m_bench.json"))
print({k: d[k] for k in ("value", "ms_per_step", "repeats")}, d["single_engine"], d["roofline"]["kernel"], round(d["roofline"]["frac"], 3))
print("tsdf", d["tsdf"]["value"], d["tsdf"]["ms_per_frame"], d["tsdf"]["kernel_ms_per_frame"], d["tsdf"].get("cpu_baseline", {}).get("openmp"))
print("loop", {k: (v.get("keyframes_per_s"), v.get("ms_per_keyframe")) for k, v in d["tandem_loop"].items() if isinstance(v, dict)})
print("tracker", d["tracker"]["calc_res_ms"], d["tracker"]["calc_g_ms"], d["tracker"]["per_call"])
print("cpu", d["cpu_baseline"])
PY
