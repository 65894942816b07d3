#!/bin/bash
# rocprofv3 --kernel-trace --stats of the DRIVER'S OWN bench command (3 engines in flight), for the cross-check of roofline.avg_launch_ms.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf gpurun_out/prof_driver
timeout 1200 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_driver -o bench -- python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04_bench_driver_under_rocprof.json 2> gpurun_out/prof_driver.err
python tools/rocprof_summary.py $(find gpurun_out/prof_driver -name "*_results.db" | head -1) > gpurun_out/r04_bench_kernel_stats_driver_cmd.txt 2>&1; head -16 gpurun_out/r04_bench_kernel_stats_driver_cmd.txt
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04_bench_driver_under_rocprof.json").read().strip().splitlines()[-1])
print({k: d.get(k) for k in ("value", "ms_per_step")}, {k: d["roofline"].get(k) for k in ("kernel", "avg_launch_ms", "frac")})
PY
rm -rf gpurun_out/prof_driver
