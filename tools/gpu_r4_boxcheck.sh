#!/bin/bash
# Which kind of box is this?  The depth leg of the bench line only (value + single window), twice; GPU clocks as rocm-smi reports them.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
rocm-smi --showclocks --showpower --showperflevel 2>/dev/null | grep -E "sclk|mclk|fclk|Power|Performance" | head -8
for i in 1 2; do timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu --no-tsdf --no-boundary --no-loop 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print("%.1f /s, single %.3f ms" % (d["value"], d["single_window_ms"]))'; done
rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power" | head -4
