#!/bin/bash
# Round 3: depth-maps/s against the number of independent windows in flight per GPU.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
: > gpurun_out/r3_engines.txt
for E in 1 2 3 4 5 6; do
  timeout 300 python bench.py --steps 120 --warmup 10 --engines $E --no-cpu --no-tsdf --no-boundary --no-loop 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('engines', $E, 'value', round(d['value'],1), 'ms_per_step', round(d['ms_per_step'],4), d.get('repeats'))" | tee -a gpurun_out/r3_engines.txt
done
