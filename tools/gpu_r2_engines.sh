#!/bin/bash
# depth-maps/s against the number of windows in flight (DrMvsnet engines per GPU)
cd "$(dirname "$0")/.."
for e in 1 2 3 4 6; do
  echo -n "engines $e: "; timeout 300 python bench.py --steps 240 --warmup 6 --engines $e --no-tsdf --no-boundary --no-loop --no-cpu 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), 'depth-maps/s', round(d['ms_per_step'],3), 'ms')"
done
