#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for cfg in "3 0" "3 1" "4 1" "2 1" "4 0"; do
  set -- $cfg
  E=""; [ $2 = 1 ] && E="DR_MVS_NO_SIDE_STREAM=1"
  env $E timeout 300 python bench.py --steps 60 --warmup 10 --engines $1 --no-cpu --no-tsdf --no-boundary --no-loop 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('engines $1 no_side_stream $2: value', round(d['value'],1), 'ms_per_step', round(d['ms_per_step'],4), d['single_engine']['ms_per_depth_map'])"
done
