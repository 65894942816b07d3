#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_mvsnet_gpu.py -q --no-header -p no:cacheprovider -m gpu --maxfail=5 -k "golden or full_size or plain_variance" > gpurun_out/r3q_mvs.log 2>&1; grep -E "passed|failed|^FAILED|^ERROR" gpurun_out/r3q_mvs.log | tail -8
for r in 1 2; do echo "--- $(DR_MVS_NO_SIDE_STREAM=1 timeout 300 python tools/profile_ops.py 'costvol' 2>&1 | grep -v amdgpu.ids | tail -1)"; done
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu --no-tsdf --no-boundary --no-loop 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('value', round(d['value'],1), 'ms_per_step', round(d['ms_per_step'],4), d['repeats'], d['single_engine'])"
