#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -m pytest tests/test_fusion_gpu.py -m gpu -q --no-header -p no:cacheprovider -x 2>&1 | tail -8
python bench.py --steps 5 --warmup 2 --no-cpu --tsdf-scans 25 --tsdf-cycles 8 > gpurun_out/bench_quick.json 2>gpurun_out/bench_quick.err
python - <<'PY'
import json
b=json.load(open('gpurun_out/bench_quick.json'))
t=b['tsdf']; print('mvs', b['value'], 'ms', b['ms_per_step']); print({k:t[k] for k in ('value','ms_per_scan','voxels_per_scan','blocks','raycast_ms_incl_d2h')}); print(t['roofline'])
PY
