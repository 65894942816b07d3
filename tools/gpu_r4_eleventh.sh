#!/bin/bash
# Round 4, eleventh call: the radix select's scan moved into the histogram kernel's last workgroup (9 -> 5 launches per edge filter):
# the filter's exactness cases, the fixtures, per-op times.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_mvsnet_gpu.py -m gpu -q --no-header -p no:cacheprovider -k "edge_filter or golden_fixture or full_size or pipelined or concurrent or textureless" > gpurun_out/r4k_tests.log 2>&1
echo "== filter / fixture cases: $(grep -E 'passed|failed' gpurun_out/r4k_tests.log | tail -1)"; grep -E "^FAILED|^ERROR" gpurun_out/r4k_tests.log | head -8
for i in 1 2; do timeout 300 python tools/profile_ops.py 'filter|regress|prob' 2>&1 | tail -1 | cut -c1-900; done | tee gpurun_out/r4k_ops.txt
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu --no-tsdf --no-boundary --no-loop 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print("%.1f /s, single %.3f ms" % (d["value"], d["single_window_ms"]))'
