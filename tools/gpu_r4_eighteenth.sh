#!/bin/bash
# Round 4, eighteenth call: stage 1's cost volume as two 16-channel halves -- depth / view-shard suites, A/B against the interleaved layout.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_mvsnet_gpu.py tests/test_view_shard_gpu.py tests/test_shim.py -m gpu -q --no-header -p no:cacheprovider -k "not bf16x3" > gpurun_out/r4r_tests.log 2>&1
echo "== depth / shard suites: $(grep -E 'passed|failed' gpurun_out/r4r_tests.log | tail -1)"; grep -E "^FAILED|^ERROR" gpurun_out/r4r_tests.log | head -8
for v in DR_VOL_NO_SPLIT X DR_VOL_NO_SPLIT X; do
  echo "-- $v=1: $(env $v=1 timeout 300 python tools/profile_ops.py 's1.costvol|s1.conv0' 2>&1 | tail -1 | cut -c1-300)"
  echo "      bench $(env $v=1 timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu --no-tsdf --no-boundary --no-loop 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print("%.1f /s, single %.3f ms" % (d["value"], d["single_window_ms"]))')"
done | tee gpurun_out/r4r_split_ab.txt
for v in DR_VOL_NO_SPLIT X; do echo "-- shipped, $v=1: $(env $v=1 timeout 300 python bench.py --config shipped --steps 120 --no-tsdf --no-loop --no-cpu --no-boundary 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print("%.1f /s, single %.3f ms" % (d["value"], d["single_window_ms"]))')"; done | tee -a gpurun_out/r4r_split_ab.txt
