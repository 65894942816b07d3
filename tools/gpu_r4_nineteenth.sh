#!/bin/bash
# Round 4, nineteenth call: k_conv_r (residual operands prefetched before the staging) -- conv suite, within-box A/B against k_conv (DR_CONV_NO_RPRE=1).
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_conv_gpu.py -m gpu -q --no-header -p no:cacheprovider -k "not bf16x3" > gpurun_out/r4s_tests.log 2>&1
echo "== conv suite: $(grep -E 'passed|failed' gpurun_out/r4s_tests.log | tail -1)"; grep -E "^FAILED|^ERROR" gpurun_out/r4s_tests.log | head -8
for v in DR_CONV_NO_RPRE X DR_CONV_NO_RPRE X; do
  echo "-- $v=1: $(env $v=1 timeout 300 python tools/profile_ops.py 'conv7|conv9|conv11|skip2|out3b|out3c' 2>&1 | tail -1 | cut -c1-700)"
  echo "      bench $(env $v=1 timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu --no-tsdf --no-boundary --no-loop 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print("%.1f /s, single %.3f ms" % (d["value"], d["single_window_ms"]))')"
done | tee gpurun_out/r4s_rpre_ab.txt
