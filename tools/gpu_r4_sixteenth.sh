#!/bin/bash
# Round 4, sixteenth call: k_conv_a issues the next unit's DMA before a store-only epilogue: conv suite, within-box A/B against the old order.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_conv_gpu.py -m gpu -q --no-header -p no:cacheprovider -k "not bf16x3" > gpurun_out/r4p_tests.log 2>&1
echo "== conv suite: $(grep -E 'passed|failed' gpurun_out/r4p_tests.log | tail -1)"; grep -E "^FAILED|^ERROR" gpurun_out/r4p_tests.log | head -8
for lib in build/ab/libdr_issuelate.so "" build/ab/libdr_issuelate.so ""; do
  echo "-- lib '$lib': $(env ${lib:+DR_MI355X_LIB=$PWD/$lib} timeout 300 python tools/profile_ops.py 'conv1$|fn.conv0.0|fn.conv1.0|fn.conv2.0|skip2' 2>&1 | tail -1 | cut -c1-900)"
  echo "      bench $(env ${lib:+DR_MI355X_LIB=$PWD/$lib} timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu --no-tsdf --no-boundary --no-loop 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print("%.1f /s, single %.3f ms" % (d["value"], d["single_window_ms"]))')"
done | tee gpurun_out/r4p_issue_ab.txt
