#!/bin/bash
# Round 4, twelfth call: residual operands fetched per group of four entries before the group's first store (conv_epilogue): the conv
# suite, then a within-box A/B against a build with the per-entry form.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_conv_gpu.py -m gpu -q --no-header -p no:cacheprovider -k "not bf16x3" > gpurun_out/r4l_tests.log 2>&1
echo "== conv suite: $(grep -E 'passed|failed' gpurun_out/r4l_tests.log | tail -1)"; grep -E "^FAILED|^ERROR" gpurun_out/r4l_tests.log | head -8
for lib in build/ab/libdr_perentry.so "" build/ab/libdr_perentry.so ""; do
  echo "-- lib '$lib': $(env ${lib:+DR_MI355X_LIB=$PWD/$lib} timeout 300 python tools/profile_ops.py 'conv7|conv9|conv11|skip2|out3|out2' 2>&1 | tail -1 | cut -c1-900)"
  echo "      bench $(env ${lib:+DR_MI355X_LIB=$PWD/$lib} timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu --no-tsdf --no-boundary --no-loop 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print("%.1f /s, single %.3f ms" % (d["value"], d["single_window_ms"]))')"
done | tee gpurun_out/r4l_epilogue_ab.txt
