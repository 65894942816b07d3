#!/bin/bash
# Round 4, sixth call: the narrow-wave K loop (operands two chunks ahead, two accumulators) -- conv suite, A/B against a build without
# it -- and autotune passes at both tuned shapes on top of it.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_conv_gpu.py tests/test_mvsnet_gpu.py -m gpu -q --no-header -p no:cacheprovider -k "not bf16x3 and not lds_staged" > gpurun_out/r4f_tests.log 2>&1
echo "== conv + mvsnet suites: $(grep -E 'passed|failed' gpurun_out/r4f_tests.log | tail -1)"; grep -E "^FAILED|^ERROR" gpurun_out/r4f_tests.log | head -8
for lib in build/ab/libdr_nonarrow.so "" build/ab/libdr_nonarrow.so ""; do
  echo "-- lib '$lib': $(env ${lib:+DR_MI355X_LIB=$PWD/$lib} timeout 300 python tools/profile_ops.py 'conv[3-7]|conv9|conv1$' 2>&1 | tail -1 | cut -c1-900)"
  echo "      bench $(env ${lib:+DR_MI355X_LIB=$PWD/$lib} timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu --no-tsdf --no-boundary --no-loop 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print("%.1f /s, single %.3f ms" % (d["value"], d["single_window_ms"]))')"
done | tee gpurun_out/r4f_narrow_ab.txt
for r in 1 2; do
  DR_CONV_PRINT=1 timeout 600 python tools/try_autotune.py 400 320 512 48,4,4 > gpurun_out/r4f_tune_shipped_$r.txt 2>&1
  grep -E "^before|^after|^autotune\(" gpurun_out/r4f_tune_shipped_$r.txt | tr '\n' ' '; echo; grep "^TUNED" gpurun_out/r4f_tune_shipped_$r.txt
done
DR_CONV_PRINT=1 timeout 600 python tools/try_autotune.py 400 > gpurun_out/r4f_tune_headline.txt 2>&1
grep -E "^before|^after|^autotune\(" gpurun_out/r4f_tune_headline.txt | tr '\n' ' '; echo; grep "^TUNED" gpurun_out/r4f_tune_headline.txt
