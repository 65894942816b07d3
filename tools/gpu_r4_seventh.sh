#!/bin/bash
# Round 4, seventh call: k_conv_w (Winograd F(2,3) along y) on the GPU for the first time -- its conv cases and the end-to-end bounds, a
# per-op A/B with the form preferred everywhere it applies, and autotune passes with its candidates in the ranking (both tuned shapes).
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_conv_gpu.py tests/test_mvsnet_gpu.py -m gpu -q --no-header -p no:cacheprovider -k "winograd" > gpurun_out/r4g_tests.log 2>&1
echo "== winograd cases: $(grep -E 'passed|failed' gpurun_out/r4g_tests.log | tail -1)"; grep -E "^FAILED|^ERROR|Error|max\|err" gpurun_out/r4g_tests.log | head -12
for w in 0 2 0 2; do
  echo "-- untuned plans, DR_CONV_WINO=$w: $(DR_CONV_NO_TUNED=1 DR_CONV_WINO=$w timeout 300 python tools/profile_ops.py 'conv|out|skip' 2>&1 | tail -1 | cut -c1-1600)"
done | tee gpurun_out/r4g_wino_ab.txt
for r in 1 2; do
  DR_CONV_WINO=1 DR_CONV_PRINT=2 timeout 700 python tools/try_autotune.py 400 > gpurun_out/r4g_tune_headline_$r.txt 2>&1
  grep -E "^before|^after|^autotune\(" gpurun_out/r4g_tune_headline_$r.txt | tr '\n' ' '; echo; grep "^TUNED" gpurun_out/r4g_tune_headline_$r.txt
done
for r in 1 2; do
  DR_CONV_WINO=1 DR_CONV_PRINT=2 timeout 700 python tools/try_autotune.py 400 320 512 48,4,4 > gpurun_out/r4g_tune_shipped_$r.txt 2>&1
  grep -E "^before|^after|^autotune\(" gpurun_out/r4g_tune_shipped_$r.txt | tr '\n' ' '; echo; grep "^TUNED" gpurun_out/r4g_tune_shipped_$r.txt
done
