#!/bin/bash
# Round 4, thirteenth call: conv + depth suites on the grouped epilogues, then a last autotune round of both tuned shapes (two runs each).
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_conv_gpu.py tests/test_mvsnet_gpu.py -m gpu -q --no-header -p no:cacheprovider -k "not bf16x3 and not lds_staged" > gpurun_out/r4m_tests.log 2>&1
echo "== conv + mvsnet suites: $(grep -E 'passed|failed' gpurun_out/r4m_tests.log | tail -1)"; grep -E "^FAILED|^ERROR" gpurun_out/r4m_tests.log | head -8
for r in 1 2; do
  DR_CONV_PRINT=1 timeout 700 python tools/try_autotune.py 400 > gpurun_out/r4m_tune_headline_$r.txt 2>&1
  grep -E "^before|^after|^autotune\(" gpurun_out/r4m_tune_headline_$r.txt | tr '\n' ' '; echo; grep "^TUNED" gpurun_out/r4m_tune_headline_$r.txt
done
for r in 1 2; do
  DR_CONV_PRINT=1 timeout 700 python tools/try_autotune.py 400 320 512 48,4,4 > gpurun_out/r4m_tune_shipped_$r.txt 2>&1
  grep -E "^before|^after|^autotune\(" gpurun_out/r4m_tune_shipped_$r.txt | tr '\n' ' '; echo; grep "^TUNED" gpurun_out/r4m_tune_shipped_$r.txt
done
