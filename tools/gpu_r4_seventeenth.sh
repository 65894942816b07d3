#!/bin/bash
# Round 4, seventeenth call: the tile ring on the GPU for the first time -- its conv cases, then autotune with its candidates ranked (both shapes).
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_conv_gpu.py -m gpu -q --no-header -p no:cacheprovider -k "ring or marching or row_march" > gpurun_out/r4q_tests.log 2>&1
echo "== ring / march cases: $(grep -E 'passed|failed' gpurun_out/r4q_tests.log | tail -1)"; grep -E "^FAILED|^ERROR|Error" gpurun_out/r4q_tests.log | head -8
for r in 1 2; do
  DR_CONV_PRINT=2 timeout 700 python tools/try_autotune.py 400 > gpurun_out/r4q_tune_headline_$r.txt 2>&1
  grep -E "^before|^after|^autotune\(" gpurun_out/r4q_tune_headline_$r.txt | tr '\n' ' '; echo; grep "^TUNED" gpurun_out/r4q_tune_headline_$r.txt
done
for l in s2.conv1 s3.conv1 fn.conv1.0 fn.conv2.0; do grep -E "cand $l " gpurun_out/r4q_tune_headline_1.txt | sort -t: -k2 -n | head -4; done
for r in 1 2; do
  DR_CONV_PRINT=2 timeout 700 python tools/try_autotune.py 400 320 512 48,4,4 > gpurun_out/r4q_tune_shipped_$r.txt 2>&1
  grep -E "^before|^after|^autotune\(" gpurun_out/r4q_tune_shipped_$r.txt | tr '\n' ' '; echo; grep "^TUNED" gpurun_out/r4q_tune_shipped_$r.txt
done
