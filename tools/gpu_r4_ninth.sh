#!/bin/bash
# Round 4, ninth call: autotune with the rank permutation fixed (the cost model's own first choice of a tuned layer -- the Winograd march for
# s1.conv0 / s2.conv0 -- is now timed), both tuned shapes twice; bench line.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
for r in 1 2; do
  DR_CONV_PRINT=2 timeout 700 python tools/try_autotune.py 400 > gpurun_out/r4i_tune_headline_$r.txt 2>&1
  grep -E "^before|^after|^autotune\(" gpurun_out/r4i_tune_headline_$r.txt | tr '\n' ' '; echo; grep "^TUNED" gpurun_out/r4i_tune_headline_$r.txt
done
for l in s1.conv0 s2.conv0 s1.conv2; do grep -E "cand $l " gpurun_out/r4i_tune_headline_1.txt | sort -t: -k2 -n | head -5; done
for r in 1 2; do
  DR_CONV_PRINT=2 timeout 700 python tools/try_autotune.py 400 320 512 48,4,4 > gpurun_out/r4i_tune_shipped_$r.txt 2>&1
  grep -E "^before|^after|^autotune\(" gpurun_out/r4i_tune_shipped_$r.txt | tr '\n' ' '; echo; grep "^TUNED" gpurun_out/r4i_tune_shipped_$r.txt
done
for l in s1.conv0 s2.conv0 s3.conv0; do grep -E "cand $l " gpurun_out/r4i_tune_shipped_1.txt | sort -t: -k2 -n | head -4; done
timeout 300 python tools/profile_ops.py . 2>&1 | tail -1 | cut -c1-2500 | tee gpurun_out/r4i_ops.txt
