#!/bin/bash
# Round 3: pipelined DMA producer (m.depth) -- correctness of all marching tests, then A/B of the producer depth on the layers that march.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
timeout 900 python -m pytest tests/test_conv_gpu.py -q --no-header -p no:cacheprovider --maxfail=8 -k "row_march or marching" > $O/r3s_conv.log 2>&1; grep -E "passed|failed|^FAILED|^ERROR" $O/r3s_conv.log | tail -12
for D in 1 2 3 4; do
  echo "== producer depth $D"
  DR_MARCH_PDEPTH=$D DR_AUTOTUNE_ONLY=${ONLY:-conv} DR_CONV_NO_TUNED=1 DR_CONV_PRINT=2 timeout 600 python tools/try_autotune.py 400 > $O/r3s_tune_$D.log 2>&1
  grep -E "before|after" $O/r3s_tune_$D.log
  grep -E "cand .*(rowmarch|march)" $O/r3s_tune_$D.log | awk '{k=$2" "$5; t=$(NF-1); if (!(k in b) || t<b[k]) {b[k]=t; l[k]=$0}} END {for (k in b) print l[k]}' | sort -k2,2 | cut -c1-150
done
