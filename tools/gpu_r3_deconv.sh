#!/bin/bash
# Round 3: parity forms of the transposed layers -- correctness, then per-form timing of conv7 / conv9 / conv11 (autotuned per form).
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
timeout 900 python -m pytest tests/test_conv_gpu.py -q --no-header -p no:cacheprovider --maxfail=8 -k "parity_forms or deconv" > $O/r3t_conv.log 2>&1; grep -E "passed|failed|^FAILED|^ERROR" $O/r3t_conv.log | tail -12
for F in 0 1 2; do
  echo "== form $F"
  DR_DECONV_FORM=$F DR_AUTOTUNE_ONLY=conv DR_CONV_NO_TUNED=1 DR_CONV_PRINT=1 timeout 600 python tools/try_autotune.py 400 > $O/r3t_tune_$F.log 2>&1
  grep -E "before|after" $O/r3t_tune_$F.log
  grep -E "^autotune s[123].conv(7|9|11) " $O/r3t_tune_$F.log | cut -c1-200
done
