#!/bin/bash
# Round 3: the row march of the 2-D layers -- correctness, then the per-candidate autotune log of FeatureNet's layers.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
timeout 900 python -m pytest tests/test_conv_gpu.py -q --no-header -p no:cacheprovider --maxfail=8 -k "row_march or marching" > $O/r3r_conv.log 2>&1; grep -E "passed|failed|^FAILED|^ERROR" $O/r3r_conv.log | tail -12
DR_AUTOTUNE_ONLY=fn. DR_CONV_NO_TUNED=1 DR_CONV_PRINT=2 timeout 600 python tools/try_autotune.py 400 > $O/r3r_tune.log 2>&1; grep -E "^autotune|before|after" $O/r3r_tune.log | cut -c1-220
grep -E "cand .*rowmarch" $O/r3r_tune.log | sort -k2,2 -k15,15n | cut -c1-150
