#!/bin/bash
# Round 3, fifth GPU run: 12-consumer-wave marching instances -- correctness, then the per-candidate autotune log.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
timeout 900 python -m pytest tests/test_conv_gpu.py -q --no-header -p no:cacheprovider --maxfail=8 -k "march or sweep" > $O/r3e_conv.log 2>&1; grep -E "passed|failed|^FAILED|^ERROR" $O/r3e_conv.log | tail -8
DR_CONV_NO_TUNED=1 DR_CONV_PRINT=2 timeout 600 python tools/try_autotune.py 400 > $O/r3e_tune.log 2>&1; grep -E "^autotune.*(conv0|conv2 |out3|out2|conv0.1|conv1.1)|before|after" $O/r3e_tune.log | cut -c1-200
grep -E "cand (s1.conv0|s2.conv0|s3.conv0|fn.out3) .*march" $O/r3e_tune.log | sort -k2,2 -k15,15n | cut -c1-140
