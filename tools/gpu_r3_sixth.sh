#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
timeout 600 python -m pytest tests/test_mvsnet_gpu.py -q --no-header -p no:cacheprovider --maxfail=8 -k "fused" > $O/r3f_mvs.log 2>&1; grep -E "passed|failed|^FAILED|^ERROR|Error" $O/r3f_mvs.log | tail -8
DR_CONV_NO_TUNED=1 DR_CONV_MARCH=2 timeout 300 python tools/profile_ops.py 'out3|conv0' 2>&1 | tail -1
timeout 300 python tools/profile_ops.py 'out3|conv0' 2>&1 | tail -1
