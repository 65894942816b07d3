#!/bin/bash
# Round 3, fourth GPU run: K-loop microbenchmark (prefetch depth / waves / PT), depth-2 marching kernel, LDS-staged k_prob2.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
timeout 120 tools/ubench/march_kloop 2>&1 | tee $O/r3d_ubench.txt
timeout 900 python -m pytest tests/test_conv_gpu.py -q --no-header -p no:cacheprovider --maxfail=8 -k "march or sweep" > $O/r3d_conv.log 2>&1; grep -E "passed|failed|^FAILED|^ERROR" $O/r3d_conv.log | tail -8
timeout 600 python -m pytest tests/test_mvsnet_gpu.py -q --no-header -p no:cacheprovider --maxfail=8 > $O/r3d_mvs.log 2>&1; grep -E "passed|failed|^FAILED|^ERROR|Error" $O/r3d_mvs.log | tail -8
timeout 300 python tools/profile_ops.py . > $O/r3d_ops.txt 2>&1; tr ' ' '\n' < $O/r3d_ops.txt | grep -E "out3|conv0=|conv2=|prob|forward|ms" | tr '\n' ' '; echo
DR_PROB_V1=1 timeout 300 python tools/profile_ops.py prob 2>&1 | tail -1
