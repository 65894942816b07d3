#!/bin/bash
# correctness of the depth pipeline, then the full per-op table
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_conv_gpu.py tests/test_mvsnet_gpu.py -m gpu -q --no-header -p no:cacheprovider -x 2>&1 | grep -E "passed|failed|Error" | tail -4
timeout 300 python tools/profile_ops.py '.' 2>&1 | grep -v amdgpu.ids | tr ' ' '\n' | tee gpurun_out/r2_ops.txt | sort -t= -k2 -n -r | head -70
