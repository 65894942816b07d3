#!/bin/bash
# k_conv_a (persistent, LDS-DMA staged): correctness over the conv cases and every plan candidate, then per-op A/B
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
DR_CONV_ASYNC=1 timeout 600 python -m pytest tests/test_conv_gpu.py -m gpu -q --no-header -p no:cacheprovider -x --durations=3 2>&1 | grep -E "passed|failed|Error|assert|s call" | tail -8
for v in 0 1; do
  echo "--- DR_CONV_ASYNC=$v"; DR_CONV_ASYNC=$v timeout 300 python tools/profile_ops.py "conv|out|skip" 2>&1 | grep -v amdgpu.ids | tr ' ' '\n' | sort -t= -k2 -n -r | head -${TOP:-24} | tr '\n' ' '; echo
done
