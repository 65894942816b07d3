#!/bin/bash
# k_conv_a iteration: correctness over the conv cases and every plan candidate (async family forced), then the per-op table of the big layers
cd "$(dirname "$0")/.."
DR_CONV_ASYNC=1 timeout 600 python -m pytest tests/test_conv_gpu.py -m gpu -q --no-header -p no:cacheprovider -x 2>&1 | grep -E "passed|failed|Error|assert" | tail -4
timeout 600 python -m pytest tests/test_conv_gpu.py tests/test_mvsnet_gpu.py -m gpu -q --no-header -p no:cacheprovider -x 2>&1 | grep -E "passed|failed|Error|assert" | tail -4
for i in 1 2; do timeout 300 python tools/profile_ops.py "conv0|conv6|forward" 2>&1 | grep -v amdgpu.ids; done
