#!/bin/bash
# conv iteration: correctness (conv cases + pipeline fixtures), then per-op profile (best of 3) twice
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_conv_gpu.py tests/test_mvsnet_gpu.py -m gpu -q --no-header -p no:cacheprovider -x 2>&1 | tail -6
for v in 1 2; do
  timeout 300 python tools/profile_ops.py "${1:-conv0|out3|out2|conv2\$|conv11}" 2>&1 | grep -v amdgpu.ids
done
