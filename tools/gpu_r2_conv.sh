#!/bin/bash
# conv iteration: correctness (conv cases + pipeline fixtures), then per-op profile A/B via env hooks
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_conv_gpu.py tests/test_mvsnet_gpu.py -m gpu -q --no-header -p no:cacheprovider -x 2>&1 | tail -6
for v in "X=1" "${AB:-DR_COSTVOL_CPL=4}" "X=1" "${AB:-DR_COSTVOL_CPL=4}"; do
  echo "--- $v"; env $v timeout 300 python tools/profile_ops.py "${1:-conv|out|skip}" 2>&1 | grep -v amdgpu.ids
done
