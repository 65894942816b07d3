#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_mvsnet_gpu.py tests/test_shim.py -q --no-header -p no:cacheprovider -m gpu --maxfail=5 > gpurun_out/r3y_mvs.log 2>&1; grep -E "passed|failed|^FAILED|^ERROR" gpurun_out/r3y_mvs.log | tail -8
for r in 1 2; do echo "--- $(DR_MVS_NO_SIDE_STREAM=1 timeout 300 python tools/profile_ops.py 'prob|filter|regress' 2>&1 | grep -v amdgpu.ids | tail -1)"; done
