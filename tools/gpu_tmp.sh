#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_mvsnet_gpu.py -q --no-header -p no:cacheprovider -m gpu --maxfail=5 -k "shared_setup or golden or full_size" > gpurun_out/r3q_mvs.log 2>&1; grep -E "passed|failed|^FAILED|^ERROR|Error" gpurun_out/r3q_mvs.log | tail -8
for v in 3 2 3 2; do E=""; [ $v = 2 ] && E="DR_COSTVOL_V2=1"; echo "--- costvol$v $(env $E DR_MVS_NO_SIDE_STREAM=1 timeout 300 python tools/profile_ops.py 'costvol' 2>&1 | grep -v amdgpu.ids | tail -1)"; done
