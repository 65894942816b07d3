#!/bin/bash
# Generic A/B: the tree's library against build/libdr_prev.so, per-op profile of a sequential forward (3 rounds), conv tests first.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_conv_gpu.py -q --no-header -p no:cacheprovider --maxfail=5 -x > gpurun_out/r3ab_conv.log 2>&1; grep -E "passed|failed|^FAILED|^ERROR" gpurun_out/r3ab_conv.log | tail -6
for v in new prev new prev new prev; do
  lib=$PWD/tandem_amd/libdr_mi355x.so; [ $v = prev ] && lib=$PWD/build/libdr_prev.so
  echo "--- $v: $(DR_MVS_NO_SIDE_STREAM=1 DR_MI355X_LIB=$lib timeout 300 python tools/profile_ops.py "${PAT:-zzz}" 2>&1 | grep -v amdgpu.ids | tail -1)"
done | tee gpurun_out/r3_ab.txt
