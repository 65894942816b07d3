#!/bin/bash
# fused-skip staging batch size (DR_FZ_BATCH builds in build/): fn.out3 inside a forward
cd "$(dirname "$0")/.."
export DR_CONV_NO_TUNED=1
for v in 4 6 10; do
  lib=$PWD/build/libdr_fzb_$v.so; [ $v = 4 ] && lib=$PWD/tandem_amd/libdr_mi355x.so
  echo "--- DR_FZ_BATCH=$v"; DR_MI355X_LIB=$lib timeout 300 python tools/profile_ops.py 'fn.out3|fn.out2' 2>&1 | grep -v amdgpu.ids
  DR_MVS_NO_SIDE_STREAM=1 DR_MI355X_LIB=$lib timeout 300 python tools/profile_ops.py 'fn.out3|fn.out2' 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/r3_fzb.txt
