#!/bin/bash
# k_conv ablations (DR_ABL_NO_STAGE / DR_ABL_NO_KLOOP builds in build/): where fn.out3's time goes
cd "$(dirname "$0")/.."
export DR_CONV_NO_TUNED=1 DR_MVS_NO_SIDE_STREAM=1
for v in FULL NO_STAGE NO_KLOOP; do
  lib=$PWD/build/libdr_abl_$v.so; [ $v = FULL ] && lib=$PWD/tandem_amd/libdr_mi355x.so
  echo "--- $v"; DR_MI355X_LIB=$lib timeout 300 python tools/profile_ops.py 'fn.out3|s2.conv1$|s2.conv9|s2.conv11|s2.conv6' 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/r3_fzabl.txt
