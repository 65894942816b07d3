#!/bin/bash
# Timing ablations of k_conv_m (libraries built by tools/build_ablate.sh into build/): which part of the kernel costs what.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export DR_CONV_NO_TUNED=1 DR_CONV_MARCH=2
for v in FULL NO_WAIT NO_EPI NO_DMA NO_KLOOP FREE; do
  lib=$PWD/build/libdr_mabl_$v.so; [ $v = FULL ] && lib=$PWD/tandem_amd/libdr_mi355x.so
  echo "--- $v"; DR_MI355X_LIB=$lib timeout 300 python tools/profile_ops.py 's[123].conv0|conv0.1|s[23].conv2$' 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/r3c_ablate.txt
