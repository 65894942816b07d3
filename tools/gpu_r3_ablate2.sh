#!/bin/bash
# Timing ablations of the row march (libraries from tools/build_ablate.sh): which part of a step costs what.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export DR_CONV_NO_TUNED=1 DR_CONV_ROWMARCH=2
for v in FULL NO_WAIT NO_EPI NO_DMA NO_KLOOP FREE; do
  lib=$PWD/build/libdr_mabl_$v.so; [ $v = FULL ] && lib=$PWD/tandem_amd/libdr_mi355x.so
  echo "--- $v"; DR_MI355X_LIB=$lib timeout 300 python tools/profile_ops.py 'fn.conv0.1|fn.conv1.1|fn.conv2.1|fn.out2' 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/r3_ablate_rows.txt
