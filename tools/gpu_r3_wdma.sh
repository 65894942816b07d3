#!/bin/bash
# Round 3: k_conv's packed weights by LDS-DMA ahead of the tile staging -- correctness, then A/B per layer (build/libdr_nowdma.so = the register path).
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_conv_gpu.py -q --no-header -p no:cacheprovider --maxfail=5 > gpurun_out/r3w_conv.log 2>&1; grep -E "passed|failed|^FAILED|^ERROR" gpurun_out/r3w_conv.log | tail -6
for v in dma regs dma regs; do
  lib=$PWD/tandem_amd/libdr_mi355x.so; [ $v = regs ] && lib=$PWD/build/libdr_nowdma.so
  echo "--- weights by $v"; DR_MVS_NO_SIDE_STREAM=1 DR_MI355X_LIB=$lib timeout 300 python tools/profile_ops.py 'conv[1-9]|conv11|fn.conv2.0|fn.out3' 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/r3_wdma.txt
