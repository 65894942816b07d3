#!/bin/bash
# Round 3: out.stage3 folded (composed 8->8 layer + phase layers over inter2 at half resolution + border term) -- unit tests of the up2 layer,
# the depth-pipeline suite, then timing against the fused-skip form.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_conv_gpu.py -q --no-header -p no:cacheprovider -k "upsampled" > gpurun_out/r3z_up2.log 2>&1; grep -E "passed|failed|^FAILED|^ERROR|Error" gpurun_out/r3z_up2.log | tail -6
timeout 1500 python -m pytest tests/test_mvsnet_gpu.py tests/test_shim.py -q --no-header -p no:cacheprovider -m gpu --maxfail=5 > gpurun_out/r3z_mvs.log 2>&1; grep -E "passed|failed|^FAILED|^ERROR" gpurun_out/r3z_mvs.log | tail -8
for v in 1 0 1 0; do
  echo "--- folded=$v: $(DR_OUT3_FOLDED=$v DR_MVS_NO_SIDE_STREAM=1 timeout 300 python tools/profile_ops.py 'fn.out3|fn.out2' 2>&1 | grep -v amdgpu.ids | tail -1)"
done | tee gpurun_out/r3_fold.txt
DR_AUTOTUNE_ONLY=fn.out3 DR_CONV_NO_TUNED=1 DR_CONV_PRINT=1 timeout 600 python tools/try_autotune.py 400 2>&1 | grep -E "^autotune|^TUNED|before|after"
