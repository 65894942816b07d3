#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for fb in 2 4; do
  export DR_MI355X_LIB=$PWD/build_ab/lib_fb$fb.so
  echo "=== FB=$fb"
  timeout 300 python -m pytest tests/test_mvsnet_gpu.py -m gpu -q --no-header -p no:cacheprovider -x -k "fused_skip" --durations=3 2>&1 | grep -E "passed|failed|Error|assert|s call" | tail -6
  timeout 300 python tools/profile_ops.py "fn.out|fn.skip" 2>&1 | grep -v amdgpu.ids
  DR_CONV_PRINT=1 DR_AUTOTUNE_ONLY=fn.out3 timeout 300 python tools/try_autotune.py 40 2>&1 | grep -E "autotune|TUNED|before|after"
done
