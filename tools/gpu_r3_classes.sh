#!/bin/bash
# Round 3: k_conv_a with several parity classes per staged tile -- correctness, then the autotuner's verdict on the layers it applies to.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_conv_gpu.py -q --no-header -p no:cacheprovider --maxfail=5 -k "parity_forms or upsampled or conv_matches or every_plan" > gpurun_out/r3c_conv.log 2>&1; grep -E "passed|failed|^FAILED|^ERROR" gpurun_out/r3c_conv.log | tail -6
timeout 900 python -m pytest tests/test_mvsnet_gpu.py -q --no-header -p no:cacheprovider -m gpu --maxfail=5 -k "folded or golden" > gpurun_out/r3c_mvs.log 2>&1; grep -E "passed|failed|^FAILED|^ERROR" gpurun_out/r3c_mvs.log | tail -4
DR_CONV_NO_TUNED=1 DR_CONV_PRINT=1 timeout 600 python tools/try_autotune.py 400 2> gpurun_out/r3c_tune.err | tail -3
grep -E "^autotune (s[123].conv(7|9|11)|fn.out3b)|^TUNED.*(conv7|conv9|conv11|out3b)" gpurun_out/r3c_tune.err | cut -c1-190
