#!/bin/bash
# Round 4: first GPU contact of k_conv_b (csrc/conv_bf3.h, DR_CONV_BF16X3=1) -- the gated unit and end-to-end cases, then its time
# against the fp32 engine (single window, profile_ops per layer).
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
DR_TEST_BF16X3=1 timeout 900 python -m pytest tests/test_conv_gpu.py -q --no-header -p no:cacheprovider -k "bf16x3" --maxfail=8 > gpurun_out/r4b_conv.log 2>&1
grep -E "passed|failed|^FAILED|^ERROR|Error" gpurun_out/r4b_conv.log | tail -12
DR_TEST_BF16X3=1 timeout 900 python -m pytest tests/test_mvsnet_gpu.py -q --no-header -p no:cacheprovider -k "bf16x3" --maxfail=4 > gpurun_out/r4b_mvs.log 2>&1
grep -E "passed|failed|^FAILED|^ERROR|Error" gpurun_out/r4b_mvs.log | tail -8
for v in "" 1 "" 1; do
  echo "--- bf16x3='$v': $(env ${v:+DR_CONV_BF16X3=$v} DR_MVS_NO_SIDE_STREAM=1 timeout 300 python tools/profile_ops.py '.' 2>&1 | grep -v amdgpu.ids | tail -1)"
  echo "    bench: $(env ${v:+DR_CONV_BF16X3=$v} timeout 600 python bench.py --steps 100 --warmup 10 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d["value"], d["ms_per_step"], d["roofline"]["top"][:3])')"
done | tee gpurun_out/r4b_ab.txt
