#!/bin/bash
# Round 4, first call: the one-launch form of the folded stage-3 head's phase layer (DR_OUT3_ONE_LAUNCH=1; profiles/r03_experiments.txt, 18) --
# the suites that cover it, its tuned row, then the A/B against the two-launch default on the same box.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
DR_OUT3_ONE_LAUNCH=1 timeout 1500 python -m pytest tests/test_mvsnet_gpu.py tests/test_shim.py -q --no-header -p no:cacheprovider -m gpu --maxfail=5 > gpurun_out/r4a_mvs.log 2>&1
grep -E "passed|failed|^FAILED|^ERROR" gpurun_out/r4a_mvs.log | tail -8
DR_OUT3_ONE_LAUNCH=1 DR_AUTOTUNE_ONLY=fn.out3 DR_CONV_NO_TUNED=1 DR_CONV_PRINT=1 timeout 600 python tools/try_autotune.py 400 2>&1 | grep -E "^autotune|^TUNED|before|after" | tee gpurun_out/r4a_tune.txt
for v in "" 1 "" 1; do
  echo "--- one_launch='$v': $(env ${v:+DR_OUT3_ONE_LAUNCH=$v} DR_MVS_NO_SIDE_STREAM=1 timeout 300 python tools/profile_ops.py 'fn.out3' 2>&1 | grep -v amdgpu.ids | tail -1)"
  echo "    bench: $(env ${v:+DR_OUT3_ONE_LAUNCH=$v} timeout 600 python bench.py --steps 100 --warmup 10 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d["value"], d["ms_per_step"])')"
done | tee gpurun_out/r4a_ab.txt
