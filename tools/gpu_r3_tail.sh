#!/bin/bash
# Round 3: register-resident regression -- the whole depth-pipeline suite, then timing A/B.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_mvsnet_gpu.py tests/test_shim.py tests/test_view_shard_gpu.py -q --no-header -p no:cacheprovider -m gpu --maxfail=5 > gpurun_out/r3x_mvs.log 2>&1; grep -E "passed|failed|^FAILED|^ERROR" gpurun_out/r3x_mvs.log | tail -8
for v in new old; do
  E=""; [ $v = old ] && E="DR_REGRESS_GENERIC=1"
  echo "--- $v: $(env $E DR_MVS_NO_SIDE_STREAM=1 timeout 300 python tools/profile_ops.py 'regress' 2>&1 | grep -v amdgpu.ids | tail -1)"
done | tee gpurun_out/r3_tail.txt
