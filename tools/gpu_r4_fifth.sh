#!/bin/bash
# Round 4, fifth call: neighbour-tap borrowing in k_costvol3 (bit-identical to k_costvol2 by the existing test; A/B against
# DR_COSTVOL_NO_BORROW=1), the allocation pass of scan k + 1 beside the ray-cast of scan k (fusion suite: bit-exact against the oracle
# on the bench workload; the probe's per-frame split), the operator boundary with the per-view upload pipeline.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_mvsnet_gpu.py tests/test_fusion_gpu.py tests/test_view_shard_gpu.py -m gpu -q --no-header -p no:cacheprovider -k "cost_volume or golden or fusion or shard or raycast or full_size" -x > gpurun_out/r4e_tests.log 2>&1
echo "== suites: $(grep -E 'passed|failed' gpurun_out/r4e_tests.log | tail -1)"; grep -E "^FAILED|^ERROR" gpurun_out/r4e_tests.log | head -8
for v in "DR_COSTVOL_NO_BORROW=1" "X=1" "DR_COSTVOL_NO_BORROW=1" "X=1"; do
  echo "-- $v: $(env $v DR_MVS_NO_SIDE_STREAM=1 timeout 300 python tools/profile_ops.py 'costvol' 2>&1 | tail -1)"
done | tee gpurun_out/r4e_costvol_ab.txt
for r in 1 2; do echo "-- tsdf probe: $(timeout 300 python tools/gpu_tsdf_probe.py 200 2>&1 | grep '^lap [12]' | tr '\n' '|')"; done | tee gpurun_out/r4e_tsdf.txt
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu --no-loop --tsdf-frames 300 --no-tsdf-native > gpurun_out/r4e_bench.json 2> gpurun_out/r4e_bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4e_bench.json").readlines()[-1])
print({k: d.get(k) for k in ("value", "ms_per_step", "single_window_ms", "boundary_single_engine_ms")}, d["tsdf"]["value"], d["tsdf"]["ms_per_frame"], d["tsdf"]["kernel_ms_per_frame"], d["boundary"]["engines_1"], d.get("bf16x3_mode"))
PY
