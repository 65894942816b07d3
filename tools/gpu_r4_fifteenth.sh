#!/bin/bash
# Round 4, fifteenth call: the boundary's host copies split with a helper thread -- protocol / pipelining / concurrency cases and the boundary leg.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_mvsnet_gpu.py tests/test_shim.py tests/test_view_shard_gpu.py -m gpu -q --no-header -p no:cacheprovider -k "protocol or pipelined or concurrent or pinned or golden_fixture or shim or resolution or ref_index or sharded" > gpurun_out/r4o_tests.log 2>&1
echo "== boundary cases: $(grep -E 'passed|failed' gpurun_out/r4o_tests.log | tail -1)"; grep -E "^FAILED|^ERROR" gpurun_out/r4o_tests.log | head -8
for i in 1 2; do
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu --no-tsdf --no-loop > gpurun_out/r4o_bench_$i.json 2> gpurun_out/r4o_bench.err
python - <<PY
import json
d = json.loads(open("gpurun_out/r4o_bench_$i.json").read().strip().splitlines()[-1])
print({k: d.get(k) for k in ("value", "single_window_ms", "boundary_single_engine_ms", "boundary_pinned_single_engine_ms")}, d["boundary"]["engines_1"]["call_async_ms"], d["boundary"]["engines_3"]["depth_maps_per_s"])
PY
done
