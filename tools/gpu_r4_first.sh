#!/bin/bash
# Round 4, first call: (1) the whole -m gpu suite on the tree with ADVICE r3's fixes, (2) the code that round 3 shipped without
# ever running it -- the bf16x3 cases (DR_TEST_BF16X3=1) and the one-launch phase layer (DR_OUT3_ONE_LAUNCH=1) -- with an A/B of each
# against the default on the same box, (3) the driver's bench command on the SURVEY 8(d) depth range, twice (run-to-run spread).
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
Q="--steps 20 --warmup 5 --no-cpu --no-tsdf --no-boundary --no-loop"
line() { python -c 'import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print("value %.1f /s  ms/step %.3f  single %.3f ms  scene-range single %.3f ms  top %s" % (d["value"], d["ms_per_step"], d["single_window_ms"], d["scene_depth_range"]["single_window_ms"], [(t["kernel"], t["ms"], t["frac"]) for t in d["roofline"]["top"][:4]]))'; }
timeout 1500 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider -x > gpurun_out/r4a_gpu_tests.log 2>&1
echo "== gpu suite: $(grep -E 'passed|failed' gpurun_out/r4a_gpu_tests.log | tail -1)"; grep -E "^FAILED|^ERROR" gpurun_out/r4a_gpu_tests.log | head -5
DR_TEST_BF16X3=1 timeout 900 python -m pytest tests/test_conv_gpu.py tests/test_mvsnet_gpu.py -q --no-header -p no:cacheprovider -k "bf16x3" > gpurun_out/r4a_bf3.log 2>&1
echo "== bf16x3 gated cases: $(grep -E 'passed|failed' gpurun_out/r4a_bf3.log | tail -1)"; grep -E "^FAILED|^ERROR" gpurun_out/r4a_bf3.log | head -8
DR_OUT3_ONE_LAUNCH=1 timeout 900 python -m pytest tests/test_mvsnet_gpu.py tests/test_shim.py -q --no-header -p no:cacheprovider -m gpu > gpurun_out/r4a_one.log 2>&1
echo "== one-launch phase layer: $(grep -E 'passed|failed' gpurun_out/r4a_one.log | tail -1)"; grep -E "^FAILED|^ERROR" gpurun_out/r4a_one.log | head -5
for v in "X=1" "DR_OUT3_ONE_LAUNCH=1" "DR_CONV_BF16X3=1" "X=1"; do
  echo "-- $v: $(env $v timeout 400 python bench.py $Q 2>gpurun_out/r4a_ab.err | line)"
done | tee gpurun_out/r4a_ab.txt
python tools/profile_ops.py . > gpurun_out/r4a_ops.txt 2>&1; tail -1 gpurun_out/r4a_ops.txt | tr ' ' '\n' | paste -sd' ' | cut -c1-1500
for r in 1; do
  timeout 1200 python bench.py --steps 20 --warmup 5 > gpurun_out/r4a_bench_driver_$r.json 2> gpurun_out/r4a_bench_driver_$r.err; echo "driver-command bench $r rc=$?"
  python - <<EOF
import json
d = json.loads(open("gpurun_out/r4a_bench_driver_$r.json").readlines()[-1])
print({k: d.get(k) for k in ("value", "ms_per_step", "single_window_ms", "boundary_single_engine_ms")}, d["scene_depth_range"], d["roofline"]["frac"], d["tsdf"]["value"], d["tsdf"]["kernel_ms_per_frame"], d["tandem_loop"]["640x480_5mm"] if "tandem_loop" in d else None)
EOF
done
