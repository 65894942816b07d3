#!/bin/bash
# Round 4, second call: (1) the paired-gather ray-cast sampler -- the fusion suite (bit-exact against the oracle, all sampler generations
# against each other) and the per-kernel split of the BASELINE configs[3] loop for samplers 2 / 1; (2) k_conv_a at PT = 1 -- the conv
# suite (every plan candidate against torch), then two autotune passes over the headline shape with the tuned table in place;
# (3) the de-phased k_conv A/B builds; (4) the reference's TandemBackend as integration driver.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_fusion_gpu.py tests/test_shim.py -m gpu -q --no-header -p no:cacheprovider > gpurun_out/r4b_fusion_tests.log 2>&1
echo "== fusion + shim suites: $(grep -E 'passed|failed' gpurun_out/r4b_fusion_tests.log | tail -1)"; grep -E "^FAILED|^ERROR" gpurun_out/r4b_fusion_tests.log | head -8
for v in 2 1 2 1; do echo "-- sampler $v: $(DR_RAYCAST_SAMPLER=$v timeout 300 python tools/gpu_tsdf_probe.py 200 2>&1 | grep '^lap' | tr '\n' '|')"; done | tee gpurun_out/r4b_raycast_ab.txt
timeout 900 python -m pytest tests/test_conv_gpu.py -q --no-header -p no:cacheprovider > gpurun_out/r4b_conv_tests.log 2>&1
echo "== conv suite: $(grep -E 'passed|failed' gpurun_out/r4b_conv_tests.log | tail -1)"; grep -E "^FAILED|^ERROR" gpurun_out/r4b_conv_tests.log | head -8
for r in 1 2; do
  DR_CONV_PRINT=1 timeout 600 python tools/try_autotune.py 400 > gpurun_out/r4b_tune_$r.txt 2>&1
  grep -E "^before|^after|^autotune\(" gpurun_out/r4b_tune_$r.txt | tr '\n' ' '; echo; grep "^TUNED" gpurun_out/r4b_tune_$r.txt
done
for lib in "" build/ab/libdr_dephase1.so build/ab/libdr_dephase3.so "" build/ab/libdr_dephase1.so build/ab/libdr_dephase3.so; do
  echo "-- lib '$lib': $(env ${lib:+DR_MI355X_LIB=$PWD/$lib} timeout 300 python tools/profile_ops.py 'NONE' 2>&1 | tail -1)  bench $(env ${lib:+DR_MI355X_LIB=$PWD/$lib} timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu --no-tsdf --no-boundary --no-loop 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print("%.1f /s, single %.3f ms" % (d["value"], d["single_window_ms"]))')"
done | tee gpurun_out/r4b_dephase_ab.txt
