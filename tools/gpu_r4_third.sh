#!/bin/bash
# Round 4, third call: the whole -m gpu suite on the tree with the parity build split off (libdr_mi355x_hooks.so), k_costvol4 (LDS-staged
# taps) against k_costvol3 per stage, the centre-from-corners ray-cast sampler against round 3's, the re-tuned plan table.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider > gpurun_out/r4c_gpu_tests.log 2>&1
echo "== gpu suite: $(grep -E 'passed|failed' gpurun_out/r4c_gpu_tests.log | tail -1)"; grep -E "^FAILED|^ERROR" gpurun_out/r4c_gpu_tests.log | head -12
for v in "X=1" "DR_COSTVOL_V3=1" "X=1" "DR_COSTVOL_V3=1"; do
  echo "-- $v: $(env $v DR_MVS_NO_SIDE_STREAM=1 timeout 300 python tools/profile_ops.py 'costvol' 2>&1 | tail -1)"
done | tee gpurun_out/r4c_costvol_ab.txt
H=$PWD/tandem_amd/libdr_mi355x_hooks.so
for v in 2 1 2 1; do echo "-- sampler $v: $(DR_MI355X_LIB=$H DR_RAYCAST_SAMPLER=$v timeout 300 python tools/gpu_tsdf_probe.py 200 2>&1 | grep '^lap [12]' | tr '\n' '|')"; done | tee gpurun_out/r4c_raycast_ab.txt
for r in 1 2; do
  echo "-- bench $r: $(timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu --no-tsdf --no-boundary --no-loop 2>gpurun_out/r4c_bench.err | python -c 'import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print("%.1f /s, single %.3f ms, top %s" % (d["value"], d["single_window_ms"], [(t["kernel"], t["ms"], t["frac"]) for t in d["roofline"]["top"][:6]]))')"
done | tee gpurun_out/r4c_bench.txt
python tools/profile_ops.py . > gpurun_out/r4c_ops.txt 2>&1; tail -1 gpurun_out/r4c_ops.txt | cut -c1-1800
