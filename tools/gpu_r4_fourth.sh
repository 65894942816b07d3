#!/bin/bash
# Round 4, fourth call: k_costvol4 after its instruction diet (per stage, 4 / 8 planes per step, against k_costvol3), the ray-cast
# sampler that selects the centre voxel from the corners (mismatching lanes no longer go to the literal pass), the affected suites.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_mvsnet_gpu.py tests/test_fusion_gpu.py -m gpu -q --no-header -p no:cacheprovider -k "cost_volume or raycast or bf16x3 or golden or bench_workload or full_size" > gpurun_out/r4d_tests.log 2>&1
echo "== suites: $(grep -E 'passed|failed' gpurun_out/r4d_tests.log | tail -1)"; grep -E "^FAILED|^ERROR" gpurun_out/r4d_tests.log | head -8
for v in "DR_COSTVOL_V3=1" "DR_CV4_SP8=0" "DR_CV4_SP8=2" "DR_CV4_SP8=6" "DR_COSTVOL_V3=1" "DR_CV4_SP8=0" "DR_CV4_SP8=6"; do
  echo "-- $v: $(env $v DR_MVS_NO_SIDE_STREAM=1 timeout 300 python tools/profile_ops.py 'costvol' 2>&1 | tail -1)"
done | tee gpurun_out/r4d_costvol_ab.txt
H=$PWD/tandem_amd/libdr_mi355x_hooks.so
for v in 2 1 2 1; do echo "-- sampler $v: $(DR_MI355X_LIB=$H DR_RAYCAST_SAMPLER=$v timeout 300 python tools/gpu_tsdf_probe.py 200 2>&1 | grep '^lap [12]' | tr '\n' '|')"; done | tee gpurun_out/r4d_raycast_ab.txt
