#!/bin/bash
# fusion iteration: parity tests of the fusion path, then the configs[3] loop probe (new vs old ray-caster)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_fusion_gpu.py tests/test_mesh_gpu.py tests/test_tracker_gpu.py -m gpu -q --no-header -p no:cacheprovider -x 2>&1 | tail -15
echo "--- probe (default)"; timeout 300 python tools/gpu_tsdf_probe.py 200 2>&1 | grep -v amdgpu.ids
echo "--- probe (DR_RAYCAST_V1)"; DR_RAYCAST_V1=1 timeout 300 python tools/gpu_tsdf_probe.py 200 2>&1 | grep -v amdgpu.ids
