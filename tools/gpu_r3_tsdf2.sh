#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_fusion_gpu.py -q --no-header -p no:cacheprovider --maxfail=8 > gpurun_out/r3h_fusion.log 2>&1; grep -E "passed|failed|^FAILED|^ERROR|Error" gpurun_out/r3h_fusion.log | tail -8
timeout 300 python tools/gpu_tsdf_probe.py 300 2>&1 | grep -v amdgpu.ids | tail -2
DR_RAYCAST_UNSTAGED=1 timeout 300 python tools/gpu_tsdf_probe.py 300 2>&1 | grep -v amdgpu.ids | tail -1
