#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 300 python tools/gpu_tsdf_probe.py 300 2>&1 | grep -v amdgpu.ids | tail -3
DR_RAYCAST_STATS=1 timeout 300 python tools/gpu_tsdf_probe.py 200 > gpurun_out/r3g_raystats.txt 2>&1; grep -A1 "raycast stats" gpurun_out/r3g_raystats.txt | tail -12
