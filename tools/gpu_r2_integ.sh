#!/bin/bash
# fusion iteration: parity tests, then the configs[3] loop probe with and without the ray-cast skip
cd "$(dirname "$0")/.."
timeout 900 python -m pytest tests/test_fusion_gpu.py -m gpu -q --no-header -p no:cacheprovider -x 2>&1 | grep -E "passed|failed|Error|assert" | tail -5
echo "--- default"; timeout 300 python tools/gpu_tsdf_probe.py 300 2>&1 | grep "lap 2"
echo "--- DR_RAYCAST_NO_SKIP=1"; DR_RAYCAST_NO_SKIP=1 timeout 300 python tools/gpu_tsdf_probe.py 300 2>&1 | grep "lap 2"
