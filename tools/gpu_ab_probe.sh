#!/bin/bash
# A/B of library builds on the configs[3] probe: bash tools/gpu_ab_probe.sh build/libdr_X.so ...
cd "$(dirname "$0")/.."
for lib in "$@"; do echo "--- $lib"; DR_MI355X_LIB=$PWD/$lib timeout 300 python tools/gpu_tsdf_probe.py ${NFR:-120} 2>&1 | grep -v amdgpu.ids | head -2; done
