#!/bin/bash
# the full per-op table of one forward (hipEvents around every launch)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 300 python tools/profile_ops.py '.' 2>&1 | grep -v amdgpu.ids | tr ' ' '\n' | tee gpurun_out/r2_ops.txt | sort -t= -k2 -n -r | head -${TOP:-70}
