#!/bin/bash
# builds and runs the fp32-MFMA calibration microbenchmarks (on a GPU box): peak issue rate and rate with LDS operands
cd "$(dirname "$0")"
for f in mfma_rate mfma_lds mfma_bf3; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 $f.hip -o $f && ./$f; done
