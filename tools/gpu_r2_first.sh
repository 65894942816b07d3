#!/bin/bash
# round 2, call 1: the new parity tests (reference builds, exhaustive Combine, full-size bit-exact) + the whole -m gpu suite
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
ls -la oracle/_ref/ > gpurun_out/r2_ref_ls.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider --durations=15 2>&1 | tail -40 > gpurun_out/r2_gpu_tests.log
tail -25 gpurun_out/r2_gpu_tests.log
