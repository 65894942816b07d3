#!/bin/bash
# first GPU bring-up: conv unit tests, then the whole-pipeline diagnostic
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -m pytest tests/test_conv_gpu.py -m gpu -q -x --no-header -p no:cacheprovider 2>&1 | tail -40 > gpurun_out/conv_tests.log
cat gpurun_out/conv_tests.log | tail -30
timeout 600 python tools/check_mvsnet.py 64 96 3 > gpurun_out/check_small.log 2>&1; tail -70 gpurun_out/check_small.log
