#!/bin/bash
# optimisation loop: correctness (conv cases + small pipeline fixtures), then full-size profile
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -m pytest tests/test_conv_gpu.py tests/test_mvsnet_gpu.py -m gpu -q --no-header -p no:cacheprovider -x 2>&1 | tail -15
timeout 600 python tools/check_mvsnet.py 480 640 7 > gpurun_out/check_full.log 2>&1
grep -E "^(depth3|conf3|mask|reference|work|5 forw|hip)" gpurun_out/check_full.log
