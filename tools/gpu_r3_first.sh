#!/bin/bash
# Round 3, first GPU run: correctness of the marching convolution kernel, the bordered-feature cost volume and the
# XCD-band prob order on hardware, then A/B timings (same box): old plans vs marching kernel, per-candidate autotune log.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
timeout 1200 python -m pytest tests/test_conv_gpu.py -q --no-header -p no:cacheprovider --maxfail=8 --durations=8 > $O/r3a_conv.log 2>&1; grep -E "passed|failed|^FAILED|^ERROR" $O/r3a_conv.log | tail -12
timeout 600 python -m pytest tests/test_mvsnet_gpu.py -q --no-header -p no:cacheprovider --maxfail=8 > $O/r3a_mvs.log 2>&1; grep -E "passed|failed|^FAILED|^ERROR" $O/r3a_mvs.log | tail -12
DR_COSTVOL_V1=1 DR_PROB_LAUNCH_ORDER=1 DR_CONV_MARCH=0 timeout 400 python tools/check_mvsnet.py 480 640 7 > $O/r3a_check_old.log 2>&1; grep -E "costvol|prob|5 forwards|criterion" $O/r3a_check_old.log
timeout 400 python tools/check_mvsnet.py 480 640 7 > $O/r3a_check_new.log 2>&1; grep -E "costvol|prob|5 forwards|criterion" $O/r3a_check_new.log
DR_CONV_NO_TUNED=1 DR_CONV_MARCH=2 timeout 400 python tools/check_mvsnet.py 480 640 7 > $O/r3a_check_march.log 2>&1; grep -E "k_conv_m|5 forwards|criterion" $O/r3a_check_march.log
DR_CONV_NO_TUNED=1 DR_CONV_PRINT=2 timeout 600 python tools/try_autotune.py 400 > $O/r3a_tune.log 2>&1; grep -E "^autotune|before|after" $O/r3a_tune.log | cut -c1-200
