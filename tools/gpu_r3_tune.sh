#!/bin/bash
# Round 3: regenerate the tuned-plan rows for both tuned shapes (merged by hand into tandem_amd/csrc/conv_tuned.h).
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
bash tools/tune_conv.sh > gpurun_out/conv_tuned_480x640.h 2> gpurun_out/tune_480.err
bash tools/tune_conv.sh 320 512 48,4,4 > gpurun_out/conv_tuned_320x512.h 2> gpurun_out/tune_320.err
wc -l gpurun_out/conv_tuned_480x640.h gpurun_out/conv_tuned_320x512.h
