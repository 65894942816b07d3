#!/bin/bash
# k_prob (vector pipe) against the MFMA X8 form of the same layer, per stage, and the pipeline fixtures through the X8 form
cd "$(dirname "$0")/.."
for v in "X=1" "DR_PROB_ON_CONV=1" "DR_PROB_ON_CONV=1 DR_CONV_ASYNC=0"; do echo "--- $v"; env $v timeout 200 python tools/profile_ops.py "prob" 2>&1 | grep -v amdgpu.ids; done
DR_PROB_ON_CONV=1 timeout 200 python -m pytest tests/test_mvsnet_gpu.py -m gpu -q --no-header -p no:cacheprovider -x -k "golden" 2>&1 | grep -E "passed|failed|Error|assert" | tail -3
