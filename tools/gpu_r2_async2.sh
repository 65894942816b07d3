#!/bin/bash
cd "$(dirname "$0")/.."
DR_CONV_ASYNC=1 timeout 600 python -m pytest tests/test_conv_gpu.py -m gpu -q --no-header -p no:cacheprovider -x 2>&1 | grep -E "passed|failed|Error|assert" | tail -4
DR_CONV_ASYNC=${POLICY:-1} DR_CONV_PRINT=${PRINT:-1} DR_AUTOTUNE_ONLY=${ONLY:-conv} timeout 600 python tools/try_autotune.py ${K:-60} 2>&1 | grep -E "autotune|cand|before|after|TUNED"
