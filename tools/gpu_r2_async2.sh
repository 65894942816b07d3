#!/bin/bash
# autotune selected layers over both kernel families (DR_CONV_NO_TUNED: ignore the baked table), per-candidate print
cd "$(dirname "$0")/.."
DR_CONV_NO_TUNED=1 DR_CONV_ASYNC=${POLICY:-2} DR_CONV_PRINT=${PRINT:-2} DR_AUTOTUNE_ONLY=${ONLY:-conv0} timeout 600 python tools/try_autotune.py ${K:-400} 2>&1 | grep -E "autotune|cand|before|after"
