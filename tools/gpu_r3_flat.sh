#!/bin/bash
# Round 3: cross-section pipelined K loop (march_kloop3) -- correctness, then timing of the layers that march.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
timeout 900 python -m pytest tests/test_conv_gpu.py -q --no-header -p no:cacheprovider --maxfail=8 -k "row_march or marching" > $O/r3u_conv.log 2>&1; grep -E "passed|failed|^FAILED|^ERROR" $O/r3u_conv.log | tail -12
DR_AUTOTUNE_ONLY=${ONLY:-conv} DR_CONV_NO_TUNED=1 DR_CONV_PRINT=2 timeout 600 python tools/try_autotune.py 400 > $O/r3u_tune.log 2>&1
grep -E "before|after|^autotune\(" $O/r3u_tune.log
grep -E "cand (s[123].conv[024]|fn.conv1.1|fn.conv0.1) .*(rowmarch|march)" $O/r3u_tune.log | awk '{k=$2" "$5; t=$(NF-1); if (!(k in b) || t<b[k]) {b[k]=t; l[k]=$0}} END {for (k in b) print l[k]}' | sort -k2,2 | cut -c1-150
grep -E "^autotune s[123]" $O/r3u_tune.log | cut -c1-170
