#!/bin/bash
# timing ablations of k_conv (DR_CONV_DBG bits: 1 no refetch, 2 no K loop, 4 no epilogue); results are wrong by design
cd "$(dirname "$0")/.."
for v in 0 1 2 4 3 6 7; do
  DR_CONV_DBG=$v timeout 300 python tools/check_mvsnet.py 480 640 7 2>/dev/null | grep -E "^  (s2.conv0|s1.conv0|fn.out3|s2.conv2|s3.conv0|s2.conv11) |5 forwards" | awk -v ab="[dbg=$v]" '{printf "%s %s %s | ", ab, $1, $2} END{print ""}'
done
