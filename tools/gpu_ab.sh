#!/bin/bash
# within-box A/B of an env-selected kernel variant: interleaved runs, per-op times of the hot convs
cd "$(dirname "$0")/.."
for r in 1 2 3; do
  for v in "" "$1"; do
    env $v timeout 300 python tools/check_mvsnet.py 480 640 7 2>/dev/null | grep -E "^  (s2.conv0|s1.conv0|fn.out3|s2.conv2|s3.conv0) |5 forwards" | awk -v ab="[$v]" '{printf "%s %s %s | ", ab, $1, $2} END{print ""}'
  done
done
