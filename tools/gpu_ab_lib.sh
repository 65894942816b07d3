#!/bin/bash
# within-box A/B of two builds of the library (build/old/libdr_old.so = baseline, in-tree = candidate): interleaved runs
cd "$(dirname "$0")/.."
OLD=${1:-$PWD/build/old/libdr_old.so}
for r in 1 2; do
  for v in "DR_MI355X_LIB=$OLD" "X=1"; do
    env $v timeout 300 python tools/check_mvsnet.py 480 640 7 2>/dev/null > gpurun_out/ab_${r}_${v%%=*}.log
    grep -E "^  (fn|s1|s2|s3)\..*k_conv" gpurun_out/ab_${r}_${v%%=*}.log | awk '{s+=$2} END{printf "conv total %.3f ms | ", s}'
    grep -E "^  (s2.conv0|s1.conv0|fn.out3|s3.conv0|fn.skip3|s2.conv2|s2.conv11) |5 forwards" gpurun_out/ab_${r}_${v%%=*}.log | awk -v ab="[${v%%=*}]" '{printf "%s %s %s | ", ab, $1, $2} END{print ""}'
  done
done
paste <(grep -E "^  (fn|s1|s2|s3)\." gpurun_out/ab_2_DR_MI355X_LIB.log | awk '{print $1, $2, $4}') <(grep -E "^  (fn|s1|s2|s3)\." gpurun_out/ab_2_X.log | awk '{print $2, $4}')
