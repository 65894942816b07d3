#!/bin/bash
cd "$(dirname "$0")/.."
for v in "" NO_STAGE NO_KLOOP NO_EPI; do
  if [ -n "$v" ]; then export DR_MI355X_LIB=$PWD/build_ab/lib_$v.so; fi
  echo "--- async ${v:-full}"; DR_CONV_ASYNC=1 timeout 300 python tools/profile_ops.py "conv0|conv11|conv2$" 2>&1 | grep -v amdgpu.ids
done
