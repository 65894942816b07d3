#!/bin/bash
# timing ablations of k_conv (builds with -DDR_ABL_NO_KLOOP / -DDR_ABL_NO_STAGE / both; results are wrong by design)
cd "$(dirname "$0")/.."
for lib in tandem_amd/libdr_mi355x.so build/libdr_NO_KLOOP.so build/libdr_NO_STAGE.so build/libdr_NONE.so; do
  echo "--- $lib"; DR_MI355X_LIB=$PWD/$lib timeout 300 python tools/profile_ops.py 'conv0|out3|out2|conv2$|conv11|skip3' 2>&1 | grep -v amdgpu.ids
done
