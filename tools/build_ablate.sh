#!/bin/bash
# Builds the k_conv_m timing-ablation libraries (build/libdr_mabl_<V>.so); fusion/tracker objects are shared with the product build.
cd "$(dirname "$0")/.." && mkdir -p build
for v in NO_WAIT NO_EPI NO_DMA NO_KLOOP FREE; do
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -Wno-unused-function -Wno-pass-failed -DDR_MABL_$v -c tandem_amd/csrc/dr_mvsnet.hip -o build/m_mabl_$v.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build/m_mabl_$v.o tandem_amd/csrc/dr_fusion.o tandem_amd/csrc/dr_tracker.o -o build/libdr_mabl_$v.so -lpthread ) &
done
wait; ls -la build/libdr_mabl_*.so
