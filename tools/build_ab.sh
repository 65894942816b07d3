#!/bin/bash
# A/B builds of the library: build/ab/libdr_<name>.so = the tree's sources with extra -D flags on dr_mvsnet.hip (the other two
# objects are the tree's).  Loaded through DR_MI355X_LIB for within-box comparisons; never the product library.
#   usage: tools/build_ab.sh name "-DDR_CONV_DEPHASE=2" [name2 "flags2" ...]
cd "$(dirname "$0")/.."
mkdir -p build/ab
make -s -j8 all
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -Wall -Wno-unused-function -Wno-pass-failed $flags -c tandem_amd/csrc/dr_mvsnet.hip -o build/ab/dr_mvsnet_$name.o &&
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build/ab/dr_mvsnet_$name.o tandem_amd/csrc/dr_fusion.o tandem_amd/csrc/dr_tracker.o -o build/ab/libdr_$name.so -lpthread && echo "built build/ab/libdr_$name.so ($flags)"
done
