#!/bin/bash
cd "$(dirname "$0")/.."
for xo in 4 2 1; do for b in 256 128; do for z in 0 8; do
  v="DR_PROB_XO=$xo DR_PROB_BLOCK=$b"; if [ $z != 0 ]; then v="$v DR_PROB_ZCHUNK=$z"; fi
  echo -n "$v : "; env $v timeout 300 python tools/profile_ops.py 'prob' 2>&1 | grep -v amdgpu.ids
done; done; done
