#!/bin/bash
# parameter sweep of the vector-pipe kernels' depth chunking (tuning hooks DR_CV_DCHUNK{1,2,3}, DR_PROB_ZCHUNK)
cd "$(dirname "$0")/.."
run() { env "$@" python tools/check_mvsnet.py 480 640 7 2>/dev/null | grep -E "^  s[123]\.(costvol|prob) " | awk -v tag="$*" '{printf "%s=%s ", $1, $2} END{print " [" tag "]"}'; }
run X=1
for c in 1 2 3 6 8 12 16 24 48; do run DR_CV_DCHUNK1=$c; done
for c in 1 2 4 16 32; do run DR_CV_DCHUNK2=$c; done
for c in 1 2 4; do run DR_CV_DCHUNK3=$c; done
for z in 1 2 4 8 16 32 48; do run DR_PROB_ZCHUNK=$z; done
