#!/bin/bash
# Round 3, second GPU run: (1) conv_tuned.h rows for both tuned shapes with the marching kernel among the candidates
# (3 autotune runs each, majority), (2) PMC passes over a sequential forward with the marching kernel wherever it applies.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
bash tools/tune_conv.sh > $O/conv_tuned_480x640.h 2> $O/tune_480.err
bash tools/tune_conv.sh 320 512 48,4,4 > $O/conv_tuned_320x512.h 2> $O/tune_320.err
wc -l $O/conv_tuned_*.h
export DR_MVS_NO_SIDE_STREAM=1 DR_CONV_NO_TUNED=1 DR_CONV_MARCH=2
rm -rf $O/pm3 $O/pm4
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pm3 -o pmc -- python tools/profile_ops.py conv0 > $O/pm3.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAIT_INST_LDS TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pm4 -o pmc -- python tools/profile_ops.py conv0 > $O/pm4.log 2>&1
for i in 3 4; do d=$(dirname $(find $O/pm$i -name "pmc_counter_collection.csv" | head -1)); echo "== pass $i"; python tools/pmc_summary.py $d 2>&1 | grep -v "^at::\|elementwise\|    .*at::\|rocclr\|^void at"; done > $O/r3b_pmc_summary.txt
tail -1 $O/pm3.log; wc -l $O/r3b_pmc_summary.txt
rm -rf $O/pm3 $O/pm4
