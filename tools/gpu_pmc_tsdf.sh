#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
A="--steps 1 --warmup 0 --no-cpu --tsdf-scans 10 --tsdf-cycles 2"
rm -rf gpurun_out/pt1 gpurun_out/pt2 gpurun_out/pt3 gpurun_out/pt4
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pt1 -o pmc -- python bench.py $A > gpurun_out/pt1.log 2>&1
rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pt2 -o pmc -- python bench.py $A > gpurun_out/pt2.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pt3 -o pmc -- python bench.py $A > gpurun_out/pt3.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES MemUnitStalled GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pt4 -o pmc -- python bench.py $A > gpurun_out/pt4.log 2>&1
ls gpurun_out/pt1 gpurun_out/pt2 gpurun_out/pt3 gpurun_out/pt4; tail -2 gpurun_out/pt3.log
