#!/bin/bash
# HBM traffic counters (separate passes, as MI355X_MICROARCH.md prescribes: FETCH_SIZE takes 3 TCC slots, WRITE_SIZE 2)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
export DR_MVS_NO_SIDE_STREAM=1  # strictly sequential kernels
A="--steps 3 --warmup 1 --no-cpu --engines 1 --tsdf-scans 50 --tsdf-cycles 1"
rm -rf gpurun_out/tr1 gpurun_out/tr2 gpurun_out/tr3
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/tr1 -o pmc -- python bench.py $A > gpurun_out/tr1.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/tr2 -o pmc -- python bench.py $A > gpurun_out/tr2.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/tr3 -o pmc -- python bench.py $A > gpurun_out/tr3.log 2>&1
ls gpurun_out/tr1 gpurun_out/tr2 gpurun_out/tr3 | head -20; tail -2 gpurun_out/tr1.log
