#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf gpurun_out/pmc1 gpurun_out/pmc2
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc1 -o pmc -- python bench.py --steps 3 --warmup 1 --no-cpu $PMC_ARGS > gpurun_out/pmc1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_RD GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc2 -o pmc -- python bench.py --steps 3 --warmup 1 --no-cpu $PMC_ARGS > gpurun_out/pmc2.log 2>&1
ls -la gpurun_out/pmc1 gpurun_out/pmc2; tail -3 gpurun_out/pmc1.log
