#!/bin/bash
# PMC passes (counters + kernel-trace only, one pass per counter group as MI355X_MICROARCH.md prescribes) over the depth
# pipeline (strictly sequential kernels: one engine, side stream off) and over the TSDF configs[3] loop.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
export DR_MVS_NO_SIDE_STREAM=1
A="--steps 3 --warmup 1 --no-cpu --engines 1 --no-tsdf --no-boundary --no-loop"
rm -rf gpurun_out/pm1 gpurun_out/pm2 gpurun_out/pm3 gpurun_out/pm4
rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pm1 -o pmc -- python bench.py $A > gpurun_out/pm1.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pm2 -o pmc -- python bench.py $A > gpurun_out/pm2.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pm3 -o pmc -- python bench.py $A > gpurun_out/pm3.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAIT_INST_LDS TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pm4 -o pmc -- python bench.py $A > gpurun_out/pm4.log 2>&1
for i in 1 2 3 4; do d=$(dirname $(find gpurun_out/pm$i -name "pmc_counter_collection.csv" | head -1)); echo "== pass $i"; python tools/pmc_summary.py $d 2>&1 | grep -v "^at::\|elementwise\|    .*at::" ; done > gpurun_out/r2_pmc_mvsnet.txt
python tools/pmc_to_json.py gpurun_out/r2_pmc_traffic_mvsnet.json $(for i in 1 2 3; do dirname $(find gpurun_out/pm$i -name "pmc_counter_collection.csv" | head -1); done)
tail -2 gpurun_out/pm4.log; wc -l gpurun_out/r2_pmc_mvsnet.txt
