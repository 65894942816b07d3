#!/bin/bash
# PMC passes (counters only + kernel-trace) over the configs[3] loop probe: what bounds k_raycast2 / k_integrate / k_allocate
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
N=${1:-60}
rm -rf gpurun_out/pf1 gpurun_out/pf2 gpurun_out/pf3 gpurun_out/pf4
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pf1 -o pmc -- python tools/gpu_tsdf_probe.py $N > gpurun_out/pf1.log 2>&1
rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pf2 -o pmc -- python tools/gpu_tsdf_probe.py $N > gpurun_out/pf2.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pf3 -o pmc -- python tools/gpu_tsdf_probe.py $N > gpurun_out/pf3.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pf4 -o pmc -- python tools/gpu_tsdf_probe.py $N > gpurun_out/pf4.log 2>&1
for i in 1 2 3 4; do d=$(dirname $(find gpurun_out/pf$i -name "pmc_counter_collection.csv" | head -1)); echo "== pass $i ($d)"; python tools/pmc_summary.py $d 2>&1 | head -60; done > gpurun_out/r2_pmc_fusion.txt
tail -3 gpurun_out/pf4.log
wc -l gpurun_out/r2_pmc_fusion.txt
