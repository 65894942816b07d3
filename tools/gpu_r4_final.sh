#!/bin/bash
# Round-4 evidence, in the order bench.py needs it:
#   1. PMC passes over bench.py itself (one counter group per pass, --kernel-trace only; strictly sequential kernels: one engine, side
#      stream off) -> profiles/r04_pmc_traffic.json, stamped with this tree's source hash (bench.py refuses any other stamp)
#   2. the whole -m gpu suite, smoke()
#   3. the driver's bench command; the shipped-model line
#   4. rocprofv3 --kernel-trace --stats of the sequential configuration (per-kernel durations without overlap)
# No product source changes after this has run.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out profiles
export TMPDIR=/tmp
S="--no-cpu --engines 1 --no-boundary --no-loop --no-tsdf-native"
if [ -z "$SKIP_PMC" ]; then
export DR_MVS_NO_SIDE_STREAM=1
rm -rf gpurun_out/pm1 gpurun_out/pm2 gpurun_out/pm3 gpurun_out/pm4 gpurun_out/prof
A="--steps 3 --warmup 1 --tsdf-frames 60 $S"
timeout 600 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pm1 -o pmc -- python bench.py $A > gpurun_out/pm1.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pm2 -o pmc -- python bench.py $A > gpurun_out/pm2.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pm3 -o pmc -- python bench.py $A > gpurun_out/pm3.log 2>&1
timeout 600 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAIT_INST_LDS TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pm4 -o pmc -- python bench.py $A > gpurun_out/pm4.log 2>&1
for i in 1 2 3 4; do d=$(dirname $(find gpurun_out/pm$i -name "pmc_counter_collection.csv" | head -1)); echo "== pass $i"; python tools/pmc_summary.py $d 2>&1 | grep -v "^at::\|elementwise\|    .*at::\|rocclr\|^void at" ; done > gpurun_out/r04_pmc_summary.txt
python tools/pmc_to_json.py profiles/r04_pmc_traffic.json $(for i in 1 2 3; do dirname $(find gpurun_out/pm$i -name "pmc_counter_collection.csv" | head -1); done)
cp profiles/r04_pmc_traffic.json gpurun_out/
unset DR_MVS_NO_SIDE_STREAM
fi
timeout 1500 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider --durations=8 > gpurun_out/r04_gpu_tests.log 2>&1; grep -E "passed|failed|^FAILED|^ERROR" gpurun_out/r04_gpu_tests.log | tail -5
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee gpurun_out/r04_smoke.txt
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04_bench_driver.json 2> gpurun_out/r04_bench_driver.err; echo "driver-command bench rc=$?"; head -c 500 gpurun_out/r04_bench_driver.json; echo
timeout 600 python bench.py --config shipped --steps 240 --no-tsdf --no-loop --no-cpu > gpurun_out/r04_bench_shipped.json 2> gpurun_out/r04_bench_shipped.err; echo "shipped rc=$?"; head -c 300 gpurun_out/r04_bench_shipped.json; echo
DR_MVS_NO_SIDE_STREAM=1 timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o bench -- python bench.py --steps 20 --warmup 3 --tsdf-frames 200 $S > gpurun_out/r04_bench_prof.json 2> gpurun_out/prof.err
python tools/rocprof_summary.py $(find gpurun_out/prof -name "*_results.db" | head -1) > gpurun_out/r04_bench_kernel_stats.txt 2>&1; head -14 gpurun_out/r04_bench_kernel_stats.txt
rm -rf gpurun_out/pm1 gpurun_out/pm2 gpurun_out/pm3 gpurun_out/pm4 gpurun_out/prof
