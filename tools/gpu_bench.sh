#!/bin/bash
# bench + rocprofv3 kernel-trace of the same command; summaries land in gpurun_out/
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
python bench.py --steps 50 --warmup 5 > gpurun_out/bench.json 2> gpurun_out/bench.err; tail -c 6000 gpurun_out/bench.json; tail -5 gpurun_out/bench.err
rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o bench -- python bench.py --steps 20 --warmup 3 --no-cpu > gpurun_out/bench_prof.json 2> gpurun_out/prof.err
ls -R gpurun_out/prof | head -30
