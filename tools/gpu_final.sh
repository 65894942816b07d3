#!/bin/bash
# round-end evidence: full GPU test suite, smoke, bench (with CPU baselines), rocprofv3 kernel stats.
# The kernel stats come from `DR_MVS_NO_SIDE_STREAM=1 bench.py --engines 1`, i.e. strictly sequential kernels: with the
# default 3 engines per GPU (and the engine's side stream) kernels of different windows / branches overlap and every
# per-kernel duration is inflated by its neighbours; bench.py's own roofline figures are measured the same sequential
# way (engine 0 alone, hipEvents around each launch of an un-forked forward), so the two agree.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
python -m pytest tests -m gpu -q --no-header -p no:cacheprovider 2>&1 | tail -5 > gpurun_out/gpu_tests.log; cat gpurun_out/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err; tail -c 4000 gpurun_out/bench.json
rm -rf gpurun_out/prof; DR_MVS_NO_SIDE_STREAM=1 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o bench -- python bench.py --steps 20 --warmup 3 --no-cpu --engines 1 > gpurun_out/bench_prof.json 2> gpurun_out/prof.err
ls gpurun_out/prof
