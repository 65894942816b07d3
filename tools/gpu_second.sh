#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -m pytest tests -m gpu -q --no-header -p no:cacheprovider -x 2>&1 | tail -40 > gpurun_out/gpu_tests.log
tail -40 gpurun_out/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5
