#!/bin/bash
# Round 4, fourteenth call: two knobs on the final kernels -- DMA loads a marching producer keeps in flight (DR_MARCH_PDEPTH, plan time) and
# engines in flight for the headline leg.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
for d in 2 3 4 1 2 3; do
  echo "-- DR_MARCH_PDEPTH=$d: $(DR_MARCH_PDEPTH=$d timeout 300 python tools/profile_ops.py 'conv0$|conv2$' 2>&1 | tail -1 | cut -c1-400)"
done | tee gpurun_out/r4n_pdepth.txt
for e in 3 4 2 3 4; do
  echo "-- engines $e: $(timeout 400 python bench.py --steps 20 --warmup 5 --engines $e --no-cpu --no-tsdf --no-boundary --no-loop 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print("%.1f /s, single %.3f ms" % (d["value"], d["single_window_ms"]))')"
done | tee gpurun_out/r4n_engines.txt
