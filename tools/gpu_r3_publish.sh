#!/bin/bash
# Round 3: results handed to the host by a kernel (k_publish / k_publish4) instead of copy-engine transfers -- tests, then A/B in the
# TandemBackend-shaped loop and in the TSDF frame loop.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out /tmp/loop
export TMPDIR=/tmp
O=gpurun_out/r3_publish.txt; : > $O
timeout 900 python -m pytest tests/test_fusion_gpu.py tests/test_shim.py tests/test_mvsnet_gpu.py -q --no-header -p no:cacheprovider -m gpu --maxfail=5 -k "not full_size and not bench_workload and not maximum" > gpurun_out/r3p_tests.log 2>&1; grep -E "passed|failed|^FAILED|^ERROR" gpurun_out/r3p_tests.log | tail -8
g++ -std=c++14 -O2 -Iinclude -Itandem_amd/libdr tools/tandem_loop.cpp -o /tmp/loop/tandem_loop -Ltandem_amd -ldr_mi355x -Wl,-rpath,$PWD/tandem_amd || exit 1
python - <<'PY'
import sys, numpy as np
sys.path.insert(0, "tools")
from synth import scene
from export_fixture import write_tdms
win = scene.make_window(480, 640, 7, seed=5)
z = np.zeros((480, 640), np.float32)
write_tdms("/tmp/loop/w.tdms", np.stack(win["bgrs"]), win["K"], win["c2ws"], win["ref_index"], win["depth_min"], win["depth_max"], 10.0, z, z)
PY
run() { echo "== $*" >> $O; env "$@" timeout 300 /tmp/loop/tandem_loop weights/tandem_va.tdmw /tmp/loop/w.tdms 100 $VS 0 1 >> $O 2>&1; }
for VS in 0.005 0.01; do
  for rep in 1 2; do
    run DR_RENDER_D2H=copy DR_MVS_D2H=copy
    run DR_RENDER_D2H=kernel DR_MVS_D2H=copy
    run DR_RENDER_D2H=kernel DR_MVS_D2H=kernel
  done
done
python - <<'PY'
import json
for l in open("gpurun_out/r3_publish.txt"):
    if l.startswith("=="): tag = l.strip()
    elif l.startswith("{"):
        d = json.loads(l); print(tag, "vs", d["voxel_size"], "ms/kf", d["ms_per_keyframe"], d["mean_ms"])
    else: print(l.strip())
PY
for M in copy kernel; do echo "TSDF frame loop, DR_RENDER_D2H=$M"; DR_RENDER_D2H=$M timeout 300 python tools/gpu_tsdf_probe.py 2>&1 | tail -2; done
