#!/bin/bash
# Round 3: TandemBackend-shaped loop, stream-priority A/B.  Output: gpurun_out/r3_loop.txt
cd "$(dirname "$0")/.."
mkdir -p gpurun_out /tmp/loop
export TMPDIR=/tmp
O=gpurun_out/r3_loop.txt; : > $O
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
  for P in low normal high low normal high; do run DR_FUSION_PRIORITY=$P; done
  run DR_FUSION_PRIORITY=low TANDEM_LOOP_SERIAL=1
  run DR_FUSION_PRIORITY=high DR_CONV_MARCH=0
  run DR_FUSION_PRIORITY=low DR_CONV_MARCH=0
done
python - <<'PY'
import json
for l in open("gpurun_out/r3_loop.txt"):
    if l.startswith("=="): tag = l.strip()
    elif l.startswith("{"):
        d = json.loads(l); print(tag, "vs", d["voxel_size"], "ms/kf", d["ms_per_keyframe"], d["mean_ms"])
    else: print(l.strip())
PY
