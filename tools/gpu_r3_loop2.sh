#!/bin/bash
# Round 3: per-kernel GPU time of the TandemBackend-shaped loop, overlapped (TANDEM's order) vs serialised, 5 mm.
cd "$(dirname "$0")/.."
R=$PWD
mkdir -p gpurun_out /tmp/loop
export TMPDIR=/tmp
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
cd /tmp
for M in overlap serial; do
  rm -rf /tmp/loop/prof_$M
  if [ $M = serial ]; then export TANDEM_LOOP_SERIAL=1; else unset TANDEM_LOOP_SERIAL; fi
  DR_FUSION_PRIORITY=normal timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/loop/prof_$M -o p -- /tmp/loop/tandem_loop $R/weights/tandem_va.tdmw /tmp/loop/w.tdms 100 0.005 0 1 > $R/gpurun_out/r3_loop_prof_$M.txt 2>&1
  f=$(find /tmp/loop/prof_$M -name "*kernel_stats.csv" | head -1)
  cp "$f" $R/gpurun_out/r3_loop_kernel_stats_$M.csv
done
cd $R
python - <<'PY'
import csv
def load(p):
    d = {}
    for r in csv.DictReader(open(p)):
        d[r["Name"].split("(")[0][:60]] = (int(r["Calls"]), float(r["TotalDurationNs"]) / 1e6)
    return d
a, b = load("gpurun_out/r3_loop_kernel_stats_overlap.csv"), load("gpurun_out/r3_loop_kernel_stats_serial.csv")
ta = tb = 0
for k in sorted(a, key=lambda k: -a[k][1]):
    ca, ma = a[k]; cb, mb = b.get(k, (0, 0.0))
    ta += ma; tb += mb
    if ma > 2: print("%-60s calls %5d overlap %8.2f ms serial %8.2f ms  x%.2f" % (k, ca, ma, mb, ma / max(mb, 1e-9)))
print("total kernel ms: overlap %.1f serial %.1f  (103 keyframes)" % (ta, tb))
PY
tail -n 1 gpurun_out/r3_loop_prof_overlap.txt | cut -c1-300; tail -n 1 gpurun_out/r3_loop_prof_serial.txt | cut -c1-300
