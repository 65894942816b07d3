#!/bin/bash
# Round 5's GPU script: ONE parameterised entry point for every gpurun call (the r4 per-call scripts were folded into this form).
#   tools/gpu_r5.sh <tag> <step> [<step> ...]       results under gpurun_out/<tag>_*
# steps: build_ubench fused_sweep batch_proxy tests tests_fast bench bench_quick prof pmc smoke autotune
set -u
cd "$(dirname "$0")/.."
TAG=$1; shift
OUT=gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
for step in "$@"; do
  echo "=== $step ($(date +%T))"
  case $step in
    fused_sweep)
      (cd tools/ubench && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=fast -Wno-unused-function -Wno-pass-failed fused_sweep.hip -o fused_sweep 2>&1 | tail -3
       timeout 300 ./fused_sweep) > $OUT/${TAG}_fused_sweep.txt 2>&1; tail -20 $OUT/${TAG}_fused_sweep.txt ;;
    fused_prio)
      (cd tools/ubench && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=fast -Wno-unused-function -Wno-pass-failed fused_sweep.hip -o fused_sweep 2>&1 | grep -E "error" ; FS_PRIO=1 timeout 300 ./fused_sweep) > $OUT/${TAG}_fused_sweep_prio.txt 2>&1; tail -8 $OUT/${TAG}_fused_sweep_prio.txt ;;
    corun)
      (cd tools/ubench && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 mfma_valu_corun.hip -o mfma_valu_corun 2>&1 | grep -E "error"; timeout 120 ./mfma_valu_corun) > $OUT/${TAG}_corun.txt 2>&1; cat $OUT/${TAG}_corun.txt ;;
    batch_proxy) timeout 900 python tools/exp_batch_proxy.py > $OUT/${TAG}_batch_proxy.txt 2>&1; tail -12 $OUT/${TAG}_batch_proxy.txt ;;
    tests) timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/${TAG}_gpu_tests.log 2>&1; tail -5 $OUT/${TAG}_gpu_tests.log ;;
    tests_fast) timeout 1200 python -m pytest tests -m gpu -x -q -k "${DR_TESTS_K:-mvsnet or conv}" > $OUT/${TAG}_gpu_tests_fast.log 2>&1; tail -5 $OUT/${TAG}_gpu_tests_fast.log ;;
    smoke) timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/${TAG}_smoke.txt 2>&1; tail -2 $OUT/${TAG}_smoke.txt ;;
    bench) timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err; tail -c 600 $OUT/${TAG}_bench.json ;;
    bench_quick) timeout 900 python bench.py --gpus 1 --steps 60 --warmup 5 --no-cpu --no-loop --no-boundary --no-tsdf > $OUT/${TAG}_bench_quick.json 2> $OUT/${TAG}_bench_quick.err
                 python - <<PY
import json
d=json.load(open("$OUT/${TAG}_bench_quick.json"))
print("value %.1f /s  ms_per_step %.3f  single_window %.3f ms" % (d["value"], d["ms_per_step"], d["single_window_ms"]))
print(json.dumps(d["pipeline"]["kernels"]))
PY
      ;;
    ops) timeout 600 python tools/profile_ops.py > $OUT/${TAG}_ops.txt 2>&1; tail -4 $OUT/${TAG}_ops.txt ;;
    prof)
      rm -rf /tmp/prof_$TAG
      (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -- python $OLDPWD/bench.py --gpus 1 --steps 20 --warmup 5 --engines 1 --no-cpu --no-loop --no-boundary --no-tsdf-native > /dev/null 2> $OLDPWD/$OUT/${TAG}_prof.err)
      python tools/rocprof_summary.py /tmp/prof_$TAG > $OUT/${TAG}_kernel_stats.txt 2>&1; head -40 $OUT/${TAG}_kernel_stats.txt ;;
    *) echo "unknown step $step" ;;
  esac
done
