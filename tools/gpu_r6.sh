#!/bin/bash
# Round 6's GPU script (round 5's, carried on): ONE parameterised entry point for every gpurun call.
#   tools/gpu_r6.sh <tag> <step> [<step> ...]       results under gpurun_out/<tag>_*
# steps: ab_costvol abl_costvol tests_cv tests_fc tests_conv fc_time loop_sliding ab_classloop tsdf16m corun engines (DR_ENGINES_LIST) fused_sweep fused_prio batch_proxy ops ab_front ab_head3 fn_quick ab_ops ab_tail ops_tail ops_prob cv_sweep loop_raycast bench_host tune tests tests_fast smoke bench bench_quick shipped pmc prof_seq prof_driver
#   the round's closing evidence, in the order bench.py needs it:  tools/gpu_r6.sh final pmc tests smoke bench shipped prof_seq prof_driver
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
    ab_costvol)  # k_costvol5 (product) against k_costvol3 (DR_COSTVOL_V3, parity build) and against its own A/B forms, headline shape and depth range 0.01 .. 10
      export DR_MI355X_LIB=$PWD/tandem_amd/libdr_mi355x_hooks.so DR_OPS_RANGE=0.01,10
      for r in 1 2; do
        echo "k_costvol5 (chunks of 4 planes, footprint reuse in one tap set, 4-row tiles): $(timeout 300 python tools/profile_ops.py "costvol" 2>&1 | tail -1)"
        echo "k_costvol5 one-row tiles everywhere: $(DR_CV5_ROWS=1 timeout 300 python tools/profile_ops.py "costvol" 2>&1 | tail -1)"
        echo "k_costvol5 four-row tiles everywhere: $(DR_CV5_ROWS=4 timeout 300 python tools/profile_ops.py "costvol" 2>&1 | tail -1)"
        echo "k_costvol5 chunks of 8 planes: $(DR_CV_DCHUNK1=8 DR_CV_DCHUNK2=8 DR_CV_DCHUNK3=8 timeout 300 python tools/profile_ops.py "costvol" 2>&1 | tail -1)"
        echo "k_costvol5 without footprint reuse: $(DR_CV5_REUSE=0 timeout 300 python tools/profile_ops.py "costvol" 2>&1 | tail -1)"
        echo "k_costvol3: $(DR_COSTVOL_V3=1 timeout 300 python tools/profile_ops.py "costvol" 2>&1 | tail -1)"
      done | tee $OUT/${TAG}_costvol_ab.txt
      DR_OPS_RANGE=0.5,5 bash -c 'echo "scene range 0.5 .. 5: k_costvol5: $(timeout 300 python tools/profile_ops.py costvol 2>&1 | tail -1)"; echo "scene range 0.5 .. 5: k_costvol3: $(DR_COSTVOL_V3=1 timeout 300 python tools/profile_ops.py costvol 2>&1 | tail -1)"' | tee -a $OUT/${TAG}_costvol_ab.txt
      unset DR_MI355X_LIB DR_OPS_RANGE ;;
    abl_costvol)  # where k_costvol5's time goes: ablations (parity build), depth chunks of 4, one-row tiles
      export DR_MI355X_LIB=$PWD/tandem_amd/libdr_mi355x_hooks.so DR_OPS_RANGE=0.01,10 
      for abl in 0 1 2 4 3 5 7; do echo "DR_CV5_ABL=$abl (1 no gathers, 2 no stores, 4 no tap arithmetic): $(DR_CV5_ABL=$abl timeout 300 python tools/profile_ops.py "costvol" 2>&1 | tail -1)"; done | tee $OUT/${TAG}_costvol_abl.txt
      unset DR_MI355X_LIB DR_OPS_RANGE ;;
    tsdf16m)  # SURVEY 8(d)'s fusion configuration once: a pool of 16 M voxel blocks (64 GB), everything else as the bench runs it
      timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-boundary --no-loop --no-tsdf-native --no-tsdf-boundary --tsdf-blocks 16000000 > $OUT/${TAG}_bench_tsdf16m.json 2> $OUT/${TAG}_bench_tsdf16m.err
      python -c "import json; d=json.load(open('$OUT/${TAG}_bench_tsdf16m.json'))['tsdf']; print({k: d[k] for k in ('value','ms_per_frame','blocks','kernel_ms_per_frame')}); print(d['config']['workload'][:400])" ;;
    tests_fc) timeout 1200 python -m pytest tests/test_mvsnet_gpu.py -m gpu -x -q -k "feature_cache or golden" > $OUT/${TAG}_gpu_tests_fc.log 2>&1; tail -15 $OUT/${TAG}_gpu_tests_fc.log ;;
    fc_time) timeout 600 python tools/time_feature_cache.py 2>&1 | tee $OUT/${TAG}_feature_cache_time.txt ;;
    loop_sliding)  # the TandemBackend loop on a sliding window, feature cache off / on
      timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-tsdf --no-boundary > $OUT/${TAG}_bench_loop.json 2> $OUT/${TAG}_bench_loop.err
      python -c "import json; d=json.load(open('$OUT/${TAG}_bench_loop.json'))['tandem_loop']; print({k: (v.get('keyframes_per_s'), v.get('mean_ms')) for k, v in d.items() if isinstance(v, dict) and 'keyframes_per_s' in v}); print(json.dumps(d.get('sliding_window'), indent=1))" ;;
    ab_classloop)  # k_conv's class loop (one workgroup per tile walks all parity classes of a single-pass transposed layer) against class-per-workgroup launches
      export DR_MI355X_LIB=$PWD/tandem_amd/libdr_mi355x_hooks.so DR_OPS_RANGE=0.01,10
      for r in 1 2; do
        echo "class loop: $(timeout 300 python tools/profile_ops.py "conv11|conv9|conv7" 2>&1 | tail -1)"
        echo "class per workgroup: $(DR_CONV_NO_CLASS_LOOP=1 timeout 300 python tools/profile_ops.py "conv11|conv9|conv7" 2>&1 | tail -1)"
      done | tee $OUT/${TAG}_classloop_ab.txt
      unset DR_MI355X_LIB DR_OPS_RANGE ;;
    tests_conv) timeout 1500 python -m pytest tests/test_conv_gpu.py tests/test_mvsnet_gpu.py -m gpu -x -q -k "transposed or deconv or golden or conv_matches or plan_candidate" > $OUT/${TAG}_gpu_tests_conv.log 2>&1; tail -4 $OUT/${TAG}_gpu_tests_conv.log ;;
    tests_cv) timeout 1200 python -m pytest tests/test_mvsnet_gpu.py -m gpu -x -q -k "cost_volume or golden" > $OUT/${TAG}_gpu_tests_cv.log 2>&1; tail -5 $OUT/${TAG}_gpu_tests_cv.log ;;
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
    ab_ops)  # per-op profile with every A/B library under build/ab (tools/build_ab.sh), the tree's own first
      timeout 300 python tools/profile_ops.py "${DR_OPS_RE:-.}" > $OUT/${TAG}_ops_tree.txt 2>&1; tail -1 $OUT/${TAG}_ops_tree.txt
      for so in build/ab/libdr_*.so; do n=$(basename $so .so); DR_MI355X_LIB=$PWD/$so timeout 300 python tools/profile_ops.py "${DR_OPS_RE:-.}" > $OUT/${TAG}_ops_$n.txt 2>&1; echo "--- $n"; tail -1 $OUT/${TAG}_ops_$n.txt; done ;;
    loop_raycast) timeout 900 python tools/exp_loop_raycast.py 100 > $OUT/${TAG}_loop_raycast.txt 2>&1; cat $OUT/${TAG}_loop_raycast.txt ;;
    tune)  # three autotune runs per tuned shape WITH the committed table in place; merge at home: python tools/merge_tuned.py gpurun_out/<tag>_tune_*.txt
      for r in 1 2 3; do
        DR_CONV_PRINT=1 timeout 600 python tools/try_autotune.py 400 > $OUT/${TAG}_tune_head_$r.txt 2>&1; tail -3 $OUT/${TAG}_tune_head_$r.txt | cut -c1-150
        DR_CONV_PRINT=1 timeout 600 python tools/try_autotune.py 400 320 512 48,4,4 > $OUT/${TAG}_tune_ship_$r.txt 2>&1; tail -3 $OUT/${TAG}_tune_ship_$r.txt | cut -c1-150
      done ;;
    ab_tail)  # (switches of the parity build: these A/Bs run on libdr_mi355x_hooks.so)
      export DR_MI355X_LIB=$PWD/tandem_amd/libdr_mi355x_hooks.so
       # DR_TAIL_FUSED=1 per-op profile with every A/B library under build/ab
      for so in build/ab/libdr_*.so; do nme=$(basename $so .so); echo "--- $nme: $(DR_TAIL_FUSED=1 DR_MI355X_LIB=$PWD/$so timeout 300 python tools/profile_ops.py "tail" 2>&1 | tail -1)"; done 
      unset DR_MI355X_LIB ;;
    cv_sweep)  # cost-volume depth chunks and register caps (A/B libraries under build/ab)
      for cfg in "" "DR_CV_DCHUNK1=8" "DR_CV_DCHUNK1=12" "DR_CV_DCHUNK1=16" "DR_CV_DCHUNK2=4" "DR_CV_DCHUNK2=16" "DR_CV_DCHUNK3=4"; do echo "[$cfg] $(env $cfg timeout 300 python tools/profile_ops.py "costvol" 2>&1 | tail -1)"; done
      for so in build/ab/libdr_*.so; do nme=$(basename $so .so); echo "--- $nme: $(DR_MI355X_LIB=$PWD/$so timeout 300 python tools/profile_ops.py "costvol" 2>&1 | tail -1)"; done ;;
    bench_host)  # the legs that cross the operator boundary: boundary, tandem_loop (no TSDF loop, no CPU baselines)
      timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-tsdf > $OUT/${TAG}_bench_host.json 2> $OUT/${TAG}_bench_host.err
      python - <<PY
import json
d=json.load(open("$OUT/${TAG}_bench_host.json"))
print("value %.1f  single %.3f  boundary %.3f  pinned %.3f" % (d["value"], d["single_window_ms"], d["boundary_single_engine_ms"], d["boundary_pinned_single_engine_ms"]))
print({k:(v.get("keyframes_per_s"),v.get("mean_ms")) for k,v in d["tandem_loop"].items() if isinstance(v,dict)})
PY
      ;;
    ops_prob)  # (switches of the parity build: these A/Bs run on libdr_mi355x_hooks.so)
      export DR_MI355X_LIB=$PWD/tandem_amd/libdr_mi355x_hooks.so
       # k_prob2<NR>: rows per lane 1 / 2 / 4
      for f in 1 2 4; do echo "DR_PROB_ROWS=$f: $(DR_PROB_ROWS=$f timeout 300 python tools/profile_ops.py "prob" 2>&1 | tail -1)"; done 
      unset DR_MI355X_LIB ;;
    ops_tail)  # (switches of the parity build: these A/Bs run on libdr_mi355x_hooks.so)
      export DR_MI355X_LIB=$PWD/tandem_amd/libdr_mi355x_hooks.so
       # per-op profile with the fused tail forms beside the default
      for f in 0 1 2; do DR_TAIL_FUSED=$f timeout 300 python tools/profile_ops.py "tail|conv11|prob" > $OUT/${TAG}_ops_tail$f.txt 2>&1; echo "DR_TAIL_FUSED=$f: $(tail -1 $OUT/${TAG}_ops_tail$f.txt)"; done 
      unset DR_MI355X_LIB ;;
    ab_front)  # (switches of the parity build: these A/Bs run on libdr_mi355x_hooks.so)
      export DR_MI355X_LIB=$PWD/tandem_amd/libdr_mi355x_hooks.so
       # k_fn_front against the three launches it replaces: per-op times, then the quick bench line, both ways
      for f in 1 0; do echo "DR_FN_FRONT=$f: $(DR_FN_FRONT=$f timeout 300 python tools/profile_ops.py "preprocess|fn.front|fn.conv0|fn.conv1.0" 2>&1 | tail -1)"; done | tee $OUT/${TAG}_front_ops.txt
      for f in 1 0 1 0; do
        DR_FN_FRONT=$f timeout 600 python bench.py --gpus 1 --steps 60 --warmup 5 --no-cpu --no-loop --no-boundary --no-tsdf > $OUT/${TAG}_bench_front$f.json 2> $OUT/${TAG}_bench_front$f.err
        python -c "import json; d=json.load(open('$OUT/${TAG}_bench_front$f.json')); print('DR_FN_FRONT=$f: value %.1f /s  ms_per_step %.3f  single_window %.3f ms' % (d['value'], d['ms_per_step'], d['single_window_ms']))"
      done | tee -a $OUT/${TAG}_front_ops.txt 
      unset DR_MI355X_LIB ;;
    ab_head3)  # (switches of the parity build: these A/Bs run on libdr_mi355x_hooks.so)
      export DR_MI355X_LIB=$PWD/tandem_amd/libdr_mi355x_hooks.so
       # k_fn_head3 against the four launches it replaces
      for f in 1 0; do echo "DR_FN_HEAD3=$f: $(DR_FN_HEAD3=$f timeout 300 python tools/profile_ops.py "fn.head3|fn.out3|fn.out2|fn.skip2" 2>&1 | tail -1)"; done | tee $OUT/${TAG}_head3_ops.txt
      for f in 1 0 1 0; do
        DR_FN_HEAD3=$f timeout 600 python bench.py --gpus 1 --steps 60 --warmup 5 --no-cpu --no-loop --no-boundary --no-tsdf > $OUT/${TAG}_bench_head3$f.json 2> $OUT/${TAG}_bench_head3$f.err
        python -c "import json; d=json.load(open('$OUT/${TAG}_bench_head3$f.json')); print('DR_FN_HEAD3=$f: value %.1f /s  ms_per_step %.3f  single_window %.3f ms' % (d['value'], d['ms_per_step'], d['single_window_ms']))"
      done | tee -a $OUT/${TAG}_head3_ops.txt 
      unset DR_MI355X_LIB ;;
    fn_quick)  # the two fused FeatureNet kernels as they stand: per-op times and the quick bench line, twice
      echo "$(timeout 300 python tools/profile_ops.py "fn.front|fn.head3" 2>&1 | tail -1)" | tee $OUT/${TAG}_fn_quick.txt
      for r in 1 2; do
        timeout 600 python bench.py --gpus 1 --steps 60 --warmup 5 --no-cpu --no-loop --no-boundary --no-tsdf > $OUT/${TAG}_bench_fnq.json 2> $OUT/${TAG}_bench_fnq.err
        python -c "import json; d=json.load(open('$OUT/${TAG}_bench_fnq.json')); print('value %.1f /s  ms_per_step %.3f  single_window %.3f ms' % (d['value'], d['ms_per_step'], d['single_window_ms']))"
      done | tee -a $OUT/${TAG}_fn_quick.txt ;;
    engines)  # windows in flight: 2 / 3 / 4 engines (the quick bench line each)
      for n in ${DR_ENGINES_LIST:-3 4 2 3 4}; do
        timeout 600 python bench.py --gpus 1 --steps 60 --warmup 5 --engines $n --no-cpu --no-loop --no-boundary --no-tsdf > $OUT/${TAG}_bench_eng.json 2> $OUT/${TAG}_bench_eng.err
        python -c "import json; d=json.load(open('$OUT/${TAG}_bench_eng.json')); print('engines $n: value %.1f /s  ms_per_step %.3f' % (d['value'], d['ms_per_step']))"
      done | tee $OUT/${TAG}_engines.txt ;;
    ab_launches)  # (switches of the parity build: these A/Bs run on libdr_mi355x_hooks.so)
      export DR_MI355X_LIB=$PWD/tandem_amd/libdr_mi355x_hooks.so
       # the edge filter's folded scans and prob + regression in one launch, against the forms they replace
      for f in 1 0; do echo "DR_FILTER_FUSED=$f DR_PROB_REGRESS=$f: $(DR_FILTER_FUSED=$f DR_PROB_REGRESS=$f timeout 300 python tools/profile_ops.py "filter|s3.prob|s3.regress" 2>&1 | tail -1)"; done | tee $OUT/${TAG}_launches_ops.txt
      for f in 1 0 1 0; do
        DR_FILTER_FUSED=$f DR_PROB_REGRESS=$f timeout 600 python bench.py --gpus 1 --steps 60 --warmup 5 --no-cpu --no-loop --no-boundary --no-tsdf > $OUT/${TAG}_bench_l$f.json 2> $OUT/${TAG}_bench_l$f.err
        python -c "import json; d=json.load(open('$OUT/${TAG}_bench_l$f.json')); print('fused=$f: value %.1f /s  ms_per_step %.3f  single_window %.3f ms' % (d['value'], d['ms_per_step'], d['single_window_ms']))"
      done | tee -a $OUT/${TAG}_launches_ops.txt 
      unset DR_MI355X_LIB ;;
    ops) timeout 600 python tools/profile_ops.py > $OUT/${TAG}_ops.txt 2>&1; tail -4 $OUT/${TAG}_ops.txt ;;
    pmc)  # counters in their own passes (--kernel-trace only), strictly sequential kernels (one engine, side stream off) -> profiles/r06_pmc_traffic.json, stamped with this tree's source hash
      export DR_MVS_NO_SIDE_STREAM=1
      S="--no-cpu --engines 1 --no-boundary --no-loop --no-tsdf-native --no-tsdf-boundary"; A="--steps 3 --warmup 1 --tsdf-frames 60 $S"
      rm -rf $OUT/pm1 $OUT/pm2 $OUT/pm3 $OUT/pm4
      timeout 600 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pm1 -o pmc -- python bench.py $A > $OUT/pm1.log 2>&1
      timeout 600 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pm2 -o pmc -- python bench.py $A > $OUT/pm2.log 2>&1
      timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pm3 -o pmc -- python bench.py $A > $OUT/pm3.log 2>&1
      timeout 600 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAIT_INST_LDS TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pm4 -o pmc -- python bench.py $A > $OUT/pm4.log 2>&1
      for i in 1 2 3 4; do d=$(dirname $(find $OUT/pm$i -name "pmc_counter_collection.csv" | head -1)); echo "== pass $i"; python tools/pmc_summary.py $d 2>&1 | grep -v "^at::\|elementwise\|    .*at::\|rocclr\|^void at" ; done > $OUT/${TAG}_pmc_summary.txt
      python tools/pmc_to_json.py profiles/r06_pmc_traffic.json $(for i in 1 2 3; do dirname $(find $OUT/pm$i -name "pmc_counter_collection.csv" | head -1); done)
      cp profiles/r06_pmc_traffic.json $OUT/; rm -rf $OUT/pm1 $OUT/pm2 $OUT/pm3 $OUT/pm4
      unset DR_MVS_NO_SIDE_STREAM ;;
    shipped) timeout 600 python bench.py --config shipped --steps 240 --no-tsdf --no-loop --no-cpu > $OUT/${TAG}_bench_shipped.json 2> $OUT/${TAG}_bench_shipped.err; head -c 300 $OUT/${TAG}_bench_shipped.json; echo ;;
    prof_seq)  # per-kernel durations without overlap: one engine, side stream off
      rm -rf $OUT/prof
      DR_MVS_NO_SIDE_STREAM=1 timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/prof -o bench -- python bench.py --steps 20 --warmup 3 --tsdf-frames 200 --no-cpu --engines 1 --no-boundary --no-loop --no-tsdf-native --no-tsdf-boundary > $OUT/${TAG}_bench_prof.json 2> $OUT/prof.err
      python tools/rocprof_summary.py $(find $OUT/prof -name "*_results.db" | head -1) > $OUT/${TAG}_bench_kernel_stats.txt 2>&1; head -14 $OUT/${TAG}_bench_kernel_stats.txt; rm -rf $OUT/prof ;;
    prof_driver)  # the same under the driver's own command (three engines in flight)
      rm -rf $OUT/prof
      timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/prof -o bench -- python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-boundary --no-loop --no-tsdf > $OUT/${TAG}_bench_prof_driver.json 2> $OUT/prof.err
      python tools/rocprof_summary.py $(find $OUT/prof -name "*_results.db" | head -1) > $OUT/${TAG}_bench_kernel_stats_driver_cmd.txt 2>&1; head -14 $OUT/${TAG}_bench_kernel_stats_driver_cmd.txt; rm -rf $OUT/prof ;;
    *) echo "unknown step $step" ;;
  esac
done
