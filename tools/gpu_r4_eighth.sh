#!/bin/bash
# Round 4, eighth call: the Winograd form of the marching kernel on the GPU for the first time (conv cases, end-to-end bounds, autotune
# with its candidates ranked) and the boundary extensions (pinned upload, result view): their cases and the bench's boundary leg.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_conv_gpu.py tests/test_mvsnet_gpu.py tests/test_shim.py -m gpu -q --no-header -p no:cacheprovider -k "winograd or pinned or shim" > gpurun_out/r4h_tests.log 2>&1
echo "== winograd / boundary cases: $(grep -E 'passed|failed' gpurun_out/r4h_tests.log | tail -1)"; grep -E "^FAILED|^ERROR|Error|max\|err" gpurun_out/r4h_tests.log | head -12
for r in 1 2; do
  DR_CONV_PRINT=2 timeout 700 python tools/try_autotune.py 400 > gpurun_out/r4h_tune_headline_$r.txt 2>&1
  grep -E "^before|^after|^autotune\(" gpurun_out/r4h_tune_headline_$r.txt | tr '\n' ' '; echo; grep "^TUNED" gpurun_out/r4h_tune_headline_$r.txt
done
grep -E "cand s[123].conv0 .*winomarch" gpurun_out/r4h_tune_headline_1.txt | sort -t: -k2 -n | head -12
grep -E "cand s[123].conv0 " gpurun_out/r4h_tune_headline_1.txt | grep -v wino | sort -t: -k2 -n | head -6
for r in 1 2; do
  DR_CONV_PRINT=2 timeout 700 python tools/try_autotune.py 400 320 512 48,4,4 > gpurun_out/r4h_tune_shipped_$r.txt 2>&1
  grep -E "^before|^after|^autotune\(" gpurun_out/r4h_tune_shipped_$r.txt | tr '\n' ' '; echo; grep "^TUNED" gpurun_out/r4h_tune_shipped_$r.txt
done
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu --no-tsdf --no-loop > gpurun_out/r4h_bench.json 2> gpurun_out/r4h_bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4h_bench.json").read().strip().splitlines()[-1])
print({k: d.get(k) for k in ("value", "ms_per_step", "single_window_ms", "boundary_single_engine_ms", "boundary_pinned_single_engine_ms")})
print(d.get("boundary"))
PY
