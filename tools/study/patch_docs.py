"""Rewrites the figures DESIGN.md section 5 / section 6 and README.md quote from profiles/r06_bench_driver.json and r06_bench_shipped.json (one source for the documents' numbers)."""
import json, os, re
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
d = json.load(open(root + '/profiles/r06_bench_driver.json')); sh = json.load(open(root + '/profiles/r06_bench_shipped.json'))
t = d['tsdf']; lp = d['tandem_loop']; r = d['roofline']; pl = d['pipeline']
s = ''
def sub(pattern, repl):
    global s
    new, n = re.subn(pattern, lambda m: repl, s, count=1, flags=re.S)
    assert n == 1, pattern
    s = new
p = root + '/DESIGN.md'; s = open(p).read()
sub(r"\*\*[\d.]+ depth maps/s\*\* = [\d.]+ ms per depth map \(median of 9 repeats, [\d.]+ … [\d.]+\) on the box of the final run, whose single window ran at [\d.]+ ms",
    f"**{d['value']:.1f} depth maps/s** = {d['ms_per_step']:.3f} ms per depth map (median of 9 repeats, {d['repeats']['ms_per_step_min']:.3f} … {d['repeats']['ms_per_step_max']:.3f}) on the box of the final run, whose single window ran at {d['single_window_ms']:.2f} ms")
sub(r"\| `single_window_ms` \(one engine, TANDEM's usage\) \| \*\*[\d.]+ ms\*\* in this run", f"| `single_window_ms` (one engine, TANDEM's usage) | **{d['single_window_ms']:.3f} ms** in this run")
sub(r"\| \*\*[\d.]+ ms\*\* = \d+ /s; three engines \d+ /s \|", f"| **{d['boundary_single_engine_ms']:.2f} ms** = {1e3/d['boundary_single_engine_ms']:.0f} /s; three engines {d['boundary']['engines_3']['depth_maps_per_s']:.0f} /s |")
sub(r"\(page-locked images in place, result views\) \| [\d.]+ ms \|", f"(page-locked images in place, result views) | {d['boundary_pinned_single_engine_ms']:.2f} ms |")
sub(r"\| whole pipeline \| [\d.]+ GFLOP / [\d.]+ ms = [\d.]+ TFLOP/s = [\d.]+ of the fp32 MFMA peak; [\d.]+ GB = [\d.]+ of HBM \|",
    f"| whole pipeline | {pl['gflop_per_depth_map']:.1f} GFLOP / {d['ms_per_step']:.3f} ms = {pl['tflops']:.1f} TFLOP/s = {pl['frac_mfma']:.2f} of the fp32 MFMA peak; {pl['gb_per_depth_map']:.2f} GB = {pl['frac_hbm']:.2f} of HBM |")
sub(r"2 launches per depth map, [\d.]+ ms each by hipEvents in this run: [\d.]+ TFLOP/s = \*\*[\d.]+\*\* of 157.3", f"2 launches per depth map, {r['avg_launch_ms']:.3f} ms each by hipEvents in this run: {r['achieved']:.1f} TFLOP/s = **{r['frac']:.2f}** of 157.3")
sub(r"\| `cpu_baseline` \| [\d.]+ depth-maps/s on \d+ cores", f"| `cpu_baseline` | {d['cpu_baseline']['value']:.3f} depth-maps/s on {d['cpu_baseline']['cores']} cores")
sub(r"\| [\d.]+ G voxels/s, [\d.]+ ms per frame \(allocate [\d.]+, integrate [\d.]+, ray-cast [\d.]+, hand-over [\d.]+\); `k_integrate` = \*\*[\d.]+ of HBM\*\*, counter traffic \d+ MB",
    f"| {t['value']/1e9:.2f} G voxels/s, {t['ms_per_frame']:.3f} ms per frame (allocate {t['kernel_ms_per_frame']['allocate_commit_cull']:.3f}, integrate {t['kernel_ms_per_frame']['integrate']:.3f}, ray-cast {t['kernel_ms_per_frame']['raycast']:.3f}, hand-over {t['kernel_ms_per_frame']['render_d2h']:.3f}); `k_integrate` = **{t['roofline']['frac']:.3f} of HBM**, counter traffic {t['roofline']['traffic']/1e6:.0f} MB")
sub(r"\| \*\*[\d.]+ ms per frame, [\d.]+ G voxels/s\*\*; host time per frame: IntegrateScanAsync [\d.]+, RenderAsync [\d.]+, GetRenderResult \(waits\) [\d.]+;",
    f"| **{t['boundary']['ms_per_frame']:.3f} ms per frame, {t['boundary']['value']/1e9:.1f} G voxels/s**; host time per frame: IntegrateScanAsync {t['boundary']['host_ms_per_frame']['IntegrateScanAsync']:.3f}, RenderAsync {t['boundary']['host_ms_per_frame']['RenderAsync']:.3f}, GetRenderResult (waits) {t['boundary']['host_ms_per_frame']['GetRenderResult']:.3f};")
sub(r"640x480_5mm: \*\*\d+ keyframes/s\*\*; 640x480_10mm: \*\*\d+\*\*; `sliding_window` \(new: each key frame drops the oldest image and adds a new one; 1 cm\): \d+ without, \*\*\d+ with the feature cache\*\* \(× [\d.]+;",
    f"640x480_5mm: **{lp['640x480_5mm']['keyframes_per_s']:.0f} keyframes/s**; 640x480_10mm: **{lp['640x480_10mm']['keyframes_per_s']:.0f}**; `sliding_window` (new: each key frame drops the oldest image and adds a new one; 1 cm): {lp['sliding_window']['keyframes_per_s_cache_off']:.0f} without, **{lp['sliding_window']['keyframes_per_s_cache_on']:.0f} with the feature cache** (× {lp['sliding_window']['speedup']:.2f};")
sub(r"\| `tracker` \| \d+ Gauss-Newton iterations/s \|", f"| `tracker` | {d['tracker']['gauss_newton_iterations_per_s']:.0f} Gauss-Newton iterations/s |")
sub(r"\| \*\*\d+ depth maps/s\*\* \(4 engines\), single window [\d.]+ ms \(`profiles/r06_bench_shipped.json`\)", f"| **{sh['value']:.0f} depth maps/s** (4 engines), single window {sh['single_window_ms']:.3f} ms (`profiles/r06_bench_shipped.json`)")
sub(r"≥ 600 depth maps/s: \*\*\d+ on the final run's box, 578–580 on the others\*\*", f"≥ 600 depth maps/s: **{d['value']:.0f} on the final run's box, 578–580 on the others**")
sub(r"Single window ≤ 2.0 ms: \*\*[\d.]+ ms in the final run \(2.14–2.16 on faster boxes\)\*\*", f"Single window ≤ 2.0 ms: **{d['single_window_ms']:.2f} ms in the final run (2.14–2.16 on faster boxes)**")
open(p, 'w').write(s)
p = root + '/README.md'; s = open(p).read()
sub(r"\*\*\d+ depth maps/s\*\* at 640×480×7 views, planes 48/32/8, fp32 \(4 windows in flight; 578–580 in the quick line on the round's other boxes, `profiles/r06_queues_side_stream.txt`; [\d.]+ ms for a",
    f"**{d['value']:.0f} depth maps/s** at 640×480×7 views, planes 48/32/8, fp32 (4 windows in flight; 578–580 in the quick line on the round's other boxes, `profiles/r06_queues_side_stream.txt`; {d['single_window_ms']:.2f} ms for a")
sub(r"views; [\d.]+ ms through `CallAsync`/`GetResult` with host buffers,\n[\d.]+ ms with page-locked images and result views\); \d+ /s \([\d.]+ ms\) for the shipped",
    f"views; {d['boundary_single_engine_ms']:.2f} ms through `CallAsync`/`GetResult` with host buffers,\n{d['boundary_pinned_single_engine_ms']:.2f} ms with page-locked images and result views); {sh['value']:.0f} /s ({sh['single_window_ms']:.2f} ms) for the shipped")
sub(r"\n[\d.]+ ms per 640×480 frame into a 5 mm grid \([\d.]+ G voxels/s device-resident, [\d.]+ ms / [\d.]+ G voxels/s through",
    f"\n{t['ms_per_frame']:.2f} ms per 640×480 frame into a 5 mm grid ({t['value']/1e9:.1f} G voxels/s device-resident, {t['boundary']['ms_per_frame']:.2f} ms / {t['boundary']['value']/1e9:.1f} G voxels/s through")
sub(r"driving all operators: \d+ keyframes/s at 5 mm,\n\d+ at TANDEM's 1 cm, \d+ on a sliding window with the feature cache \(\d+ without\)",
    f"driving all operators: {lp['640x480_5mm']['keyframes_per_s']:.0f} keyframes/s at 5 mm,\n{lp['640x480_10mm']['keyframes_per_s']:.0f} at TANDEM's 1 cm, {lp['sliding_window']['keyframes_per_s_cache_on']:.0f} on a sliding window with the feature cache ({lp['sliding_window']['keyframes_per_s_cache_off']:.0f} without)")
sub(r"on the box's 128 host cores: [\d.]+ depth maps/s;\nTSDF integration with OpenMP on the same cores: \d+ M voxels/s", f"on the box's 128 host cores: {d['cpu_baseline']['value']:.2f} depth maps/s;\nTSDF integration with OpenMP on the same cores: {t['cpu_baseline']['value']/1e6:.0f} M voxels/s")
open(p, 'w').write(s)
print('documents patched from', 'profiles/r06_bench_driver.json:', '%.1f /s' % d['value'])
