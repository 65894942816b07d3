"""Markdown table of a bench line for DESIGN.md section 5: python tools/fill_numbers.py bench_driver.json [bench_shipped.json]"""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r, t, b = d["roofline"], d.get("tsdf", {}), d.get("boundary", {})
rows = [
    ("`value` (3 engines in flight, fp32, inputs resident)", "%.1f depth maps/s = %.3f ms per depth map" % (d["value"], d["ms_per_step"])),
    ("`single_window_ms` (one engine, TANDEM's usage)", "%.3f ms" % d["single_window_ms"]),
    ("`boundary_single_engine_ms` (CallAsync(host u8) → GetResult(host maps))", "%.2f ms = %.0f /s" % (d["boundary_single_engine_ms"], d["boundary_single_engine_depth_maps_per_s"])),
    ("`boundary_pinned_single_engine_ms` (page-locked images in place, result views)", "%.2f ms" % d.get("boundary_pinned_single_engine_ms", float("nan"))),
    ("whole pipeline", "%.1f GFLOP / %.3f ms = %.1f TFLOP/s = %.2f of the fp32 MFMA peak; %.2f GB = %.2f of HBM" % (
        d["pipeline"]["gflop_per_depth_map"], d["ms_per_step"], d["pipeline"]["tflops"], d["pipeline"]["frac_mfma"], d["pipeline"]["gb_per_depth_map"], d["pipeline"]["frac_hbm"])),
    ("`roofline` (dominant kernel)", "`%s`, %d launches per depth map, %.3f ms each: %.1f TFLOP/s = **%.2f** of %.1f; traffic %s" % (r["kernel"], r["launches_per_step"], r["avg_launch_ms"], r["achieved"], r["frac"], r["peak"], r.get("traffic"))),
    ("next kernels by time", "; ".join("`%s` ×%d %.3f ms %.2f" % (k["kernel"], k["launches"], k["ms"], k["frac"]) for k in r.get("top", [])[1:6])),
    ("`cpu_baseline`", "%.3f %s on %d cores (%s)" % (d["cpu_baseline"]["value"], d["cpu_baseline"]["unit"], d["cpu_baseline"]["cores"], d["cpu_baseline"]["kind"])),
    ("`bf16x3_mode` (opt-in, never `value`)", "%.0f /s, %.2f ms per window" % (d["bf16x3_mode"]["depth_maps_per_s_3_engines"], d["bf16x3_mode"]["single_window_ms"])) if d.get("bf16x3_mode") else None,
    ("`tsdf` (1000 frames into an empty 5 mm map)", "%.2f G voxels/s, %.3f ms per frame (allocate %.3f, integrate %.3f, ray-cast %.3f, hand-over %.3f); `k_integrate` %.2f of HBM" % (
        t["value"] / 1e9, t["ms_per_frame"], t["kernel_ms_per_frame"]["allocate_commit_cull"], t["kernel_ms_per_frame"]["integrate"], t["kernel_ms_per_frame"]["raycast"],
        t["kernel_ms_per_frame"]["render_d2h"], t["roofline"]["frac"])) if t else None,
    ("`tandem_loop` (the reference's `tandem_backend.cpp`, unchanged)", "; ".join("%s: %.0f keyframes/s" % (k, v["keyframes_per_s"]) for k, v in d.get("tandem_loop", {}).items() if isinstance(v, dict))),
    ("`tracker`", "%.0f Gauss-Newton iterations/s (calcRes %.3f ms + calcG %.3f ms)" % (d["tracker"]["gauss_newton_iterations_per_s"], d["tracker"]["calc_res_ms"], d["tracker"]["calc_g_ms"])) if d.get("tracker") else None,
]
if len(sys.argv) > 2:
    s = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    rows.append(("shipped model (320×512×7, planes 48/4/4; `--config shipped`)", "%.0f depth maps/s, single window %.3f ms" % (s["value"], s["single_window_ms"])))
print("| leg | measured |\n|---|---|")
for row in rows:
    if row:
        print("| %s | %s |" % row)
