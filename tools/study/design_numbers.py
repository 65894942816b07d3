"""Prints the figures DESIGN.md section 5 / README quote, from profiles/r06_bench_driver.json and r06_bench_shipped.json (so that the documents are edited from one source)."""
import json, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
d = json.load(open(os.path.join(ROOT, "profiles", "r06_bench_driver.json")))
sh = json.load(open(os.path.join(ROOT, "profiles", "r06_bench_shipped.json")))
t, lp, r, pl = d["tsdf"], d["tandem_loop"], d["roofline"], d["pipeline"]
print("value %.1f /s  ms_per_step %.3f (%.3f .. %.3f)  single %.3f ms  engines %d" % (d["value"], d["ms_per_step"], d["repeats"]["ms_per_step_min"], d["repeats"]["ms_per_step_max"], d["single_window_ms"], d["engines_per_gpu"]))
print("boundary %.2f ms (%.0f /s)  pinned %.2f ms  3 engines %.0f /s" % (d["boundary_single_engine_ms"], 1e3 / d["boundary_single_engine_ms"], d["boundary_pinned_single_engine_ms"], d["boundary"]["engines_3"]["depth_maps_per_s"]))
print("pipeline %.1f GFLOP %.1f TFLOP/s %.2f mfma  %.2f GB %.2f hbm" % (pl["gflop_per_depth_map"], pl["tflops"], pl["frac_mfma"], pl["gb_per_depth_map"], pl["frac_hbm"]))
print("roofline %s x%d %.3f ms %.1f TFLOP/s frac %.2f traffic %s" % (r["kernel"], r["launches_per_step"], r["avg_launch_ms"], r["achieved"], r["frac"], r["traffic"]))
print("parity mean %.2e max %.2e flips %.1e" % (d["parity"]["mean_abs_err"], d["parity"]["max_err"], d["parity"]["mask_flips"]))
print("cpu %.3f /s on %d cores" % (d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"]))
print("tsdf %.2f G vox/s %.3f ms" % (t["value"] / 1e9, t["ms_per_frame"]), t["kernel_ms_per_frame"], "integrate frac %.3f traffic %s" % (t["roofline"]["frac"], t["roofline"]["traffic"]))
print("tsdf boundary %.3f ms %.1f G vox/s" % (t["boundary"]["ms_per_frame"], t["boundary"]["value"] / 1e9), t["boundary"]["host_ms_per_frame"])
rc = t["roofline_raycast"]
print("raycast %.3f ms  %.1f samples/ray  %.2f GB  %.0f GB/s  frac %.2f  traffic %s / %s  counter frac %s" % (rc["avg_launch_ms"], rc["steps"]["samples_per_ray"], rc["bytes_per_launch"] / 1e9, rc["achieved"], rc["frac"], rc["traffic"], rc["traffic_if_fetch_doubled"], rc.get("hbm_frac_of_counter_traffic")))
print("tsdf parity", t["parity"]["tsdf_frames_equal"], t["parity"]["blocks"], " cpu %.0f M vox/s" % (t["cpu_baseline"]["value"] / 1e6))
print("loop 5mm %.0f  10mm %.0f  sliding %.0f -> %.0f (x%.2f)" % (lp["640x480_5mm"]["keyframes_per_s"], lp["640x480_10mm"]["keyframes_per_s"], lp["sliding_window"]["keyframes_per_s_cache_off"], lp["sliding_window"]["keyframes_per_s_cache_on"], lp["sliding_window"]["speedup"]))
print("tracker %.0f it/s   shipped %.0f /s single %.3f ms" % (d["tracker"]["gauss_newton_iterations_per_s"], sh["value"], sh["single_window_ms"]))
for row in pl["launches"]:
    if row["op"] in ("s2.conv0", "s2.conv1", "s2.conv11", "s2.prob", "s2.costvol", "fn.head3", "s1.costvol", "s3.costvol"):
        print("  ", row)
