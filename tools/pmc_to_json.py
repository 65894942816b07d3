"""rocprofv3 --pmc CSVs -> profiles/rNN_pmc_traffic.json (per-kernel means per launch).
Usage: python tools/pmc_to_json.py out.json dir [dir ...]
Units: FETCH_SIZE / WRITE_SIZE are KiB.  MI355X_MICROARCH.md (HBM): on gfx950 FETCH_SIZE reports half the bytes of a
wide (16 B/lane) coalesced stream -> `fetch_bytes_corrected` doubles it; other widths are uncalibrated."""
import collections
import csv
import json
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[2:]:
    for row in csv.DictReader(open(d + "/pmc_counter_collection.csv")):
        name = row["Kernel_Name"].split("(")[0].replace("void ", "").replace("dr::", "").replace(" ", "")
        acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
out = {}
for n, c in acc.items():
    e = {"launches_sampled": max(len(v) for v in c.values())}
    for k, v in c.items():
        e[k] = sum(v) / len(v)
    if "FETCH_SIZE" in e:
        e["fetch_bytes_raw"] = e["FETCH_SIZE"] * 1024
        e["fetch_bytes_corrected"] = e["FETCH_SIZE"] * 2048
    if "WRITE_SIZE" in e:
        e["write_bytes"] = e["WRITE_SIZE"] * 1024
    if "SQ_VALU_MFMA_BUSY_CYCLES" in e and e.get("GRBM_GUI_ACTIVE"):
        e["mfma_util"] = e["SQ_VALU_MFMA_BUSY_CYCLES"] / (e["GRBM_GUI_ACTIVE"] / 8 * 1024)  # GUI_ACTIVE is summed over 8 XCDs
    out[n] = e
# stamp: the sources these counters were taken on (bench.py refuses a profile whose stamp differs from its own tree's)
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
out["_meta"] = {"source_stamp": bench.source_stamp(), "kernels": sorted(out)}
json.dump(out, open(sys.argv[1], "w"), indent=1, sort_keys=True)
print("wrote", sys.argv[1], len(out), "kernels")
