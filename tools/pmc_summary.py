"""Per-kernel means of rocprofv3 --pmc counter CSVs.  Usage: python tools/pmc_summary.py dir [dir ...]"""
import collections
import csv
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    for row in csv.DictReader(open(d + "/pmc_counter_collection.csv")):
        name = row["Kernel_Name"].split("(")[0].replace("void ", "")
        acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
names = sorted(acc, key=lambda n: -sum(acc[n].get("GRBM_GUI_ACTIVE", [0])))
ctrs = sorted({c for n in acc for c in acc[n]})
for n in names:
    calls = len(acc[n].get("GRBM_GUI_ACTIVE", [])) or 1
    print("%s  (dispatch records %d)" % (n, calls))
    for c in ctrs:
        v = acc[n].get(c)
        if v:
            print("    %-28s mean %14.1f  total %16.0f" % (c, sum(v) / len(v), sum(v)))
