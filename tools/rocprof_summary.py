"""Turns a rocprofv3 results .db (rocpd sqlite) into the per-kernel summary text committed under profiles/.
Usage: python tools/rocprof_summary.py gpurun_out/prof/bench_results.db > profiles/rNN_*.txt"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
rows = list(c.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
print("# rocprofv3 --kernel-trace --stats summary (durations in microseconds)")
print("%-100s %8s %14s %12s %7s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
for name, calls, tot, avg, pct in rows:
    print("%-100s %8d %14.1f %12.3f %6.2f%%" % (name[:100], calls, tot, avg, pct))
