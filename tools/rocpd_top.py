#!/usr/bin/env python3
"""Per-kernel summary (calls, total / average duration in us, share) of a rocprofv3 rocpd database -- the same table as
`rocprofv3 --stats` writes in csv mode.   python3 tools/rocpd_top.py out/bench_results.db [rows]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 30
print("%-110s %8s %14s %12s %7s" % ("kernel", "calls", "total_us", "average_us", "%"))
for name, calls, total, avg, pct in db.execute("select name, total_calls, total_duration, average, percentage from top_kernels limit %d" % rows):
    print("%-110s %8d %14.1f %12.3f %7.2f" % (name[:110], calls, total, avg, pct))
