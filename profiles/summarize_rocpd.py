#!/usr/bin/env python3
"""Turn a rocprofv3 (ROCm 7.2 rocpd sqlite) kernel trace into the text summary kept under profiles/.
usage: summarize_rocpd.py results.db [title]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
print("# %s" % (sys.argv[2] if len(sys.argv) > 2 else sys.argv[1]))
print("# rocprofv3 --kernel-trace --stats; durations in microseconds")
print("%-64s %8s %14s %12s %8s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
for name, calls, total, avg, pct in db.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
    short = name.split("(")[0]
    if len(short) > 62:
        short = short[:59] + "..."
    print("%-64s %8d %14.1f %12.3f %8.2f" % (short, calls, total, avg, pct))
