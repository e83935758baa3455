#!/usr/bin/env python3
"""Dumps the per-kernel summary (`top_kernels` view) of a rocprofv3 rocpd .db as CSV.
usage: python tools/rocprof_summary.py gpurun_out/prof_rNN/x_results.db > profiles/rNN_kernel_stats.csv"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
print("kernel,calls,total_us,avg_us,percent")
for name, calls, total, avg, pct in c.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
    print('"%s",%d,%.3f,%.3f,%.3f' % (name.split("(")[0], calls, total, avg, pct))
