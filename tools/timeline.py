#!/usr/bin/env python3
"""Kernel and copy timeline of the LAST packed call of a rocprofv3 --kernel-trace --memory-copy-trace run (rocpd .db): start / end in ms
relative to the first event shown, so that overlap (or its absence) between the copy-out of one part and the kernels of the next is visible.
usage: python tools/timeline.py x_results.db [last_ms=80] [anchor [back_ms=8]]   (anchor: a kernel name -- the window then starts back_ms before the LAST
launch of that kernel instead of last_ms before the end of the run)"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
span = float(sys.argv[2]) if len(sys.argv) > 2 else 80.0
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
ev = []
kt = [t for t in tabs if t.startswith("kernels")] or [t for t in tabs if "kernel_dispatch" in t]
mt = [t for t in tabs if t.startswith("memory_copies")] or [t for t in tabs if "memory_copy" in t]
if kt:
    cols = [r[1] for r in c.execute("pragma table_info(%s)" % kt[0])]
    nm = "name" if "name" in cols else "kernel_name"
    for name, s, e in c.execute("select %s, start, end from %s" % (nm, kt[0])):
        ev.append((s, e, "K " + str(name).split("(")[0]))
if mt:
    cols = [r[1] for r in c.execute("pragma table_info(%s)" % mt[0])]
    nm = "name" if "name" in cols else cols[1]
    sz = "size" if "size" in cols else None
    q = "select %s, start, end%s from %s" % (nm, (", " + sz) if sz else "", mt[0])
    for row in c.execute(q):
        ev.append((row[1], row[2], "C %s %s" % (row[0], ("%.1f MB" % (row[3] / 1e6)) if sz else "")))
ev.sort()
if not ev:
    print("no events; tables:", tabs)
    sys.exit(0)
t_end = ev[-1][1]
if len(sys.argv) > 3:
    hits = [x for x in ev if x[2][2:].startswith(sys.argv[3])]
    if hits:
        t_end = hits[-1][0] - (float(sys.argv[4]) if len(sys.argv) > 4 else 8.0) * 1e6 + span * 1e6
sel = [x for x in ev if t_end - span * 1e6 <= x[0] <= t_end and (x[1] - x[0]) > 50e3]
t0 = sel[0][0]
for s, e, n in sel:
    print("%9.3f %9.3f  %7.3f ms  %s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, n))
