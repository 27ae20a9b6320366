#!/usr/bin/env python3
"""Per-phase view of a rocprofv3 kernel trace: consecutive dispatches of the same kernel sequence are grouped, so that a
probe that times configuration A, then B, ... shows one line per (configuration, kernel).
    python tools/rocprof_runs.py gpurun_out/<tag>/r_results.db [name filter]"""
import sqlite3
import sys
from collections import OrderedDict

c = sqlite3.connect(sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
rows = c.execute("select name, start, end, grid_x from kernels order by start").fetchall()
rows = [r for r in rows if flt in r[0]]
groups = []          # (key, {name: [durs]})
for name, s, e, gx in rows:
    key = (name.split("(")[0][:40], gx)
    if groups and key in groups[-1][1]:
        groups[-1][1][key].append((e - s) / 1e3)
        continue
    if groups and len(groups[-1][1]) < 4 and all(len(v) <= 1 for v in groups[-1][1].values()):
        groups[-1][1][key] = [(e - s) / 1e3]
        continue
    groups.append((None, OrderedDict([(key, [(e - s) / 1e3])])))
for _, g in groups:
    print(" | ".join("%s grid %d: n=%d avg %.1f min %.1f max %.1f us" % (k[0], k[1], len(v), sum(v) / len(v), min(v), max(v)) for k, v in g.items()))
