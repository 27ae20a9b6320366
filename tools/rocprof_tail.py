#!/usr/bin/env python3
"""Per-kernel totals over the LAST part of a rocprofv3 kernel trace (steady state: skips warm-up / MIOpen find kernels):
    python tools/rocprof_tail.py <results.db> <fraction of the time span, e.g. 0.4> <steps in that part>"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
frac, steps = float(sys.argv[2]), float(sys.argv[3])
t0, t1 = c.execute("select min(start), max(end) from kernels").fetchone()
cut = t1 - (t1 - t0) * frac
rows = c.execute("select name, count(*), sum(end-start) from kernels where start >= ? group by name order by 3 desc", (cut,)).fetchall()
tot = sum(r[2] for r in rows)
print("window %.1f ms, kernel time %.1f ms, per step: kernel %.2f ms of %.2f ms wall" % ((t1 - cut) / 1e6, tot / 1e6, tot / 1e6 / steps, (t1 - cut) / 1e6 / steps))
for r in rows[:40]:
    print("%-90s n/step %6.1f  ms/step %7.3f  %5.1f%%" % (r[0][:90], r[1] / steps, r[2] / 1e6 / steps, 100.0 * r[2] / tot))
