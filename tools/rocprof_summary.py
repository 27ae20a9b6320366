#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls / avg / min / max / share.

    python tools/rocprof_summary.py gpurun_out/prof/x_results.db > profiles/rNN_name.txt
"""
import sqlite3
import sys


def main(path):
    c = sqlite3.connect(path)
    rows = c.execute("select name, count(*), avg(end-start), min(end-start), max(end-start), sum(end-start), "
                     "max(vgpr_count), max(sgpr_count), max(lds_size), max(grid_x), max(workgroup_x) "
                     "from kernels group by name order by sum(end-start) desc").fetchall()
    tot = sum(r[5] for r in rows) or 1
    print("%-64s %7s %10s %10s %10s %7s %5s %5s %7s" % ("kernel", "calls", "avg_us", "min_us", "max_us", "share%", "vgpr", "sgpr", "lds_B"))
    for r in rows:
        print("%-64s %7d %10.2f %10.2f %10.2f %7.1f %5s %5s %7s" % (r[0][:64], r[1], r[2] / 1e3, r[3] / 1e3, r[4] / 1e3,
                                                                   100.0 * r[5] / tot, r[6], r[7], r[8]))
    print("total kernel time: %.3f ms over %d dispatches" % (tot / 1e6, sum(r[1] for r in rows)))


if __name__ == "__main__":
    main(sys.argv[1])
