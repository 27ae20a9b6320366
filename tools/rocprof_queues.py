#!/usr/bin/env python3
"""How many kernels of the path run at a time (rocprofv3 kernel trace of `bench.py --launch path`):

    python tools/rocprof_queues.py gpurun_out/<tag>/r_results.db

Over the steady part of the trace (between the first and the last RoiPoolGrad launch): the share of time with 0 / 1 / 2 / 3 / 4 / 5+ kernels
running, the average number running, the sum of kernel durations per batch and -- if the table has one -- the busy share of every queue.
Evidence for EXPERIMENTS R6.17: eight streams share four hardware queues that run one kernel at a time."""
import sqlite3
import sys


def main(path):
    c = sqlite3.connect(path)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)").fetchall()]
    qcol = next((k for k in ("queue_id", "queue", "stream_id", "stream") if k in cols), None)
    rows = c.execute("select name, start, end%s from kernels order by start" % ((", " + qcol) if qcol else "")).fetchall()
    bwd = [(r[1], r[2]) for r in rows if "roi_pair_tiles" in r[0]]
    if len(bwd) < 4:
        print("no steady part in the trace (columns: %s)" % cols)
        return
    t0, t1 = bwd[2][0], bwd[-2][1]
    rows = [r for r in rows if r[1] >= t0 and r[2] <= t1]
    ev = sorted([(r[1], 1) for r in rows] + [(r[2], -1) for r in rows])
    hist, cur, last = {}, 0, t0
    for t, d in ev:
        hist[min(cur, 5)] = hist.get(min(cur, 5), 0) + (t - last)
        cur += d
        last = t
    tot = float(t1 - t0)
    nb = sum(1 for r in rows if "roi_pair_tiles" in r[0])
    busy = sum(r[2] - r[1] for r in rows)
    print("columns: %s" % cols)
    print("window %.2f ms, %d kernels, %d batches: %.1f us per batch, %.1f us of kernel durations per batch, %.2f kernels running on average"
          % (tot / 1e6, len(rows), nb, tot / 1e3 / nb, busy / 1e3 / nb, busy / tot))
    print("time with n kernels running: " + "  ".join("%s: %.1f %%" % (("%d" % k) if k < 5 else "5+", 100.0 * hist.get(k, 0) / tot) for k in range(6)))
    if qcol:
        per = {}
        for r in rows:
            per[r[3]] = per.get(r[3], 0) + (r[2] - r[1])
        print("busy share per %s: " % qcol + "  ".join("%s: %.0f %%" % (k, 100.0 * v / tot) for k, v in sorted(per.items(), key=lambda kv: str(kv[0]))))


if __name__ == "__main__":
    main(sys.argv[1])
