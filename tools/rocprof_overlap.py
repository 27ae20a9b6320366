#!/usr/bin/env python3
"""How the chip-filling launches of the path overlap in time (rocprofv3 kernel trace of `bench.py --launch path`):

    python tools/rocprof_overlap.py gpurun_out/<tag>/r_results.db

For the two RoiPool kernels: the share of every launch's duration during which ANOTHER RoiPool launch (forward or backward, any stream)
was running too, the average number of RoiPool launches running at a time, and the machine's view -- over the traced window, the
time with 0 / 1 / 2 / 3+ RoiPool launches in flight."""
import sqlite3
import sys

import numpy as np


def main(path):
    c = sqlite3.connect(path)
    rows = c.execute("select name, start, end from kernels order by start").fetchall()
    big = [(s, e, "bwd" if "roi_pair_tiles" in n else "fwd") for n, s, e in rows if "roi_pair_tiles" in n or "roi_pool_fwd_pair" in n]
    if not big:
        print("no RoiPool launches in the trace")
        return
    t0, t1 = min(s for s, _, _ in big), max(e for _, e, _ in big)
    ev = sorted([(s, 1) for s, _, _ in big] + [(e, -1) for _, e, _ in big])
    hist, cur, last = {}, 0, t0
    for t, d in ev:
        hist[cur] = hist.get(cur, 0) + (t - last)
        cur += d
        last = t
    tot = float(t1 - t0)
    print("window %.1f ms, %d RoiPool launches (%d forward, %d backward)" % (tot / 1e6, len(big), sum(k == "fwd" for _, _, k in big), sum(k == "bwd" for _, _, k in big)))
    print("time with n RoiPool launches running:", ", ".join("%d: %.1f %%" % (n, 100.0 * hist[n] / tot) for n in sorted(hist)))
    starts = np.array([s for s, _, _ in big]); ends = np.array([e for _, e, _ in big])
    for kind in ("fwd", "bwd"):
        sel = [i for i, (_, _, k) in enumerate(big) if k == kind]
        dur = np.array([ends[i] - starts[i] for i in sel], dtype=np.float64)
        ov = []
        for i in sel:
            o = np.minimum(ends, ends[i]) - np.maximum(starts, starts[i])
            o[i] = 0
            ov.append(np.clip(o, 0, None).sum() / max(ends[i] - starts[i], 1))
        print("%s: avg %.1f us (min %.1f, max %.1f); on average %.2f other RoiPool launches alongside" % (kind, dur.mean() / 1e3, dur.min() / 1e3, dur.max() / 1e3, float(np.mean(ov))))
    period = (t1 - t0) / (len(big) / 2.0)
    print("batch period %.1f us (window / batches)" % (period / 1e3))


if __name__ == "__main__":
    main(sys.argv[1])
