#!/usr/bin/env python3
"""Per (kernel, launch shape) table of a set of rocprofv3 PMC passes over ONE probe command (tools/gpu_r05_pmc.sh):

    python tools/pmc_kernel_table.py <dir with <pass>/r_results.db> [kernel-name filter]

One row per distinct (kernel name, grid size): launches, average duration, and what the passes' counters give:
  MFMA busy %   SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES)        (100 % = the matrix pipes of the busy CUs never idle)
  CU busy %     SQ_BUSY_CU_CYCLES / (256 CUs x GRBM_GUI_ACTIVE)                   (share of the launch the CUs hold waves at all)
  wait %        SQ_WAIT_ANY / SQ_WAVE_CYCLES   (waves parked at s_waitcnt / s_barrier), inst-stall % = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES
  LDS cnfl %    SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
  HBM rd / wr   FETCH_SIZE x 2 (gfx950 wide-read correction, MI355X_MICROARCH.md "HBM") and WRITE_SIZE, MB per launch
Counters of different passes are matched through (kernel, grid): the probe launches the same sequence in every pass."""
import os
import sqlite3
import sys

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
rows = {}
for sub in sorted(os.listdir(src)):
    db = os.path.join(src, sub, "r_results.db")
    if not os.path.exists(db):
        continue
    c = sqlite3.connect(db)
    try:
        it = c.execute("select kernel_name, grid_size, counter_name, count(*), avg(value), avg(duration) from counters_collection "
                       "group by kernel_name, grid_size, counter_name")
    except sqlite3.Error as e:
        print("# %s: %s" % (sub, e))
        continue
    for name, grid, ctr, n, v, d in it:
        if flt and flt not in name:
            continue
        r = rows.setdefault((name.split("(")[0], grid), {"n": n, "dur": {}})
        r[ctr] = v
        r["dur"][sub] = d
        r["n"] = n
g = lambda r, k: r.get(k, 0.0) or 0.0
print("%-64s %9s %5s %9s %7s %7s %6s %7s %6s %9s %9s" % ("kernel", "grid", "n", "avg us", "MFMA %", "CUbusy%", "wait %", "stall %", "LDSc %", "HBM rd MB", "HBM wr MB"))
for (name, grid), r in sorted(rows.items(), key=lambda kv: (kv[0][0], kv[0][1])):
    dur = min(r["dur"].values()) / 1e3 if r["dur"] else 0.0            # (the pass with the fewest counters perturbs least)
    mf = 100 * g(r, "SQ_VALU_MFMA_BUSY_CYCLES") / max(4 * g(r, "SQ_BUSY_CU_CYCLES"), 1) if "SQ_VALU_MFMA_BUSY_CYCLES" in r else float("nan")
    cu = 100 * g(r, "SQ_BUSY_CU_CYCLES") / max(256 * g(r, "GRBM_GUI_ACTIVE"), 1) if "GRBM_GUI_ACTIVE" in r and "SQ_BUSY_CU_CYCLES" in r else float("nan")
    wt = 100 * g(r, "SQ_WAIT_ANY") / max(g(r, "SQ_WAVE_CYCLES"), 1) if "SQ_WAIT_ANY" in r else float("nan")
    st = 100 * g(r, "SQ_WAIT_INST_ANY") / max(g(r, "SQ_WAVE_CYCLES"), 1) if "SQ_WAIT_INST_ANY" in r else float("nan")
    lc = 100 * g(r, "SQ_LDS_BANK_CONFLICT") / max(g(r, "SQ_LDS_IDX_ACTIVE"), 1) if "SQ_LDS_IDX_ACTIVE" in r else float("nan")
    rd = 2 * g(r, "FETCH_SIZE") * 1024 / 1e6 if "FETCH_SIZE" in r else float("nan")
    wr = g(r, "WRITE_SIZE") * 1024 / 1e6 if "WRITE_SIZE" in r else float("nan")
    print("%-64s %9d %5d %9.1f %7.1f %7.1f %6.1f %7.1f %6.1f %9.1f %9.1f" % (name[:64], grid, r["n"], dur, mf, cu, wt, st, lc, rd, wr))
