#!/usr/bin/env python3
"""MFMA utilisation of the dense (torch / MIOpen / CK) kernels of a PMC pass over tools/trunk_step.py:
    python tools/mfma_summary.py gpurun_out/pmc_trunk/r_results.db
MIOpen's find step times candidate solvers (incl. its naive reference kernels) on the first call of every shape; those are
listed separately.  util = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES)."""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
rows = c.execute("select kernel_name, counter_name, count(*), sum(value) from counters_collection group by kernel_name, counter_name").fetchall()
d = {}
for k, cn, n, v in rows:
    d.setdefault(k, {})[cn] = (n, v)
g = lambda v, k: v.get(k, (0, 0.0))[1]
dense = {k: v for k, v in d.items() if g(v, "SQ_INSTS_VALU_MFMA_MOPS_F32") > 0}
naive = {k: v for k, v in d.items() if "naive_conv" in k}
busy = sum(g(v, "SQ_BUSY_CU_CYCLES") for v in dense.values())
mf = sum(g(v, "SQ_VALU_MFMA_BUSY_CYCLES") for v in dense.values())
print("# kernels that execute MFMA instructions (MIOpen igemm / CK xdlops convolutions, rocBLAS GEMMs of the FC head)")
print("%-84s %6s %12s %8s" % ("kernel", "calls", "mfma MOPS f32", "util"))
for k, v in sorted(dense.items(), key=lambda kv: -g(kv[1], "SQ_BUSY_CU_CYCLES"))[:18]:
    print("%-84s %6d %12.3e %7.1f%%" % (k[:84], v["SQ_WAVES"][0] if "SQ_WAVES" in v else 0, g(v, "SQ_INSTS_VALU_MFMA_MOPS_F32"),
                                        100 * g(v, "SQ_VALU_MFMA_BUSY_CYCLES") / max(4 * g(v, "SQ_BUSY_CU_CYCLES"), 1)))
print("all MFMA kernels: MFMA busy / (4 x CU busy) = %.1f%%  (%d kernels, %.3e MFMA MOPS f32)"
      % (100 * mf / max(4 * busy, 1), len(dense), sum(g(v, "SQ_INSTS_VALU_MFMA_MOPS_F32") for v in dense.values())))
print("MIOpen find-step reference kernels in this capture (first call per shape only): %d naive_conv kernels, %.3e CU-busy cycles"
      % (len(naive), sum(g(v, "SQ_BUSY_CU_CYCLES") for v in naive.values())))
