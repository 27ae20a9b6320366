#!/bin/bash
# GPU box: counters of the pair's kernels on the probe's batches
set -u
cd $GRAFT_REPO_ROOT
export PAIR_ONLY=1 NB=8 ROUNDS=2
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05i; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { sub=$1; shift; timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d "$OUT/$sub" -o r -- python $GRAFT_REPO_ROOT/tools/roi_pair_probe.py > "$OUT/$sub.log" 2>&1; }
run wait SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES
run inst SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
run lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA
run mem TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_sum
cd $GRAFT_REPO_ROOT
python - "$OUT" <<'PY'
import sqlite3, sys, os
for sub in ("wait", "inst", "lds", "mem"):
    db = os.path.join(sys.argv[1], sub, "r_results.db")
    if not os.path.exists(db):
        print(sub, "missing"); continue
    c = sqlite3.connect(db)
    rows = c.execute("select kernel_name, counter_name, count(*), avg(value), avg(duration) from counters_collection where kernel_name like '%roi_p%' group by kernel_name, counter_name").fetchall()
    for r in rows:
        print("%-6s %-44s %-26s n=%3d avg %16.1f  dur %8.1f us" % (sub, r[0].split('(')[0][:44], r[1], r[2], r[3], r[4] / 1e3))
PY
