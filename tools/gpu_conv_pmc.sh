#!/bin/bash
# Runs on the GPU box: PMC passes (kernel-trace only, one counter group per pass, FETCH_SIZE and WRITE_SIZE in passes of their
# own -- together they hung the box's profiler once -- each under a timeout) over tools/conv_probe.py --no-torch.
#   tools/gpu_conv_pmc.sh <tag>   ->  gpurun_out/<tag>/{mfma,lds,fetch}/r_results.db + a printed per-kernel summary
set -u
TAG=${1:-conv_pmc}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_WAVES -d "$OUT/mfma" -o r -- python "$GRAFT_REPO_ROOT/tools/conv_probe.py" 16 --no-torch > "$OUT/mfma.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS -d "$OUT/lds" -o r -- python "$GRAFT_REPO_ROOT/tools/conv_probe.py" 16 --no-torch > "$OUT/lds.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/fetch" -o r -- python "$GRAFT_REPO_ROOT/tools/conv_probe.py" 16 --no-torch > "$OUT/fetch.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/write" -o r -- python "$GRAFT_REPO_ROOT/tools/conv_probe.py" 16 --no-torch > "$OUT/write.log" 2>&1
python - "$OUT" <<'PY'
import sqlite3, sys, os
for sub in ("mfma", "lds", "fetch", "write"):
    db = os.path.join(sys.argv[1], sub, "r_results.db")
    if not os.path.exists(db):
        print(sub, "missing"); continue
    c = sqlite3.connect(db)
    # per launch configuration (grid size tells the layers apart)
    rows = c.execute("select kernel_name, grid_size, counter_name, count(*), avg(value) from counters_collection where kernel_name like '%conv3x3%' group by kernel_name, grid_size, counter_name").fetchall()
    for r in rows:
        print("%-6s %-58s grid %9d %-28s n=%3d avg %16.1f" % (sub, r[0][16:74], r[1], r[2], r[3], r[4]))
PY
