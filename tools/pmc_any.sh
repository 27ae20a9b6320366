#!/bin/bash
# Runs on the GPU box: one rocprofv3 PMC pass (kernel-trace only) of an arbitrary python script, then prints the per-kernel
# average of every collected counter.   tools/pmc_any.sh <tag> "<COUNTER COUNTER ...>" <script> [args...]
set -u
TAG=$1; shift
CTRS=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
SCRIPT=$GRAFT_REPO_ROOT/$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc $CTRS -d "$OUT" -o r -- python "$SCRIPT" "$@" > "$OUT/run.log" 2>&1
python - "$OUT/r_results.db" "${FILTER:-}" <<'PY'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
flt = sys.argv[2]
rows = c.execute("select kernel_name, counter_name, count(*), avg(value), max(value) from counters_collection group by kernel_name, counter_name").fetchall()
for r in rows:
    if flt in r[0]:
        print("%-48s %-22s n=%4d avg %14.1f max %14.1f" % (r[0].split("(")[0][:48], r[1], r[2], r[3], r[4]))
PY
