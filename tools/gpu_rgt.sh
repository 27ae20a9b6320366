#!/bin/bash
# GPU box: RoiPoolGrad gather scheduling (cost ranges vs the round-2 item striding): parity tests, probe, kernel trace.
set -u
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/rgt; mkdir -p $OUT
MV3D_HIPCC_FLAGS=-DMV3D_TUNING python -m mv3d_tf_amd.build --force > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
if [ "${TESTS:-0}" = 1 ]; then timeout 300 python tools/rgt_debug.py 2>&1 | grep -v "^   " | tail -10; timeout 900 python -m pytest tests/test_roipool_pin.py tests/test_gpu_configs.py tests/test_train_stream.py -m gpu -x -q 2>&1 | tail -5 | tee $OUT/tests.log; fi
for V in bev+rgb+fv fv rgb; do
  for OLD in 0 1; do
    export ONLY=$V; if [ $OLD = 1 ]; then export MV3D_BWG_STRIDED=1; else unset MV3D_BWG_STRIDED; fi
    echo "--- $V strided=$OLD"; timeout 300 python tools/roi_bwd_probe.py 2>&1 | tail -1
    (cd /tmp && export TMPDIR=/tmp && rm -rf $GRAFT_REPO_ROOT/$OUT/prof && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof -o r -- python $GRAFT_REPO_ROOT/tools/roi_bwd_probe.py > /dev/null 2>&1)
    python - $OUT/prof/r_results.db <<'PY'
import sqlite3, sys, statistics
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, end-start from kernels order by start").fetchall()
by = {}
for n, d in rows:
    if "roi_bwd" in n:
        by.setdefault(n.split("(")[0], []).append(d / 1e3)
for n, v in by.items():
    v.sort()
    top = v[len(v) // 2:]
    print("   %-44s calls %4d  median of upper half %7.2f us  max %7.2f" % (n[:44], len(v), statistics.median(top), v[-1]))
PY
  done
done
