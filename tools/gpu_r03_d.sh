#!/bin/bash
# GPU box, end of round 3: the default bench line, a kernel trace of the headline path and one of the serving step with the
# MFMA trunk; only text summaries are left under gpurun_out/ (the rocpd databases are hundreds of MB).
set -u
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03d; mkdir -p $OUT
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
tools/gpu_profile.sh r03d/ks --steps 4 --warmup 1 --batches-per-step 64 --no-cpu-baseline --no-secondary > /dev/null
python tools/rocprof_summary.py $OUT/ks/r_results.db > $OUT/kernel_stats.txt 2>&1; rm -rf $OUT/ks
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/sv -o r -- python $GRAFT_REPO_ROOT/tools/serve_probe.py fp16_mfma 6 > $GRAFT_REPO_ROOT/$OUT/serve.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocprof_summary.py $OUT/sv/r_results.db > $OUT/serve_kernel_stats.txt 2>&1; rm -rf $OUT/sv
head -12 $OUT/kernel_stats.txt; head -14 $OUT/serve_kernel_stats.txt
