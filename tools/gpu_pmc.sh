#!/bin/bash
# Runs on the GPU box: PMC passes (own runs, kernel-trace only, one counter each) for the bench workload, eager launches on one
# stream.   tools/gpu_pmc.sh <tag> [bench args...]  ->  gpurun_out/<tag>/{fetch,write}/r_results.db
set -u
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--launch eager --streams 1 --steps 2 --warmup 1 --batches-per-step 64 --no-cpu-baseline --no-secondary"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/fetch" -o r -- python "$GRAFT_REPO_ROOT/bench.py" $ARGS "$@" > "$OUT/fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/write" -o r -- python "$GRAFT_REPO_ROOT/bench.py" $ARGS "$@" > "$OUT/write.log" 2>&1
ls -la "$OUT/fetch" "$OUT/write" | head; tail -1 "$OUT/fetch.log" | cut -c1-300
