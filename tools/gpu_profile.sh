#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace profile of one bench.py invocation.
#   tools/gpu_profile.sh <tag> [bench args...]   ->  gpurun_out/<tag>/{r_results.db,bench.log}
set -u
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$OUT" -o r -- python "$GRAFT_REPO_ROOT/bench.py" "$@" > "$OUT/bench.log" 2>&1
grep '^{' "$OUT/bench.log" | cut -c1-700
