#!/bin/bash
# Runs on the GPU box: kernel-trace profile of an arbitrary python script.
#   tools/gpu_profile_cmd.sh <tag> <script> [args...]  ->  gpurun_out/<tag>/{r_results.db,run.log}
set -u
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
SCRIPT=$GRAFT_REPO_ROOT/$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$OUT" -o r -- python "$SCRIPT" "$@" > "$OUT/run.log" 2>&1
tail -8 "$OUT/run.log"
