#!/bin/bash
# GPU box: PMC passes (kernel-trace only; one counter group per pass, FETCH_SIZE and WRITE_SIZE in passes of their own) over ONE probe
# command, then the per-kernel table.     tools/gpu_r05_pmc.sh <tag> <probe command ...>     -> gpurun_out/<tag>/table.txt
set -u
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() { sub=$1; shift; timeout 420 rocprofv3 --kernel-trace --pmc "$@" -d "$OUT/$sub" -o r -- "${CMD[@]}" > "$OUT/$sub.log" 2>&1; }
CMD=("$@")
run mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE
run wait SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
run lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS
run fetch FETCH_SIZE
run write WRITE_SIZE
cd $GRAFT_REPO_ROOT
python tools/pmc_kernel_table.py "$OUT" > "$OUT/table.txt" 2>&1
for s in mfma wait lds fetch write; do rm -rf "$OUT/$s"; done
