#!/bin/bash
# Runs on the GPU box: why the order-relaxed RoiPoolGrad scatter is slow -- atomic counters of the L2 / fabric interface.
# tools/gpu_scatter_pmc.sh  ->  gpurun_out/scatter_pmc/
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/scatter_pmc
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -i -o "TCC[A-Z0-9_]*ATOMIC[A-Za-z0-9_]*\|TCP[A-Z0-9_]*ATOMIC[A-Za-z0-9_]*" | sort -u > "$OUT/atomic_counters.txt"
cat "$OUT/atomic_counters.txt"
export SCATTER=1 ONLY=bev+rgb+fv ROUNDS=2
for C in "TCC_ATOMIC_sum TCC_EA0_ATOMIC_sum" "TCC_EA0_ATOMIC_LEVEL_sum TCC_EA0_WRREQ_sum" "TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  T=$(echo $C | tr ' ' '_' | cut -c1-40)
  FILTER=roi_scat $GRAFT_REPO_ROOT/tools/pmc_any.sh scatter_pmc/$T "$C" tools/roi_bwd_probe.py 2>&1 | tail -6
done
