#!/bin/bash
# GPU box: ablation of the pair's index launches (experiment build; MV3D_IDX_DBG bits: 1 no fill, 2 no prefix loads, 4 no list writes,
# 8 no pruning runs)
set -u
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05h; mkdir -p $OUT
for d in 0 1 2 4 8 7 15; do
  echo "== MV3D_IDX_DBG=$d" >> $OUT/ablate.txt
  PAIR_ONLY=1 NB=8 ROUNDS=4 MV3D_IDX_DBG=$d timeout 200 python tools/roi_pair_probe.py --lib build_variants/libmv3d_tuning.so 2>&1 | grep "pair " >> $OUT/ablate.txt
done
cat $OUT/ablate.txt
cd /tmp && export TMPDIR=/tmp
PAIR_ONLY=1 NB=8 ROUNDS=2 timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/ks -o r -- python $GRAFT_REPO_ROOT/tools/roi_pair_probe.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; python tools/rocprof_summary.py $OUT/ks/r_results.db > $OUT/kernel_stats.txt 2>&1; rm -rf $OUT/ks; head -8 $OUT/kernel_stats.txt
