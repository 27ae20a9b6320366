#!/bin/bash
# (EXPERIMENTS R6.16) the planned tile kernel at six waves per SIMD (-DRGT_WAVES6: 80 VGPRs, three spilled) against five (87 VGPRs): production
# flags both, the pair alone, then pair tests on the variant
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/occ
run() { echo "-- $1"; PAIR_ONLY=1 PAIR_NO_WS=1 NB=8 ROUNDS=4 timeout 300 python tools/roi_pair_probe.py --lib $1 2>&1 | grep "pair \|differ\|rror" | tail -1; }
{ for r in 1 2 3; do run mv3d_tf_amd/libmv3d_hip.so; run build_variants/libmv3d_w6.so; done; } 2>&1 | tee gpurun_out/occ/occ.txt
