cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/wsweep
TUN=build_variants/libmv3d_tuning.so
run() { echo "-- $*"; env "$@" MV3D_IDX_DBG=1 PAIR_ONLY=1 PAIR_NO_WS=1 NB=8 ROUNDS=4 timeout 300 python tools/roi_pair_probe.py --lib $TUN 2>&1 | grep "pair \|differ\|rror" | tail -1; }
{ for r in 1 2; do run MV3D_RGT_W=16; run MV3D_RGT_W=12; run MV3D_RGT_W=8; done; } 2>&1 | tee gpurun_out/wsweep/w.txt
