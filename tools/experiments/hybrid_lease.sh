cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/hyb1; mkdir -p $OUT
TUN=build_variants/libmv3d_tuning.so
run() { echo "-- $*"; env "$@" PAIR_ONLY=1 PAIR_NO_WS=1 NB=8 ROUNDS=4 timeout 300 python tools/roi_pair_probe.py --lib $TUN 2>&1 | grep "pair \|differ\|rror" | tail -1; }
{
run MV3D_RGT_MODE=0
for h in 0 8 16 24 32 48 64; do run MV3D_RGT_MODE=1 MV3D_RGT_HOT=$h; done
run MV3D_RGT_MODE=0
for h in 16 32; do run MV3D_RGT_MODE=1 MV3D_RGT_HOT=$h MV3D_RGT_PX=1052673;  run MV3D_RGT_MODE=1 MV3D_RGT_HOT=$h MV3D_RGT_PX=1052674; done
} 2>&1 | tee $OUT/hyb.txt
