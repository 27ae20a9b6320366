#!/bin/bash
# GPU box, round 5: the one-launch RoiPoolGrad inside the bench's path mode: loads in flight and tile sizes (experiment build)
set -u
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05at; mkdir -p $OUT
cp mv3d_tf_amd/libmv3d_hip.so /tmp/ship.so; cp build_variants/libmv3d_tuning.so mv3d_tf_amd/libmv3d_hip.so
run() { MV3D_BENCH_ROI_GRAD_NO_WS=1 timeout 600 python bench.py --steps 10 --warmup 2 --no-secondary --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['in_flight']['forward_us'], d['roofline']['in_flight']['backward_us'], d.get('verified',{}).get('bit_exact'))"; }
{
# tile pixels: fv | rgb << 8 | bev << 16 (launch order)
for r in 1 2; do
  echo "== default (W 16, 4 / 8 / 16 px) run $r"; run
  for w in 8 24; do echo "== W=$w run $r"; MV3D_RGT_W=$w run; done
  for px in $((2 + (8<<8) + (16<<16))) $((4 + (16<<8) + (16<<16))) $((4 + (8<<8) + (8<<16))) $((4 + (4<<8) + (8<<16))) $((8 + (8<<8) + (16<<16))); do echo "== PX=$(printf %x $px) run $r"; MV3D_RGT_PX=$px run; done
done
} 2>&1 | tee $OUT/tiles_in_path.txt
cp /tmp/ship.so mv3d_tf_amd/libmv3d_hip.so
