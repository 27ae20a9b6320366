#!/bin/bash
# GPU box, round 5: the one-launch RoiPoolGrad with its waves per SIMD capped by dynamic LDS padding (6 -> 5 / 4 / 3), in the bench's path mode
set -u
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05ax; mkdir -p $OUT
cp mv3d_tf_amd/libmv3d_hip.so /tmp/ship.so; cp build_variants/libmv3d_tuning.so mv3d_tf_amd/libmv3d_hip.so
run() { timeout 600 python bench.py --steps 10 --warmup 2 --no-secondary --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['avg_launch_us'], d['roofline']['in_flight']['forward_us'], d['roofline']['in_flight']['backward_us'], d.get('verified',{}).get('bit_exact'))"; }
{
for r in 1 2; do for pad in 0 2800 4800 8200; do echo "== MV3D_RGT_LDS_PAD=$pad run $r"; MV3D_RGT_LDS_PAD=$pad run; done; done
} 2>&1 | tee $OUT/tiles_occupancy_in_path.txt
cp /tmp/ship.so mv3d_tf_amd/libmv3d_hip.so
