#!/bin/bash
# GPU box, round 5: the one-launch RoiPoolGrad with 1 / 2 / 4 independent waves (adjacent tiles of one slice) per workgroup: fewer workgroups to dispatch
set -u
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05aw; mkdir -p $OUT
cp mv3d_tf_amd/libmv3d_hip.so /tmp/ship.so; cp build_variants/libmv3d_tuning.so mv3d_tf_amd/libmv3d_hip.so
run() { timeout 600 python bench.py --steps 10 --warmup 2 --no-secondary --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['avg_launch_us'], d['roofline']['in_flight']['forward_us'], d['roofline']['in_flight']['backward_us'], d.get('verified',{}).get('bit_exact'))"; }
{
for nw in 2 4; do echo "== pair tests NW=$nw"; MV3D_RGT_NW=$nw timeout 600 python -m pytest tests/test_roi_pair.py -x -q -m gpu 2>&1 | tail -2; done
for r in 1 2 3; do for nw in 1 2 4; do echo "== MV3D_RGT_NW=$nw run $r"; MV3D_RGT_NW=$nw run; done; done
} 2>&1 | tee $OUT/tiles_nw_in_path.txt
cp /tmp/ship.so mv3d_tf_amd/libmv3d_hip.so
