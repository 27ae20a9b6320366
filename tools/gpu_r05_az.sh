#!/bin/bash
# GPU box, round 5: the pair's forward with four passes per workgroup (half as many, twice as long workgroups) on the training batch, in the path
set -u
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05az; mkdir -p $OUT
cp mv3d_tf_amd/libmv3d_hip.so /tmp/ship.so; cp build_variants/libmv3d_tuning.so mv3d_tf_amd/libmv3d_hip.so
run() { timeout 600 python bench.py --steps 10 --warmup 2 --no-secondary --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline_kernels'][0]['avg_launch_us'], d['roofline']['in_flight']['forward_us'], d['roofline']['in_flight']['backward_us'], d.get('verified',{}).get('bit_exact'))"; }
{
for r in 1 2 3; do for ps in 2 4; do echo "== MV3D_FWD_PASSES=$ps run $r"; MV3D_FWD_PASSES=$ps run; done; done
} 2>&1 | tee $OUT/fwd_passes_in_path.txt
cp /tmp/ship.so mv3d_tf_amd/libmv3d_hip.so
