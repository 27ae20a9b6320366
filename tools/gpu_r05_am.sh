#!/bin/bash
# GPU box, round 5: the one-wave tile RoiPoolGrad with its defaults (W = 32, 2x2 / 2x4 / 4x4 tiles) against index + gather: probe, pair tests, bench path mode
set -u
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05am; mkdir -p $OUT
L=build_variants/libmv3d_tuning.so
{
for r in 1 2; do for t in 0 1; do echo "== probe MV3D_PAIR_TILES=$t run $r"; PAIR_ONLY=1 MV3D_PAIR_TILES=$t timeout 200 python tools/roi_pair_probe.py --lib $L 2>&1 | grep "pair " | tail -1; done; done
cp mv3d_tf_amd/libmv3d_hip.so /tmp/ship.so; cp $L mv3d_tf_amd/libmv3d_hip.so
echo "== pytest"; MV3D_PAIR_TILES=1 timeout 900 python -m pytest tests/test_roi_pair.py tests/test_roipool_pin.py tests/test_gpu_configs.py -x -q -m gpu 2>&1 | tail -3
for r in 1 2 3; do for t in 0 1; do
  echo "== bench path mode, MV3D_PAIR_TILES=$t run $r"
  MV3D_PAIR_TILES=$t timeout 600 python bench.py --steps 10 --warmup 2 --no-secondary --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['avg_launch_us'], d['roofline']['in_flight'], d['roofline_kernels'][0].get('avg_launch_us'), d.get('verified',{}).get('bit_exact'))"
done; done
cp /tmp/ship.so mv3d_tf_amd/libmv3d_hip.so
} 2>&1 | tee $OUT/tiles_default_ab.txt
