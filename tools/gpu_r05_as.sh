#!/bin/bash
# GPU box, round 5: the bench's path mode with RoiPoolGrad through index + gather (workspace) against the one-launch tile kernel (no workspace), alternating; f32 conv4_1 re-check
set -u
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05as; mkdir -p $OUT
{
for r in 1 2 3 4; do for t in "" 1; do
  echo "== bench path mode, no-workspace=${t:-0} run $r"
  MV3D_BENCH_ROI_GRAD_NO_WS=$t timeout 600 python bench.py --steps 10 --warmup 2 --no-secondary --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['avg_launch_us'], d['roofline']['in_flight']['forward_us'], d['roofline']['in_flight']['backward_us'], d.get('verified',{}).get('bit_exact'))"
done; done
echo "== exact-f32 conv4_1 / conv4_2"; timeout 300 python tools/conv_probe.py 16 --no-torch --f32 --only conv4_1,conv4_2 2>&1 | grep "conv4"
} 2>&1 | tee $OUT/bench_ws_ab.txt
