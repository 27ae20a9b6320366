#!/bin/bash
# GPU box, round 5: every batch's RoiPool calls on ONE shared stream (serialised chip-filling launches) against the batch's own stream
set -u
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05ay; mkdir -p $OUT
run() { timeout 600 python bench.py --steps 10 --warmup 2 --no-secondary --no-cpu-baseline "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['in_flight']['forward_us'], d['roofline']['in_flight']['backward_us'], d.get('verified',{}).get('bit_exact'))"; }
{
for r in 1 2 3; do
  echo "== own streams run $r"; run
  echo "== shared RoiPool stream run $r"; MV3D_BENCH_ROI_STREAM=1 run
  echo "== shared RoiPool stream, 12 in flight run $r"; MV3D_BENCH_ROI_STREAM=1 run --streams 12
done
} 2>&1 | tee $OUT/roi_stream_ab.txt
