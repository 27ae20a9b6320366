#!/bin/bash
# GPU box, round 5: the train-graph test that checks the pair's arguments, the rest of the suite after it, and batches in flight with the one-launch RoiPoolGrad
set -u
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05au; mkdir -p $OUT
{
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
for r in 1 2; do for st in 4 6 8 12; do
  echo "== --streams $st run $r"; timeout 600 python bench.py --steps 10 --warmup 2 --streams $st --no-secondary --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['in_flight']['forward_us'], d['roofline']['in_flight']['backward_us'], d.get('verified',{}).get('bit_exact'))"
done; done
} 2>&1 | tee $OUT/streams_tiles.txt
