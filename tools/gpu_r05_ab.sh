#!/bin/bash
# GPU box, round 5: batches in flight (= HIP streams of the path driver)
set -u
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05ab; mkdir -p $OUT
for r in 1 2; do for s in 4 6 8 10 12 16; do
  echo -n "streams $s run $r: "
  timeout 600 python bench.py --no-secondary --no-cpu-baseline --steps 8 --warmup 2 --streams $s 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value', d['value'], [ (e['avg_launch_us'], e.get('in_flight_us')) for e in d['roofline_kernels']])"
done; done | tee $OUT/streams.txt
