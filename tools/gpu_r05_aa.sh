#!/bin/bash
# GPU box, round 5: the headline with the host cores a rank gets at N = 8 on this box (16 usable cores / 8 ranks = 2)
set -u
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05aa; mkdir -p $OUT
nproc; python -c "import os; print(len(os.sched_getaffinity(0)))"
for c in "" "0-3" "0-2" "0-1" "0"; do
  echo "== cores: ${c:-all}"
  if [ -z "$c" ]; then cmd="python"; else cmd="taskset -c $c python"; fi
  timeout 600 $cmd bench.py --no-secondary --no-cpu-baseline --steps 8 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value', d['value'], 'host_cpu_s_per_step', d['config'].get('host_cpu_s_per_step'), 'ms_per_step', d['ms_per_step'], d['config'].get('host_cores',{}).get('usable'))"
done | tee $OUT/host_cores.txt
echo "== graph launch mode, 2 cores"; taskset -c 0-1 timeout 600 python bench.py --no-secondary --no-cpu-baseline --steps 8 --warmup 2 --launch graph 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value', d['value'], 'host_cpu_s_per_step', d['config'].get('host_cpu_s_per_step'))" | tee -a $OUT/host_cores.txt
