#!/bin/bash
# (EXPERIMENTS R6.17) runtime knobs for the path's small copies and launches: one bench line per argument ("ENV=.. ENV=.."), the list twice
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/$1; shift; mkdir -p $OUT
line() { env "$@" timeout 600 python bench.py --steps 10 --warmup 2 --no-secondary --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); f=d['roofline'].get('in_flight') or {}; print(d['value'], 'bwd alone', d['roofline'].get('avg_launch_us'), 'in flight', f.get('forward_us'), f.get('backward_us'), d['verified']['bit_exact'], 'host cpu s/step', d['config'].get('host_cpu_s_per_step'))"; }
{ for r in 1 2; do for cfg in "$@"; do echo "-- $cfg"; line $cfg; done; done; } 2>&1 | tee $OUT/env.txt
