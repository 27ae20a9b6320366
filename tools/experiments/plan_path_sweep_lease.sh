#!/bin/bash
# (EXPERIMENTS R6.16) bench.py path mode on the TUNING build (environment hooks live): planned mode, cut threshold, tile shapes.
# usage: plan_path_sweep_lease.sh <tag> "ENV=.. ENV=.." "ENV=.." ...   (one bench line per argument, the whole list twice)
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/$1; shift; mkdir -p $OUT
cp mv3d_tf_amd/libmv3d_hip.so /tmp/shipped.so; cp build_variants/libmv3d_tuning.so mv3d_tf_amd/libmv3d_hip.so
line() { env "$@" timeout 600 python bench.py --steps 10 --warmup 2 --no-secondary --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); f=d['roofline'].get('in_flight') or {}; print(d['value'], 'bwd alone', d['roofline'].get('avg_launch_us'), 'fwd alone', d['roofline_kernels'][0].get('avg_launch_us'), 'in flight', f.get('forward_us'), f.get('backward_us'), d['verified']['bit_exact'])"; }
{ for r in 1 2; do for cfg in "$@"; do echo "-- $cfg"; line $cfg; done; done; } 2>&1 | tee $OUT/sweep.txt
cp /tmp/shipped.so mv3d_tf_amd/libmv3d_hip.so
