#!/bin/bash
# (EXPERIMENTS R6.17) what bounds the path rate now that RoiPoolGrad is a third shorter?  (a) tuning build: the backward reduced to its
# write-out (MV3D_RGT_DBG=16: wrong results, the RATE is the question); (b) production build: streams x hardware queues re-swept
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/${1:-pathbound}; mkdir -p $OUT
line() { timeout 600 env "$@" python bench.py --steps 10 --warmup 2 --no-secondary --no-cpu-baseline $ARGS 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); f=d['roofline'].get('in_flight') or {}; print(d['value'], 'bwd alone', d['roofline'].get('avg_launch_us'), 'in flight', f.get('forward_us'), f.get('backward_us'), d['verified']['bit_exact'], 'host cpu s/step', d['config'].get('host_cpu_s_per_step'))"; }
{
cp mv3d_tf_amd/libmv3d_hip.so /tmp/shipped.so; cp build_variants/libmv3d_tuning.so mv3d_tf_amd/libmv3d_hip.so
ARGS=""
for r in 1 2; do echo "-- tuning build"; line A=1; echo "-- tuning build, backward = write-out only"; line MV3D_RGT_DBG=16; done
cp /tmp/shipped.so mv3d_tf_amd/libmv3d_hip.so
for r in 1 2; do for cfg in $PATH_SWEEP; do ARGS="--streams ${cfg##*:}"; echo "-- GPU_MAX_HW_QUEUES=${cfg%%:*}, ${cfg##*:} streams"; line GPU_MAX_HW_QUEUES=${cfg%%:*}; done; done
} 2>&1 | tee $OUT/pathbound.txt
