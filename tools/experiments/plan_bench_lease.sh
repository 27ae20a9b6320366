#!/bin/bash
# (EXPERIMENTS R6.16) bench.py path mode, planned against the static tile grid, alternating.  Production flags both; build the static variant first:
#   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math -pthread -DRGT_PLAN_DEFAULT=0 \
#         mv3d_tf_amd/csrc/*.hip -o build_variants/libmv3d_static.so
cd "$GRAFT_REPO_ROOT"; OUT=gpurun_out/${1:-planbench}; mkdir -p $OUT
cp mv3d_tf_amd/libmv3d_hip.so /tmp/planned.so
line() { timeout 600 python bench.py --steps 10 --warmup 2 --no-secondary --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline'].get('avg_launch_us'), d['roofline'].get('in_flight'), [k.get('avg_launch_us') for k in d.get('roofline_kernels', [])], d.get('verified'))"; }
{ for r in 1 2 3; do
echo "-- planned"; cp /tmp/planned.so mv3d_tf_amd/libmv3d_hip.so; line
echo "-- static"; cp build_variants/libmv3d_static.so mv3d_tf_amd/libmv3d_hip.so; line
done; cp /tmp/planned.so mv3d_tf_amd/libmv3d_hip.so; } 2>&1 | tee $OUT/planbench.txt
