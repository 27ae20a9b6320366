#!/bin/bash
# (EXPERIMENTS R6.16; NO_BENCH=1 skips the path A / B, which needs build_variants/libmv3d_static.so: see plan_bench_lease.sh)
# one round of the planned RoiPoolGrad: pair / pin / config tests, the pair alone (production build), per-wave stamps
# of the planned launch (tuning build), bench.py path mode planned against static
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/${1:-planround}
{
timeout 900 python -m pytest tests/test_roi_pair.py tests/test_roipool_pin.py tests/test_gpu_configs.py -x -q -m gpu 2>&1 | tail -2
echo "-- production build alone"; PAIR_ONLY=1 PAIR_NO_WS=1 NB=8 ROUNDS=4 timeout 300 python tools/roi_pair_probe.py 2>&1 | grep "pair \|differ\|rror" | tail -1
echo "-- per-wave stamps, planned (tuning build)"; timeout 300 python tools/roi_tiles_trace.py --lib build_variants/libmv3d_tuning.so 2>&1 | grep -v amdgpu.ids
[ -z "${NO_BENCH:-}" ] && [ -f build_variants/libmv3d_static.so ] && bash tools/experiments/plan_bench_lease.sh ${1:-planround}_bench 2>&1 | cut -c1-200
} 2>&1 | tee gpurun_out/${1:-planround}/round.txt
