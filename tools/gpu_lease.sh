#!/bin/bash
# One parametrised script for everything run on a GPU lease (gpurun): tools/gpu_lease.sh <tag> <recipe> [recipe ...]
# Output under gpurun_out/<tag>/.  Recipes:
#   pair_tests     tests/test_roi_pair.py + tests/test_roipool_pin.py (-m gpu)
#   tests          the whole -m gpu suite
#   probe          tools/roi_pair_probe.py (RoiPool pair alone: index + gather against the one-launch tiles), PROBE_ENV="K=V ..." optional
#   trace          tools/roi_tiles_trace.py on build_variants/libmv3d_tuning.so (per-wave stamps of the tile kernel)
#   ablate         the tile kernel with phases switched off (tuning build, MV3D_RGT_DBG = 8 / 1 / 2)
#   bench          python bench.py $BENCH_ARGS -> bench.json
#   bench_path     bench.py path-only, no secondary legs, 3 runs
#   stats          rocprofv3 --kernel-trace --stats of a short bench run -> kernel_stats.txt
#   overlap        kernel trace of the path mode: how many RoiPool launches run at a time (tools/rocprof_overlap.py)
#   pmc            HBM traffic counters of the default training workload (tools/gpu_pmc.sh)
set -u
cd "$GRAFT_REPO_ROOT"
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p "$OUT"
TUN=build_variants/libmv3d_tuning.so
for recipe in "$@"; do
  echo "=== $recipe"
  case $recipe in
    pair_tests) timeout 1200 python -m pytest tests/test_roi_pair.py tests/test_roipool_pin.py -x -q -m gpu > $OUT/pair_tests.log 2>&1; tail -4 $OUT/pair_tests.log ;;
    tests) timeout 2700 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; tail -4 $OUT/pytest_gpu.log ;;
    probe) for r in 1 2; do
             echo "-- workspace (index + gather)"; env ${PROBE_ENV:-} PAIR_ONLY=1 timeout 300 python tools/roi_pair_probe.py 2>&1 | grep "pair \|differ\|rror" | tail -1
             echo "-- no workspace (tiles)"; env ${PROBE_ENV:-} PAIR_ONLY=1 PAIR_NO_WS=1 timeout 300 python tools/roi_pair_probe.py 2>&1 | grep "pair \|differ\|rror" | tail -1
           done 2>&1 | tee $OUT/probe.txt ;;
    trace) timeout 300 python tools/roi_tiles_trace.py --lib $TUN 2>&1 | grep -v amdgpu.ids | tee $OUT/trace.txt ;;
    ablate) for d in 0 8 2; do echo "-- MV3D_RGT_DBG=$d"; MV3D_RGT_DBG=$d MV3D_IDX_DBG=1 PAIR_ONLY=1 PAIR_NO_WS=1 NB=8 ROUNDS=4 timeout 300 python tools/roi_pair_probe.py --lib $TUN 2>&1 | grep "pair \|rror" | tail -1; done 2>&1 | tee $OUT/ablate.txt ;;
    bench) timeout 1800 python bench.py ${BENCH_ARGS:-} > $OUT/bench.json 2> $OUT/bench.err; tail -c 600 $OUT/bench.json; tail -2 $OUT/bench.err ;;
    bench_path) for r in 1 2 3; do timeout 600 python bench.py --steps 10 --warmup 2 --no-secondary --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline'].get('avg_launch_us'), d['roofline'].get('in_flight'), [k.get('avg_launch_us') for k in d.get('roofline_kernels', [])], d.get('verified'))"; done 2>&1 | tee $OUT/bench_path.txt ;;
    stats) tools/gpu_profile.sh $TAG/ks --steps 4 --warmup 1 --batches-per-step 64 --no-cpu-baseline --no-secondary > /dev/null 2>&1
           python tools/rocprof_summary.py $OUT/ks/r_results.db > $OUT/kernel_stats.txt 2>&1; rm -rf $OUT/ks; head -12 $OUT/kernel_stats.txt ;;
    overlap) tools/gpu_profile.sh $TAG/ov --steps 3 --warmup 1 --batches-per-step 128 --no-cpu-baseline --no-secondary > /dev/null 2>&1
           python tools/rocprof_overlap.py $OUT/ov/r_results.db 2>&1 | tee $OUT/overlap.txt; rm -rf $OUT/ov ;;
    pmc) rm -f $OUT/pmc_traffic.json $OUT/pmc_traffic.txt
         tools/gpu_pmc.sh $TAG/pmc_train > /dev/null 2>&1; python tools/pmc_summary.py $OUT/pmc_train $OUT/pmc_traffic train/b2/r256/peaky/pair-tiles-planned | head -8; rm -rf $OUT/pmc_train
         [ -n "${PMC_TRAIN_ONLY:-}" ] || { tools/gpu_pmc.sh $TAG/pmc_test --workload test > /dev/null 2>&1; python tools/pmc_summary.py $OUT/pmc_test $OUT/pmc_traffic test/b16/r4800/peaky/top-only | head -8; rm -rf $OUT/pmc_test; } ;;
    *) echo "unknown recipe $recipe" ;;
  esac
done
