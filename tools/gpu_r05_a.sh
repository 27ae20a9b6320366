#!/bin/bash
# GPU box, round 5, call A: the indexed RoiPool pair -- parity tests, A / B probe, a short bench with the kernel trace.
set -u
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05a; mkdir -p $OUT
timeout 900 python -m pytest tests/test_roi_indexed.py tests/test_roipool_pin.py -x -q -m gpu > $OUT/pytest_roi.log 2>&1; tail -5 $OUT/pytest_roi.log
timeout 300 python tools/roi_pair_probe.py > $OUT/pair_probe.txt 2>&1; tail -8 $OUT/pair_probe.txt
timeout 900 python -m pytest tests/test_bench_cli.py tests/test_train_stream.py tests/test_gpu_configs.py -x -q -m gpu > $OUT/pytest_b.log 2>&1; tail -5 $OUT/pytest_b.log
timeout 600 python bench.py --no-secondary > $OUT/bench.json 2> $OUT/bench.err; tail -c 2500 $OUT/bench.json; tail -3 $OUT/bench.err
tools/gpu_profile.sh r05a/ks --steps 4 --warmup 1 --batches-per-step 64 --no-cpu-baseline --no-secondary > /dev/null
python tools/rocprof_summary.py $OUT/ks/r_results.db > $OUT/kernel_stats.txt 2>&1; rm -rf $OUT/ks; head -16 $OUT/kernel_stats.txt
