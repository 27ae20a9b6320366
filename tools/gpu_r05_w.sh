#!/bin/bash
# GPU box, round 5, final code: the whole -m gpu suite, the default bench line, kernel stats of the default workload, PMC traffic
set -u
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05w; mkdir -p $OUT
timeout 2400 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; tail -3 $OUT/pytest_gpu.log
timeout 1500 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -c 300 $OUT/bench_default.json; tail -2 $OUT/bench_default.err
tools/gpu_profile.sh r05w/ks --steps 4 --warmup 1 --batches-per-step 64 --no-cpu-baseline --no-secondary > /dev/null 2>&1
python tools/rocprof_summary.py $OUT/ks/r_results.db > $OUT/kernel_stats.txt 2>&1; rm -rf $OUT/ks; head -8 $OUT/kernel_stats.txt
tools/gpu_pmc.sh r05w/pmc_train > /dev/null 2>&1
python tools/pmc_summary.py $OUT/pmc_train $OUT/pmc_traffic train/b2/r256/peaky | head -12
rm -rf $OUT/pmc_train
