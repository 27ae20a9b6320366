#!/bin/bash
# GPU box, round 3: GPU tests of the bench CLI / train entry, the default bench line, kernel trace, PMC passes.
set -u
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03a; mkdir -p $OUT
python -m mv3d_tf_amd.build --force > /dev/null 2>&1
timeout 1500 python -m pytest tests/test_bench_cli.py tests/test_train_entry.py -m gpu -x -q 2>&1 | tail -8 | tee $OUT/tests.log
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -c 3000 $OUT/bench_default.json
tools/gpu_profile.sh r03a/ks --steps 4 --warmup 1 --batches-per-step 64 --no-cpu-baseline --no-secondary
python tools/rocprof_summary.py $OUT/ks/r_results.db > $OUT/kernel_stats.txt 2>&1; head -30 $OUT/kernel_stats.txt
tools/gpu_pmc.sh r03a/pmc_train
tools/gpu_pmc.sh r03a/pmc_test --workload test --batches-per-step 8
