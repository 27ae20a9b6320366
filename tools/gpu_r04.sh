#!/bin/bash
# GPU box, round 4: the default bench line, kernel trace of the headline path, PMC passes, steady-state traces of the training steps.
set -u
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r04; mkdir -p $OUT
timeout 1200 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -c 1500 $OUT/bench_default.json
tools/gpu_profile.sh r04/ks --steps 4 --warmup 1 --batches-per-step 64 --no-cpu-baseline --no-secondary > /dev/null
python tools/rocprof_summary.py $OUT/ks/r_results.db > $OUT/kernel_stats.txt 2>&1; rm -rf $OUT/ks; head -14 $OUT/kernel_stats.txt
tools/gpu_pmc.sh r04/pmc_train > /dev/null 2>&1
tools/gpu_pmc.sh r04/pmc_test --workload test --batches-per-step 8 > /dev/null 2>&1
python tools/pmc_summary.py $OUT/pmc_train $OUT/pmc_traffic train/b2/r256/peaky > /dev/null
python tools/pmc_summary.py $OUT/pmc_test $OUT/pmc_traffic test/b16/r4800/peaky > /dev/null
rm -rf $OUT/pmc_train $OUT/pmc_test
tools/gpu_train_tail.sh r04 bf16_mfma 6 > /dev/null; tools/gpu_train_tail.sh r04 fp32_mfma 6 > /dev/null
head -3 $OUT/bf16_mfma_tail.txt $OUT/fp32_mfma_tail.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/sv -o r -- python $GRAFT_REPO_ROOT/tools/serve_probe.py fp16_mfma 6 > $GRAFT_REPO_ROOT/$OUT/serve.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocprof_summary.py $OUT/sv/r_results.db > $OUT/serve_kernel_stats.txt 2>&1; rm -rf $OUT/sv; head -8 $OUT/serve_kernel_stats.txt
