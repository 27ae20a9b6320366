#!/bin/bash
# GPU box, round 5, final code (input-layer kernel, workspace-free RoiPoolGrad): the whole -m gpu suite, the default bench line, kernel stats of the default workload, PMC traffic
set -u
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05zz; mkdir -p $OUT
timeout 2400 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; tail -3 $OUT/pytest_gpu.log
timeout 1500 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -c 300 $OUT/bench_default.json; tail -2 $OUT/bench_default.err
tools/gpu_profile.sh r05zz/ks --steps 4 --warmup 1 --batches-per-step 64 --no-cpu-baseline --no-secondary > /dev/null 2>&1
python tools/rocprof_summary.py $OUT/ks/r_results.db > $OUT/kernel_stats.txt 2>&1; rm -rf $OUT/ks; head -8 $OUT/kernel_stats.txt
tools/gpu_pmc.sh r05zz/pmc_train > /dev/null 2>&1
python tools/pmc_summary.py $OUT/pmc_train $OUT/pmc_traffic train/b2/r256/peaky | head -12
rm -rf $OUT/pmc_train
# the input layer's kernel under the counters (MFMA busy, wait, LDS conflicts, HBM bytes) next to the general kernel's conv1_2
tools/gpu_r05_pmc.sh r05zz/conv_input python $GRAFT_REPO_ROOT/tools/conv_probe.py 16 --no-torch --only conv1_1,conv1_2 > /dev/null 2>&1
head -30 $OUT/conv_input/table.txt
# the no-workspace RoiPoolGrad: kernel stats of the pair probe
timeout 300 python tools/roi_pair_subset_probe.py 2>&1 | grep -v amdgpu.ids | tail -12 > $OUT/roi_pair_subset_probe.txt; cat $OUT/roi_pair_subset_probe.txt
