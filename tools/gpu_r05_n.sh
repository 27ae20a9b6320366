#!/bin/bash
# GPU box, round 5: the whole -m gpu suite, the default bench line (with secondary), PMC traffic of the default workload
set -u
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05n; mkdir -p $OUT
timeout 2400 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; tail -4 $OUT/pytest_gpu.log
timeout 1500 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -c 600 $OUT/bench_default.json; tail -2 $OUT/bench_default.err
tools/gpu_pmc.sh r05n/pmc_train > /dev/null 2>&1
python tools/pmc_summary.py $OUT/pmc_train $OUT/pmc_traffic train/b2/r256/peaky | head -12
rm -rf $OUT/pmc_train
