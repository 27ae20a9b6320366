#!/bin/bash
# GPU box, round 5, call C: the indexed RoiPool pair with compact argmax codes and the single-pass in-forward index
set -u
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05m; mkdir -p $OUT
timeout 900 python -m pytest tests/test_roi_pair.py -x -q -m gpu > $OUT/pytest_roi.log 2>&1; tail -15 $OUT/pytest_roi.log
timeout 300 python tools/roi_pair_probe.py > $OUT/pair_probe.txt 2>&1; tail -6 $OUT/pair_probe.txt
timeout 900 python -m pytest tests/test_bench_cli.py::test_path_driver_depth8_equals_the_oracle tests/test_train_stream.py tests/test_gpu_configs.py tests/test_roipool_pin.py -x -q -m gpu > $OUT/pytest_b.log 2>&1; tail -5 $OUT/pytest_b.log
timeout 600 python bench.py --no-secondary --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r05m/bench.json') if l.startswith('{')][0])
print('value', d['value'], 'ms/step', d['ms_per_step'], 'verified', d['verified']['bit_exact'], d['verified']['mismatches'])
for e in d['roofline_kernels']: print(e['kernel'][:40], e['avg_launch_us'], e['frac'], e.get('in_flight_us'))
PY
tail -3 $OUT/bench.err
tools/gpu_profile.sh r05m/ks --steps 4 --warmup 1 --batches-per-step 64 --no-cpu-baseline --no-secondary > /dev/null
python tools/rocprof_summary.py $OUT/ks/r_results.db > $OUT/kernel_stats.txt 2>&1; rm -rf $OUT/ks; head -8 $OUT/kernel_stats.txt
