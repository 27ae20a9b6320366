#!/bin/bash
# GPU box, round 5: the pair's one-byte argmax codes -- parity, probe, headline
set -u
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05s; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_roi_pair.py tests/test_gpu_configs.py tests/test_train_stream.py tests/test_bench_cli.py tests/test_abi.py -x -q -m gpu > $OUT/pytest_roi.log 2>&1; tail -3 $OUT/pytest_roi.log
for r in 1 2; do timeout 200 python tools/roi_pair_probe.py 2>&1 | grep "pair \|plain \|identical"; done | tee $OUT/roi_pair_probe.txt
timeout 900 python bench.py --no-secondary > $OUT/bench_nosec.json 2> $OUT/bench_nosec.err; python - <<'PY'
import json
d=json.load(open("gpurun_out/r05s/bench_nosec.json"))
print("value", d["value"], "verified", d.get("verified",{}).get("bit_exact"))
for e in d["roofline_kernels"]: print(e["kernel"][:40], e["avg_launch_us"], e["frac"], e.get("in_flight_us"))
PY
tail -2 $OUT/bench_nosec.err
