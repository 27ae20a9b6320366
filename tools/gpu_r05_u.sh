#!/bin/bash
# GPU box, round 5: forward row steps with clamped offsets (no one-pixel tail loop) -- parity, probe, TEST-cfg + headline
set -u
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05u; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_roi_pair.py tests/test_gpu_configs.py tests/test_gpu_parity.py tests/test_train_stream.py -x -q -m gpu > $OUT/pytest_roi.log 2>&1; tail -3 $OUT/pytest_roi.log
for r in 1 2; do timeout 200 python tools/roi_pair_probe.py 2>&1 | grep "pair \|plain \|identical"; done | tee $OUT/roi_pair_probe.txt
timeout 900 python bench.py --no-trunk --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; python - <<'PY'
import json
d=json.load(open("gpurun_out/r05u/bench.json"))
print("value", d["value"], "verified", d.get("verified",{}).get("bit_exact"), "test_cfg", d["secondary"]["test_cfg"]["frames_per_s"])
for e in d["roofline_kernels"]: print(e["kernel"][:40], e["avg_launch_us"], e["frac"], e.get("in_flight_us"))
for e in d["secondary"]["test_cfg"]["roofline_kernels"]: print(e["kernel"][:40], e["avg_launch_us"], e["frac"])
PY
tail -2 $OUT/bench.err
