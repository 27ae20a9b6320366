#!/bin/bash
# GPU box: what the HOST does while the device idles in a training step -- kernel + HIP runtime + memory-copy trace of tools/train_probe.py
# (no counters), then for every device-idle gap > 100 us of the timed steps the HIP API calls and copies that overlap it.
#   tools/gpu_train_api_gaps.sh <tag> <bf16_mfma|fp32_mfma> [steps]
set -u
TAG=$1; CFG=$2; STEPS=${3:-6}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --hip-runtime-trace --memory-copy-trace -d $OUT/tr -o r -- python $GRAFT_REPO_ROOT/tools/train_probe.py $CFG $STEPS > $OUT/$CFG.log 2>&1
tail -1 $OUT/$CFG.log
python - $OUT/tr/r_results.db $STEPS > $OUT/${CFG}_api_gaps.txt <<'PY'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1]); steps = float(sys.argv[2])
k = c.execute("select name, start, end from kernels order by start").fetchall()
cp = c.execute("select name, start, end, size from memory_copies order by start").fetchall()
adam = [r[1] for r in k if "multi_tensor_apply" in r[0]]
per = len(adam) / (steps + 2)
cut = adam[int(round(2 * per)) - 1] if adam else k[0][1]
dev = sorted([(s, e, n) for n, s, e in k if s > cut] + [(s, e, "COPY %s %d B" % (n, sz)) for n, s, e, sz in cp if s > cut])
api = c.execute("select name, start, end from regions where start > ? order by start", (cut - 10_000_000,)).fetchall()
gaps = []
end = dev[0][1]
prev = dev[0][2]
for s, e, n in dev[1:]:
    if s - end > 100_000:
        gaps.append((s - end, end, s, prev, n))
    if e > end:
        end, prev = e, n
print("device-idle gaps > 100 us (kernels AND copies on the timeline): %d, %.2f ms per step" % (len(gaps), sum(g[0] for g in gaps) / 1e6 / steps))
for g, t0, t1, pn, nn in sorted(gaps, key=lambda t: -t[0])[:16]:
    inside = [(min(e, t1) - max(s, t0), n) for n, s, e in api if s < t1 and e > t0]
    agg = {}
    for d, n in inside:
        a = agg.setdefault(n, [0, 0]); a[0] += 1; a[1] += d
    top = sorted(agg.items(), key=lambda kv: -kv[1][1])[:6]
    print("%7.0f us  %-44s -> %-44s | %s" % (g / 1e3, pn[:44], nn[:44], ", ".join("%s x%d %.0f us" % (n, v[0], v[1] / 1e3) for n, v in top)))
PY
rm -rf $OUT/tr
cat $OUT/${CFG}_api_gaps.txt | cut -c1-400
