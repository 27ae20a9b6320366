#!/bin/bash
# GPU box: steady-state kernel totals of one training-step configuration (rocprofv3 kernel trace of tools/train_probe.py, the last
# 40 % of the trace = the timed steps).   tools/gpu_train_tail.sh <tag> <fp32|bf16|bf16_mfma|fp32_mfma> [steps]
set -u
TAG=$1; CFG=$2; STEPS=${3:-8}; SFX=${4:-}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace -d $OUT/tr -o r -- python $GRAFT_REPO_ROOT/tools/train_probe.py $CFG $STEPS > $OUT/$CFG.log 2>&1
tail -1 $OUT/$CFG.log
python - $OUT/tr/r_results.db $STEPS > $OUT/${CFG}${SFX}_tail.txt <<'PY'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1]); steps = float(sys.argv[2])
rows = c.execute("select name, start, end from kernels order by start").fetchall()
# the timed steps = the last `steps` of steps + 2 warm-up: cut at the Adam launch (multi_tensor_apply) that ends warm-up step 2
adam = [r[1] for r in rows if "adam_step_kernel" in r[0] or ("multi_tensor_apply" in r[0] and "Adam" in r[0])]        # (not the multi-tensor casts)
per = len(adam) / (steps + 2)
cut = adam[int(round(2 * per)) - 1] if adam else rows[0][1]
agg = {}
for n, s, e in rows:
    if s > cut:
        a = agg.setdefault(n, [0, 0]); a[0] += 1; a[1] += e - s
tot = sum(v[1] for v in agg.values()); span = rows[-1][2] - cut
print("timed window %.1f ms, kernel time %.1f ms; per step: kernel %.2f ms of %.2f ms wall" % (span / 1e6, tot / 1e6, tot / 1e6 / steps, span / 1e6 / steps))
for n, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:32]:
    print("%-100s n/step %6.1f  ms/step %7.3f  %5.1f%%" % (n[:100], v[0] / steps, v[1] / 1e6 / steps, 100.0 * v[1] / tot))
# where the device idles: gaps between consecutive kernels of the timed window (one stream: the end of one to the start of the next)
tw = [(s, e, n) for n, s, e in rows if s > cut]
gaps = {}
end = tw[0][1]
for (s, e, n), (ps, pe, pn) in zip(tw[1:], tw[:-1]):
    g = s - end
    end = max(end, e)
    if g > 0:
        k = (pn[:60], n[:60])
        a = gaps.setdefault(k, [0, 0]); a[0] += 1; a[1] += g
# ... and by who owns the two launches: the library's kernels (mv3d_* / roi_* / loss_* / at_* / pt_* / nms_*) or torch / rocBLAS
own = lambda n: any(t in n for t in ("mv3d", "roi_", "loss_", "at_", "pt_", "nms_", "proposal", "rgt_", "adam_step"))
cls = {}
for (pn, n), v in gaps.items():
    k = ("library" if own(pn) else "torch") + " -> " + ("library" if own(n) else "torch")
    a = cls.setdefault(k, [0, 0]); a[0] += v[0]; a[1] += v[1]
for k, v in sorted(cls.items()):
    print("gaps %-20s %6.1f / step, %.3f ms / step" % (k, v[0] / steps, v[1] / 1e6 / steps))
print("idle between kernels: %.2f ms per step; largest contributors (previous kernel -> next kernel):" % (sum(v[1] for v in gaps.values()) / 1e6 / steps))
for k, v in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:14]:
    print("  %6.1f us x %5.1f / step   %s  ->  %s" % (v[1] / v[0] / 1e3, v[0] / steps, k[0], k[1]))
PY
rm -rf $OUT/tr
head -24 $OUT/${CFG}${SFX}_tail.txt | cut -c1-170; grep -A15 '^idle between' $OUT/${CFG}${SFX}_tail.txt | cut -c1-200
