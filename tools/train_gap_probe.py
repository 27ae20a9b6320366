"""Which host work the device waits for in a training step: torch.profiler over two steady-state steps of bench_train_step's loop, then
for every device-idle gap > 80 us the host ops that ran inside it.   python tools/train_gap_probe.py [bf16_mfma|fp32_mfma]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mv3d_tf_amd.fast_rcnn import train_mv  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "bf16_mfma"
out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "train_gap_trace.json")
os.makedirs(os.path.dirname(out), exist_ok=True)
state = {"n": 0, "prof": None}
orig_sync = torch.cuda.synchronize


def run():
    from torch.profiler import ProfilerActivity, profile
    # bench_train_step(steps, warmup): profile its last two timed steps by wrapping the step function it calls
    r = train_mv.bench_train_step(0, 1, None, steps=6, warmup=3, amp=None if name.startswith("fp32") else torch.bfloat16,
                                  mfma=name.endswith("_mfma"), step_hook=hook)
    print(json.dumps({"ms_per_step": r["ms_per_step"]}))


def hook(i, phase):
    """called by bench_train_step before (phase 0) / after (phase 1) timed step i"""
    from torch.profiler import ProfilerActivity, profile
    if phase == 0 and i == 3:
        state["prof"] = profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA])
        state["prof"].__enter__()
    if phase == 1 and i == 4:
        orig_sync()
        state["prof"].__exit__(None, None, None)
        state["prof"].export_chrome_trace(out)


run()
ev = json.load(open(out))["traceEvents"]
gpu = sorted([e for e in ev if e.get("cat") in ("kernel", "gpu_memcpy", "gpu_memset") and "dur" in e], key=lambda e: e["ts"])
cpu = [e for e in ev if e.get("cat") in ("cpu_op", "user_annotation", "python_function") and "dur" in e]
gaps = []
for a, b in zip(gpu, gpu[1:]):
    g = b["ts"] - (a["ts"] + a["dur"])
    if g > 80:
        gaps.append((g, a, b))
print("device-idle gaps > 80 us: %d, %.2f ms in two steps" % (len(gaps), sum(g for g, _, _ in gaps) / 1e3))
for g, a, b in sorted(gaps, key=lambda t: -t[0])[:14]:
    t0, t1 = a["ts"] + a["dur"], b["ts"]
    inside = sorted([c for c in cpu if c["ts"] < t1 and c["ts"] + c["dur"] > t0], key=lambda c: -min(c["ts"] + c["dur"], t1) + max(c["ts"], t0))
    names = ["%s(%.0f)" % (c["name"][:40], min(c["ts"] + c["dur"], t1) - max(c["ts"], t0)) for c in inside[:6]]
    print("%7.0f us  %-36s -> %-36s | %s" % (g, a["name"][:36], b["name"][:36], ", ".join(names)))
