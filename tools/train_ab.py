#!/usr/bin/env python3
"""A / B of the bf16 training step (bench_train_step, 2 frames, 3 views, MFMA trunks): the fusion head as one autograd function vs op by
op, the optimizer step as one launch of mv3d_adam_step vs torch's fused Adam.  Alternating runs, >= 1 s windows, min / median / mean."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from mv3d_tf_amd.fast_rcnn import train_mv

for rnd in range(int(os.environ.get("ROUNDS", "2"))):
    for fh, ka in ((True, True), (False, True), (True, False), (False, False)):
        r = train_mv.bench_train_step(0, 1, None, amp=torch.bfloat16, mfma=True, fused_head=fh, kernel_adam=ka)
        print("fused_head=%d kernel_adam=%d  ms/step mean %.3f min %.3f median %.3f (%d steps)" % (fh, ka, r["ms_per_step"], r["ms_per_step_min"], r["ms_per_step_median"], r["steps_timed"]), flush=True)
        torch.cuda.empty_cache()
