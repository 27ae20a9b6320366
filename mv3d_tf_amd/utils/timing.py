"""Timing of a repeated device step for the bench legs: wall time of the whole window (what frames/s is computed from) and, from one
HIP event per step on the current stream, every step's own duration -- so a leg reports its minimum and median next to the mean."""
import time

import numpy as np
import torch


def timed_steps(step, steps, barrier, hook=None):
    """Runs `step()` `steps` times between two `barrier()` calls.  Returns (wall seconds, per-step ms array of length steps)."""
    barrier()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    t0 = time.perf_counter()
    evs[0].record()
    for i in range(steps):
        if hook is not None:
            hook(i, 0)
        step()
        if hook is not None:
            hook(i, 1)
        evs[i + 1].record()
    barrier()
    dt = time.perf_counter() - t0
    ms = np.array([evs[i].elapsed_time(evs[i + 1]) for i in range(steps)])
    return dt, ms


def step_stats(ms):
    """min / median of the per-step device times (ms), rounded for the bench line"""
    return {"steps_timed": int(len(ms)), "ms_per_step_min": round(float(ms.min()), 3), "ms_per_step_median": round(float(np.median(ms)), 3)}
