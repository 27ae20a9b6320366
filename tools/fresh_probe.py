"""Sweep of the fresh-input line (bench.fresh_inputs_line) over batches in flight / streams:  python tools/fresh_probe.py [d,s ...]"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402

if __name__ == "__main__":
    combos = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]] or [(2, 0), (2, 2), (3, 3), (4, 4)]
    for depth, ns in combos:
        r = bench.fresh_inputs_line(0, "peaky", seconds=1.0, depth=depth, nstreams=ns)
        print(json.dumps({"depth": depth, "streams": ns, "frames_per_s": r["frames_per_s"], "draw_ms_per_frame": r["host_draw_ms_per_frame"],
                          "wait_ms_per_frame": r["host_wait_for_device_ms_per_frame"]}), flush=True)
