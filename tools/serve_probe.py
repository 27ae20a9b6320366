"""One serving configuration of bench_serve_step on its own (for rocprofv3):  python tools/serve_probe.py fp16_mfma[,fp16_mfma_graph] [steps]"""
import json
import sys

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if "--lib" in sys.argv:                      # an experiment build of the library (tools only)
    from mv3d_tf_amd import _lib
    _i = sys.argv.index("--lib")
    _lib.LIB_PATH = os.path.abspath(sys.argv[_i + 1])
    del sys.argv[_i:_i + 2]
from mv3d_tf_amd.fast_rcnn.test_mv import bench_serve_step  # noqa: E402

names = tuple(sys.argv[1].split(",")) if len(sys.argv) > 1 else ("fp16_mfma",)
steps = int(sys.argv[2]) if len(sys.argv) > 2 else None          # None: a >= 1 s window per variant
print(json.dumps(bench_serve_step(0, 1, None, steps=steps, warmup=2, dtypes=names)))
