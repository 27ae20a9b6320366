"""One serving configuration of bench_serve_step on its own (for rocprofv3):  python tools/serve_probe.py fp16_mfma [steps]"""
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

name = sys.argv[1] if len(sys.argv) > 1 else "fp16_mfma"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
print(json.dumps(bench_serve_step(0, 1, None, steps=steps, warmup=2, dtypes=(name,))))
