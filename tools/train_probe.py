"""One training-step configuration of bench_train_step on its own (for rocprofv3):  python tools/train_probe.py [fp32|bf16|bf16_mfma|fp32_mfma] [steps]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if "--lib" in sys.argv:                      # an experiment build of the library (tools only)
    from mv3d_tf_amd import _lib
    _i = sys.argv.index("--lib")
    _lib.LIB_PATH = os.path.abspath(sys.argv[_i + 1])
    del sys.argv[_i:_i + 2]
from mv3d_tf_amd.fast_rcnn.train_mv import bench_train_step  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "bf16_mfma"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
cm = os.environ.get("MV3D_CAST_MANY", "1") == "1"      # (A / B of the multi-tensor parameter casts)
r = bench_train_step(0, 1, None, steps=steps, warmup=2, cast_many=cm, amp=None if name.startswith("fp32") else torch.bfloat16, mfma=name.endswith("_mfma"))
print(json.dumps({"config": name, "ms_per_step": r["ms_per_step"], "frames_per_s": r["frames_per_s"]}))
