#!/usr/bin/env python3
"""One-GPU full MV3D_train steps incl. the torch (MIOpen / rocBLAS) VGG16 trunks -- what bench.py --with-trunk times -- as a
stand-alone script for rocprofv3 (kernel trace / MFMA counters of the dense layers; reporting only, no hand-written conv)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mv3d_tf_amd import build
from mv3d_tf_amd.fast_rcnn import train_mv
from mv3d_tf_amd.fast_rcnn.config import apply_end2end_yml

build.build()
apply_end2end_yml()
print(json.dumps(train_mv.bench_train_step(0, 1, None, steps=int(os.environ.get("STEPS", "3")), warmup=1)))
