#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in e6 e7; do echo "== variant $v"; cp build_variants/libmv3d_hip_$v.so mv3d_tf_amd/libmv3d_hip.so; timeout 300 python tools/conv_input_check.py 2>&1 | grep -v amdgpu.ids | grep -A4 "^view (2, 608" | grep "^view\|channel hist\|pixel-in" | cut -c1-330; done
