#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p tools/bin
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 tools/permlane_probe.hip -o tools/bin/permlane_probe && tools/bin/permlane_probe
python -m mv3d_tf_amd.build --force > /dev/null 2>&1
timeout 300 python tools/rgt_debug.py 2>&1 | tail -30
