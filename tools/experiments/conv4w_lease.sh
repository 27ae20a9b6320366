cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06r
L=build_variants/libmv3d_conv4w.so
for r in 1 2; do
echo "== pp (shipped)"; timeout 300 python tools/conv_variant_check.py --lib $L 2>&1 | grep -v amdgpu.ids
echo "== 4 waves 128x128 plain loop"; MV3D_CONV_BIG_MIN=100000000 MV3D_CONV_TILE=2563 timeout 300 python tools/conv_variant_check.py --lib $L 2>&1 | grep -v amdgpu.ids
echo "== 4 waves 128x128 pipelined"; MV3D_CONV_BIG_MIN=100000000 MV3D_CONV_TILE=2564 timeout 300 python tools/conv_variant_check.py --lib $L 2>&1 | grep -v amdgpu.ids
done > gpurun_out/r06r/conv4w.txt 2>&1
cat gpurun_out/r06r/conv4w.txt
