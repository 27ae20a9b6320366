cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06s
for r in 1 2; do
for a in 0 1 2 3 4 5; do
echo "== WG_ABL=$a (0 shipped, 1 no B-fragment reads, 2 no A-fragment reads, 3 no operand reads, 4 a quarter of the MFMAs, 5 no DMA)"
timeout 300 python tools/wgrad_probe.py --lib build_variants/libmv3d_wgabl$a.so 2>&1 | grep -v amdgpu.ids
done; done > gpurun_out/r06s/wgabl.txt 2>&1
cat gpurun_out/r06s/wgabl.txt
