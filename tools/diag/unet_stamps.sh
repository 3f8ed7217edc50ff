# per-phase stamps of the U-Net stage kernels (diagnostic builds tools/diag/libralenet_stamp{A,B,C}.so, see stamp_unet_bwd.py)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r4
{
echo "Per-phase cycles (s_memtime, ~2.4 GHz... see DESIGN 3 'Round 4') of ONE workgroup of a U-Net stage kernel, batch 2048 x 2 x 512."
for s in "A:k_unet_*_t<16, 8, 4, 2> (ConvTranspose1d 16 -> 8), workgroup 0" "B:k_unet_*_t<32, 32, 3, 1> (bottleneck k3), workgroup 0" "C:k_unet_*_t<32, 32, 3, 1>, workgroup 511 (the last one)"; do
  echo; echo "== ${s#*:}"
  STAMP=${s%%:*} python3 tools/diag/stamp_unet_bwd.py 2>&1 | grep -v amdgpu.ids | tail -14
done
} > gpurun_out/r4/unet_stage_stamps.txt
cat gpurun_out/r4/unet_stage_stamps.txt
