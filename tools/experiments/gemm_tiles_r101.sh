# Which workgroup tile serves the ResNet-101-C4 1x1 convolutions / linear layers best in the pair modes?
#   bash tools/experiments/gemm_tiles_r101.sh > gpurun_out/r3_gemm_tiles_r101.txt
# (SFOD_GEMM_TILE forces a tile shape in launch_conv_fwd_ut: 1 128x64, 2 128x128, 3 256x128, 4 256x64, 5 256x256; 0 = the
# library's own choice)
for dt in f16x3 bf16x3; do
  for t in 0 1 2 3 4 5; do
    echo "== dtype $dt tile $t"
    SFOD_GEMM_TILE=$t python3 tools/bench_gemm.py --dtype $dt --only r101 2>&1 | grep -v amdgpu.ids
  done
done
