# Time floors of the bf16x3 halo-patch convolution (k_conv3x3_patch<.., float, true>): variant libraries with one part of
# the kernel knocked out (csrc/conv3x3_patch.hip, P3_KO_*: results are wrong by construction), timed on the VGG layers.
#   bash tools/experiments/conv_knockout.sh build      (no GPU needed)
#   bash tools/experiments/conv_knockout.sh run > gpurun_out/x.txt      (GPU box)
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
KO=$ROOT/tools/experiments/_ko
OBJ=$ROOT/simple-sfod_amd/lib/obj
VARS="FULL MFMA READS PATCHDMA WDMA EPI MFMA_READS PATCHDMA_WDMA"
if [ "$1" = build ]; then
  mkdir -p $KO
  for v in $VARS; do
    defs=""
    for part in $(echo $v | tr '_' ' '); do [ "$part" != FULL ] && defs="$defs -DP3_KO_$part"; done
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-value $defs -c $ROOT/simple-sfod_amd/csrc/conv3x3_patch.hip -o $KO/p3_$v.o || exit 1
    objs=$(ls $OBJ/*.o | grep -v conv3x3_patch.o)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $KO/libsfod_p3_$v.so $objs $KO/p3_$v.o || exit 1
    echo built $v
  done
  exit 0
fi
for v in $VARS; do
  echo "== $v"
  SFOD_HIP_LIB=$KO/libsfod_p3_$v.so python3 $ROOT/tools/experiments/conv_knockout_time.py
done
