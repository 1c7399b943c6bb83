# Time floors / schedule variants of the 16x16x32 halo-patch convolution (k_conv3x3_m16): variant libraries with one part
# of the kernel knocked out (csrc/conv3x3_patch.hip, M16_KO_*: results are wrong by construction) or re-scheduled (M16_V_*).
#   bash tools/experiments/m16_knockout.sh build               (no GPU needed)
#   bash tools/experiments/m16_knockout.sh run [randn|zeros] > gpurun_out/x.txt      (GPU box)
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
KO=$ROOT/tools/experiments/_ko
OBJ=$ROOT/simple-sfod_amd/lib/obj
VARS=${M16_VARS:-"FULL KO_MFMA KO_WDMA KO_PDMA KO_WDMA+KO_PDMA KO_BAR KO_EPI V_LATEW"}
if [ "$1" = build ]; then
  mkdir -p $KO
  for v in $VARS; do
    defs=""
    for part in $(echo $v | tr '+' ' '); do [ "$part" != FULL ] && defs="$defs -DM16_$part"; done
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-value $defs -c $ROOT/simple-sfod_amd/csrc/conv3x3_patch.hip -o $KO/m16_$v.o || exit 1
    objs=$(ls $OBJ/*.o | grep -v conv3x3_patch.o)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $KO/libsfod_m16_$v.so $objs $KO/m16_$v.o || exit 1
    echo built $v
  done
  exit 0
fi
for v in $VARS; do
  echo "== $v"
  SFOD_HIP_LIB=$KO/libsfod_m16_$v.so python3 $ROOT/tools/experiments/m16_knockout_time.py ${2:-randn} ${3:-5}
done
