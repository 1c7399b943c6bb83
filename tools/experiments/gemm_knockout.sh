# Time floors of the generic implicit-GEMM kernel on the ResNet-101-C4 1x1 shapes (round 6): variant libraries with one
# part of k_conv_fwd knocked out (csrc/gemm_conv.hip GEMM_KO_*: results are wrong by construction).
#   bash tools/experiments/gemm_knockout.sh build               (no GPU needed)
#   bash tools/experiments/gemm_knockout.sh run > gpurun_out/x.txt      (GPU box)
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
KO=$ROOT/tools/experiments/_ko
OBJ=$ROOT/simple-sfod_amd/lib/obj
VARS=${GEMM_VARS:-"FULL KO_MFMA KO_STORE KO_STATS KO_STORE+KO_STATS KO_MFMA+KO_STORE+KO_STATS"}
if [ "$1" = build ]; then
  mkdir -p $KO
  for v in $VARS; do
    defs=""
    for part in $(echo $v | tr '+' ' '); do [ "$part" != FULL ] && defs="$defs -DGEMM_$part"; done
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-value $defs -c $ROOT/simple-sfod_amd/csrc/gemm_conv.hip -o $KO/gemm_$v.o || exit 1
    objs=$(ls $OBJ/*.o | grep -v gemm_conv.o)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $KO/libsfod_gemm_$v.so $objs $KO/gemm_$v.o || exit 1
    echo built $v
  done
  exit 0
fi
for v in $VARS; do
  echo "== $v"
  SFOD_HIP_LIB=$KO/libsfod_gemm_$v.so python3 $ROOT/tools/experiments/gemm_knockout_time.py ${2:-f16x3}
done
