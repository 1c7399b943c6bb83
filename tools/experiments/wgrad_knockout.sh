# Time floors of the pipelined bf16x3 weight-gradient loop (k_wgrad3x3_patch<4, true, true>): builds variant libraries with
# one part of the loop knocked out (csrc/wgrad3x3_patch.hip, W3_KO_*: results are wrong by construction) and times three
# layers with each.  Build here (no GPU needed): bash tools/experiments/wgrad_knockout.sh build
# Run on the GPU box:                             bash tools/experiments/wgrad_knockout.sh run > gpurun_out/x.txt
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
KO=$ROOT/tools/experiments/_ko
OBJ=$ROOT/simple-sfod_amd/lib/obj
VARS="FULL MFMA XREADS DMA BARRIER MFMA_XREADS"
if [ "$1" = build ]; then
  mkdir -p $KO
  for v in $VARS; do
    defs=""
    case $v in
      MFMA) defs="-DW3_KO_MFMA";; XREADS) defs="-DW3_KO_XREADS";; DMA) defs="-DW3_KO_DMA";; BARRIER) defs="-DW3_KO_BARRIER";;
      MFMA_XREADS) defs="-DW3_KO_MFMA -DW3_KO_XREADS";;
    esac
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-value $defs -c $ROOT/simple-sfod_amd/csrc/wgrad3x3_patch.hip -o $KO/w3_$v.o || exit 1
    objs=$(ls $OBJ/*.o | grep -v wgrad3x3_patch.o)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $KO/libsfod_$v.so $objs $KO/w3_$v.o || exit 1
    echo built $v
  done
  exit 0
fi
for v in $VARS; do
  echo "== $v"
  SFOD_HIP_LIB=$KO/libsfod_$v.so python3 $ROOT/tools/experiments/wgrad_knockout_time.py
done
