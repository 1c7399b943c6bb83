# Does the ISSUE ORDER of the three products of a split-precision MFMA group matter under the power cap?  k_conv3x3_m16 issues, per
# pixel tile, four (w_hi, x_lo) MFMAs, four (w_lo, x_hi), four (w_hi, x_hi) on four independent accumulators.  Variants (a scratch copy
# of conv3x3_patch.hip with -DM16_V_ORDER=n under tools/experiments/_ko/src, written and built by `build` below; results: profiles/r6_m16_issue_order.txt):
#   0 the product order   1 per accumulator back to back (hl, hh, lh)   2 groups (wh,cl) (wh,ch) (wl,ch)   3 two accumulators interleaved
#   bash tools/experiments/m16_order.sh build             (no GPU needed: writes the scratch copy and builds the four libraries)
#   bash tools/experiments/m16_order.sh <out.txt>        (GPU box)
export TMPDIR=/tmp
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
KO=$ROOT/tools/experiments/_ko
if [ "$1" = build ]; then
  mkdir -p $KO/src
  python3 - "$ROOT" <<'PY' || exit 1
import sys
root = sys.argv[1]
s = open(root + "/simple-sfod_amd/csrc/conv3x3_patch.hip").read()
grp = lambda a, b: f"#pragma unroll\n      for (int ic = 0; ic < 4; ++ic) acc[ic][ip] = mfma16<FMT>({a}[ic], {b}, acc[ic][ip]);\n"
old = grp("wh", "cl") + grp("wl", "ch") + grp("wh", "ch") + "#endif"
assert s.count(old) == 1, "the MFMA groups of k_conv3x3_m16 moved: update this script"
one = lambda a, b, i="ic": f"acc[{i}][ip] = mfma16<FMT>({a}[{i}], {b}, acc[{i}][ip]);"
new = ("#if M16_V_ORDER == 1\n#pragma unroll\n      for (int ic = 0; ic < 4; ++ic) { " + one("wh", "cl") + " " + one("wh", "ch") + " " + one("wl", "ch") + " }\n"
       "#elif M16_V_ORDER == 2\n" + grp("wh", "cl") + grp("wh", "ch") + grp("wl", "ch") +
       "#elif M16_V_ORDER == 3\n#pragma unroll\n      for (int ic = 0; ic < 4; ic += 2) { " +
       " ".join(one(a, b, i) for a, b in (("wh", "cl"), ("wh", "ch"), ("wl", "ch")) for i in ("ic", "ic + 1")) + " }\n"
       "#else\n" + grp("wh", "cl") + grp("wl", "ch") + grp("wh", "ch") + "#endif\n#endif")
s = s.replace(old, new, 1).replace('#include "conv_internal.h"', '#include "../../../../simple-sfod_amd/csrc/conv_internal.h"', 1)
open(root + "/tools/experiments/_ko/src/conv3x3_patch_order.hip", "w").write(s)
PY
  OBJ=$ROOT/simple-sfod_amd/lib/obj
  for o in 0 1 2 3; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-value -DM16_V_ORDER=$o -I$ROOT/simple-sfod_amd/csrc -c $KO/src/conv3x3_patch_order.hip -o $KO/m16_ORDER$o.o || exit 1
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $KO/libsfod_m16_ORDER$o.so $(ls $OBJ/*.o | grep -v conv3x3_patch.o) $KO/m16_ORDER$o.o || exit 1
    echo built ORDER$o
  done
  exit 0
fi
for rep in 1 2 3; do
  for o in 0 1 2 3; do
    echo "== ORDER$o"
    SFOD_HIP_LIB=$KO/libsfod_m16_ORDER$o.so python3 $ROOT/tools/experiments/m16_knockout_time.py randn 5
  done
done > $1 2>/dev/null
