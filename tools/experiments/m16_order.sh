# Does the ISSUE ORDER of the three products of a split-precision MFMA group matter under the power cap?  k_conv3x3_m16 issues, per
# pixel tile, four (w_hi, x_lo) MFMAs, four (w_lo, x_hi), four (w_hi, x_hi) on four independent accumulators.  Variants (a scratch copy
# of conv3x3_patch.hip with -DM16_V_ORDER=n under tools/experiments/_ko/src, built by hand -- see profiles/r6_m16_issue_order.txt):
#   0 the product order   1 per accumulator back to back (hl, hh, lh)   2 groups (wh,cl) (wh,ch) (wl,ch)   3 two accumulators interleaved
#   bash tools/experiments/m16_order.sh <out.txt>        (GPU box)
export TMPDIR=/tmp
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
KO=$ROOT/tools/experiments/_ko
for rep in 1 2 3; do
  for o in 0 1 2 3; do
    echo "== ORDER$o"
    SFOD_HIP_LIB=$KO/libsfod_m16_ORDER$o.so python3 $ROOT/tools/experiments/m16_knockout_time.py randn 5
  done
done > $1 2>/dev/null
