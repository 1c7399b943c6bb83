# Round 6, review item 5: what do the two named suspects of k_conv3x3_m16 cost?
#  (a) LDS bank conflicts: V_NOCONF (fragment rows forced conflict-free) against FULL -- time, and SQ_LDS_BANK_CONFLICT of both
#  (b) the non-MFMA VALU skeleton (~66 instructions per 48 MFMAs): V_EXTRAVALU=n adds n per stage; the slope is their price
#   bash tools/experiments/m16_round6.sh <outdir>     (GPU box; build the variants first: M16_VARS="..." m16_knockout.sh build)
export TMPDIR=/tmp
OUT=$1; mkdir -p $OUT
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
KO=$ROOT/tools/experiments/_ko
for rep in 1 2; do
  M16_VARS="FULL V_EXTRAVALU=16 V_EXTRAVALU=32 V_EXTRAVALU=64 V_NOCONF" bash $ROOT/tools/experiments/m16_knockout.sh run randn >> $OUT/time.txt 2>/dev/null
done
for v in FULL V_NOCONF; do
  rm -rf $OUT/pmc_$v
  SFOD_HIP_LIB=$KO/libsfod_m16_$v.so rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA -d $OUT/pmc_$v -o p --output-format csv -- python3 $ROOT/tools/experiments/m16_knockout_time.py randn > /dev/null 2> $OUT/pmc_$v.err
  python3 - $OUT/pmc_$v $v >> $OUT/counters.txt <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "m16" not in r["Kernel_Name"]: continue
        key = (int(r["Grid_Size"]) if "Grid_Size" in r else 0)
        agg[key][r["Counter_Name"]] += float(r["Counter_Value"]); n[(key, r["Counter_Name"])] += 1
print("==", sys.argv[2])
for key in sorted(agg):
    c = {k: v / max(n[(key, k)], 1) for k, v in agg[key].items()}
    print(f"grid {key}: LDS instrs {c.get('SQ_INSTS_LDS',0):.3e}  IDX_ACTIVE {c.get('SQ_LDS_IDX_ACTIVE',0):.3e}  BANK_CONFLICT {c.get('SQ_LDS_BANK_CONFLICT',0):.3e}"
          f"  share {c.get('SQ_LDS_BANK_CONFLICT',0) / max(c.get('SQ_LDS_IDX_ACTIVE',1),1):.3f}  VALU/MFMA {(c.get('SQ_INSTS_VALU',0) - c.get('SQ_INSTS_MFMA',0)) / max(c.get('SQ_INSTS_MFMA',1),1):.2f}")
PY
  rm -rf $OUT/pmc_$v
done
