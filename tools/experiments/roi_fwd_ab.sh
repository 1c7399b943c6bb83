# ROIAlign forward A/B (round 6): channel-block width x non-temporal output stores, time + HBM read bytes.
# usage (GPU box): bash tools/experiments/roi_fwd_ab.sh <outdir>
export TMPDIR=/tmp
OUT=$1; mkdir -p $OUT
for nt in 0 1; do for cb in 0 256 128 64; do
  SFOD_ROI_CBLK=$cb SFOD_ROI_NT=$nt python3 tools/experiments/time_roi_align.py >> $OUT/time.txt 2>/dev/null
done; done
for cfg in "256 0" "256 1" "128 1" "64 1" "128 0"; do
  set -- $cfg
  export SFOD_ROI_CBLK=$1 SFOD_ROI_NT=$2
  rm -rf $OUT/pmc_f $OUT/pmc_w
  rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_f -o f --output-format csv -- python3 tools/experiments/time_roi_align.py > /dev/null 2> $OUT/pmc_f.err
  rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_w -o w --output-format csv -- python3 tools/experiments/time_roi_align.py > /dev/null 2> $OUT/pmc_w.err
  echo "== SFOD_ROI_CBLK=$1 SFOD_ROI_NT=$2" >> $OUT/pmc.txt
  python3 tools/pmc_summary.py $OUT/pmc_f $OUT/pmc_w | grep roi_align >> $OUT/pmc.txt
done
rm -rf $OUT/pmc_f $OUT/pmc_w
