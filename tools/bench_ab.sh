#!/bin/bash
# Same-box alternating A/B of bench.py under different environment settings (what the 21 one-off tools/r5_session*.sh of
# round 5 did by hand; those are in the history at 2886d1a).
#   bash tools/bench_ab.sh <outdir> <reps> "<bench.py args>" "<ENV=...>" ["<ENV2=... ENV3=...>" ...]
# e.g. bash tools/bench_ab.sh gpurun_out/ab 2 "--steps 60" "SFOD_SHARED_SIDE_STREAM=1" "SFOD_SHARED_SIDE_STREAM=0"
#      bash tools/bench_ab.sh gpurun_out/ab 2 "--model r101 --steps 30" "X=1" "SFOD_ROI_NT=0 SFOD_ROI_CBLK=256"
# Every run: bench.py --no-cpu-baseline --no-secondary --no-kernel-timer <args>; the settings alternate inside each
# repetition (A B A B ...), the device is given WAIT seconds (default 3; 12 for --res full) to return its memory in between.
export TMPDIR=/tmp
O=$1; REPS=$2; ARGS=$3; shift 3
mkdir -p $O
WAIT=${WAIT:-3}
case "$ARGS" in *"--res full"*) WAIT=${WAIT_FULL:-12};; esac
for rep in $(seq 1 $REPS); do
  i=0
  for envs in "$@"; do
    ( export $envs; python3 bench.py --no-cpu-baseline --no-secondary --no-kernel-timer --no-smi $ARGS > $O/ab_${i}_$rep.json 2> $O/ab_${i}_$rep.err )
    sleep $WAIT
    i=$((i+1))
  done
done
i=0
for envs in "$@"; do
  python3 - "$envs" $O/ab_${i}_*.json <<'PY'
import json, sys
vals = []
for f in sys.argv[2:]:
    t = open(f).read().strip()
    if t:
        d = json.loads(t.splitlines()[-1]); vals.append((d["value"], d["ms_per_step"]))
print(f"{sys.argv[1]:50s} images/s {[v[0] for v in vals]}  ms/step {[v[1] for v in vals]}")
PY
  i=$((i+1))
done
