#!/bin/bash
# round 5, GPU session 16: stream priorities of the teacher stream / the weight-gradient side stream (throwaway A/B)
export TMPDIR=/tmp
O=gpurun_out/r5s16; mkdir -p $O
for i in 1 2; do
 for cfg in "-1 0" "0 0" "-1 -1" "0 -1"; do
  set -- $cfg
  tag="t${1}_w${2}"
  SFOD_TEACHER_PRIO=$1 SFOD_WGRAD_PRIO=$2 python bench.py --no-cpu-baseline --no-secondary --no-kernel-timer --steps 60 > $O/vgg_${tag}_$i.json 2>/dev/null
  SFOD_TEACHER_PRIO=$1 SFOD_WGRAD_PRIO=$2 python bench.py --no-cpu-baseline --no-secondary --no-kernel-timer --batch 1 --steps 300 > $O/b1_${tag}_$i.json 2>/dev/null
  SFOD_TEACHER_PRIO=$1 SFOD_WGRAD_PRIO=$2 python bench.py --no-cpu-baseline --no-secondary --no-kernel-timer --model r101 --steps 30 > $O/r101_${tag}_$i.json 2>/dev/null
 done
done
for f in $O/*.json; do python - $f <<'PY'
import json,sys
t=open(sys.argv[1]).read().strip()
if not t: print(sys.argv[1], "EMPTY"); sys.exit()
d=json.loads(t.splitlines()[-1]); print(sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'])
PY
done
