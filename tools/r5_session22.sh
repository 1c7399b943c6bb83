#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r5s22; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-secondary --no-kernel-timer"
for i in 1 2 3; do
  $B --batch 1 --steps 300 > $O/b1_base_$i.json 2>/dev/null
  SFOD_MAIN_PRIO=-1 $B --batch 1 --steps 300 > $O/b1_mainhi_$i.json 2>/dev/null
done
for i in 1 2; do
  $B --steps 60 > $O/vgg_base_$i.json 2>/dev/null
  SFOD_MAIN_PRIO=-1 $B --steps 60 > $O/vgg_mainhi_$i.json 2>/dev/null
  $B --model r101 --steps 30 > $O/r101_base_$i.json 2>/dev/null
  SFOD_MAIN_PRIO=-1 $B --model r101 --steps 30 > $O/r101_mainhi_$i.json 2>/dev/null
done
for f in $O/*.json; do python - $f <<'PY'
import json,sys
t=open(sys.argv[1]).read().strip()
if not t: print(sys.argv[1], "EMPTY"); sys.exit()
d=json.loads(t.splitlines()[-1]); print(sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'])
PY
done
