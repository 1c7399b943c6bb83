#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r5s26; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-secondary --no-kernel-timer"
for i in 1 2 3; do
  $B --steps 60 > $O/vgg_cur_$i.json 2>/dev/null
  SFOD_WGRAD_PRIO=-1 $B --steps 60 > $O/vgg_allhi_$i.json 2>/dev/null
done
for i in 1 2; do
  $B --model r101 --steps 30 > $O/r101_cur_$i.json 2>/dev/null
  SFOD_WGRAD_PRIO=-1 $B --model r101 --steps 30 > $O/r101_allhi_$i.json 2>/dev/null
  $B --batch 1 --steps 300 > $O/b1_cur_$i.json 2>/dev/null
  SFOD_WGRAD_PRIO=-1 $B --batch 1 --steps 300 > $O/b1_allhi_$i.json 2>/dev/null
done
for f in $O/*.json; do python - $f <<'PY'
import json,sys
t=open(sys.argv[1]).read().strip()
if not t: print(sys.argv[1], "EMPTY"); sys.exit()
d=json.loads(t.splitlines()[-1]); print(sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'])
PY
done
