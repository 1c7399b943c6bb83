#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r5s32; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-secondary --no-kernel-timer"
for i in 1 2; do
  $B --steps 60 > $O/vgg_q4_$i.json 2>/dev/null
  GPU_MAX_HW_QUEUES=8 $B --steps 60 > $O/vgg_q8_$i.json 2>/dev/null
  $B --model r101 --steps 30 > $O/r101_q4_$i.json 2>/dev/null
  GPU_MAX_HW_QUEUES=8 $B --model r101 --steps 30 > $O/r101_q8_$i.json 2>/dev/null
  $B --batch 1 --steps 300 > $O/b1_q4_$i.json 2>/dev/null
  GPU_MAX_HW_QUEUES=8 $B --batch 1 --steps 300 > $O/b1_q8_$i.json 2>/dev/null
done
sleep 5
$B --res full --steps 30 > $O/full_q4.json 2>/dev/null; sleep 12
GPU_MAX_HW_QUEUES=8 $B --res full --steps 30 > $O/full_q8.json 2>/dev/null; sleep 12
GPU_MAX_HW_QUEUES=16 $B --res full --steps 30 > $O/full_q16.json 2>/dev/null; sleep 12
GPU_MAX_HW_QUEUES=16 $B --steps 60 > $O/vgg_q16_1.json 2>/dev/null
for f in $O/*.json; do python - $f <<'PY'
import json,sys
t=open(sys.argv[1]).read().strip()
if not t: print(sys.argv[1], "EMPTY"); sys.exit()
d=json.loads(t.splitlines()[-1]); print(sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'])
PY
done
