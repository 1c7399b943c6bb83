#!/bin/bash
# round 5, GPU session 29: the teacher's box-head packing on a third stream (tests + A/B); --res full with / without the off-chain work
export TMPDIR=/tmp
O=gpurun_out/r5s29; mkdir -p $O
python -m pytest tests/test_gpu_model.py tests/test_gpu_trajectory.py -m gpu -q -x > $O/tests.txt 2>&1; echo "rc=$?" >> $O/tests.txt
B="python bench.py --no-cpu-baseline --no-secondary --no-kernel-timer"
for i in 1 2 3; do
  $B --steps 60 > $O/vgg_new_$i.json 2>/dev/null
  SFOD_NO_TEACHER_PACK=1 $B --steps 60 > $O/vgg_old_$i.json 2>/dev/null
  $B --batch 1 --steps 300 > $O/b1_new_$i.json 2>/dev/null
  SFOD_NO_TEACHER_PACK=1 $B --batch 1 --steps 300 > $O/b1_old_$i.json 2>/dev/null
done
for i in 1 2; do
  $B --model r101 --steps 30 > $O/r101_new_$i.json 2>/dev/null
  SFOD_NO_TEACHER_PACK=1 $B --model r101 --steps 30 > $O/r101_old_$i.json 2>/dev/null
done
sleep 5
$B --res full --steps 30 > $O/full_all.json 2>/dev/null; sleep 15
SFOD_HEAD_WGRAD_STREAM=0 $B --res full --steps 30 > $O/full_nooffchain.json 2>/dev/null; sleep 15
$B --res full --steps 30 --opts SFOD.STEP_STREAM_PRIORITY 0 > $O/full_nostepstream.json 2>/dev/null; sleep 15
SFOD_NO_PREFETCH_RPN=1 SFOD_HEAD_WGRAD_STREAM=0 $B --res full --steps 30 --opts SFOD.STEP_STREAM_PRIORITY 0 > $O/full_none.json 2>/dev/null
tail -3 $O/tests.txt
for f in $O/*.json; do python - $f <<'PY'
import json,sys
t=open(sys.argv[1]).read().strip()
if not t: print(sys.argv[1], "EMPTY"); sys.exit()
d=json.loads(t.splitlines()[-1]); print(sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'], d['config'].get('peak_hbm_reserved_GB'))
PY
done
