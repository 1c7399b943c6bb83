#!/bin/bash
# round 5, GPU session 19: the student's RPN head + proposals ahead of the pseudo labels -- bitwise test, A/B
export TMPDIR=/tmp
O=gpurun_out/r5s19; mkdir -p $O
python -m pytest tests/test_gpu_model.py tests/test_gpu_trajectory.py -m gpu -q -x -k "ahead_of or in_flight or three_steps or reproducible" > $O/tests.txt 2>&1; echo "rc=$?" >> $O/tests.txt
B="python bench.py --no-cpu-baseline --no-secondary --no-kernel-timer"
for i in 1 2 3; do
  $B --batch 1 --steps 300 > $O/b1_new_$i.json 2>/dev/null
  SFOD_NO_PREFETCH_RPN=1 $B --batch 1 --steps 300 > $O/b1_old_$i.json 2>/dev/null
done
for i in 1 2; do
  $B --steps 60 > $O/vgg_new_$i.json 2>/dev/null
  SFOD_NO_PREFETCH_RPN=1 $B --steps 60 > $O/vgg_old_$i.json 2>/dev/null
  $B --model r101 --steps 30 > $O/r101_new_$i.json 2>/dev/null
  SFOD_NO_PREFETCH_RPN=1 $B --model r101 --steps 30 > $O/r101_old_$i.json 2>/dev/null
done
tail -4 $O/tests.txt
for f in $O/*.json; do python - $f <<'PY'
import json,sys
t=open(sys.argv[1]).read().strip()
if not t: print(sys.argv[1], "EMPTY"); sys.exit()
d=json.loads(t.splitlines()[-1]); print(sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'])
PY
done
