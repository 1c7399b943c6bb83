#!/bin/bash
# round 5, GPU session 23: the step on the trainer's own high-priority stream -- tests, then A/B (SFOD.STEP_STREAM_PRIORITY 0 = before)
export TMPDIR=/tmp
O=gpurun_out/r5s23; mkdir -p $O
python -m pytest tests/test_gpu_model.py tests/test_gpu_trajectory.py tests/test_gpu_two_rank.py tests/test_gpu_resnet.py -m gpu -q -x > $O/tests.txt 2>&1; echo "rc=$?" >> $O/tests.txt
B="python bench.py --no-cpu-baseline --no-secondary --no-kernel-timer"
for i in 1 2 3; do
  $B --steps 60 > $O/vgg_new_$i.json 2>/dev/null
  $B --steps 60 --opts SFOD.STEP_STREAM_PRIORITY 0 > $O/vgg_old_$i.json 2>/dev/null
done
for i in 1 2; do
  $B --model r101 --steps 30 > $O/r101_new_$i.json 2>/dev/null
  $B --model r101 --steps 30 --opts SFOD.STEP_STREAM_PRIORITY 0 > $O/r101_old_$i.json 2>/dev/null
  $B --batch 1 --steps 300 > $O/b1_new_$i.json 2>/dev/null
  $B --batch 1 --steps 300 --opts SFOD.STEP_STREAM_PRIORITY 0 > $O/b1_old_$i.json 2>/dev/null
  $B --trainer base --steps 60 > $O/base_new_$i.json 2>/dev/null
  $B --trainer base --steps 60 --opts SFOD.STEP_STREAM_PRIORITY 0 > $O/base_old_$i.json 2>/dev/null
done
tail -4 $O/tests.txt
for f in $O/*.json; do python - $f <<'PY'
import json,sys
t=open(sys.argv[1]).read().strip()
if not t: print(sys.argv[1], "EMPTY"); sys.exit()
d=json.loads(t.splitlines()[-1]); print(sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'], d['config'].get('peak_hbm_reserved_GB'))
PY
done
