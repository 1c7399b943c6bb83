#!/bin/bash
# round 5, GPU session 33: ONE shared side stream for all off-chain work vs a stream per module
export TMPDIR=/tmp
O=gpurun_out/r5s33; mkdir -p $O
python -m pytest tests/test_gpu_model.py tests/test_gpu_trajectory.py tests/test_gpu_resnet.py -m gpu -q -x > $O/tests.txt 2>&1; echo "rc=$?" >> $O/tests.txt
B="python bench.py --no-cpu-baseline --no-secondary --no-kernel-timer"
for i in 1 2; do
  $B --steps 60 > $O/vgg_shared_$i.json 2>/dev/null
  SFOD_SHARED_SIDE_STREAM=0 $B --steps 60 > $O/vgg_permod_$i.json 2>/dev/null
  $B --model r101 --steps 30 > $O/r101_shared_$i.json 2>/dev/null
  SFOD_SHARED_SIDE_STREAM=0 $B --model r101 --steps 30 > $O/r101_permod_$i.json 2>/dev/null
  $B --batch 1 --steps 300 > $O/b1_shared_$i.json 2>/dev/null
  SFOD_SHARED_SIDE_STREAM=0 $B --batch 1 --steps 300 > $O/b1_permod_$i.json 2>/dev/null
done
sleep 5
$B --res full --steps 30 > $O/full_shared.json 2>/dev/null; sleep 12
SFOD_SHARED_SIDE_STREAM=0 $B --res full --steps 30 > $O/full_permod.json 2>/dev/null; sleep 12
SFOD_HEAD_WGRAD_STREAM=0 $B --res full --steps 30 > $O/full_nooffchain.json 2>/dev/null
tail -3 $O/tests.txt
for f in $O/*.json; do python - $f <<'PY'
import json,sys
t=open(sys.argv[1]).read().strip()
if not t: print(sys.argv[1], "EMPTY"); sys.exit()
d=json.loads(t.splitlines()[-1]); print(sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'])
PY
done
