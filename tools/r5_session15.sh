#!/bin/bash
# round 5, GPU session 15: EXPERIMENT -- CU-partitioned streams (SFOD_CU_PARTITION): correctness under the detours, then A/B
export TMPDIR=/tmp
O=gpurun_out/r5s15; mkdir -p $O
SFOD_CU_PARTITION=32 python -m pytest tests/test_gpu_trajectory.py tests/test_gpu_model.py -m gpu -q -x -k "vgg-bf16x3 or backbone_matches or student_losses" > $O/tests_partition.txt 2>&1; echo "rc=$?" >> $O/tests_partition.txt
B="python bench.py --no-cpu-baseline --no-secondary --steps 60"
for i in 1 2; do
  $B > $O/base_$i.json 2> $O/base_$i.err
  for n in 24 32 48 64; do
    SFOD_CU_PARTITION=$n $B > $O/part${n}_$i.json 2> $O/part${n}_$i.err
  done
done
tail -3 $O/tests_partition.txt
for f in $O/*.json; do python - $f <<'PY'
import json,sys
t=open(sys.argv[1]).read().strip()
if not t: print(sys.argv[1], "EMPTY"); sys.exit()
d=json.loads(t.splitlines()[-1]); c=d['config']; print(sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'], c.get('peak_hbm_reserved_GB'), (d.get('gpu_fill') or {}).get('gpu_ms_per_step_single_stream'))
PY
done
