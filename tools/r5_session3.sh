#!/bin/bash
# round 5, GPU session 3: published ROIAlign vector, loader/glue tests, single-launch BatchNorm finalize (bitwise test, A/B)
export TMPDIR=/tmp
O=gpurun_out/r5s3; mkdir -p $O
python -m pytest tests/test_gpu_ops.py tests/test_gpu_glue.py -m gpu -q -k "bn_finalize or published or bn_fused or glue or conv_bn_relu" > $O/new_tests.txt 2>&1
echo "new tests rc=$?" >> $O/new_tests.txt
python -m pytest tests/test_gpu_trajectory.py tests/test_gpu_resnet.py -m gpu -q -x > $O/traj_tests.txt 2>&1
echo "trajectory tests rc=$?" >> $O/traj_tests.txt
R="python bench.py --model r101 --steps 30 --warmup 5 --no-cpu-baseline --no-secondary"
V="python bench.py --batch 1 --steps 200 --warmup 20 --no-cpu-baseline --no-secondary"
for i in 1 2; do
  $R > $O/r101_fused_$i.json 2> $O/r101_fused_$i.err
  SFOD_BN_FINALIZE_FUSED=0 $R > $O/r101_two_$i.json 2> $O/r101_two_$i.err
  $V > $O/b1_fused_$i.json 2> $O/b1_fused_$i.err
  SFOD_BN_FINALIZE_FUSED=0 $V > $O/b1_two_$i.json 2> $O/b1_two_$i.err
done
tail -3 $O/new_tests.txt; tail -3 $O/traj_tests.txt
for f in $O/r101_*.json $O/b1_*.json; do python - $f <<'PY'
import json,sys
t=open(sys.argv[1]).read().strip()
if not t: print(sys.argv[1], "EMPTY"); sys.exit()
d=json.loads(t.splitlines()[-1]); print(sys.argv[1], d['value'], d['ms_per_step'], d.get('gpu_fill',{}).get('native_launches_per_step'), d['config'].get('pseudo_labels_per_image'))
PY
done
