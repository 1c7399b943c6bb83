#!/bin/bash
# round 5, GPU session 2: fixed glue tests, the frozen-stage fusions (bitwise tests, r101 parity gates, bench A/B)
export TMPDIR=/tmp
O=gpurun_out/r5s2; mkdir -p $O
python -m pytest tests/test_gpu_glue.py "tests/test_gpu_resnet.py::test_stem_without_the_im2col_matrix_is_bit_identical" "tests/test_gpu_resnet.py::test_frozen_bottleneck_join_in_the_conv_epilogue_is_bit_identical" tests/test_gpu_model.py::test_batchnorm_fold_gate_at_the_default_config -m gpu -q -s > $O/new_tests.txt 2>&1
echo "new tests rc=$?" >> $O/new_tests.txt
python -m pytest tests/test_gpu_resnet.py "tests/test_gpu_fullsize.py::test_r101_yaml_teacher_and_student_at_600x1200" "tests/test_gpu_trajectory.py" -m "gpu" -q -k "r101 or resnet or bottleneck" > $O/r101_tests.txt 2>&1
echo "r101 tests rc=$?" >> $O/r101_tests.txt
B="python bench.py --model r101 --steps 30 --warmup 5 --no-cpu-baseline --no-secondary"
for i in 1 2; do
  $B > $O/r101_new_$i.json 2> $O/r101_new_$i.err
  SFOD_NO_FUSE_STEM=1 SFOD_NO_FUSE_FROZEN_JOIN=1 $B > $O/r101_old_$i.json 2> $O/r101_old_$i.err
done
SFOD_NO_FUSE_STEM=1 $B > $O/r101_nostem.json 2> $O/r101_nostem.err
python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-secondary > $O/vgg_default.json 2> $O/vgg_default.err
tail -3 $O/new_tests.txt; tail -3 $O/r101_tests.txt
for f in $O/r101_*.json $O/vgg_default.json; do python - $f <<'PY'
import json,sys
t=open(sys.argv[1]).read().strip()
if not t: print(sys.argv[1], "EMPTY"); sys.exit()
d=json.loads(t.splitlines()[-1]); print(sys.argv[1], d['value'], d['ms_per_step'], d.get('gpu_fill',{}).get('native_launches_per_step'), d['config'].get('pseudo_labels_per_image'))
PY
done
