#!/bin/bash
# round 5, GPU session 1: new parity tests first, full suite with durations, planted-label calibration sweep
export TMPDIR=/tmp
O=gpurun_out/r5s1; mkdir -p $O
python -m pytest tests/test_gpu_glue.py tests/test_gpu_two_rank.py -m gpu -q -s > $O/new_tests.txt 2>&1
echo "new tests rc=$?" >> $O/new_tests.txt
python -m pytest tests -m gpu -q --durations=150 --maxfail=40 > $O/gpu_suite.txt 2>&1
echo "suite rc=$?" >> $O/gpu_suite.txt
for S in 1 4 8 16 30; do
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-kernel-timer --plant-scale $S > $O/plant_vgg_$S.json 2> $O/plant_vgg_$S.err
done
for S in 1 2 4; do
  python bench.py --model r101 --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-kernel-timer --plant-scale $S > $O/plant_r101_$S.json 2> $O/plant_r101_$S.err
done
tail -3 $O/new_tests.txt; tail -5 $O/gpu_suite.txt; grep -h "planted labels" $O/plant_*.err
