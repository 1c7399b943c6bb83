#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r5s21; mkdir -p $O
bias_of() { python3 bench.py "$@" --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-kernel-timer 2>/dev/null | python3 -c "import json,sys; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['config']['planted_labels']['background_bias'])"; }
vb=$(bias_of --batch 1)
rm -rf $O/prof
rocprofv3 --kernel-trace -d $O/prof -o kt -- python3 bench.py --batch 1 --steps 40 --warmup 10 --plant-bias $vb --no-cpu-baseline --no-secondary --no-kernel-timer > $O/b1.json 2> $O/b1.err
DB=$(find $O/prof -name "*.db" | head -1)
python3 tools/step_timeline.py $DB 3 10 > $O/b1_timeline.txt 2>&1
rm -rf $O/prof
head -5 $O/b1_timeline.txt
