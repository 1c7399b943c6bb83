#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r5s35; mkdir -p $O
bias_of() { python3 bench.py "$@" --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-kernel-timer 2>/dev/null | python3 -c "import json,sys; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['config']['planted_labels']['background_bias'])"; }
vb=$(bias_of)
rm -rf $O/prof
rocprofv3 --kernel-trace -d $O/prof -o kt -- python3 bench.py --steps 24 --warmup 6 --plant-bias $vb --no-cpu-baseline --no-secondary --no-kernel-timer > $O/b8.json 2> $O/b8.err
DB=$(find $O/prof -name "*.db" | head -1)
python3 tools/step_timeline.py $DB 3 30 > $O/b8_timeline.txt 2>&1
python3 tools/busy_fraction.py $DB 0.5 0.95 > $O/b8_busy.txt 2>&1
rm -rf $O/prof
head -4 $O/b8_timeline.txt | tail -3; cat $O/b8_busy.txt | head -3
