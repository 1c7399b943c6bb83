#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r5s31; mkdir -p $O
bias_of() { python3 bench.py "$@" --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-kernel-timer 2>/dev/null | python3 -c "import json,sys; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['config']['planted_labels']['background_bias'])"; }
vb=$(bias_of --res full)
for tag in all nooff; do
  rm -rf $O/prof
  if [ $tag = nooff ]; then export SFOD_HEAD_WGRAD_STREAM=0; fi
  rocprofv3 --kernel-trace -d $O/prof -o kt -- python3 bench.py --res full --steps 12 --warmup 4 --plant-bias $vb --no-cpu-baseline --no-secondary --no-kernel-timer > $O/$tag.json 2> $O/$tag.err
  DB=$(find $O/prof -name "*.db" | head -1)
  python3 tools/step_timeline.py $DB 3 200 > $O/timeline_$tag.txt 2>&1
  rm -rf $O/prof
  sleep 12
done
head -4 $O/timeline_all.txt; head -4 $O/timeline_nooff.txt
