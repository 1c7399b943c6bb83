#!/bin/bash
# round 5, GPU session 13: is the GPU ever idle in the real (two-stream) step?  kernel trace of the default-mode timed steps
export TMPDIR=/tmp
O=gpurun_out/r5s13; mkdir -p $O
bias_of() { python3 bench.py "$@" --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-kernel-timer 2>/dev/null | python3 -c "import json,sys; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['config']['planted_labels']['background_bias'])"; }
run() {   # name, bench args...
  name=$1; shift
  vb=$(bias_of "$@")
  rm -rf $O/prof
  rocprofv3 --kernel-trace -d $O/prof -o kt -- python3 bench.py "$@" --plant-bias $vb --no-cpu-baseline --no-secondary --no-kernel-timer > $O/${name}.json 2> $O/${name}.err
  DB=$(find $O/prof -name "*.db" | head -1)
  echo "== $name: bench.py $* (profiled)" >> $O/busy.txt
  python3 -c "import json; d=json.loads(open('$O/${name}.json').read().strip().splitlines()[-1]); print('   ', d['value'], 'images/s', d['ms_per_step'], 'ms/step under the profiler')" >> $O/busy.txt
  python3 tools/busy_fraction.py $DB 0.5 0.95 >> $O/busy.txt
  rm -rf $O/prof
}
rm -f $O/busy.txt
run vgg_b8 --steps 60 --warmup 10
run vgg_b1 --batch 1 --steps 300 --warmup 20
run r101_b8 --model r101 --steps 40 --warmup 5
cat $O/busy.txt
