#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r5s30; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-secondary --no-kernel-timer"
for i in 1 2; do
  $B --res full --steps 30 > $O/full_all_$i.json 2>/dev/null; sleep 12
  SFOD_NO_RPN_BESIDE=1 $B --res full --steps 30 > $O/full_paramonly_$i.json 2>/dev/null; sleep 12
  SFOD_HEAD_WGRAD_STREAM=0 $B --res full --steps 30 > $O/full_nooffchain_$i.json 2>/dev/null; sleep 12
done
for f in $O/*.json; do python - $f <<'PY'
import json,sys
t=open(sys.argv[1]).read().strip()
if not t: print(sys.argv[1], "EMPTY"); sys.exit()
d=json.loads(t.splitlines()[-1]); print(sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'], d['config'].get('peak_hbm_reserved_GB'))
PY
done
