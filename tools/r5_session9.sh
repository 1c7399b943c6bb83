#!/bin/bash
# round 5, GPU session 9: the steps-in-flight bound -- memory and throughput A/B (alternating), the windowed workload check
export TMPDIR=/tmp
O=gpurun_out/r5s9; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-secondary"
for i in 1 2; do
  $B --steps 60 > $O/vgg_bound_$i.json 2> $O/vgg_bound_$i.err
  $B --steps 60 --opts SFOD.MAX_STEPS_IN_FLIGHT 0 > $O/vgg_unbound_$i.json 2> $O/vgg_unbound_$i.err
  $B --batch 1 --steps 200 > $O/b1_bound_$i.json 2> $O/b1_bound_$i.err
  $B --batch 1 --steps 200 --opts SFOD.MAX_STEPS_IN_FLIGHT 0 > $O/b1_unbound_$i.json 2> $O/b1_unbound_$i.err
  $B --model r101 --steps 30 > $O/r101_bound_$i.json 2> $O/r101_bound_$i.err
  $B --model r101 --steps 30 --opts SFOD.MAX_STEPS_IN_FLIGHT 0 > $O/r101_unbound_$i.json 2> $O/r101_unbound_$i.err
done
$B --opts SFOD.MAX_STEPS_IN_FLIGHT 1 --steps 60 > $O/vgg_depth1.json 2> $O/vgg_depth1.err
$B --opts SFOD.MAX_STEPS_IN_FLIGHT 3 --steps 60 > $O/vgg_depth3.json 2> $O/vgg_depth3.err
$B --res full --steps 30 > $O/full_bound.json 2> $O/full_bound.err
sleep 12
$B --trainer base --steps 60 > $O/base_bound.json 2> $O/base_bound.err
$B --steps 300 --dtype bf16 > $O/bf16_300.json 2> $O/bf16_300.err; echo "rc=$?" >> $O/bf16_300.err
$B --steps 400 > $O/vgg_400.json 2> $O/vgg_400.err; echo "rc=$?" >> $O/vgg_400.err
for f in $O/*.json; do python - $f <<'PY'
import json,sys
t=open(sys.argv[1]).read().strip()
if not t: print(sys.argv[1], "EMPTY"); sys.exit()
d=json.loads(t.splitlines()[-1]); c=d['config']; print(sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'], c.get('peak_hbm_reserved_GB'), c.get('peak_hbm_allocated_GB'), c.get('pseudo_labels_per_image',{}).get('mean'), (c.get('workload_check') or {}).get('pseudo_labels_per_image',{}).get('mean'), (d.get('gpu_fill') or {}).get('host_enqueue_ms_per_step'))
PY
done
