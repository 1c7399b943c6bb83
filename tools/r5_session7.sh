#!/bin/bash
# round 5, GPU session 7: the 1024x2048 bench lines (each after the device is empty), package power beside the bench,
# the full default GPU suite and the sweep suite
export TMPDIR=/tmp
O=gpurun_out/r5s7; mkdir -p $O
wait_vram() {
  for i in $(seq 1 90); do
    used=$(rocm-smi --showmeminfo vram --json 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); c=next(iter(d.values())); print(int(next(v for k,v in c.items() if 'Used' in k)) >> 30)" 2>/dev/null || echo 0)
    echo "vram used GiB: $used" >> $O/vram.log
    [ "${used:-0}" -lt 8 ] && return 0
    sleep 1
  done
}
rocm-smi --showmeminfo vram --json > $O/meminfo_raw.json 2>&1
wait_vram
python3 bench.py --trainer base --res full --steps 40 --no-cpu-baseline --no-secondary > $O/bench_trainerbaseresfullsteps40.json 2> $O/bench_trainerbaseresfullsteps40.err
echo "after base full" >> $O/vram.log; wait_vram
python3 bench.py --res full --steps 40 --no-cpu-baseline --no-secondary > $O/bench_resfullsteps40.json 2> $O/bench_resfullsteps40.err
echo "after full" >> $O/vram.log; wait_vram
rm -f $O/power.txt
bash tools/power_trace.sh $O/power.txt --steps 150
bash tools/power_trace.sh $O/power.txt --steps 150 --dtype f16x3
bash tools/power_trace.sh $O/power.txt --steps 40 --dtype fp32
bash tools/power_trace.sh $O/power.txt --steps 300 --dtype bf16
bash tools/power_trace.sh $O/power.txt --steps 100 --model r101
bash tools/power_trace.sh $O/power.txt --steps 600 --batch 1
python -m pytest tests -m gpu -q --durations=15 > $O/gpu_suite_default.txt 2>&1; echo "rc=$?" >> $O/gpu_suite_default.txt
python -m pytest tests -m "gpu and sweep" -q --durations=10 > $O/gpu_suite_sweep.txt 2>&1; echo "rc=$?" >> $O/gpu_suite_sweep.txt
cat $O/power.txt; tail -5 $O/gpu_suite_default.txt; tail -5 $O/gpu_suite_sweep.txt
for f in $O/bench_*.json; do python - $f <<'PY'
import json,sys
t=open(sys.argv[1]).read().strip()
if not t: print(sys.argv[1], "EMPTY"); sys.exit()
d=json.loads(t.splitlines()[-1]); print(sys.argv[1], d['value'], d['ms_per_step'], d['config'].get('peak_hbm_reserved_GB'), d['config'].get('peak_hbm_allocated_GB'))
PY
done
