# usage (GPU box): bash tools/experiments/wgrad_stream_ab.sh    -- VGG weight gradients on a side stream: A/B at batch 8 and 1
for b in 8 1; do
  for v in 1 0 1 0; do
    st=80; [ $b = 1 ] && st=200
    SFOD_VGG_WGRAD_STREAM=$v python3 bench.py --batch $b --no-cpu-baseline --no-secondary --no-kernel-timer --no-planted --steps $st --warmup 10 2>/dev/null \
      | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('batch $b WGRAD_STREAM=$v', d['value'], d['ms_per_step'])"
  done
done
