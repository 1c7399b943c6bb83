# usage (GPU box): bash tools/experiments/bnin_ab.sh     -- in-step A/B of SFOD.FUSE_BN_INPUT (three alternating pairs)
for i in 1 2 3; do
  for on in True False; do
    python3 bench.py --no-cpu-baseline --no-secondary --no-kernel-timer --no-planted --steps 80 --warmup 10 --opts SFOD.FUSE_BN_INPUT $on 2>/dev/null \
      | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('FUSE_BN_INPUT=$on', d['value'], d['ms_per_step'])"
  done
done
