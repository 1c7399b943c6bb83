# usage (GPU box): bash tools/experiments/r101_tile_ab.sh   -- R101 step with the generic GEMM's tile shape forced (SFOD_GEMM_TILE)
for t in 0 1 2 3 4; do
  SFOD_GEMM_TILE=$t python3 bench.py --model r101 --no-cpu-baseline --no-secondary --no-kernel-timer --no-planted --steps 30 --warmup 6 2>/dev/null \
    | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('SFOD_GEMM_TILE=$t', d['value'], d['ms_per_step'])"
done
