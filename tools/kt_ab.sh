# usage (GPU box): bash tools/kt_ab.sh <tag> "<ENV=...>" ["<ENV2=...>" ...]
# rocprofv3 kernel trace of the default bench workload (single stream, 25 steps) once per environment setting, e.g.
#   bash tools/kt_ab.sh r4 "SFOD_P3_M16=1" "SFOD_P3_M16=0"   -> gpurun_out/<tag>_kt_<i>.csv (ms per step per kernel)
# BENCH_ARGS="--batch 1" in the environment: extra bench.py arguments for every run
export TMPDIR=/tmp
TAG=$1; shift
i=0
for envs in "$@"; do
  rm -rf gpurun_out/prof_kt
  ( export $envs; rocprofv3 --kernel-trace --stats -d gpurun_out/prof_kt -o kt -- python3 bench.py --no-overlap --no-cpu-baseline --no-secondary --no-kernel-timer --no-smi --steps 20 --warmup 5 $BENCH_ARGS > /dev/null 2> gpurun_out/${TAG}_kt_$i.err )
  DB=$(find gpurun_out/prof_kt -name "*.db" | head -1)
  python3 tools/rocpd_stats.py $DB 25 > gpurun_out/${TAG}_kt_$i.csv
  echo "== $envs"; head -14 gpurun_out/${TAG}_kt_$i.csv | cut -c1-150
  rm -rf gpurun_out/prof_kt
  i=$((i+1))
done
