export TMPDIR=/tmp
rm -rf gpurun_out/prof_kt
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_kt -o kt -- python3 bench.py --no-overlap --no-cpu-baseline --no-secondary --no-kernel-timer --steps 20 --warmup 5 > /dev/null 2> gpurun_out/kt.err
DB=$(find gpurun_out/prof_kt -name "*.db" | head -1)
python3 tools/rocpd_stats.py $DB 25 > gpurun_out/r2f_kernel_stats.csv
rm -rf gpurun_out/prof_kt
