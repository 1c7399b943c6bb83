export TMPDIR=/tmp
python tools/profile_host.py --model r101 --batch 8 --steps 5 > gpurun_out/r2f_r101_host.txt 2>&1
python tools/profile_torch_ops.py --model r101 --batch 8 --steps 2 > gpurun_out/r2f_r101_torch_ops.txt 2>&1
rm -rf gpurun_out/prof_r101; rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r101 -o r101 -- python3 bench.py --model r101 --steps 15 --warmup 3 --no-cpu-baseline --no-secondary --no-kernel-timer > gpurun_out/r2f_r101_prof_bench.json 2> gpurun_out/r2f_r101_prof.err
DB=$(find gpurun_out/prof_r101 -name "*.db" | head -1)
python tools/rocpd_stats.py $DB 18 > gpurun_out/r2f_r101_kernel_stats.csv
python tools/busy_fraction.py $DB 0.4 > gpurun_out/r2f_r101_busy.txt
rm -rf gpurun_out/prof_r101
tail -5 gpurun_out/r2f_r101_busy.txt; head -30 gpurun_out/r2f_r101_host.txt
