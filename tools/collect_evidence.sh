# Collect the per-round evidence set on the GPU box (one gpurun call; ~12 GPU-minutes):
#   bash tools/collect_evidence.sh <tag> [pmc|bench|r101|all]
# pmc:   rocprofv3 kernel trace + SQ counters + HBM traffic counters of the default bench workload (writes the traffic file
#        bench.py reads, gpurun_out/<tag>_pmc_hbm_traffic.json -> copy to profiles/pmc_hbm_traffic_latest.json BEFORE the
#        bench part so that the bench line carries it)
# bench: the default bench line (with cpu_baseline and the secondary block) and the side configurations
export TMPDIR=/tmp
TAG=${1:-rX}; WHAT=${2:-all}
O=gpurun_out
# the planted-label background bias is calibrated at start-up (~50 teacher passes): profiled runs take the calibrated value
# of an unprofiled run of the same configuration instead, so that the kernel trace holds the steps only
bias_of() { python3 bench.py "$@" --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-kernel-timer 2>/dev/null | python3 -c "import json,sys; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['config']['planted_labels']['background_bias'])"; }
# A process that held > 100 GB (the 1024x2048 configurations) is still giving its VRAM back for several seconds after it
# exits; a bench started meanwhile finds little free memory and spends its steps in the allocator's free-and-retry path
# (round 5: the --res full line came back at 15 images/s twice, 64 when run alone).  Wait until the device is empty.
wait_vram() {
  for i in $(seq 1 60); do
    used=$(rocm-smi --showmeminfo vram --json 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); c=next(iter(d.values())); print(int(next(v for k,v in c.items() if 'Used' in k)) >> 30)" 2>/dev/null || echo 0)
    [ "${used:-0}" -lt 8 ] && return 0
    sleep 1
  done
  echo "wait_vram: still ${used} GiB in use after 60 s" >&2
}
VB=$(bias_of); RB=$(bias_of --model r101)
echo "planted background bias: vgg $VB r101 $RB"
if [ "$WHAT" = pmc ] || [ "$WHAT" = all ]; then
  rm -rf $O/prof_kt
  rocprofv3 --kernel-trace --stats -d $O/prof_kt -o kt -- python3 bench.py --plant-bias $VB --no-overlap --no-cpu-baseline --no-secondary --no-kernel-timer --no-smi --steps 20 --warmup 5 > /dev/null 2> $O/${TAG}_kt.err
  DB=$(find $O/prof_kt -name "*.db" | head -1)
  python3 tools/rocpd_stats.py $DB 25 > $O/${TAG}_bf16x3_B8_r600_single_stream_kernel_stats.csv
  rm -rf $O/prof_kt
  bash tools/pmc_sq_run.sh $O/${TAG}_sq_counters.json --plant-bias $VB > $O/${TAG}_sq.log 2>&1
  bash tools/pmc_hbm_run.sh $O/${TAG}_pmc_hbm_traffic_bf16x3.json --plant-bias $VB > $O/${TAG}_hbm.log 2>&1
  python3 tools/bench_conv.py --dtype bf16x3 2>&1 | grep -v amdgpu.ids > $O/${TAG}_bf16x3_conv_layers.txt
  python3 tools/bench_conv.py --dtype bf16x3 --wgrad 2>&1 | grep -v amdgpu.ids > $O/${TAG}_bf16x3_conv_layers_wgrad.txt
fi
if [ "$WHAT" = all ] && [ -s $O/${TAG}_pmc_hbm_traffic_bf16x3.json ]; then
  # the bench lines below carry THIS capture; the .meta beside it (tools/pmc_summary.py) is the fingerprint of the kernel
  # sources the counters were taken on -- bench.py drops roofline.traffic when the sources have changed since
  cp $O/${TAG}_pmc_hbm_traffic_bf16x3.json profiles/pmc_hbm_traffic_latest.json
  cp $O/${TAG}_pmc_hbm_traffic_bf16x3.json.meta profiles/pmc_hbm_traffic_latest.json.meta
  bash tools/pmc_hbm_run.sh $O/${TAG}_pmc_hbm_traffic_r101_f16x3.json --model r101 --plant-bias $RB > $O/${TAG}_hbm_r101.log 2>&1
  if [ -s $O/${TAG}_pmc_hbm_traffic_r101_f16x3.json ]; then
    cp $O/${TAG}_pmc_hbm_traffic_r101_f16x3.json profiles/pmc_hbm_traffic_r101_latest.json
    cp $O/${TAG}_pmc_hbm_traffic_r101_f16x3.json.meta profiles/pmc_hbm_traffic_r101_latest.json.meta
  fi
fi
if [ "$WHAT" = bench ] || [ "$WHAT" = all ]; then
  python3 bench.py > $O/${TAG}_bench_default.json 2> $O/${TAG}_bench_default.err
  python3 bench.py --kernel-table --steps 30 --no-cpu-baseline --no-secondary > /dev/null 2> $O/${TAG}_kernel_table.txt
  for cfg in "--trainer base" "--trainer base --res full --steps 40" "--res full --steps 40" "--batch 1 --steps 200" "--opts SFOD.ELIDE_DEAD_BRANCHES False" "--dtype f16x3"; do
    name=$(echo $cfg | tr -d ' -' | tr '.' '_')
    wait_vram
    python3 bench.py $cfg --no-cpu-baseline --no-secondary > $O/${TAG}_bench_${name}.json 2> $O/${TAG}_bench_${name}.err
  done
  # config #5 with its labelled blocks (other parity mode fp32; reduced-precision bf16x3 / bf16) from child processes
  python3 bench.py --model r101 --steps 40 --no-cpu-baseline > $O/${TAG}_bench_modelr101steps40.json 2> $O/${TAG}_bench_modelr101steps40.err
fi
if [ "$WHAT" = r101 ] || [ "$WHAT" = all ]; then
  # config #5 in its parity mode (f16x3): kernel trace of single-stream steps, GPU busy fraction (union of kernel intervals
  # over the span of the timed steps, and the un-profiled step time of the bench line beside the profiled kernel-time sum)
  rm -rf $O/prof_r101
  rocprofv3 --kernel-trace --stats -d $O/prof_r101 -o r101 -- python3 bench.py --model r101 --plant-bias $RB --no-overlap --steps 12 --warmup 3 --no-cpu-baseline --no-secondary --no-kernel-timer --no-smi > $O/${TAG}_r101_f16x3_profiled_bench.json 2> $O/${TAG}_r101_prof.err
  DB=$(find $O/prof_r101 -name "*.db" | head -1)
  python3 tools/rocpd_stats.py $DB 15 > $O/${TAG}_r101_f16x3_kernel_stats.csv
  python3 tools/busy_fraction.py $DB 0.3 > $O/${TAG}_r101_f16x3_busy_fraction.txt
  rm -rf $O/prof_r101
  python3 bench.py --model r101 --no-overlap --steps 20 --no-cpu-baseline --no-secondary > $O/${TAG}_bench_r101_f16x3_single_stream.json 2> /dev/null
fi
ls -la $O | grep ${TAG}_ | awk '{print $5, $9}'
