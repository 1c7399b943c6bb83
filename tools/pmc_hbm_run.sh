# HBM traffic per launch of the step's kernels: rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE in SEPARATE passes (they do not
# fit one pass: MI355X_MICROARCH.md "rocprofv3 PMC slots"), summarised with the gfx950 corrections by tools/pmc_summary.py.
# usage (GPU box): bash tools/pmc_hbm_run.sh <out.json> [bench.py args]
export TMPDIR=/tmp
OUT=$1; shift
B="python3 bench.py --no-overlap --no-cpu-baseline --no-secondary --no-kernel-timer --no-smi --steps 3 --warmup 2 $@"
rm -rf gpurun_out/pmc_hbm; mkdir -p gpurun_out/pmc_hbm
rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pmc_hbm/f -o f --output-format csv -- $B > /dev/null 2> gpurun_out/pmc_hbm/f.err
rocprofv3 --pmc WRITE_SIZE -d gpurun_out/pmc_hbm/w -o w --output-format csv -- $B > /dev/null 2> gpurun_out/pmc_hbm/w.err
python3 tools/pmc_summary.py gpurun_out/pmc_hbm/f gpurun_out/pmc_hbm/w $OUT
find gpurun_out/pmc_hbm -name "*.csv" -size +20M -delete
