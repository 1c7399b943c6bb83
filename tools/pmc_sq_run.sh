# usage (GPU box): bash tools/pmc_sq_run.sh <out.json> [extra bench.py args, e.g. --dtype f16x3 | --model r101]
export TMPDIR=/tmp
OUT=${1:-gpurun_out/sq_counters.json}; shift
B="python3 bench.py --no-overlap --no-cpu-baseline --no-secondary --no-kernel-timer --no-smi --steps 3 --warmup 2 $@"
rm -rf gpurun_out/pmc_sq; mkdir -p gpurun_out/pmc_sq
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -d gpurun_out/pmc_sq/p1 -o p1 --output-format csv -- $B > /dev/null 2> gpurun_out/pmc_sq/p1.err
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM -d gpurun_out/pmc_sq/p2 -o p2 --output-format csv -- $B > /dev/null 2> gpurun_out/pmc_sq/p2.err
tail -2 gpurun_out/pmc_sq/p1.err
python3 tools/pmc_sq_summary.py $OUT gpurun_out/pmc_sq/p1 gpurun_out/pmc_sq/p2
find gpurun_out/pmc_sq -name "*.csv" -size +20M -delete
du -sh gpurun_out/pmc_sq
