# usage (GPU box): bash tools/pmc_conv_run.sh <out.json> <bench_conv.py args ...>
# SQ / GRBM counters per kernel of a tools/bench_conv.py run (e.g. --variants --layers conv4_2): clock held, matrix-pipe busy
# share, parked / issue-stalled wave cycles, LDS bank conflicts -- one line per kernel instantiation.
export TMPDIR=/tmp
OUT=${1:-gpurun_out/sq_conv.json}; shift
B="python3 tools/bench_conv.py --rounds 3 $@"
rm -rf gpurun_out/pmc_conv; mkdir -p gpurun_out/pmc_conv
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -d gpurun_out/pmc_conv/p1 -o p1 --output-format csv -- $B > /dev/null 2> gpurun_out/pmc_conv/p1.err
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM -d gpurun_out/pmc_conv/p2 -o p2 --output-format csv -- $B > /dev/null 2> gpurun_out/pmc_conv/p2.err
tail -2 gpurun_out/pmc_conv/p1.err
python3 tools/pmc_sq_summary.py $OUT gpurun_out/pmc_conv/p1 gpurun_out/pmc_conv/p2
rm -rf gpurun_out/pmc_conv
