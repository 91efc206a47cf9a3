#!/bin/bash
# Run on the GPU box: hardware counters per kernel of one tool run, three separate --pmc passes (8 SQ slots / 4 TCC slots
# per pass, MI355X_MICROARCH.md "rocprofv3 PMC slots"), never combined with a trace.  The program goes directly after `--`.
#   tools/pmc_run.sh <tag> tools/rehearse_rank.py --n 1024 --ranks 8 --p1 1
#   tools/pmc_run.sh <tag> tools/shape_probe.py 2048,256,2048 f32 0 2
# summary -> gpurun_out/pmc_<tag>.txt (copy into profiles/ to keep it)
export TMPDIR=/tmp
TAG=$1; shift
OUT=gpurun_out/pmc_$TAG; rm -rf $OUT; mkdir -p $OUT
PASS_a="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
PASS_b="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE"
PASS_c="TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCC_HIT_sum TCC_MISS_sum"
for p in a b c; do
  v=PASS_$p
  rocprofv3 --pmc ${!v} --output-format csv -d $OUT/$p -- python3 "$@" > $OUT/$p.log 2>&1 || { echo "pass $p failed"; tail -5 $OUT/$p.log; [ $p = c ] || exit 1; }
done
python3 tools/summarize_pmc.py $OUT "$*" | tee gpurun_out/pmc_$TAG.txt
find $OUT -name "*.db" -delete
