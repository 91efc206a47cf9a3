#!/bin/bash
# Run on the GPU box: kernel traces of one rank's share of the multi-GPU schedules (tools/rehearse_rank.py), summarised
# per kernel (tools/summarize_rehearsal.py) into gpurun_out/rehearse_<tag>.txt -- copy into profiles/ to keep.
#   tools/profile_rehearsal.sh <tag> <rehearse_rank.py args...>
export TMPDIR=/tmp
TAG=$1; shift
OUT=gpurun_out/rehearse_$TAG; rm -rf $OUT; mkdir -p $OUT
python3 tools/rehearse_rank.py "$@" --reps 5 > $OUT/plain.log 2>&1 || { tail -5 $OUT/plain.log; exit 1; }
OFFT_LOG_PASSES=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/rehearse_rank.py "$@" --reps 6 > $OUT/trace.log 2>&1 || { tail -5 $OUT/trace.log; exit 1; }
{ echo "# tools/rehearse_rank.py $*"; grep rehearsal $OUT/plain.log; python3 tools/summarize_rehearsal.py $OUT/trace.log $OUT/trace; } | tee gpurun_out/rehearse_$TAG.txt
find $OUT -name "*.db" -delete
