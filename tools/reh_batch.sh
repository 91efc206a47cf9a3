#!/bin/bash
# Run on the GPU box: a batch of rehearsals (tools/rehearse_rank.py), one line each -> gpurun_out/reh_batch_<tag>.txt
#   tools/reh_batch.sh <tag> "<args of run 1>" "<args of run 2>" ...     (an argument may start with ENV=val words)
TAG=$1; shift
OUT=gpurun_out/reh_batch_$TAG.txt; : > $OUT
for a in "$@"; do
  envs=""; args=""
  for w in $a; do case "$w" in [A-Z_]*=*) envs="$envs $w";; *) args="$args $w";; esac; done
  echo "# $a" >> $OUT
  env $envs python3 tools/rehearse_rank.py $args 2>&1 | grep rehearsal >> $OUT || echo "FAILED: $a" >> $OUT
done
cat $OUT
