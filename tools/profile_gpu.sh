#!/bin/bash
# Run on the GPU box (via gpurun): kernel-trace stats + separate PMC passes for HBM traffic.
# Outputs under gpurun_out/prof_*; summaries are copied into profiles/ afterwards (tools/summarize_profiles.py).
set -o pipefail
export TMPDIR=/tmp
TAG=${1:-r01}
BENCH_ARGS=${2:-"--steps 5 --warmup 1 --no-cpu-baseline"}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py $BENCH_ARGS > $OUT/stats_bench.log 2>&1 || { tail -5 $OUT/stats_bench.log; exit 1; }
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/pmc_fetch_bench.log 2>&1 || { tail -5 $OUT/pmc_fetch_bench.log; exit 1; }
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/pmc_write_bench.log 2>&1 || { tail -5 $OUT/pmc_write_bench.log; exit 1; }
find $OUT -name "*.csv" | head -20
# keep the merged-back payload small: drop the raw per-dispatch traces except the stats + counter csv
find $OUT -name "*.db" -delete
du -sh $OUT
