#!/bin/bash
# developer probe: any-length kernel timing (OFFT_MIX_THREADS overrides the workgroup size)
for n in ${@:-768 1000 384}; do
  timeout -k 10 120 python tools/dev_perf.py $n f64 2>&1 | grep "^N=" | head -1 || exit 1
done
