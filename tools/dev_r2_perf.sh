#!/bin/bash
# developer A/B of the 2048-point kernels: tools/dev_r2_perf.sh <lib> [reps]
export OFFT_AMD_LIB=$1
R=${2:-4}
for P in f64 f32; do
  python3 tools/dev_shape.py 2048,256,2048 $P 0 $R 2>/dev/null | grep total
  python3 tools/dev_shape.py 256,2048,2048 $P 0 $R 2>/dev/null | grep total
  OFFT_FORCE_PIPELINE=1 python3 tools/dev_shape.py 2048,256,2048 $P 0 $R 2>/dev/null | grep total
  OFFT_FORCE_PIPELINE=1 python3 tools/dev_shape.py 256,2048,2048 $P 0 $R 2>/dev/null | grep total
done
python3 tools/dev_shape.py 1024,1024,1024 f64 0 6 2>/dev/null | grep total
python3 tools/dev_shape.py 1024,1024,1024 f64 1 4 2>/dev/null | grep total
