#!/bin/bash
# developer A/B of two dev builds on the 2048 / 1024 kernels: tools/dev_r2_ab2.sh <libA> <libB>
for rep in 1 2; do
for lib in "$@"; do
  export OFFT_AMD_LIB=$lib
  echo "== $lib (rep $rep)"
  python3 tools/dev_shape.py 2048,256,2048 f64 0 4 2>/dev/null | grep total
  python3 tools/dev_shape.py 256,2048,2048 f64 0 4 2>/dev/null | grep total
  python3 tools/dev_shape.py 2048,256,2048 f32 0 4 2,-1,4 2>/dev/null | grep total
  python3 tools/dev_shape.py 256,2048,2048 f32 0 4 -1,4,4 2>/dev/null | grep total
  python3 tools/dev_shape.py 1024,1024,1024 f64 0 6 2>/dev/null | grep total
done
done
