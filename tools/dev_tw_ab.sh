#!/bin/bash
# developer A/B: twiddle tables fetched before / committed after the panel loads (tw_new) against the r01 load-and-store
# loop at kernel entry (tw_old); tw_copy = everything but the memory accesses compiled out
for rep in 1 2; do
for v in tw_old tw_new tw_copy; do
  export OFFT_AMD_LIB=build/dev/$v/liboffthip.so
  echo "== $v (rep $rep)"
  python3 tools/dev_shape.py 1024,1024,1024 f64 0 5 2>/dev/null | grep total
  python3 tools/dev_shape.py 1024,1024,1024 f32 0 4 2>/dev/null | grep total
  python3 tools/dev_shape.py 1024,1024,1024 f32 0 4 -1,0,0 2>/dev/null | grep total
  python3 tools/dev_shape.py 2048,256,2048 f64 0 3 2>/dev/null | grep total
  python3 tools/dev_shape.py 256,2048,2048 f64 0 3 2>/dev/null | grep total
  python3 tools/dev_shape.py 256,2048,2048 f32 0 3 2>/dev/null | grep total
done
done
