#!/bin/bash
# developer A/B: the contiguous-in / strided-out flavour of the single-precision 1024 kernel on 16-column panels (variant 0,
# 128-B store segments) against the 8-column packed panel (variant 1, 64-B segments, the r01 choice)
for v in -1,-1,-1 -1,0,0 -1,1,1; do
  python3 tools/dev_shape.py 1024,1024,1024 f32 0 4 $v 2>/dev/null | grep total
done
for lib in base copy; do
  export OFFT_AMD_LIB=build/dev/abl_$lib/liboffthip.so
  echo "== ablation $lib, variant 0 on z and y"
  python3 tools/dev_shape.py 1024,1024,1024 f32 0 4 -1,0,0 2>/dev/null | grep total
done
