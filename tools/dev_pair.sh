#!/bin/bash
# developer A/B: single-precision column-pair kernels (descriptor variant 200 + id) against the one-column defaults
export OFFT_AMD_LIB=${OFFT_AMD_LIB:-build/dev/pair/liboffthip.so}
R=${R:-4}
echo "== 1024^3 f32: default, pair 0 (32x32), pair 1 (16x16x4) on z,y / on all"
for v in -1,-1,-1 -1,200,200 -1,201,201 200,200,200 201,201,201 200,-1,-1; do python3 tools/dev_shape.py 1024,1024,1024 f32 0 $R $v 2>/dev/null | grep total; done
echo "== 512^3 f32"
for v in -1,-1,-1 -1,200,200 -1,201,201 200,200,200; do python3 tools/dev_shape.py 512,512,512 f32 0 $R $v 2>/dev/null | grep total; done
echo "== 256^3 f32"
for v in -1,-1,-1 -1,200,200 200,200,200; do python3 tools/dev_shape.py 256,256,256 f32 0 $R $v 2>/dev/null | grep total; done
echo "== 2048 slabs f32 (x,y,z variants)"
for v in -1,-1,-1 -1,-1,200 -1,-1,201 -1,-1,202 200,-1,200 201,-1,200; do python3 tools/dev_shape.py 2048,256,2048 f32 0 3 $v 2>/dev/null | grep total; done
for v in -1,-1,-1 -1,200,200 -1,202,202 -1,201,201; do python3 tools/dev_shape.py 256,2048,2048 f32 0 3 $v 2>/dev/null | grep total; done
