#!/bin/bash
# developer ablation study: which part of a panel kernel costs what.  Builds dev libraries with the twiddles
# (OFFT_ABL_NOTW), the register butterflies (OFFT_ABL_NOBF) and the LDS exchanges (OFFT_ABL_NOEX) compiled out, one at
# a time and all together (= a copy with the kernel's access pattern, occupancy and registers), then
#   tools/dev_ablate.sh run      on the GPU box times them on the 1024 / 2048 shapes.
# Results of ablated kernels are WRONG by construction; only the durations mean something.
set -e
if [ "$1" = build ]; then
  tools/dev_build_variant.sh abl_base -DOFFT_DEV_ABL > /dev/null
  tools/dev_build_variant.sh abl_notw -DOFFT_DEV_ABL -DOFFT_ABL_NOTW > /dev/null
  tools/dev_build_variant.sh abl_nobf -DOFFT_DEV_ABL -DOFFT_ABL_NOBF > /dev/null
  tools/dev_build_variant.sh abl_noex -DOFFT_DEV_ABL -DOFFT_ABL_NOEX > /dev/null
  tools/dev_build_variant.sh abl_copy -DOFFT_DEV_ABL -DOFFT_ABL_NOTW -DOFFT_ABL_NOBF -DOFFT_ABL_NOEX > /dev/null
  exit 0
fi
for v in base notw nobf noex copy; do
  export OFFT_AMD_LIB=build/dev/abl_$v/liboffthip.so
  echo "== $v"
  python3 tools/dev_shape.py 1024,1024,1024 f64 0 4 2>/dev/null | grep total
  python3 tools/dev_shape.py 1024,1024,1024 f32 0 4 2>/dev/null | grep total
  python3 tools/dev_shape.py 2048,256,2048 f64 0 3 2>/dev/null | grep total
  python3 tools/dev_shape.py 256,2048,2048 f64 0 3 2>/dev/null | grep total
  python3 tools/dev_shape.py 2048,256,2048 f32 0 3 2>/dev/null | grep total
  python3 tools/dev_shape.py 256,2048,2048 f32 0 3 2>/dev/null | grep total
done
