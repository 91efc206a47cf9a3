#!/bin/bash
# developer A/B: y and x passes alternating over groups of z-planes (OFFT_ZGROUP_MIB, 0 = off) with cache-keeping y stores,
# x launches on a second stream (OFFT_ZGROUP_STREAMS=2, default) or on the same stream (1)
for st in 2 1; do
for mib in ${MIBS:-0 32 64 128 256}; do
  echo "== OFFT_ZGROUP_MIB=$mib OFFT_ZGROUP_STREAMS=$st"
  export OFFT_ZGROUP_MIB=$mib OFFT_ZGROUP_STREAMS=$st
  python3 tools/dev_shape.py 1024,1024,1024 f64 0 5 2>/dev/null | grep total
  python3 tools/dev_shape.py 1024,1024,1024 f32 0 5 2>/dev/null | grep total
  python3 tools/dev_shape.py 512,512,512 f64 0 5 2>/dev/null | grep total
  python3 tools/dev_shape.py 2048,2048,2048 f32 0 2 2>/dev/null | grep total
done
done
