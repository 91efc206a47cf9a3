#!/bin/bash
# developer A/B: plane-group alternation (OFFT_ZGROUP_MIB, 0 = off) on the x-y-z (S=1) and y-z-x (is_equalxy) layouts
for z in 256 0; do
  echo "== OFFT_ZGROUP_MIB=$z"
  for n in 512 1024; do for p in F64 F32; do
    OFFT_ZGROUP_MIB=$z python3 -c "
import sys; sys.path.insert(0, 'tools'); sys.path.insert(0, '.')
from dev_shape import run
from offt_amd import api
run(($n, $n, $n), api.$p, 1, 3)
run(($n, $n, $n), api.$p, 0, 3, eq=1)" 2>/dev/null | grep total
  done; done
done
