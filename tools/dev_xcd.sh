#!/bin/bash
# developer probe: XCD-aware panel order (0 off, 1 whole chunk per XCD, G = runs of G panels), default 1024 kernels, interleaved repeats
export OFFT_AMD_LIB=${OFFT_AMD_LIB:-build/dev/xcd/liboffthip.so}
for rep in 1 2 3; do
for x in ${XCD_LIST:-0 1 64 1024 32 128 512 2048}; do
  echo -n "rep $rep OFFT_XCD_REMAP=$x  "
  OFFT_XCD_REMAP=$x timeout -k 10 300 python tools/dev_perf.py ${1:-1024} zyx 2>&1 | grep -E "^N=" | sed -E "s/.*total/total/" | cut -c1-150 || exit 1
done
done
