#!/bin/bash
# developer A/B on one box: bench.py under different environment settings, interleaved.
#   tools/dev_ab.sh "OFFT_XCD_REMAP=0" "OFFT_XCD_REMAP=32" ...
for rep in 1 2 3; do
  for cfg in "$@"; do
    echo -n "rep $rep [$cfg] "
    env $cfg timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['pass_ms'], d['roofline']['transform_frac'])" || exit 1
  done
done
