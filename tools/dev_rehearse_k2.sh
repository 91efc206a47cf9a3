#!/bin/bash
# developer probe: the y pass (K2) of the slab schedule at the 8-rank shapes, kernel variants and chunk thickness
for a in "--n 2048 --dtype f32" "--n 1024 --dtype f32" "--n 1024 --dtype f64"; do
  for v in "" "-1,202,-1" "-1,201,-1" "-1,0,-1" "-1,1,-1"; do
    python3 tools/rehearse_rank.py $a --ranks 8 --p1 1 --reps 4 ${v:+--variants=$v} 2>/dev/null | grep rehearsal
  done
  for t2 in 8 16 64; do python3 tools/rehearse_rank.py $a --ranks 8 --p1 1 --reps 4 --t2 $t2 2>/dev/null | grep rehearsal; done
done
