#!/bin/bash
# developer A/B: plane pad of the scratch volume W (elements) for single precision -- 72 elements = 576 B puts every odd
# plane 64 B off the 128-B lines, so the 128-B store segments of the z pass straddle two lines
for w in 72 0 16 80 144 272; do
  echo "== OFFT_WPAD=$w"
  OFFT_WPAD=$w python3 tools/dev_shape.py 1024,1024,1024 f32 0 4 2>/dev/null | grep total
  OFFT_WPAD=$w python3 tools/dev_shape.py 1024,1024,1024 f32 0 4 -1,0,0 2>/dev/null | grep total
  OFFT_WPAD=$w python3 tools/dev_shape.py 256,2048,2048 f32 0 3 2>/dev/null | grep total
  OFFT_WPAD=$w python3 tools/dev_shape.py 2048,256,2048 f32 0 3 2>/dev/null | grep total
  OFFT_WPAD=$w python3 tools/dev_shape.py 512,512,512 f32 0 4 2>/dev/null | grep total
done
