for g in 0 4 8 16 32 64 128 256; do
  echo "== OFFT_XCD_REMAP=$g"
  OFFT_XCD_REMAP=$g python3 tools/dev_shape.py 1024,1024,1024 f32 0 4 2>/dev/null | grep total
  OFFT_XCD_REMAP=$g python3 tools/dev_shape.py 256,2048,2048 f32 0 3 2>/dev/null | grep total
  OFFT_XCD_REMAP=$g python3 tools/dev_shape.py 256,2048,2048 f64 0 3 2>/dev/null | grep total
  OFFT_XCD_REMAP=$g python3 tools/dev_shape.py 512,512,512 f32 0 4 2>/dev/null | grep total
done
