// offt_reg_mixed_f32_a.hip -- single-precision mixed-radix (2^a 3^b 5^c) panel kernels (single precision is an
// extension of this library, so only the commonest lengths get a register kernel; the others run on the
// any-length kernel).  Shapes are the winners of the static sweep, profiles/r01_mixed_sweep_f32.txt.
#include "offt_panel.hpp"

namespace offtk {

void reg_mixed_f32_a() {
  reg_variantx<float, 384, 16, 8, 8, 6, 16, true>(0);  // 69.4 % of 8 TB/s on the 384-point passes
  reg_variantx<float, 640, 40, 16, 8, 5, 16, true>(0);  // 47.8 % of 8 TB/s on the 640-point passes
  reg_variantx<float, 960, 64, 15, 8, 8, 16, true>(0);  // 53.4 % of 8 TB/s on the 960-point passes
  reg_variantx<float, 1152, 48, 12, 8, 12, 16, true>(0);  // 49.5 % of 8 TB/s on the 1152-point passes
  reg_variantx<float, 1280, 44, 16, 8, 10, 16, true>(0);  // 42.9 % of 8 TB/s on the 1280-point passes
  reg_variantx<float, 1920, 64, 32, 10, 6, 16, true>(0);  // 43.3 % of 8 TB/s on the 1920-point passes
}

}  // namespace offtk
