// offt_reg_mixed_f32_b.hip -- single-precision mixed-radix (2^a 3^b 5^c) panel kernels (single precision is an
// extension of this library, so only the commonest lengths get a register kernel; the others run on the
// any-length kernel).  Shapes are the winners of the static sweep, profiles/r01_mixed_sweep_f32.txt.
#include "offt_panel.hpp"

namespace offtk {

void reg_mixed_f32_b() {
  reg_variantx<float, 576, 48, 12, 12, 4, 16, true>(0);  // 67.6 % of 8 TB/s on the 576-point passes
  reg_variantx<float, 768, 32, 8, 8, 12, 16, true>(0);  // 68.5 % of 8 TB/s on the 768-point passes
  reg_variantx<float, 1000, 40, 25, 8, 5, 16, true>(0);  // 43.1 % of 8 TB/s on the 1000-point passes
  reg_variantx<float, 1200, 40, 15, 8, 10, 16, true>(0);  // 46.5 % of 8 TB/s on the 1200-point passes
  reg_variantx<float, 1536, 64, 8, 8, 24, 16, true>(0);  // 48.2 % of 8 TB/s on the 1536-point passes
  reg_variantx<float, 3072, 104, 32, 32, 3, 8, true>(0);  // 41.7 % of 8 TB/s on the 3072-point passes
}

}  // namespace offtk
