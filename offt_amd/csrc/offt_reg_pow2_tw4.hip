// offt_reg_pow2_tw4.hip -- strided / strided panel kernels of the short power-of-two lengths with the twiddles of a LONG
// line on their stores: the first sub-pass of the four-step decomposition (offt_kernels.hip, four_pass) -- 8192 = 64 x 128,
// 16384 = 128 x 128, 32768 = 128 x 256, 65536 = 256 x 256.  Same shapes as the lengths' default kernels.
#include "offt_panel.hpp"

namespace offtk {

void reg_pow2_tw4() {
  // (2 ... 16 points: for lengths like 6000 = 4 x 1500 or 12000 = 4 x 3000, whose long factor has a mixed-radix register kernel)
  reg_variant_tw4<double, 2, 2, 2, 1, 1, 64, false>(0);
  reg_variant_tw4<double, 4, 4, 4, 1, 1, 64, false>(0);
  reg_variant_tw4<double, 8, 8, 8, 1, 1, 64, false>(0);
  reg_variant_tw4<double, 16, 16, 16, 1, 1, 64, false>(0);
  reg_variant_tw4<double, 32, 32, 32, 1, 1, 64, false>(0);
  reg_variant_tw4<double, 64, 8, 8, 8, 1, 8, false>(0);
  reg_variant_tw4<double, 128, 16, 16, 8, 1, 8, false>(0);
  reg_variant_tw4<double, 256, 16, 16, 16, 1, 8, false>(0);
  reg_variant_tw4<float, 2, 2, 2, 1, 1, 64, false>(0);
  reg_variant_tw4<float, 4, 4, 4, 1, 1, 64, false>(0);
  reg_variant_tw4<float, 8, 8, 8, 1, 1, 64, false>(0);
  reg_variant_tw4<float, 16, 16, 16, 1, 1, 64, false>(0);
  reg_variant_tw4<float, 32, 32, 32, 1, 1, 64, false>(0);
  reg_variant_tw4<float, 64, 8, 8, 8, 1, 16, false>(0);
  reg_variant_tw4<float, 128, 16, 16, 8, 1, 16, false>(0);
  reg_variant_tw4<float, 256, 16, 16, 16, 1, 16, false>(0);
}

}  // namespace offtk
