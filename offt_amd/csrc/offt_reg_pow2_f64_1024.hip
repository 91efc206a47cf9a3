// offt_reg_pow2_f64_1024.hip -- the double-precision 1024-point panel kernels (the headline length)
#include "offt_panel.hpp"

namespace offtk {

void reg_pow2_f64_1024() {
  // static sweep result (profiles/r01_sweep.txt): E=16 (radix 16x16x4, 4 waves/SIMD, no
  // spills) and E=32 (radix 32x32, one exchange fewer but 256 VGPRs, 2 waves/SIMD, spills in two
  // flavours) are tied within box-to-box noise at 1024^3; E=16 is the default, E=32 stays selectable.
  reg_variant<double, 1024, 32, 32, 32, 1, 8, true>(0, 0);
  reg_variant<double, 1024, 16, 16, 16, 4, 8, true>(1, F_ALL);
  reg_variant<double, 1024, 32, 32, 32, 1, 4, false>(2, 0);
}

}  // namespace offtk
