// offt_reg_pow2_f64.hip -- power-of-two double-precision panel kernels
#include "offt_panel.hpp"

namespace offtk {

void reg_pow2_f64() {
  // variant 0 of every length is the default unless a defmask says otherwise; higher ids are the
  // static sweep (LDS tile width x radix order x re/im split), see DESIGN.md section 5.
  // ---- f64 ----
  reg_variant<double, 2, 2, 2, 1, 1, 64, false>(0);
  reg_variant<double, 4, 4, 4, 1, 1, 64, false>(0);
  reg_variant<double, 8, 8, 8, 1, 1, 64, false>(0);
  reg_variant<double, 16, 16, 16, 1, 1, 64, false>(0);
  reg_variant<double, 32, 32, 32, 1, 1, 64, false>(0);
  reg_variant<double, 64, 8, 8, 8, 1, 8, false>(0);
  reg_variant<double, 128, 16, 16, 8, 1, 8, false>(0);
  reg_variant<double, 256, 16, 16, 16, 1, 8, false>(0);
  reg_variant<double, 512, 32, 32, 16, 1, 8, true>(0, 0);
  reg_variant<double, 512, 16, 16, 16, 2, 8, true>(1, F_ALL);
  // static sweep result (profiles/r01_sweep.txt): E=16 (radix 16x16x4, 4 waves/SIMD, no
  // spills) beats E=32 (radix 32x32, one exchange fewer but 256 VGPRs and 2 waves/SIMD)
  // on every flavour at 1024^3, so it is the default; E=32 stays selectable as variant 0.
  reg_variant<double, 1024, 32, 32, 32, 1, 8, true>(0, 0);
  reg_variant<double, 1024, 16, 16, 16, 4, 8, true>(1, F_ALL);
  reg_variant<double, 1024, 32, 32, 32, 1, 4, false>(2, 0);
  reg_variant<double, 2048, 32, 32, 32, 2, 8, true>(0);
  reg_variant<double, 4096, 32, 32, 32, 4, 4, true>(0);
}

}  // namespace offtk
