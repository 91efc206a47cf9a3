// offt_reg_pow2_f32_big.hip -- power-of-two single-precision panel kernels, 2048 points and up (their own translation
// unit: together with the shorter lengths they were the longest compile of the library)
#include "offt_panel.hpp"

namespace offtk {

void reg_pow2_f32_big() {
  // 2048 f32.  A strided side wants 16 columns (16 x 8 B = 128-B segments; the 8-column panel that used to be the
  // strided/strided default moved 64-B segments: 40 % of the roofline in the 8-rank rehearsal,
  // profiles/r02_rehearse_f32_2048_1x8_first.txt).  E=32 on 1024 threads (4 waves per SIMD, a few spilled registers) beats
  // E=64 on 512 threads (2 waves per SIMD) on the y pass, 3.88 vs 4.39 ms, and ties on the z pass; the
  // contiguous/contiguous flavour again prefers a narrow packed panel.
  reg_variant<float, 2048, 32, 32, 32, 2, 16, true>(0, F_SS | F_CS | F_SC);
  reg_variant<float, 2048, 64, 32, 32, 2, 16, true>(1, 0);
  reg_variant<float, 2048, 32, 32, 32, 2, 4, false>(2, F_CC);
  reg_variant<float, 2048, 32, 32, 32, 2, 8, true>(3, 0);
  reg_variant<float, 4096, 32, 32, 32, 4, 4, true>(0);
  reg_variant<float, 8192, 32, 32, 8, 32, 2, true>(0);   // long lines: correctness net, see the f64 comment
}

}  // namespace offtk
