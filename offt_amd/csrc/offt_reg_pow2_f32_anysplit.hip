// offt_reg_pow2_f32_anysplit.hip -- single-precision fft_panelx_k instances of the power-of-two lengths, for per-peer
// blocks fft_panel_k cannot address with shifts: grids split over 3, 5, 6, 7 ... ranks, or the reference's uneven
// F / F+1 blocks (offt-compute.c:132-144).  Same role as offt_reg_pow2_f64_anysplit.hip; never a default.  Without
// them such single-precision passes ran on the any-length kernel (26 % of the roofline).
#include "offt_panel.hpp"

namespace offtk {

void reg_pow2_f32_anysplit() {
  reg_variantx<float, 64, 8, 8, 8, 1, 16, false>(VARIANT_ANYSPLIT, 0);
  reg_variantx<float, 128, 8, 16, 8, 1, 16, false>(VARIANT_ANYSPLIT, 0);
  reg_variantx<float, 256, 16, 16, 16, 1, 16, false>(VARIANT_ANYSPLIT, 0);
  reg_variantx<float, 512, 16, 32, 16, 1, 16, false>(VARIANT_ANYSPLIT, 0);
  reg_variantx<float, 1024, 32, 32, 32, 1, 16, true>(VARIANT_ANYSPLIT, 0);
  reg_variantx<float, 2048, 128, 16, 16, 8, 8, true>(VARIANT_ANYSPLIT, 0);   // (radix-32 butterflies on 1024 threads spill)
  reg_variantx<float, 4096, 256, 16, 16, 16, 4, true>(VARIANT_ANYSPLIT, 0);
}

}  // namespace offtk
