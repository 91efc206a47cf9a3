// offt_reg_pow2_f64_anysplit.hip -- fft_panelx_k instances of the power-of-two lengths.  fft_panel_k addresses
// per-peer blocks with shifts, so it needs power-of-two block lengths; a grid split over 3, 5, 6, 7 ... ranks has
// blocks of any length, or the reference's uneven F / F+1 blocks (offt-compute.c:132-144).  Those passes run here
// (same radices and panel shape as the defaults, division-based block addressing); never a default.
#include "offt_panel.hpp"

namespace offtk {

void reg_pow2_f64_anysplit() {
  reg_variantx<double, 64, 8, 8, 8, 1, 8, false>(VARIANT_ANYSPLIT, 0);
  reg_variantx<double, 128, 8, 16, 8, 1, 8, false>(VARIANT_ANYSPLIT, 0);
  reg_variantx<double, 256, 16, 16, 16, 1, 8, false>(VARIANT_ANYSPLIT, 0);
  reg_variantx<double, 512, 32, 16, 16, 2, 8, true>(VARIANT_ANYSPLIT, 0);
  reg_variantx<double, 1024, 64, 16, 16, 4, 8, true>(VARIANT_ANYSPLIT, 0);
  reg_variantx<double, 2048, 128, 16, 16, 8, 8, true>(VARIANT_ANYSPLIT, 0);
  reg_variantx<double, 4096, 256, 16, 16, 16, 4, true>(VARIANT_ANYSPLIT, 0);
}

}  // namespace offtk
