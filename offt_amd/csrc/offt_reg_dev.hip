// offt_reg_dev.hip -- developer A/B builds only (tools/dev_build_variant.sh); never part of the product library
#include "offt_panel.hpp"

namespace offtk {

void reg_dev() {
#ifndef OFFT_DEV_NO_1024
  reg_variant<double, 1024, 32, 32, 32, 1, 8, true>(0, 0);
  reg_variant<double, 1024, 16, 16, 16, 4, 8, true>(1, F_ALL);
#endif
#ifdef OFFT_DEV_R2  /* round 2: the product defaults of the 2048-point kernels (+ 256 for 2048 x 256 x 2048 slabs) */
  reg_variant<double, 256, 16, 16, 16, 1, 8, false>(0);
  reg_variant<double, 2048, 16, 16, 16, 8, 8, true>(0, F_SS | F_CS | F_SC);
  reg_variant<double, 2048, 32, 32, 32, 2, 4, true>(1, F_CC);
  reg_variant<double, 2048, 16, 16, 16, 8, 4, true>(2, 0);
  reg_variant<float, 256, 16, 16, 16, 1, 16, false>(0);
  reg_variant<float, 2048, 32, 32, 32, 2, 8, true>(0, F_SS);
  reg_variant<float, 2048, 64, 32, 32, 2, 16, true>(1, F_CS | F_SC);
  reg_variant<float, 2048, 32, 32, 32, 2, 4, false>(2, F_CC);
  reg_variant<float, 2048, 32, 16, 16, 8, 16, true>(3, 0);
  reg_variant<float, 2048, 32, 32, 32, 2, 16, true>(4, 0);
#endif
#ifdef OFFT_DEV_R2F32  /* round 2: radix orders of the 2048-point single-precision kernel on 16-column panels */
  reg_variant<float, 256, 16, 16, 16, 1, 16, false>(0);
  reg_variant<float, 2048, 32, 32, 32, 2, 16, true>(0, F_SS | F_CS | F_SC);
  reg_variant<float, 2048, 32, 32, 32, 2, 4, false>(1, F_CC);
  reg_variant<float, 2048, 32, 32, 2, 32, 16, true>(2, 0);
  reg_variant<float, 2048, 32, 32, 8, 8, 16, true>(3, 0);
  reg_variant<float, 2048, 32, 8, 8, 32, 16, true>(4, 0);
  reg_variant<float, 2048, 32, 16, 16, 8, 16, true>(5, 0);
  reg_variant<float, 2048, 32, 8, 16, 16, 16, true>(6, 0);
  reg_variant<float, 2048, 32, 16, 8, 16, 16, true>(7, 0);
#endif
#ifdef OFFT_DEV_ABL  /* round 2: ablation study (tools/dev_ablate.sh) on the product shapes of 256 / 1024 / 2048 */
  reg_variant<double, 256, 16, 16, 16, 1, 8, false>(0);
  reg_variant<double, 2048, 16, 16, 16, 8, 8, true>(0, F_SS | F_CS | F_SC);
  reg_variant<double, 2048, 32, 32, 32, 2, 4, true>(1, F_CC);
  reg_variant<float, 256, 16, 16, 16, 1, 16, false>(0);
  reg_variant<float, 1024, 32, 32, 32, 1, 16, true>(0, F_SS | F_SC);
  reg_variant<float, 1024, 32, 32, 32, 1, 8, false>(1, F_CC | F_CS);
  reg_variant<float, 2048, 32, 32, 32, 2, 16, true>(0, F_SS | F_CS | F_SC);
  reg_variant<float, 2048, 32, 32, 32, 2, 4, false>(2, F_CC);
#endif
#ifdef OFFT_DEV_PAIR  /* round 2: single-precision column-pair kernels against the one-column product shapes */
  reg_variant<float, 256, 16, 16, 16, 1, 16, false>(0);
  reg_variant<float, 512, 32, 32, 16, 1, 16, false>(0, F_SS | F_CS | F_SC);
  reg_variant<float, 512, 32, 32, 16, 1, 8, false>(1, F_CC);
  reg_variant<float, 1024, 32, 32, 32, 1, 16, true>(0, F_SS | F_SC | F_CS);
  reg_variant<float, 1024, 32, 32, 32, 1, 8, false>(1, F_CC);
  reg_variant<float, 2048, 32, 32, 32, 2, 16, true>(0, F_SS | F_CS | F_SC);
  reg_variant<float, 2048, 32, 32, 32, 2, 4, false>(2, F_CC);
  reg_variant_pair<256, 16, 16, 16, 1, 8, false>(0, 0);
  reg_variant_pair<512, 16, 16, 16, 2, 8, true>(0, 0);
  reg_variant_pair<512, 32, 32, 16, 1, 8, true>(1, 0);
  reg_variant_pair<1024, 32, 32, 32, 1, 8, true>(0, 0);
  reg_variant_pair<1024, 16, 16, 16, 4, 8, true>(1, 0);
  reg_variant_pair<2048, 16, 16, 16, 8, 8, true>(0, 0);
  reg_variant_pair<2048, 32, 32, 32, 2, 4, true>(1, 0);
  reg_variant_pair<2048, 32, 32, 32, 2, 8, true>(2, 0);
#endif
#ifdef OFFT_DEV_512
  reg_variant<double, 512, 16, 16, 16, 2, 8, true>(0, F_ALL);
  reg_variant<double, 512, 16, 16, 8, 4, 8, true>(1, 0);
  reg_variant<double, 512, 16, 8, 8, 8, 8, true>(2, 0);
  reg_variant<double, 512, 8, 8, 8, 8, 8, true>(3, 0);
  reg_variant<double, 512, 16, 16, 8, 4, 16, true>(4, 0);
  reg_variant<double, 1024, 16, 16, 8, 8, 8, true>(2, 0);
  reg_variant<double, 256, 16, 16, 16, 1, 8, false>(0, F_ALL);
  reg_variant<double, 256, 16, 16, 16, 1, 8, true>(1, 0);
  reg_variant<double, 256, 16, 16, 16, 1, 16, true>(2, 0);
  reg_variant<double, 256, 8, 8, 8, 4, 8, true>(3, 0);
#endif
#ifdef OFFT_DEV_2048
  reg_variant<double, 2048, 32, 32, 32, 2, 8, true>(0, F_ALL);
  reg_variant<double, 2048, 16, 16, 16, 8, 8, true>(1, 0);
  reg_variant<double, 2048, 32, 32, 32, 2, 4, true>(2, 0);
  reg_variant<double, 2048, 16, 16, 16, 8, 4, true>(3, 0);
  reg_variantx<double, 2048, 128, 16, 16, 8, 4, true>(4, 0);
  reg_variantx<double, 2048, 128, 16, 16, 8, 8, true>(5, 0);
  reg_variant<double, 4096, 32, 32, 32, 4, 4, true>(0, F_ALL);
  reg_variant<double, 4096, 16, 16, 16, 16, 4, true>(1, 0);
  reg_variant<double, 4096, 32, 32, 32, 4, 2, true>(2, 0);
  reg_variantx<double, 4096, 256, 16, 16, 16, 2, true>(3, 0);
#endif
#ifdef OFFT_DEV_F32
  reg_variant<float, 1024, 32, 32, 32, 1, 16, true>(0, F_SS | F_CS | F_SC);
  reg_variant<float, 1024, 32, 32, 32, 1, 8, false>(1, F_CC);
  reg_variant<float, 1024, 32, 32, 32, 1, 16, false>(2, 0);
  reg_variant<float, 1024, 16, 16, 16, 4, 16, true>(3, 0);
  reg_variant<float, 1024, 16, 16, 16, 4, 16, false>(4, 0);
  reg_variant<float, 2048, 32, 32, 32, 2, 8, true>(0);
  reg_variant<float, 2048, 64, 32, 32, 2, 16, true>(1, 0);
  reg_variant<float, 2048, 32, 32, 32, 2, 4, false>(2, 0);
  reg_variant<float, 2048, 32, 32, 32, 2, 8, false>(3, 0);
  reg_variant<float, 2048, 32, 16, 16, 8, 4, false>(4, 0);
#endif
#ifdef OFFT_DEV_MIXED_LIST
  OFFT_DEV_MIXED_LIST
#endif
#ifdef OFFT_DEV_MIXED
  reg_variantx<double, 768, 32, 12, 8, 8, 8, true>(0);
  reg_variantx<double, 768, 64, 12, 8, 8, 8, true>(1);
  reg_variantx<double, 768, 48, 16, 16, 3, 8, true>(2);
  reg_variantx<double, 768, 32, 12, 8, 8, 16, true>(3);
  reg_variantx<double, 768, 64, 8, 8, 12, 8, true>(4);
  reg_variantx<double, 768, 64, 3, 16, 16, 8, true>(5);
  reg_variantx<double, 1000, 50, 10, 10, 10, 8, true>(0);
  reg_variantx<double, 1000, 100, 10, 10, 10, 8, true>(1);
  reg_variantx<double, 1000, 64, 10, 10, 10, 8, true>(2);
  reg_variantx<double, 1000, 40, 5, 8, 25, 8, true>(3);
  reg_variantx<double, 384, 16, 8, 8, 6, 16, true>(0);
  reg_variantx<double, 384, 32, 8, 8, 6, 8, true>(1);
  reg_variantx<double, 384, 16, 6, 8, 8, 16, true>(2);
  reg_variantx<double, 384, 16, 24, 16, 1, 16, true>(3);
  reg_variantx<double, 15, 1, 15, 1, 1, 64, false>(0);
  reg_variantx<double, 45, 3, 15, 3, 1, 16, false>(0);
  reg_variantx<double, 90, 6, 5, 6, 3, 16, true>(0);
#endif
#ifdef OFFT_DEV_EXTRA
  reg_variant<double, 1024, 16, 4, 16, 16, 8, true>(2, 0);
  reg_variant<double, 1024, 16, 16, 4, 16, 8, true>(3, 0);
  reg_variant<double, 1024, 16, 16, 16, 4, 16, true>(4, 0);
  reg_variant<double, 1024, 16, 16, 16, 4, 4, true>(5, 0);
  reg_variant<double, 1024, 16, 16, 16, 4, 8, false>(6, 0);
  reg_variant<double, 1024, 32, 32, 32, 1, 16, true>(7, 0);
#endif
}

}  // namespace offtk
