// offt_reg_mixed_f64_e.hip -- double-precision mixed-radix panel kernels, group e: lengths with a factor 7
// (7 * 2^k, 21 * 2^k, 35 * 2^k) and 1001 = 7 * 11 * 13; radices 7, 11, 13, 14 use the direct prime butterfly of
// dft_mixed.  Shapes from the static sweep (profiles/r01_mixed_sweep_f64.txt), as in groups a-d.
#include "offt_panel.hpp"

namespace offtk {

void reg_mixed_f64_e() {
  reg_variantx<double, 224, 16, 14, 16, 1, 8, true>(0);  // 69.0 % of 8 TB/s on the 224-point passes
  reg_variantx<double, 448, 32, 16, 2, 14, 8, true>(0);  // 75.6 % of 8 TB/s on the 448-point passes
  reg_variantx<double, 560, 40, 14, 5, 8, 8, true>(0);  // 68.1 % of 8 TB/s on the 560-point passes
  reg_variantx<double, 672, 56, 14, 12, 4, 8, true>(0);  // 73.4 % of 8 TB/s on the 672-point passes
  reg_variantx<double, 896, 64, 14, 8, 8, 8, true, F_SS | F_CS | F_SC>(0);  // 59.7 % of 8 TB/s on the 896-point passes
  reg_variantx<double, 896, 64, 14, 8, 8, 4, true, F_CC>(1, F_CC);  // contig/contig: two narrow workgroups per CU
  reg_variantx<double, 1001, 80, 13, 7, 11, 8, true, F_SS | F_CS | F_SC>(0);  // 39.8 % of 8 TB/s on the 1001-point passes
  reg_variantx<double, 1001, 96, 13, 7, 11, 4, true, F_CC>(1, F_CC);  // contig/contig: two narrow workgroups per CU
  reg_variantx<double, 1120, 80, 16, 5, 14, 8, true, F_SS | F_CS | F_SC>(0);  // 57.4 % of 8 TB/s on the 1120-point passes
  reg_variantx<double, 1120, 80, 16, 5, 14, 4, true, F_CC>(1, F_CC);  // contig/contig: two narrow workgroups per CU
  reg_variantx<double, 1344, 96, 16, 6, 14, 8, true>(0);  // 59.0 % of 8 TB/s on the 1344-point passes
  reg_variantx<double, 1792, 128, 16, 8, 14, 8, true, F_SS | F_CS | F_SC>(0);  // 60.9 % of 8 TB/s on the 1792-point passes
  reg_variantx<double, 1792, 128, 16, 8, 14, 4, true, F_CC>(1, F_CC);  // contig/contig: two narrow workgroups per CU
}

}  // namespace offtk
