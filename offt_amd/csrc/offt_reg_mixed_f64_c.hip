// offt_reg_mixed_f64_c.hip -- double-precision mixed-radix (2^a 3^b 5^c) panel kernels, group c of 4.
// One shape per length: <T, N, threads per line, R0, R1, R2, columns, split re/im exchange>, the winner of the
// static sweep over radix order x threads per line x panel width (tools/sweep_mixed.py, every candidate and
// its time in profiles/r01_mixed_sweep_f64.txt).  The percentage is algorithmic bytes / time of the passes of that
// length against 8 TB/s, measured on an N^3 grid (an N x 256 x N slab above 1600).
#include "offt_panel.hpp"

namespace offtk {

void reg_mixed_f64_c() {
  reg_variantx<double, 100, 10, 10, 10, 1, 16, true>(0);  // 33.3 % of 8 TB/s on the 100-point passes
  reg_variantx<double, 192, 8, 8, 8, 3, 16, true>(0);  // 60.9 % of 8 TB/s on the 192-point passes
  reg_variantx<double, 320, 16, 10, 8, 4, 8, true>(0);  // 66.3 % of 8 TB/s on the 320-point passes
  reg_variantx<double, 576, 24, 12, 8, 6, 8, true>(0);  // 73.0 % of 8 TB/s on the 576-point passes
  reg_variantx<double, 768, 32, 12, 8, 8, 8, true>(0);  // 70.3 % of 8 TB/s on the 768-point passes
  reg_variantx<double, 1152, 48, 24, 24, 2, 16, true, F_SS | F_CS | F_SC>(0);  // 61.8 % of 8 TB/s on the 1152-point passes
  reg_variantx<double, 1152, 48, 12, 12, 8, 4, true, F_CC>(1, F_CC);  // contig/contig: two narrow workgroups per CU
  reg_variantx<double, 1500, 104, 15, 10, 10, 8, true, F_SS | F_CS | F_SC>(0);  // 47.6 % of 8 TB/s on the 1500-point passes
  reg_variantx<double, 1500, 160, 15, 10, 10, 4, true, F_CC>(1, F_CC);  // contig/contig: two narrow workgroups per CU
  reg_variantx<double, 2000, 104, 10, 20, 10, 8, true>(0);  // 54.0 % of 8 TB/s on the 2000-point passes
  reg_variantx<double, 3000, 208, 20, 15, 10, 4, true>(0);  // 42.5 % of 8 TB/s on the 3000-point passes
}

}  // namespace offtk
