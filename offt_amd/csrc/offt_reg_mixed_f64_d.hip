// offt_reg_mixed_f64_d.hip -- double-precision mixed-radix (2^a 3^b 5^c) panel kernels, group d of 4.
// One shape per length: <T, N, threads per line, R0, R1, R2, columns, split re/im exchange>, the winner of the
// static sweep over radix order x threads per line x panel width (tools/sweep_mixed.py, every candidate and
// its time in profiles/r01_mixed_sweep_f64.txt).  The percentage is algorithmic bytes / time of the passes of that
// length against 8 TB/s, measured on an N^3 grid (an N x 256 x N slab above 1600).
#include "offt_panel.hpp"

namespace offtk {

void reg_mixed_f64_d() {
  reg_variantx<double, 96, 8, 4, 6, 4, 16, true>(0);  // 28.7 % of 8 TB/s on the 96-point passes
  reg_variantx<double, 160, 8, 4, 4, 10, 16, true>(0);  // 56.6 % of 8 TB/s on the 160-point passes
  reg_variantx<double, 288, 24, 12, 6, 4, 8, true>(0);  // 69.9 % of 8 TB/s on the 288-point passes
  reg_variantx<double, 480, 32, 16, 2, 15, 16, true>(0);  // 75.0 % of 8 TB/s on the 480-point passes
  reg_variantx<double, 720, 48, 16, 15, 3, 8, true>(0);  // 70.1 % of 8 TB/s on the 720-point passes
  reg_variantx<double, 1000, 100, 10, 10, 10, 8, true, F_SS | F_CS | F_SC>(0);  // 57.0 % of 8 TB/s on the 1000-point passes
  reg_variantx<double, 1000, 100, 10, 10, 10, 4, true, F_CC>(1, F_CC);  // contig/contig: two narrow workgroups per CU
  reg_variantx<double, 1440, 96, 16, 6, 15, 8, true>(0);  // 57.8 % of 8 TB/s on the 1440-point passes
  reg_variantx<double, 1920, 128, 16, 8, 15, 8, true, F_SS | F_CS | F_SC>(0);  // 63.9 % of 8 TB/s on the 1920-point passes
  reg_variantx<double, 1920, 128, 16, 8, 15, 4, true, F_CC>(1, F_CC);  // contig/contig: two narrow workgroups per CU
  reg_variantx<double, 2560, 160, 16, 16, 10, 4, true>(0);  // 54.9 % of 8 TB/s on the 2560-point passes
}

}  // namespace offtk
