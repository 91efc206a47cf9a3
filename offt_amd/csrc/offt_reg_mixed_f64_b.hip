// offt_reg_mixed_f64_b.hip -- double-precision mixed-radix (2^a 3^b 5^c) panel kernels, group b of 4.
// One shape per length: <T, N, threads per line, R0, R1, R2, columns, split re/im exchange>, the winner of the
// static sweep over radix order x threads per line x panel width (tools/sweep_mixed.py, every candidate and
// its time in profiles/r01_mixed_sweep_f64.txt).  The percentage is algorithmic bytes / time of the passes of that
// length against 8 TB/s, measured on an N^3 grid (an N x 256 x N slab above 1600).
#include "offt_panel.hpp"

namespace offtk {

void reg_mixed_f64_b() {
  reg_variantx<double, 120, 8, 15, 8, 1, 16, true>(0);  // 43.1 % of 8 TB/s on the 120-point passes
  reg_variantx<double, 200, 20, 10, 4, 5, 16, true>(0);  // 57.0 % of 8 TB/s on the 200-point passes
  reg_variantx<double, 384, 16, 12, 4, 8, 16, true>(0);  // 67.9 % of 8 TB/s on the 384-point passes
  reg_variantx<double, 600, 40, 15, 8, 5, 8, true>(0);  // 67.4 % of 8 TB/s on the 600-point passes
  reg_variantx<double, 800, 40, 20, 20, 2, 16, true>(0);  // 59.9 % of 8 TB/s on the 800-point passes
  reg_variantx<double, 1200, 120, 12, 10, 10, 8, true, F_SS | F_CS | F_SC>(0);  // 58.1 % of 8 TB/s on the 1200-point passes
  reg_variantx<double, 1200, 120, 12, 10, 10, 4, true, F_CC>(1, F_CC);  // contig/contig: two narrow workgroups per CU
  reg_variantx<double, 1536, 96, 16, 16, 6, 8, true>(0);  // 62.0 % of 8 TB/s on the 1536-point passes
  reg_variantx<double, 2304, 96, 24, 24, 4, 8, true>(0);  // 59.5 % of 8 TB/s on the 2304-point passes
  reg_variantx<double, 3072, 192, 16, 16, 12, 4, true>(0);  // 54.0 % of 8 TB/s on the 3072-point passes
}

}  // namespace offtk
