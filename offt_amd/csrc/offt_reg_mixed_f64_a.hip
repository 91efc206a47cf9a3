// offt_reg_mixed_f64_a.hip -- double-precision mixed-radix (2^a 3^b 5^c) panel kernels, group a of 4.
// One shape per length: <T, N, threads per line, R0, R1, R2, columns, split re/im exchange>, the winner of the
// static sweep over radix order x threads per line x panel width (tools/sweep_mixed.py, every candidate and
// its time in profiles/r01_mixed_sweep_f64.txt).  The percentage is algorithmic bytes / time of the passes of that
// length against 8 TB/s, measured on an N^3 grid (an N x 256 x N slab above 1600).
#include "offt_panel.hpp"

namespace offtk {

void reg_mixed_f64_a() {
  reg_variantx<double, 144, 12, 12, 2, 6, 16, true>(0);  // 53.8 % of 8 TB/s on the 144-point passes
  reg_variantx<double, 240, 16, 15, 16, 1, 8, true>(0);  // 67.3 % of 8 TB/s on the 240-point passes
  reg_variantx<double, 400, 40, 10, 10, 4, 8, true>(0);  // 67.4 % of 8 TB/s on the 400-point passes
  reg_variantx<double, 640, 80, 10, 8, 8, 8, true, F_SS | F_CS | F_SC>(0);  // 65.2 % of 8 TB/s on the 640-point passes
  reg_variantx<double, 640, 80, 10, 8, 8, 4, true, F_CC>(1, F_CC);  // contig/contig: two narrow workgroups per CU
  reg_variantx<double, 960, 64, 8, 8, 15, 8, true>(0);  // 71.7 % of 8 TB/s on the 960-point passes
  reg_variantx<double, 1280, 80, 16, 8, 10, 8, true>(0);  // 59.1 % of 8 TB/s on the 1280-point passes
  reg_variantx<double, 1600, 80, 20, 20, 4, 8, true>(0);  // 57.5 % of 8 TB/s on the 1600-point passes
  reg_variantx<double, 2400, 120, 5, 20, 24, 8, true>(0);  // 54.0 % of 8 TB/s on the 2400-point passes
  reg_variantx<double, 3200, 160, 20, 8, 20, 4, true>(0);  // 50.6 % of 8 TB/s on the 3200-point passes
}

}  // namespace offtk
