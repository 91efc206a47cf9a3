// offt_reg_pow2_f64.hip -- power-of-two double-precision panel kernels, lengths up to 512 and above 1024
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
  // 2048 (profiles/r01_sweep.txt): an 8-column panel owns most of a CU's LDS (one workgroup per CU); the
  // contiguous/contiguous flavour does not need wide panels, and with 4 columns two workgroups fit:
  // 5.74 vs 7.97 ms on the x pass of a 2048 x 256 x 2048 slab
  reg_variant<double, 2048, 16, 16, 16, 8, 8, true>(0, F_SS | F_CS | F_SC);
  reg_variant<double, 2048, 32, 32, 32, 2, 4, true>(1, F_CC);
  reg_variant<double, 4096, 32, 32, 32, 4, 4, true>(0);
  // 8192: one column per workgroup is all the LDS holds (64 KiB image + 32 KiB twiddles); strided sides then move
  // 16-B segments -- a correctness net for long lines (the reference's FFTW takes them), not a tuned path
  reg_variant<double, 8192, 32, 32, 8, 32, 1, true>(0);
}

}  // namespace offtk
