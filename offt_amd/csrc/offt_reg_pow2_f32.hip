// offt_reg_pow2_f32.hip -- power-of-two single-precision panel kernels
#include "offt_panel.hpp"

namespace offtk {

void reg_pow2_f32() {
  // ---- f32 ----
  reg_variant<float, 2, 2, 2, 1, 1, 64, false>(0);
  reg_variant<float, 4, 4, 4, 1, 1, 64, false>(0);
  reg_variant<float, 8, 8, 8, 1, 1, 64, false>(0);
  reg_variant<float, 16, 16, 16, 1, 1, 64, false>(0);
  reg_variant<float, 32, 32, 32, 1, 1, 64, false>(0);
  reg_variant<float, 64, 8, 8, 8, 1, 16, false>(0);
  reg_variant<float, 128, 16, 16, 8, 1, 16, false>(0);
  reg_variant<float, 256, 16, 16, 16, 1, 16, false>(0);
  // f32 moves twice the elements per HBM byte, so LDS/issue work per byte doubles: the
  // contiguous/contiguous flavour is fastest with a packed (one 8-B op per element) exchange
  // on a narrow 8-column panel (2.75 vs 3.63 ms at 1024^3); the flavours with a strided
  // side keep 16 columns (128-B segments) and the split exchange.  profiles/r01_sweep.txt
  reg_variant<float, 512, 32, 32, 16, 1, 16, false>(0, F_SS | F_CS | F_SC);
  reg_variant<float, 512, 32, 32, 16, 1, 8, false>(1, F_CC);
  // (r01 measured the narrow packed panel ahead on contig-in/strided-out as well, 3.63 vs 4.12 ms on the z pass of
  //  1024^3 -- with the scratch planes 64 B off the 128-B lines (offt_host.c, wpad); on aligned planes the 16-column
  //  panel's 128-B store segments win: 3.39-3.43 vs 3.73-3.76 ms, profiles/r02_wpad_f32.txt)
  reg_variant<float, 1024, 32, 32, 32, 1, 16, true>(0, F_SS | F_SC | F_CS);
  reg_variant<float, 1024, 32, 32, 32, 1, 8, false>(1, F_CC);
}

}  // namespace offtk
