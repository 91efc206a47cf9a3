// offt_reg_pow2_f32_pair.hip -- single-precision panel kernels on COLUMN PAIRS (T = f32x2, see offt_panel.hpp): the
// double-precision shapes, carrying two adjacent float columns per lane.  An eligible descriptor (pair_ok() in
// offt_kernels.hip) prefers them where the mask says so; measured against the one-column defaults in
// profiles/r02_pair_ab.txt: 1024^3 10.1 -> 9.3 ms, 512^3 1.32 -> 1.21 ms, the z pass of 2048 x 256 x 2048 3.77 -> 3.41 ms
// (256^3: no gain, not registered).
#include "offt_panel.hpp"

namespace offtk {

void reg_pow2_f32_pair() {
  reg_variant_pair<512, 16, 16, 16, 2, 8, true>(0, F_ALL);
  // (1024 as 32 x 32 with 32 pairs per thread runs at the 256-register limit: 9.6 ms at 1024^3 against 9.3 ms)
  reg_variant_pair<1024, 16, 16, 16, 4, 8, true>(0, F_ALL);
  reg_variant_pair<2048, 16, 16, 16, 8, 8, true>(0, F_SS | F_CS | F_SC);
  reg_variant_pair<2048, 32, 32, 32, 2, 4, true>(1, F_CC);
  reg_variant_pair<2048, 32, 32, 32, 2, 8, true>(2, 0);
  // (2048 on 4-pair panels -- two workgroups per CU, but 64-B segments on a strided side -- was tried again in round 3:
  //  56.8 % against 62.5 % on the z pass of 2048 x 256 x 2048, profiles/r03_pair2048_cols4.txt; not registered)
  // 4096: the one-column kernel fits 4 columns (32-B segments on a strided side, 19 % of the roofline); 4 pairs move 64-B
  // segments: 54 % (profiles/r02_pair_4096.txt)
  reg_variant_pair<4096, 32, 32, 32, 4, 4, true>(0, F_ALL);
}

}  // namespace offtk
