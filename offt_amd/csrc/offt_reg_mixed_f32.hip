// offt_reg_mixed_f32.hip -- single-precision mixed-radix (2^a 3^b 5^c) panel kernels.
#include "offt_panel.hpp"

namespace offtk {

void reg_mixed_f32() {
}

}  // namespace offtk
