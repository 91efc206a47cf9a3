// offt_reg_mixed_f64_b.hip -- double-precision mixed-radix (2^a 3^b 5^c) panel kernels, group b.
// Shapes (threads per line, radix order, columns) picked by the static sweep, profiles/r01_mixed_sweep.txt.
#include "offt_panel.hpp"

namespace offtk {

void reg_mixed_f64_b() {
}

}  // namespace offtk
