// offt_reg_bluestein_f32.hip -- single-precision instances of fft_bluestein_k (see offt_reg_bluestein.hip)
#include "offt_bluestein.hpp"

namespace offtk {

void reg_bluestein_f32() {
  reg_bluestein<float, 256, 16, 16, 16, 1, 16>();
  reg_bluestein<float, 512, 16, 16, 2, 16, 16>();
  reg_bluestein<float, 1024, 16, 16, 4, 16, 16>();
  reg_bluestein<float, 2048, 32, 32, 2, 32, 8>();
  reg_bluestein<float, 4096, 32, 32, 4, 32, 4>();
}

}  // namespace offtk
