// offt_reg_bluestein.hip -- double-precision instances of fft_bluestein_k: one panel shape per convolution length M.
// A line of N points (2N - 1 <= M) with no register kernel runs on the M-point instance.  Shapes are chosen for register
// pressure first (two M-point FFTs live in one kernel): 8 columns (128-B segments) up to M = 1024, 4 columns above.
#include "offt_bluestein.hpp"

namespace offtk {

void reg_bluestein_f32();

void reg_bluestein_all() {
  reg_bluestein<double, 256, 16, 16, 16, 1, 8>();
  reg_bluestein<double, 512, 16, 16, 2, 16, 8>();
  reg_bluestein<double, 1024, 16, 16, 4, 16, 8>();
  reg_bluestein<double, 2048, 16, 16, 8, 16, 4>();   // 512 threads, two workgroups per CU (8 columns: 1024 threads at 128 registers spill)
  reg_bluestein<double, 4096, 32, 32, 4, 32, 4>();
  reg_bluestein_f32();
}

}  // namespace offtk
