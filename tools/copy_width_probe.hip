// developer probe: streaming copy at 8 B vs 16 B per lane (non-temporal), chip-wide TB/s
#include <hip/hip_runtime.h>
#include <cstdio>
template <typename V> __global__ void __launch_bounds__(256) copy_k(const V *in, V *out, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    V v = __builtin_nontemporal_load(in + i);
    __builtin_nontemporal_store(v, out + i);
  }
}
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
template <typename V> void run(const char *name, void *a, void *b, size_t bytes, int grid) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  size_t n = bytes / sizeof(V);
  float best = 1e9;
  for (int r = 0; r < 6; r++) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(copy_k<V>, dim3(grid), dim3(256), 0, 0, (const V *)a, (V *)b, n);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
  }
  printf("%s grid %d: %.3f ms, %.2f TB/s (read + write)\n", name, grid, best, 2.0 * bytes / best / 1e9);
}
int main() {
  size_t bytes = 8ull << 30;
  void *a, *b; hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMemset(a, 1, bytes); hipMemset(b, 0, bytes);
  for (int grid : {256 * 8, 256 * 16, 256 * 32, 1 << 20}) { run<f2>("8 B/lane ", a, b, bytes, grid); run<f4>("16 B/lane", a, b, bytes, grid); }
  return 0;
}
