// developer probe: a strided/strided pass as a pure copy -- what does the ACCESS PATTERN alone allow?
// (the ablation builds of round 2, profiles/r02_ablation.txt, showed that the single-precision strided passes run no faster with their
// arithmetic, twiddles and LDS exchanges compiled out).  A workgroup copies a panel of N rows x SEG bytes (rows `pitch`
// bytes apart) through registers: all loads issued, then all stores, like a panel FFT kernel.  Knobs: bytes per lane,
// segment width, row pitch, rows per panel, the LDS reservation (workgroups per CU), the XCD-aware panel order and a bit
// rotation of the row a lane group takes (which rows one wave instruction touches).
//   hipcc --offload-arch=gfx950 -O3 tools/strided_width_probe.hip -o build/dev/strided_width
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned panel_of_block(unsigned bid, unsigned lim, unsigned gshift) {
  if (bid >= lim) return bid;
  const unsigned x = bid & 7u, r = bid >> 3;
  return ((r >> gshift) << (gshift + 3u)) + (x << gshift) + (r & ((1u << gshift) - 1u));
}
constexpr int ilog2(int n) { return n <= 1 ? 0 : 1 + ilog2(n >> 1); }

template <typename V, int N, int E, int LANES>
__global__ void __launch_bounds__((N / E) * LANES) copy_panel(const V *in, V *out, int ncp, size_t row, size_t batch, unsigned lim,
                                                              unsigned gshift, int rot) {
  extern __shared__ unsigned char smem[];
  constexpr int TPL = N / E, LB = ilog2(TPL);
  const int tid = threadIdx.x, c = tid % LANES;
  int j = tid / LANES;
  j = ((j << rot) | (j >> (LB - rot))) & (TPL - 1);
  const unsigned bid = panel_of_block(blockIdx.x, lim, gshift);
  const size_t base = (size_t)(bid / ncp) * batch + (size_t)(bid % ncp) * LANES + c;
  V v[E];
#pragma unroll
  for (int i = 0; i < E; i++) v[i] = __builtin_nontemporal_load(in + base + (size_t)(j + i * TPL) * row);
  if (smem[tid] == 77) v[0] = v[1];  // keep the LDS reservation alive
#pragma unroll
  for (int i = 0; i < E; i++) __builtin_nontemporal_store(v[i], out + base + (size_t)(j + i * TPL) * row);
}

// contiguous-in / strided-out: plane [c][n] -> [n][c] (the z and y passes of the z-y-x schedule).  Loads run lanes along
// the line (VL bytes per lane), stores run lanes across the COLS columns of the panel (VS bytes per lane, COLS * 8 B... the
// element is ES bytes; a VS-byte store covers VS/ES adjacent columns).
template <typename EL, typename VS, int N, int E, int COLS>
__global__ void __launch_bounds__((N / E) * COLS) transpose_panel(const EL *in, VS *out, int ncp, unsigned lim, unsigned gshift) {
  extern __shared__ unsigned char smem[];
  constexpr int TPL = N / E, K = sizeof(VS) / sizeof(EL);   // K columns per store lane
  const int tid = threadIdx.x;
  const unsigned bid = panel_of_block(blockIdx.x, lim, gshift);
  const size_t plane = (size_t)(bid / ncp) * N * N;          // square N x N planes
  const int c0 = (bid % ncp) * COLS;
  EL v[E];
  {
    const int j = tid % TPL, c = tid / TPL;
    const EL *src = in + plane + (size_t)(c0 + c) * N + j;
#pragma unroll
    for (int i = 0; i < E; i++) v[i] = __builtin_nontemporal_load(src + i * TPL);
  }
  if (smem[tid] == 77) v[0] = v[1];
  {
    // store side: thread = (column group cg of K columns, row jj), E/K rows each
    constexpr int CG = COLS / K, TPR = (N / E) * COLS / CG;  // threads along the rows
    const int cg = tid % CG, jj = tid / CG;
    VS *dst = out + (plane + (size_t)c0) / K + cg;
#pragma unroll
    for (int i = 0; i < E / K; i++) {
      VS w;
      EL *wp = reinterpret_cast<EL *>(&w);
#pragma unroll
      for (int k = 0; k < K; k++) wp[k] = v[i * K + k];
      __builtin_nontemporal_store(w, dst + (size_t)(jj + i * TPR) * (N / K));
    }
  }
}

static void *A, *B;
static size_t BYTES = 8ull << 30;

template <typename V, int N, int E, int LANES>
void run(size_t pitch, int lds, int G, int rot) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const size_t row = pitch / sizeof(V);
  const int ncp = (int)(pitch / (LANES * sizeof(V)));
  const size_t batch = row * N;
  const size_t nb = BYTES / (pitch * N);
  const unsigned nblk = (unsigned)(nb * ncp);
  unsigned gs = 0;
  while (G > 1 && (2 << gs) <= G) ++gs;
  const unsigned lim = G > 0 ? ((nblk >> (gs + 3)) << (gs + 3)) : 0u;
  (void)hipFuncSetAttribute((const void *)copy_panel<V, N, E, LANES>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  float best = 1e9;
  for (int r = 0; r < 4; r++) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((copy_panel<V, N, E, LANES>), dim3(nblk), dim3((N / E) * LANES), lds, 0, (const V *)A, (V *)B, ncp, row, batch, lim, gs, rot);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
  }
  printf("%2zu B/lane seg %4zu B rows %4d pitch %3zu KiB thr %4d lds %3d KiB G %3d rot %d: %7.3f ms  %5.1f %% of 8 TB/s\n", sizeof(V),
         LANES * sizeof(V), N, pitch >> 10, (N / E) * LANES, lds >> 10, G, rot, best, 2.0 * BYTES / (best * 1e-3) / 8e12 * 100);
  fflush(stdout);
}

template <typename EL, typename VS, int N, int E, int COLS>
void run_t(int lds, int G) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int ncp = N / COLS;
  const size_t nb = BYTES / ((size_t)N * N * sizeof(EL));
  const unsigned nblk = (unsigned)(nb * ncp);
  unsigned gs = 0;
  while (G > 1 && (2 << gs) <= G) ++gs;
  const unsigned lim = G > 0 ? ((nblk >> (gs + 3)) << (gs + 3)) : 0u;
  (void)hipFuncSetAttribute((const void *)transpose_panel<EL, VS, N, E, COLS>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  float best = 1e9;
  for (int r = 0; r < 4; r++) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((transpose_panel<EL, VS, N, E, COLS>), dim3(nblk), dim3((N / E) * COLS), lds, 0, (const EL *)A, (VS *)B, ncp, lim, gs);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
  }
  printf("transpose el %2zu B store %2zu B/lane seg %4zu B N %4d thr %4d lds %3d KiB G %3d: %7.3f ms  %5.1f %% of 8 TB/s\n", sizeof(EL), sizeof(VS),
         COLS * sizeof(EL), N, (N / E) * COLS, lds >> 10, G, best, 2.0 * BYTES / (best * 1e-3) / 8e12 * 100);
  fflush(stdout);
}

int main() {
  (void)hipMalloc(&A, BYTES); (void)hipMalloc(&B, BYTES); (void)hipMemset(A, 1, BYTES); (void)hipMemset(B, 0, BYTES);
  puts("# contiguous-in / strided-out planes (transposes)");
  run_t<f4, f4, 1024, 32, 8>(64 << 10, 32);    // f64-like: 16-B elements, 8 columns
  run_t<f2, f2, 1024, 32, 16>(64 << 10, 32);   // f32: 8-B elements, 16 columns, 8-B stores
  run_t<f2, f4, 1024, 32, 16>(64 << 10, 32);   // f32: 16-B stores of two columns
  run_t<f2, f2, 1024, 32, 8>(64 << 10, 32);    // f32: 8 columns (64-B segments), 8-B stores
  run_t<f2, f4, 1024, 32, 8>(64 << 10, 32);
  run_t<f2, f2, 1024, 32, 32>(128 << 10, 32);  // 256-B segments
  run_t<f2, f4, 1024, 32, 32>(128 << 10, 32);
  run_t<f2, f2, 1024, 32, 16>(64 << 10, 8);
  run_t<f2, f2, 1024, 32, 16>(64 << 10, 64);
  run_t<f2, f2, 1024, 32, 16>(64 << 10, 0);
  run_t<f4, f4, 2048, 32, 8>(128 << 10, 32);
  run_t<f2, f2, 2048, 32, 16>(128 << 10, 32);
  run_t<f2, f4, 2048, 32, 16>(128 << 10, 32);
  return 0;
}
int main_ss() {
  puts("# row pitch (1024 rows x 128 B, 16 B/lane, E=32, 2 workgroups/CU)");
  for (size_t p : {2048, 4096, 8192, 16384, 32768, 65536}) run<f4, 1024, 32, 8>(p, 64 << 10, 32, 0);
  puts("# 8 B/lane at the same pitches");
  for (size_t p : {8192, 16384}) run<f2, 1024, 32, 16>(p, 64 << 10, 32, 0);
  puts("# which rows one wave instruction touches (rotation of the lane group's row index)");
  for (size_t p : {8192, 16384})
    for (int rot : {1, 2, 3, 4}) run<f4, 1024, 32, 8>(p, 64 << 10, 32, rot);
  puts("# XCD run length");
  for (int G : {0, 8, 64, 128}) run<f4, 1024, 32, 8>(8192, 64 << 10, G, 0);
  puts("# segment width at pitch 8 KiB / 16 KiB");
  for (size_t p : {8192, 16384}) { run<f4, 1024, 32, 4>(p, 64 << 10, 32, 0); run<f4, 1024, 16, 16>(p, 64 << 10, 32, 0); run<f4, 1024, 32, 16>(p, 128 << 10, 32, 0); }
  puts("# workgroups per CU (LDS reservation) at pitch 8 KiB");
  for (int l : {32 << 10, 48 << 10, 64 << 10, 128 << 10}) run<f4, 1024, 32, 8>(8192, l, 32, 0);
  puts("# 2048 rows x 128 B (one workgroup per CU), pitch 16 / 32 KiB");
  for (size_t p : {16384, 32768}) {
    run<f4, 2048, 32, 8>(p, 128 << 10, 32, 0);
    run<f4, 2048, 16, 8>(p, 128 << 10, 32, 0);
    run<f4, 2048, 32, 8>(p, 64 << 10, 32, 0);
    for (int rot : {2, 3}) run<f4, 2048, 32, 8>(p, 128 << 10, 32, rot);
  }
  return 0;
}
