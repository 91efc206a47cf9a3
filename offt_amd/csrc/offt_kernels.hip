// offt_kernels.hip -- C ABI (offt_hipk.h), kernel registry, twiddle tables and launchers of the
// hand-written CDNA4 (gfx950) kernels; the panel kernels themselves are in offt_panel.hpp.
//
// What the reference does per pencil with FFTW + element-wise memcpy
// (offt-compute.c:959-963 FFTz, 1484-1494 / 1708-1710 FFTy, 2493-2495 FFTx,
//  pack/unpack 1029-1109, 1307-1385, 1773-2058, 2447-2687, transpose 625-639)
// is done here by ONE kernel family: a panel Stockham FFT.
//
//  * a workgroup owns a panel [N x COLS] of one axis: N = FFT length, COLS =
//    independent lines;
//  * every thread keeps E complex points in registers and does radix-R0/R1/R2
//    butterflies entirely in registers (radix 2..32, built from radix-2 DIF
//    stages with compile-time twiddles);
//  * between register stages the panel is exchanged through LDS (Stockham
//    autosort indexing, padded against bank conflicts; optionally re / im in
//    two half-size sweeps so that two workgroups fit the 160 KiB LDS of a CU);
//  * inter-stage twiddles come from a quarter-wave table staged in LDS
//    (exact to 0.5 ulp, no sincos recurrences);
//  * loads and stores use independent stride descriptors, so the transposes
//    and the pack/unpack of the pencil decomposition ride on the FFT's own
//    HBM traffic.  Wave lanes run along whichever dimension is unit-stride
//    (IN_CONTIG / OUT_CONTIG), 16 B per lane.
//
// No MFMA: the path is HBM-bound (1.56 flop/B), see DESIGN.md.

#include <hip/hip_runtime.h>
#include <type_traits>
#include <mutex>
#include <map>
#include <vector>
#include <string>
#include <cstdio>
#include <cmath>
#include <cstring>
#include <cstdlib>
#include <dlfcn.h>
#include <unordered_map>
#include <algorithm>
#include <hip/hiprtc.h>
#include "offt_hipk.h"
#include "offt_panel.hpp"
#include "offt_bluestein.hpp"
#include "offt_rtc_source.inc"

namespace offtk {
std::vector<Variant> &registry() {
  static std::vector<Variant> r;
  return r;
}
std::vector<BlueVariant> &blue_registry() {
  static std::vector<BlueVariant> r;
  return r;
}
}  // namespace offtk

namespace {
using namespace offtk;

thread_local char g_err[512] = "";
#define HIPK_CHECK(call)                                                          \
  do {                                                                            \
    hipError_t e_ = (call);                                                       \
    if (e_ != hipSuccess) {                                                       \
      snprintf(g_err, sizeof g_err, "%s:%d: %s -> %s", __FILE__, __LINE__, #call, \
               hipGetErrorString(e_));                                            \
      return -1;                                                                  \
    }                                                                             \
  } while (0)
// ---------------------------------------------------------------------------
// Any-length kernel: mixed-radix Stockham in LDS with run-time radices.
// It exists so that every length the reference accepts (FFTW takes any N) is
// transformed correctly and at a tolerable cost; lengths with a compile-time
// register kernel never come here, nor do power-of-two splits.
//
// N = r0 * r1 * ... (prime factors, pairs of 2 merged to 4).  One workgroup owns
// COLS lines in two ping-pong LDS images.  A work item of stage s produces ONE
// output of one radix-r butterfly:
//   y[(q-k) r + k + j Ns] = sum_t x[q + t N/r] * w_N^(k t M) * w_r^(j t),
//   k = q mod Ns, M = N / (Ns r)
// with exact table twiddles w_N^m (m = 0..N-1, staged in LDS when it fits).
// Radices 2, 3, 4, 5 are done as whole butterflies (r loads, r-1 twiddles, a
// constant-coefficient DFT, r stores); any other prime radix produces one
// output per work item with r multiply-adds, so a line costs at most
// N * sum(r_s) instead of N^2 and a prime N degenerates to the plain DFT.
// Also handles the reference's uneven A2AV per-peer splits on either side and
// the real-input z pass.
// ---------------------------------------------------------------------------
#define OFFT_MIX_MAXFAC 16
struct GenArgs {
  long long in_axis, in_col, in_b1, in_b2, in_blk;
  long long out_axis, out_col, out_b1, out_b2, out_blk;
  int in_split, in_nfloor, out_split, out_nfloor;
  int n, ncols, nb1, ncp, cols;
  int in_contig, out_contig;
  int conj;
  int real_in;
  int tw_in_lds;
  unsigned xcd_lim, xcd_gshift;
  int nfac;
  int fac[OFFT_MIX_MAXFAC];
  double scale;
  const long long *in_tab, *out_tab;  // per-block element offsets (offt_pass_desc::in_block_tab / out_block_tab) or nullptr
};

__device__ __forceinline__ long long split_off(int k, int split, int nfloor, long long blk, long long axis, const long long *tab) {
  if (split == 0 && nfloor == 0) return (long long)k * axis;  // no split
  int a, r;
  if (nfloor > 0 && k >= split * nfloor) {
    int kk = k - split * nfloor;
    a = nfloor + kk / (split + 1);
    r = kk % (split + 1);
  } else {
    a = k / split;
    r = k % split;
  }
  return (tab ? tab[a] : (long long)a * blk) + (long long)r * axis;
}

// one radix-R butterfly of the any-length kernel: inputs xin[t * is], t = 0..R-1, twiddled by
// w_N^(kM t) from the table, outputs yout[j * os]
template <typename T, int R>
__device__ __forceinline__ void mixed_butterfly(const typename vec2<T>::type *xin, typename vec2<T>::type *yout,
                                                int is, int os, int kM, const typename vec2<T>::type *tw) {
  using V2 = typename vec2<T>::type;
  cx<T> v[R];
#pragma unroll
  for (int t = 0; t < R; ++t) {
    const V2 a = xin[t * is];
    if (t == 0) { v[0] = cx<T>{a.x, a.y}; }
    else {
      const V2 w = tw[kM * t];
      v[t] = cx<T>{a.x * w.x - a.y * w.y, a.x * w.y + a.y * w.x};
    }
  }
  cx<T> o[R];
  if constexpr (R == 2) {
    o[0] = cx<T>{v[0].x + v[1].x, v[0].y + v[1].y};
    o[1] = cx<T>{v[0].x - v[1].x, v[0].y - v[1].y};
  } else if constexpr (R == 3) {
    constexpr T S3 = (T)0.86602540378443864676;  // sin(2 pi / 3)
    const cx<T> s{v[1].x + v[2].x, v[1].y + v[2].y}, d{v[1].x - v[2].x, v[1].y - v[2].y};
    const cx<T> t{v[0].x - (T)0.5 * s.x, v[0].y - (T)0.5 * s.y};
    const cx<T> e{S3 * d.y, -S3 * d.x};  // -i * S3 * d
    o[0] = cx<T>{v[0].x + s.x, v[0].y + s.y};
    o[1] = cx<T>{t.x + e.x, t.y + e.y};
    o[2] = cx<T>{t.x - e.x, t.y - e.y};
  } else if constexpr (R == 4) {
    const cx<T> a{v[0].x + v[2].x, v[0].y + v[2].y}, b{v[0].x - v[2].x, v[0].y - v[2].y};
    const cx<T> c{v[1].x + v[3].x, v[1].y + v[3].y}, d{v[1].x - v[3].x, v[1].y - v[3].y};
    o[0] = cx<T>{a.x + c.x, a.y + c.y};
    o[2] = cx<T>{a.x - c.x, a.y - c.y};
    o[1] = cx<T>{b.x + d.y, b.y - d.x};  // b - i d
    o[3] = cx<T>{b.x - d.y, b.y + d.x};  // b + i d
  } else {
    static_assert(R == 5, "radix");
    constexpr T C1 = (T)0.30901699437494742410, C2 = (T)-0.80901699437494742410;   // cos(2 pi/5), cos(4 pi/5)
    constexpr T S1 = (T)0.95105651629515357212, S2 = (T)0.58778525229247312917;    // sin(2 pi/5), sin(4 pi/5)
    const cx<T> s1{v[1].x + v[4].x, v[1].y + v[4].y}, s2{v[2].x + v[3].x, v[2].y + v[3].y};
    const cx<T> d1{v[1].x - v[4].x, v[1].y - v[4].y}, d2{v[2].x - v[3].x, v[2].y - v[3].y};
    const cx<T> p1{v[0].x + C1 * s1.x + C2 * s2.x, v[0].y + C1 * s1.y + C2 * s2.y};
    const cx<T> p2{v[0].x + C2 * s1.x + C1 * s2.x, v[0].y + C2 * s1.y + C1 * s2.y};
    const cx<T> q1{S1 * d1.x + S2 * d2.x, S1 * d1.y + S2 * d2.y};
    const cx<T> q2{S2 * d1.x - S1 * d2.x, S2 * d1.y - S1 * d2.y};
    o[0] = cx<T>{v[0].x + s1.x + s2.x, v[0].y + s1.y + s2.y};
    o[1] = cx<T>{p1.x + q1.y, p1.y - q1.x};  // p1 - i q1
    o[4] = cx<T>{p1.x - q1.y, p1.y + q1.x};
    o[2] = cx<T>{p2.x + q2.y, p2.y - q2.x};
    o[3] = cx<T>{p2.x - q2.y, p2.y + q2.x};
  }
#pragma unroll
  for (int j = 0; j < R; ++j) { V2 w; w.x = o[j].x; w.y = o[j].y; yout[j * os] = w; }
}

template <typename T>
__global__ void __launch_bounds__(1024)
fft_mixed_k(GenArgs a, const typename vec2<T>::type *in, typename vec2<T>::type *out,
            const typename vec2<T>::type *twf) {
  using V2 = typename vec2<T>::type;
  extern __shared__ __align__(16) unsigned char smem[];
  const int N = a.n, C = a.cols;
  V2 *buf0 = reinterpret_cast<V2 *>(smem);
  V2 *buf1 = buf0 + (size_t)C * N;
  V2 *twl = buf1 + (size_t)C * N;
  const V2 *tw = a.tw_in_lds ? twl : twf;
  const int tid = threadIdx.x, NT = blockDim.x;
  const unsigned bid = panel_of_block(blockIdx.x, a.xcd_lim, a.xcd_gshift);
  const int cp = bid % (unsigned)a.ncp;
  const unsigned rest = bid / (unsigned)a.ncp;
  const int b1 = rest % (unsigned)a.nb1;
  const int b2 = rest / (unsigned)a.nb1;
  const int c0 = cp * C;
  const int nc = (a.ncols - c0 < C) ? a.ncols - c0 : C;  // valid columns of this panel
  const long long ibase = (long long)b1 * a.in_b1 + (long long)b2 * a.in_b2;
  const long long obase = (long long)b1 * a.out_b1 + (long long)b2 * a.out_b2;

  if (a.tw_in_lds)
    for (int i = tid; i < N; i += NT) twl[i] = twf[i];
  const float invN = 1.0f / (float)N, invnc = 1.0f / (float)nc;
  for (int i = tid; i < nc * N; i += NT) {
    int c, n;
    if (a.in_contig) { c = fdiv(i, N, invN); n = i - c * N; } else { n = fdiv(i, nc, invnc); c = i - n * nc; }
    const V2 *src = in + ibase + (long long)(c0 + c) * a.in_col;
    V2 x;
    if (a.real_in) { x.x = reinterpret_cast<const T *>(src)[n]; x.y = 0; }
    else x = src[split_off(n, a.in_split, a.in_nfloor, a.in_blk, a.in_axis, a.in_tab)];
    if (a.conj) x.y = -x.y;
    buf0[c * N + n] = x;
  }
  __syncthreads();

  V2 *x = buf0, *y = buf1;
  int Ns = 1;
  for (int s = 0; s < a.nfac; ++s) {
    const int r = a.fac[s];
    const int nq = N / r;        // butterflies per line
    const int M = N / (Ns * r);  // twiddle step
    if (r <= 5) {
      // whole radix-2/3/4/5 butterflies: r loads, r-1 table twiddles, constant-coefficient DFT, r stores
      const float invnq = 1.0f / (float)nq, invNs = 1.0f / (float)Ns;
      for (int i = tid; i < nc * nq; i += NT) {
        const int c = fdiv(i, nq, invnq), q = i - c * nq, k = q - fdiv(q, Ns, invNs) * Ns;
        switch (r) {
          case 2: mixed_butterfly<T, 2>(x + c * N + q, y + c * N + (q - k) * 2 + k, nq, Ns, k * M, tw); break;
          case 3: mixed_butterfly<T, 3>(x + c * N + q, y + c * N + (q - k) * 3 + k, nq, Ns, k * M, tw); break;
          case 4: mixed_butterfly<T, 4>(x + c * N + q, y + c * N + (q - k) * 4 + k, nq, Ns, k * M, tw); break;
          default: mixed_butterfly<T, 5>(x + c * N + q, y + c * N + (q - k) * 5 + k, nq, Ns, k * M, tw); break;
        }
      }
    } else {
      // any other (prime) radix: one output per work item, r multiply-adds
      for (int i = tid; i < nc * N; i += NT) {
        const int c = i / N;
        const int o = i - c * N;   // (q, j) of this output
        const int q = o % nq, j = o / nq;
        const int k = q % Ns;
        const V2 *xc = x + c * N + q;
        T sr = 0, si = 0;
        int e1 = 0;                // k * t * M       (< N)
        int jt = 0;                // (j * t) mod r
        const int kM = k * M;
        for (int t = 0; t < r; ++t) {
          int e = e1 + jt * nq;    // + (N/r) * ((j t) mod r)
          if (e >= N) e -= N;
          const V2 w = tw[e];
          const V2 v = xc[t * nq];
          sr += v.x * w.x - v.y * w.y;
          si += v.x * w.y + v.y * w.x;
          e1 += kM;
          jt += j;
          if (jt >= r) jt -= r;
        }
        V2 res; res.x = sr; res.y = si;
        y[c * N + (q - k) * r + k + j * Ns] = res;
      }
    }
    __syncthreads();
    V2 *tmp = x; x = y; y = tmp;
    Ns *= r;
  }

  const int kend = a.real_in ? N / 2 + 1 : N;
  const float invkend = 1.0f / (float)kend;
  for (int i = tid; i < nc * kend; i += NT) {
    int c, k;
    if (a.out_contig) { c = fdiv(i, kend, invkend); k = i - c * kend; } else { k = fdiv(i, nc, invnc); c = i - k * nc; }
    V2 v = x[c * N + k];
    V2 w;
    w.x = v.x * (T)a.scale;
    w.y = (a.conj ? -v.y : v.y) * (T)a.scale;
    V2 *dst = out + obase + (long long)(c0 + c) * a.out_col;
    dst[split_off(k, a.out_split, a.out_nfloor, a.out_blk, a.out_axis, a.out_tab)] = w;
  }
}

// ---------------------------------------------------------------------------
// strided 3-D copy (permutation); tile-transposed through LDS when the unit
// strides of input and output sit on different dimensions.
// ---------------------------------------------------------------------------
template <typename V2>
__global__ void __launch_bounds__(256)
copy3d_k(const V2 *in, V2 *out, int n0, int n1, int n2, long long is0, long long is1, long long is2,
         long long os0, long long os1, long long os2) {
  long long total = (long long)n0 * n1 * n2;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    int i2 = (int)(i % n2);
    long long r = i / n2;
    int i1 = (int)(r % n1);
    int i0 = (int)(r / n1);
    out[i0 * os0 + i1 * os1 + i2 * os2] = in[i0 * is0 + i1 * is1 + i2 * is2];
  }
}

__device__ __forceinline__ double hash_val(int x, int y, int z, int c) {
  unsigned h = (unsigned)x * 73856093u ^ (unsigned)y * 19349663u ^ (unsigned)z * 83492791u ^
               (unsigned)c * 2654435761u;
  h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15;
  return (double)(h & 0xffffffu) / 8388608.0 - 1.0;
}

template <typename V2>
__global__ void __launch_bounds__(256)
fill_k(V2 *buf, int kind, int n0, int n1, int n2, int s0, int s1, int s2,
       long long st0, long long st1, long long st2) {
  long long total = (long long)n0 * n1 * n2;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    int i2 = (int)(i % n2);
    long long r = i / n2;
    int i1 = (int)(r % n1);
    int i0 = (int)(r / n1);
    V2 v;
    if (kind == 0) {  // run-fft.c:56-57 ramp
      v.x = (decltype(v.x))((i2 + s2) + 10 * (i1 + s1) + 100 * (i0 + s0));
      v.y = 0;
    } else {
      v.x = (decltype(v.x))hash_val(i0 + s0, i1 + s1, i2 + s2, 0);
      v.y = (decltype(v.y))hash_val(i0 + s0, i1 + s1, i2 + s2, 1);
    }
    buf[i0 * st0 + i1 * st1 + i2 * st2] = v;
  }
}

template <typename T>
__global__ void __launch_bounds__(256)
fill_real_k(T *buf, int kind, int n0, int n1, int n2, int s0, int s1, int s2, long long st0, long long st1, long long st2) {
  long long total = (long long)n0 * n1 * n2;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    int i2 = (int)(i % n2);
    long long r = i / n2;
    int i1 = (int)(r % n1), i0 = (int)(r / n1);
    buf[i0 * st0 + i1 * st1 + i2 * st2] = kind == 0 ? (T)((i2 + s2) + 10 * (i1 + s1) + 100 * (i0 + s0))
                                                    : (T)hash_val(i0 + s0, i1 + s1, i2 + s2, 0);
  }
}

// ---------------------------------------------------------------------------
// host side: kernel registry lookup, twiddle tables, launchers
// ---------------------------------------------------------------------------
std::once_flag g_reg_once;
void build_registry() {
  registry().reserve(8192);  // plan-time instances are appended later: no reallocation under a concurrent lookup
#ifdef OFFT_DEV_REGISTRY  /* the static-sweep tool (tools/sweep_mixed.py) builds the library with ONLY its candidate kernels, in a reg_dev.hip it generates */
  reg_dev();
#else
  reg_pow2_f64();
  reg_pow2_f64_1024();
  reg_pow2_f64_anysplit();
  reg_pow2_f32();
  reg_pow2_f32_big();
  reg_pow2_f32_anysplit();
  reg_pow2_f32_pair();
  reg_pow2_tw4();
  reg_mixed_f64_a();
  reg_mixed_f64_b();
  reg_mixed_f64_c();
  reg_mixed_f64_d();
  reg_mixed_f64_e();
  reg_mixed_f32_a();
  reg_mixed_f32_b();
  reg_bluestein_all();
#endif
}

// (n, precision, flavour) -> registry entries, rebuilt when the registry has grown (plan-time instances): a launch must
// not scan several hundred variants
std::mutex g_idx_mu;
std::unordered_map<unsigned long long, std::vector<int>> g_idx;
size_t g_idx_size = 0;
unsigned long long variant_key(int n, int prec, bool inc, bool outc, bool r2c, bool keep = false, bool tw4 = false) {
  return ((unsigned long long)n << 8) | (tw4 ? 64u : 0u) | (keep ? 32u : 0u) | ((unsigned long long)prec << 3) | (inc ? 4u : 0u) | (outc ? 2u : 0u) | (r2c ? 1u : 0u);
}

Variant *find_variant(int n, int prec, bool inc, bool outc, int id, bool r2c = false, bool keep = false, bool tw4 = false) {
  std::call_once(g_reg_once, build_registry);
  std::lock_guard<std::mutex> lk(g_idx_mu);
  auto &reg = registry();
  if (g_idx_size != reg.size()) {
    g_idx.clear();
    for (size_t i = 0; i < reg.size(); ++i)
      g_idx[variant_key(reg[i].n, reg[i].prec, reg[i].inc, reg[i].outc, reg[i].r2c, reg[i].keep, reg[i].tw4)].push_back((int)i);
    g_idx_size = reg.size();
  }
  auto it = g_idx.find(variant_key(n, prec, inc, outc, r2c, keep, tw4));
  if (it == g_idx.end()) return nullptr;
  Variant *def = nullptr;
  for (int i : it->second) {
    Variant &v = reg[i];
    if (v.id == id) return &v;
    if (v.is_default) def = &v;
  }
  return def;
}

// ---------------------------------------------------------------------------
// Plan-time specialisation.  A length without a precompiled panel kernel whose prime factors are <= 31 gets
// its own fft_panelx_k instances at offt_hipk_prepare(): the device part of offt_panel.hpp travels inside
// the library as a string, a shape (radix order, threads per line, panel width) is picked with the scoring of
// tools/sweep_mixed.py, hipRTC compiles the four (in_contig, out_contig) flavours and the two real-input ones (2-4 s in all) and the
// code object is loaded as a module.  OFFT_RTC=0 turns it off; any failure leaves the any-length kernel in
// charge and says why on stderr once.
// ---------------------------------------------------------------------------
struct RtcApi {
  void *h = nullptr;
  hiprtcResult (*CreateProgram)(hiprtcProgram *, const char *, const char *, int, const char **, const char **) = nullptr;
  hiprtcResult (*AddNameExpression)(hiprtcProgram, const char *) = nullptr;
  hiprtcResult (*CompileProgram)(hiprtcProgram, int, const char **) = nullptr;
  hiprtcResult (*GetProgramLogSize)(hiprtcProgram, size_t *) = nullptr;
  hiprtcResult (*GetProgramLog)(hiprtcProgram, char *) = nullptr;
  hiprtcResult (*GetLoweredName)(hiprtcProgram, const char *, const char **) = nullptr;
  hiprtcResult (*GetCodeSize)(hiprtcProgram, size_t *) = nullptr;
  hiprtcResult (*GetCode)(hiprtcProgram, char *) = nullptr;
  hiprtcResult (*DestroyProgram)(hiprtcProgram *) = nullptr;
};
RtcApi g_rtc;
std::mutex g_rtc_mu;
int g_rtc_state = 0;  // 0 untried, 1 ready, -1 unavailable

bool rtc_load() {
  if (g_rtc_state) return g_rtc_state > 0;
  g_rtc_state = -1;
  // the hipRTC that belongs to the HIP runtime this process already uses (PyTorch bundles both), else the system one
  std::vector<std::string> cand;
  Dl_info di;
  if (dladdr((void *)&hipModuleLoadData, &di) && di.dli_fname) {
    std::string dir(di.dli_fname);
    const size_t sl = dir.rfind('/');
    if (sl != std::string::npos) cand.push_back(dir.substr(0, sl + 1) + "libhiprtc.so");
  }
  if (getenv("OFFT_HIPRTC_LIB")) cand.insert(cand.begin(), getenv("OFFT_HIPRTC_LIB"));
  cand.push_back("libhiprtc.so");
  cand.push_back("/opt/rocm/lib/libhiprtc.so");
  for (auto &c : cand) {
    g_rtc.h = dlopen(c.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (g_rtc.h) break;
  }
  if (!g_rtc.h) return false;
#define RTC_SYM(name) \
  g_rtc.name = (decltype(g_rtc.name))dlsym(g_rtc.h, "hiprtc" #name); \
  if (!g_rtc.name) return false;
  RTC_SYM(CreateProgram) RTC_SYM(AddNameExpression) RTC_SYM(CompileProgram) RTC_SYM(GetProgramLogSize) RTC_SYM(GetProgramLog)
  RTC_SYM(GetLoweredName) RTC_SYM(GetCodeSize) RTC_SYM(GetCode) RTC_SYM(DestroyProgram)
#undef RTC_SYM
  g_rtc_state = 1;
  return true;
}

struct Shape { int tpl, r0, r1, r2, cols; double score; int emax; size_t lds; };

// lengths the panel kernels can factor into register radices <= 32: every prime factor <= 31
bool smooth13(int n) {
  for (int p : {2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31}) while (n % p == 0) n /= p;
  return n == 1;
}
int cdiv_i(int a, int b) { return (a + b - 1) / b; }

// the candidate scoring of tools/sweep_mixed.py (live butterfly slots, waves per CU, panel width, radix size)
bool choose_shapes(int N, int prec, int want, std::vector<Shape> *out) {
  const int esz = prec == OFFT_PREC_F64 ? 8 : 4, emax_cap = 32, emax_soft = prec == OFFT_PREC_F64 ? 24 : 32;
  std::vector<int> rad;
  for (int r = 2; r <= 32; ++r) if (smooth13(r)) rad.push_back(r);
  std::vector<Shape> all;
  auto consider = [&](int r0, int r1, int r2) {
    const int rs[3] = {r0, r1, r2};
    const int nst = r2 > 1 ? 3 : (r1 > 1 ? 2 : 1);
    for (int cols : {8, 16, 4}) {
      for (int tpl = 1; tpl <= 256; ++tpl) {
        const int nt = tpl * cols;
        if (nt > 1024 || nt < 64) continue;
        int emax = 0; double eff = 0;
        for (int s = 0; s < nst; ++s) {
          const int nb = cdiv_i(N / rs[s], tpl);
          emax = nb * rs[s] > emax ? nb * rs[s] : emax;
          eff += (double)N / ((double)tpl * nb * rs[s]);
        }
        eff /= nst;
        if ((nt % 64) && !(eff == 1.0 && nt % 16 == 0)) continue;
        const bool swz = r0 % 16 == 0 && N % 16 == 0;
        const int paddiv = (r0 % 2 == 0 && !swz) ? r0 : 0;
        const int npad = paddiv ? N + N / paddiv : N;
        const int lstride = (npad + 31) / 32 * 32 + 4;
        const size_t ex = nst > 1 ? (size_t)cols * lstride * esz : 0;
        const size_t qt = N % 4 == 0 ? N / 4 + 1 : N;
        const size_t lds = nst > 1 ? (ex + 15) / 16 * 16 + qt * 2 * esz : 0;
        if (emax > emax_cap || emax < 6 || lds > 160 * 1024 || eff < 0.74) continue;
        size_t wg = lds ? (160 * 1024) / lds : 4;
        wg = wg < 1 ? 1 : (wg > 4 ? 4 : wg);
        const int waves = (int)wg * cdiv_i(nt, 64);
        if (waves < 4) continue;
        double sc = eff * std::sqrt(waves >= 8 ? 1.0 : waves / 8.0) * (cols * esz * 2 >= 128 ? 1.0 : 0.8);
        sc *= (emax >= 12 ? 1.0 : 0.85) * (nst > 2 ? 0.97 : 1.0) * (emax > emax_soft ? 0.8 : 1.0);  // > 24 f64 points per thread: 2 waves/SIMD
        const int rmax = r0 > r1 ? (r0 > r2 ? r0 : r2) : (r1 > r2 ? r1 : r2);
        sc *= rmax > 16 ? 0.92 : 1.0;
        sc *= (double)nt / (64.0 * cdiv_i(nt, 64));
        all.push_back(Shape{tpl, r0, r1, r2, cols, sc, emax, lds});
      }
    }
  };
  for (int r0 : rad) {
    if (N % r0) continue;
    const int m = N / r0;
    if (m == 1) consider(r0, 1, 1);
    for (int r1 : rad) {
      if (m % r1) continue;
      const int r2 = m / r1;
      if (r2 == 1) consider(r0, r1, 1);
      else if (r2 <= 32 && smooth13(r2)) consider(r0, r1, r2);
    }
  }
  std::stable_sort(all.begin(), all.end(), [](const Shape &a, const Shape &b) { return a.score > b.score; });
  // the best shape, then (for a plan-time sweep, OFFT_RTC_SHAPES > 1) the next ones that differ in radix set, thread count
  // or panel width
  auto key = [](const Shape &s) { int r[3] = {s.r0, s.r1, s.r2}; std::sort(r, r + 3); return r[0] * 10000 + r[1] * 100 + r[2]; };
  for (const Shape &c : all) {
    if ((int)out->size() >= want) break;
    bool dup = false;
    int same_set = 0;
    for (const Shape &o : *out) {
      if (key(o) == key(c)) {
        ++same_set;
        if (o.cols == c.cols && std::abs(o.tpl - c.tpl) < 16 && !(o.r0 != c.r0 && o.tpl == c.tpl)) dup = true;
      }
    }
    if (!dup && same_set < 2) out->push_back(c);
  }
  return !out->empty();
}

bool rtc_enabled() {
  static const bool on = !(getenv("OFFT_RTC") && atoi(getenv("OFFT_RTC")) == 0);
  return on;
}

// compile and register fft_panelx_k<T, n, shape> for the four flavours; 0 on success
int rtc_build(int n, int prec) {
  std::lock_guard<std::mutex> lk(g_rtc_mu);
  if (find_variant(n, prec, true, true, -1)) return 0;  // somebody was faster
  static bool warned = false;
  auto fail = [&](const char *what, const std::string &detail) {
    if (!warned) fprintf(stderr, "offt(hip): no plan-time kernel for n=%d (%s%s%s); using the any-length kernel\n", n, what,
                         detail.empty() ? "" : ": ", detail.c_str());
    warned = true;
    return -1;
  };
  if (!rtc_load()) return fail("hipRTC library not found", "");
  static const int nshapes = getenv("OFFT_RTC_SHAPES") ? atoi(getenv("OFFT_RTC_SHAPES")) : 1;
  std::vector<Shape> shapes;
  if (!choose_shapes(n, prec, nshapes < 1 ? 1 : (nshapes > 8 ? 8 : nshapes), &shapes)) return fail("no panel shape fits", "");
  const char *T = prec == OFFT_PREC_F64 ? "double" : "float";
  std::string src;
  for (const char *p : k_rtc_source_pieces) src += p;
  hiprtcProgram prog;
  if (g_rtc.CreateProgram(&prog, src.c_str(), "offt_panel_rtc.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS)
    return fail("hiprtcCreateProgram failed", "");
  // four (in_contig, out_contig) flavours per shape, plus the two real-input z-pass flavours of the default shape
  const bool flav[6][3] = {{true, true, false}, {false, false, false}, {true, false, false}, {false, true, false},
                           {true, true, true}, {true, false, true}};
  struct Inst { int shape, f; };
  std::vector<Inst> inst;
  std::vector<std::string> expr;
  for (size_t k = 0; k < shapes.size(); ++k)
    for (int f = 0; f < (k == 0 ? 6 : 4); ++f) {
      const Shape &sh = shapes[k];
      char b[256];
      snprintf(b, sizeof b, "offtk::fft_panelx_k<%s, %d, %d, %d, %d, %d, %d, %s, %s, true, %s>", T, n, sh.tpl, sh.r0, sh.r1, sh.r2,
               sh.cols, flav[f][0] ? "true" : "false", flav[f][1] ? "true" : "false", flav[f][2] ? "true" : "false");
      expr.push_back(b);
      inst.push_back(Inst{(int)k, f});
      g_rtc.AddNameExpression(prog, expr.back().c_str());
    }
  const char *opts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17"};
  if (g_rtc.CompileProgram(prog, 3, opts) != HIPRTC_SUCCESS) {
    size_t ls = 0;
    g_rtc.GetProgramLogSize(prog, &ls);
    std::string log(ls, '\0');
    if (ls) g_rtc.GetProgramLog(prog, &log[0]);
    g_rtc.DestroyProgram(&prog);
    return fail("hipRTC compile failed", log.substr(0, 600));
  }
  size_t cs = 0;
  g_rtc.GetCodeSize(prog, &cs);
  std::vector<char> code(cs);
  g_rtc.GetCode(prog, code.data());
  hipModule_t mod;
  if (hipModuleLoadData(&mod, code.data()) != hipSuccess) {
    (void)hipGetLastError();
    g_rtc.DestroyProgram(&prog);
    return fail("hipModuleLoadData failed", "");
  }
  std::vector<hipFunction_t> fn(expr.size());
  for (size_t e = 0; e < expr.size(); ++e) {
    const char *low = nullptr;
    if (g_rtc.GetLoweredName(prog, expr[e].c_str(), &low) != HIPRTC_SUCCESS || hipModuleGetFunction(&fn[e], mod, low) != hipSuccess) {
      (void)hipGetLastError();
      g_rtc.DestroyProgram(&prog);
      return fail("kernel symbol not found in the compiled module", "");
    }
  }
  g_rtc.DestroyProgram(&prog);
  // variant 0 = the best-scored shape = the default; the others are there for the static sweep (offt_hip_set_variant, -l N)
  std::lock_guard<std::mutex> lk_idx(g_idx_mu);  // lookups index the registry under this lock
  for (size_t e = 0; e < inst.size(); ++e) {
    const Shape &sh = shapes[inst[e].shape];
    const bool *fl = flav[inst[e].f];
    char nm[200];
    snprintf(nm, sizeof nm, "%s N=%d mixed radix=%dx%dx%d threads/line=%d (<=%d elems/thread) cols=%d split-re/im lds=%zuB [plan-time hipRTC]",
             prec ? "f32" : "f64", n, sh.r0, sh.r1, sh.r2, sh.tpl, sh.emax, sh.cols, sh.lds);
    registry().push_back(Variant{n, prec, fl[0], fl[1], inst[e].shape, inst[e].shape == 0, fl[2], sh.cols, sh.tpl * sh.cols, sh.emax, sh.lds,
                                 nullptr, nm, false, true, n % 4 != 0, (void *)fn[e]});
  }
  return 0;
}

struct Tables {
  void *quarter = nullptr;  // w^r, r = 0..N/4      (fast path)
  void *full = nullptr;     // w^m, m = 0..N-1      (generic path)
};
std::mutex g_tab_mu;
std::map<std::pair<int, int>, Tables> g_tabs;  // (n, prec) per current device is enough: one device per process

template <typename T>
int make_tables(int n, Tables &tb) {
  using V2 = typename vec2<T>::type;
  const long double pi = 3.14159265358979323846264338327950288419716939937510L;
  auto tw = [&](long long m) {
    // exact octant reduction, then cosl/sinl on [0, pi/4]
    long long mm = ((m % n) + n) % n;
    // angle = 2 pi mm / n ; reduce by octants using exact integer arithmetic on 8*mm/n
    long long oct = (8 * mm) / n;
    long long rem = 8 * mm - oct * n;  // angle = (oct + rem/n) * pi/4
    long double c, s;
    long double a = (long double)rem / (long double)n * (pi / 4);
    switch (oct & 7) {
      case 0: c = cosl(a); s = sinl(a); break;
      case 1: { long double b = pi / 4 - a; c = sinl(b); s = cosl(b); if (rem == 0) { c = s = sqrtl(0.5L); } break; }
      case 2: c = -sinl(a); s = cosl(a); break;
      case 3: { long double b = pi / 4 - a; c = -cosl(b); s = sinl(b); if (rem == 0) { c = -sqrtl(0.5L); s = sqrtl(0.5L); } break; }
      case 4: c = -cosl(a); s = -sinl(a); break;
      case 5: { long double b = pi / 4 - a; c = -sinl(b); s = -cosl(b); if (rem == 0) { c = s = -sqrtl(0.5L); } break; }
      case 6: c = sinl(a); s = -cosl(a); break;
      default: { long double b = pi / 4 - a; c = cosl(b); s = -sinl(b); if (rem == 0) { c = sqrtl(0.5L); s = -sqrtl(0.5L); } break; }
    }
    V2 w;
    w.x = (T)c;
    w.y = (T)(-s);  // forward: exp(-i theta)
    return w;
  };
  std::vector<V2> q(n / 4 + 1), f(n);
  for (int r = 0; r <= n / 4; ++r) q[r] = tw(r);
  for (int m = 0; m < n; ++m) f[m] = tw(m);
  HIPK_CHECK(hipMalloc(&tb.quarter, q.size() * sizeof(V2)));
  HIPK_CHECK(hipMemcpy(tb.quarter, q.data(), q.size() * sizeof(V2), hipMemcpyHostToDevice));
  HIPK_CHECK(hipMalloc(&tb.full, f.size() * sizeof(V2)));
  HIPK_CHECK(hipMemcpy(tb.full, f.data(), f.size() * sizeof(V2), hipMemcpyHostToDevice));
  return 0;
}

int get_tables(int n, int prec, Tables &out, bool create) {
  std::lock_guard<std::mutex> lk(g_tab_mu);
  auto key = std::make_pair(n, prec);
  auto it = g_tabs.find(key);
  if (it != g_tabs.end()) { out = it->second; return 0; }
  if (!create) {
    snprintf(g_err, sizeof g_err, "offt_hipk: no twiddle tables for n=%d (call offt_hipk_prepare at plan time)", n);
    return -1;
  }
  Tables tb;
  int rc = prec == OFFT_PREC_F64 ? make_tables<double>(n, tb) : make_tables<float>(n, tb);
  if (rc) return rc;
  g_tabs[key] = tb;
  out = tb;
  return 0;
}

bool is_pow2(int n) { return n > 0 && (n & (n - 1)) == 0; }

// ---------------------------------------------------------------------------
// Bluestein (offt_bluestein.hpp): per (n, precision) the chirp a[n] = exp(-i pi n^2 / n_len) and the spectrum
// B^ = FFT_M(b wrapped) / M, b[m] = exp(+i pi m^2 / n_len), M = the power of two >= 2 n_len - 1.
// ---------------------------------------------------------------------------
struct BlueTab { void *chirp = nullptr, *bhat = nullptr; int m = 0; };
std::mutex g_blue_mu;
std::map<std::pair<int, int>, BlueTab> g_blue;

bool blue_enabled() {
  static const bool on = !(getenv("OFFT_BLUESTEIN") && atoi(getenv("OFFT_BLUESTEIN")) == 0);
  return on;
}
int blue_m(int n) {
  int m = 256;
  while (m < 2 * n - 1) m <<= 1;
  return m;
}
BlueVariant *find_blue(int m, int prec, bool inc, bool outc) {
  std::call_once(g_reg_once, build_registry);
  for (auto &b : blue_registry())
    if (b.m == m && b.prec == prec && b.inc == inc && b.outc == outc) return &b;
  return nullptr;
}
bool blue_lookup(int n, int prec, BlueTab *out) {
  std::lock_guard<std::mutex> lk(g_blue_mu);
  auto it = g_blue.find(std::make_pair(n, prec));
  if (it == g_blue.end()) return false;
  *out = it->second;
  return true;
}

// convolution length of a Bluestein line through scratch (long_pass): not the next power of two -- up to twice 2n - 1 -- but
// the shortest M >= 2n - 1 of the form (64 | 32) x L, L a precompiled register length: a fused four-step line (10007 points:
// M = 20480 instead of 32768).  0: none found, use the power of two
int blue_m_long(int n, int prec) {
  const long long need = 2LL * n - 1;
  const int n1 = prec == OFFT_PREC_F64 ? 64 : 32;
  if (!find_variant(n1, prec, false, false, -1, false, false, true)) return 0;
  long long best = 0;
  for (auto &v : registry()) {
    if (v.prec != prec || !v.inc || !v.outc || v.r2c || v.tw4 || v.id != 0 || v.n < 64 || v.n > 4096) continue;
    const long long m = (long long)n1 * v.n;
    if (m >= need && m < (1LL << 24) && (!best || m < best)) best = m;
  }
  return best && best < blue_m(n) ? (int)best : 0;
}

template <typename T>
int blue_build(int n, int prec, BlueTab &tb, int m_override = 0) {
  using V2 = typename vec2<T>::type;
  const int m = m_override ? m_override : blue_m(n);
  const long double pi = 3.14159265358979323846264338327950288419716939937510L;
  // angle pi k^2 / n with k^2 reduced mod 2n in integers: exact argument reduction
  auto ang = [&](long long k) { return pi * (long double)((k * k) % (2LL * n)) / (long double)n; };
  std::vector<V2> a(n), b(m);
  for (int k = 0; k < n; ++k) { a[k].x = (T)cosl(ang(k)); a[k].y = (T)(-sinl(ang(k))); }
  for (int k = 0; k < m; ++k) { b[k].x = 0; b[k].y = 0; }
  for (int k = 0; k < n; ++k) {
    V2 w; w.x = (T)cosl(ang(k)); w.y = (T)sinl(ang(k));
    b[k] = w;
    if (k) b[m - k] = w;
  }
  HIPK_CHECK(hipMalloc(&tb.chirp, a.size() * sizeof(V2)));
  HIPK_CHECK(hipMemcpy(tb.chirp, a.data(), a.size() * sizeof(V2), hipMemcpyHostToDevice));
  HIPK_CHECK(hipMalloc(&tb.bhat, b.size() * sizeof(V2)));
  HIPK_CHECK(hipMemcpy(tb.bhat, b.data(), b.size() * sizeof(V2), hipMemcpyHostToDevice));
  tb.m = m;
  // B^ = FFT_M(b) / M with the M-point panel kernel itself, in place, one line
  if (offt_hipk_prepare(m, prec)) return -1;
  offt_pass_desc d;
  memset(&d, 0, sizeof d);
  d.n = m; d.precision = prec; d.direction = -1; d.ncols = 1; d.nb1 = d.nb2 = 1;
  d.in_axis_stride = d.out_axis_stride = 1; d.in_col_stride = d.out_col_stride = m;
  d.in_contig = d.out_contig = 1; d.variant = -1; d.scale = 1.0 / (double)m;
  if (offt_hipk_fft_pass(&d, tb.bhat, tb.bhat, nullptr)) return -1;
  HIPK_CHECK(hipStreamSynchronize(nullptr));
  return 0;
}

// XCD-aware panel order (panel_of_block): runs of G = 32 neighbouring panels per XCD by default,
// OFFT_XCD_REMAP=0 turns it off, OFFT_XCD_REMAP=<power of two> sets G
void xcd_order(long long nblk, unsigned *lim, unsigned *gshift) {
  static const int env = getenv("OFFT_XCD_REMAP") ? atoi(getenv("OFFT_XCD_REMAP")) : 32;
  unsigned gs = 0;
  while (env > 1 && (2 << gs) <= env) ++gs;
  *gshift = gs;
  *lim = env > 0 ? (unsigned)((nblk >> (gs + 3)) << (gs + 3)) : 0u;
}

// Column-pair kernels (T = f32x2, offt_panel.hpp) take a descriptor with an even column count whose strided sides put the
// two columns of a pair into 16 contiguous, 16-B aligned bytes: unit column stride, every other stride even (the base
// pointers are checked at launch).  A contiguous side moves the two columns separately and needs nothing.
bool pair_ok(const offt_pass_desc *d) {
  if (d->precision != OFFT_PREC_F32 || d->real_input || (d->ncols & 1)) return false;
  auto even = [](long long x) { return (x & 1) == 0; };
  // (a per-block base table replaces the block stride; its entries are the host's to keep even: offt_host.c elem_delta)
  if (!d->in_contig && !(d->in_col_stride == 1 && even(d->in_axis_stride) && even(d->in_b1_stride) && even(d->in_b2_stride) &&
                         ((d->in_split && d->in_block_tab) || even(d->in_block_stride))))
    return false;
  if (!d->out_contig && !(d->out_col_stride == 1 && even(d->out_axis_stride) && even(d->out_b1_stride) && even(d->out_b2_stride) &&
                          ((d->out_split && d->out_block_tab) || even(d->out_block_stride))))
    return false;
  return true;
}

// the panel-kernel variant that will run this descriptor, or nullptr (-> any-length kernel)
Variant *pick_variant0(const offt_pass_desc *d, bool allow_pair) {
  if (d->real_input && (!d->in_contig || d->in_axis_stride != 1 || d->in_split || d->direction > 0)) return nullptr;
  const bool inc = d->in_contig != 0, outc = d->out_contig != 0, r2c = d->real_input != 0;
  const bool uneven = d->in_split_nfloor > 0 || d->out_split_nfloor > 0;
  const bool odd_split = (d->in_split && !is_pow2(d->in_split)) || (d->out_split && !is_pow2(d->out_split));
  static const bool pairs_on = !(getenv("OFFT_F32_PAIRS") && atoi(getenv("OFFT_F32_PAIRS")) == 0);
  int want = d->variant;
  if (want >= VARIANT_PAIR0 || (want < 0 && pairs_on && !d->no_pairs)) {
    if (allow_pair && !uneven && !odd_split && pair_ok(d)) {
      Variant *p = find_variant(d->n, OFFT_PREC_F32_PAIR, inc, outc, want < 0 ? -1 : want - VARIANT_PAIR0);
      if (p && (want >= 0 ? p->id == want - VARIANT_PAIR0 : p->is_default)) return p;
    }
    if (want >= VARIANT_PAIR0) want = -1;
  }
  Variant *v = find_variant(d->n, d->precision, inc, outc, r2c ? -1 : want, r2c);
  if (!v) return nullptr;
  if (v->mixed || !(uneven || odd_split)) return v;
  // fft_panel_k addresses per-peer blocks with shifts: other block lengths go to the length's fft_panelx_k instance
  Variant *w = find_variant(d->n, d->precision, inc, outc, VARIANT_ANYSPLIT, r2c);
  return (w && w->id == VARIANT_ANYSPLIT) ? w : nullptr;
}

// ... and its cache-keeping twin when the descriptor asks for one (out_keep) and one is registered
Variant *pick_variant(const offt_pass_desc *d, bool allow_pair = true) {
  if (d->tw4) {  // first sub-pass of a four-step line: the strided / strided kernel with the twiddles on its stores, or nothing
    if (d->in_contig || d->out_contig || d->real_input || d->in_split_nfloor || d->out_split_nfloor) return nullptr;
    if ((d->in_split && !is_pow2(d->in_split)) || (d->out_split && !is_pow2(d->out_split))) return nullptr;
    return find_variant(d->n, d->precision, false, false, -1, false, false, true);
  }
  Variant *v = pick_variant0(d, allow_pair);
  static const bool keep_on = !(getenv("OFFT_KEEP_STORES") && atoi(getenv("OFFT_KEEP_STORES")) == 0);
  if (v && d->out_keep && keep_on && !v->mixed && !v->r2c) {
    Variant *k = find_variant(v->n, v->prec, v->inc, v->outc, v->id, false, true);
    if (k && k->keep && k->id == v->id) return k;
  }
  return v;
}

bool fast_ok(const offt_pass_desc *d) { return pick_variant(d) != nullptr; }

// 1 if the descriptor, with out_keep set, runs on a kernel whose stores stay cached (a KEEP twin exists for its shape)
extern "C" int offt_hipk_keeps_output(const offt_pass_desc *d) {
  offt_pass_desc k = *d;
  k.out_keep = 1;
  const Variant *v = pick_variant(&k);
  return v && v->keep;
}

int log2i(int n) { int l = 0; while ((1 << l) < n) ++l; return l; }

// ---------------------------------------------------------------------------
// Four-step decomposition: lines no single kernel takes.
//
// FFTW plans any length (offt-compute.c:335-341, 416-425); the panel kernels end at 4096 points (8192 with one column per
// workgroup), the any-length kernel where two images of a line fill the LDS (5120 double / 10240 single points).  A longer
// line of n = n1 n2 points is transformed as  X[k1 + n1 k2] = sum_{j2} w_n^(j2 k1) [ sum_{j1} x[j1 n2 + j2] w_n1^(j1 k1) ] w_n2^(j2 k2):
//   A  n2 x (n1-point FFTs over j1)   -- any kernel of the library, reading the caller's layout (splits, block tables)
//   T  times w_n^(j2 k1)              -- the exact full-wave table of n; j2 k1 < n, so no reduction
//   C  n1 x (n2-point FFTs over j2)   -- any kernel, writing the caller's layout
// through a dense scratch S[column][k1][j2] (j2 contiguous).  A side whose unit-stride dimension is the axis takes the
// j2 / k1 index as the sub-pass's columns; a side whose columns are unit-stride keeps them as columns -- every access
// stays a 128-B segment.  For a strided-in pass step A writes S'[k1][j2][column] and T transposes on the way.
// Three sweeps over the data instead of one: a correctness net with decent bandwidth, not a tuned path.
// ---------------------------------------------------------------------------
struct FourStep { int n1 = 0, n2 = 0; void *t4 = nullptr; bool all = false; };  // all: every flavour goes this way (no register kernel for n)  // t4[k1][j2] = w_n^(k1 j2), from the exact full-wave table
std::mutex g_four_mu;
std::map<std::pair<int, int>, FourStep> g_four;
// scratch per caller stream and per NESTING DEPTH of the decomposing paths: a four-step line whose sub-pass goes through scratch
// lines (a real-input line gathered as complex; a factor that runs as a Bluestein convolution), whose power-of-two lines are
// four-step lines again ... every level keeps its own buffers while the levels below it run.  Slots 3 d + {0, 1}: four-step of
// depth d, 3 d + 2: lines through scratch of depth d
enum { SCRATCH_DEPTHS = 4 };
struct Scratch { void *p[3 * SCRATCH_DEPTHS] = {}; size_t bytes[3 * SCRATCH_DEPTHS] = {}; };
thread_local int g_scratch_depth = 0;
struct DepthGuard {
  int d;
  DepthGuard() : d(g_scratch_depth++) {}
  ~DepthGuard() { --g_scratch_depth; }
};
std::map<void *, Scratch> g_four_scratch;  // per stream

bool four_lookup(int n, int prec, FourStep *out) {
  std::lock_guard<std::mutex> lk(g_four_mu);
  auto it = g_four.find(std::make_pair(n, prec));
  if (it == g_four.end()) return false;
  if (out) *out = it->second;
  return true;
}

// a single launch can transform lines of n points: a panel kernel, the Bluestein kernel, or the any-length kernel
bool direct_ok(int n, int prec) {
  if (find_variant(n, prec, true, true, -1)) return true;
  BlueTab bt;
  if (blue_lookup(n, prec, &bt)) return true;
  const size_t esz = prec == OFFT_PREC_F64 ? sizeof(double2) : sizeof(float2);
  return 2 * (size_t)n * esz <= (size_t)160 * 1024;
}

void *four_scratch(void *stream, int which, size_t bytes) {
  std::lock_guard<std::mutex> lk(g_four_mu);
  Scratch &sc = g_four_scratch[stream];
  if (sc.bytes[which] < bytes) {
    if (sc.p[which]) { (void)hipStreamSynchronize((hipStream_t)stream); (void)hipFree(sc.p[which]); }
    sc.p[which] = nullptr; sc.bytes[which] = 0;
    if (hipMalloc(&sc.p[which], bytes) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    sc.bytes[which] = bytes;
  }
  return sc.p[which];
}

// t4[k1][j2] = tw[k1 j2] (k1 j2 < n)
template <typename V2>
__global__ void __launch_bounds__(256) four_table_k(V2 *t4, const V2 *tw, int n, int n2) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) t4[i] = tw[(long long)(i / n2) * (i % n2)];
}
// T, in place: S[q][k1][j2] *= w_n^(+-(k1 j2))
template <typename V2>
__global__ void __launch_bounds__(256) four_twiddle_k(V2 *s, const V2 *tw, long long total, int n1, int n2, int conj) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int j2 = (int)(i % n2), k1 = (int)((i / n2) % n1);
  const V2 w = tw[k1 * j2];
  const V2 x = s[i];
  const auto wy = conj ? -w.y : w.y;
  V2 r;
  r.x = x.x * w.x - x.y * wy;
  r.y = x.x * wy + x.y * w.x;
  s[i] = r;
}
// T for a strided-in pass: S'[b][m][c] (m = k1 n2 + j2 < n, c < cc) -> S[b][c][m] times w_n^(+-(k1 j2)), 32 x 32 tiles through LDS
template <typename V2>
__global__ void __launch_bounds__(256) four_twiddle_t_k(const V2 *sp, V2 *s, const V2 *tw, int n, int cc, int n2, int conj) {
  __shared__ V2 tile[32][33];
  const long long b = blockIdx.z;
  const int m0 = blockIdx.y * 32, c0 = blockIdx.x * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int r = ty; r < 32; r += 8) {
    const int m = m0 + r, c = c0 + tx;
    if (m < n && c < cc) tile[r][tx] = sp[(b * n + m) * cc + c];
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int c = c0 + r, m = m0 + tx;
    if (m < n && c < cc) {
      const V2 x = tile[tx][r];
      const int k1 = m / n2, j2 = m - k1 * n2;
      const V2 w = tw[k1 * j2];
      const auto wy = conj ? -w.y : w.y;
      V2 o;
      o.x = x.x * w.x - x.y * wy;
      o.y = x.x * wy + x.y * w.x;
      s[(b * cc + c) * n + m] = o;
    }
  }
}

int four_pass(const offt_pass_desc *d, const void *in, void *out, void *stream, const FourStep &fs);
// ---------------------------------------------------------------------------
// Lines through a dense scratch: the last net under "FFTW plans any length".
//   * a line no single launch takes and no four-step split fits (a prime beyond the any-length kernel, or a prime factor
//     too large itself) runs as a Bluestein convolution on M-point lines, M the power of two >= 2n - 1:
//       gather x[j] a[j] into U[line][0..M) (zero padded) | FFT_M | times B^ | FFT_M^-1 | scatter a[k] U[line][k], k < n
//     with the library's own M-point path (a panel kernel or the four-step decomposition) on contiguous lines;
//   * a real-input line too long for the r2c kernels is gathered as complex, transformed by the complex path of the same
//     length, and its first n/2 + 1 outputs scattered.
// Seven (three) sweeps over scratch lines of twice the length: a few per cent of the roofline -- it exists so that no
// grid the reference accepts is refused, not to be fast.
// ---------------------------------------------------------------------------
struct LongTab { void *chirp = nullptr, *bhat = nullptr; int m = 0; };
std::mutex g_long_mu;  // (its own lock: blue_build runs an M-point pass while the Bluestein registry's is held, and every pass looks here)
std::map<std::pair<int, int>, LongTab> g_long;
bool long_lookup(int n, int prec, LongTab *out) {
  std::lock_guard<std::mutex> lk(g_long_mu);
  auto it = g_long.find(std::make_pair(n, prec));
  if (it == g_long.end()) return false;
  if (out) *out = it->second;
  return true;
}

// U[l][j] = x_line(line0 + l)[j] (* chirp[j]), j < n; 0 for n <= j < m.  Lines are numbered (b2, b1, column), column fastest
template <typename T>
__global__ void __launch_bounds__(256)
long_gather_k(GenArgs a, const typename vec2<T>::type *in, typename vec2<T>::type *U, const typename vec2<T>::type *chirp, int m,
              long long line0, long long total) {
  using V2 = typename vec2<T>::type;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int j = (int)(i % m);
  const long long L = line0 + i / m;
  const int c = (int)(L % a.ncols);
  const long long r = L / a.ncols;
  const int b1 = (int)(r % a.nb1), b2 = (int)(r / a.nb1);
  V2 x; x.x = 0; x.y = 0;
  if (j < a.n) {
    const V2 *src = in + (long long)b1 * a.in_b1 + (long long)b2 * a.in_b2 + (long long)c * a.in_col;
    if (a.real_in) x.x = reinterpret_cast<const T *>(src)[j];
    else x = src[split_off(j, a.in_split, a.in_nfloor, a.in_blk, a.in_axis, a.in_tab)];
    if (a.conj) x.y = -x.y;
    if (chirp) { const V2 w = chirp[j]; V2 t; t.x = x.x * w.x - x.y * w.y; t.y = x.x * w.y + x.y * w.x; x = t; }
  }
  U[i] = x;
}
// U[l][k] *= bhat[k]
template <typename V2>
__global__ void __launch_bounds__(256) long_mul_k(V2 *U, const V2 *bhat, int m, long long total) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const V2 w = bhat[(int)(i % m)], x = U[i];
  V2 t; t.x = x.x * w.x - x.y * w.y; t.y = x.x * w.y + x.y * w.x;
  U[i] = t;
}
// out_line(line0 + l)[k] = U[l][k] (* chirp[k]) * scale, k < kend
template <typename T>
__global__ void __launch_bounds__(256)
long_scatter_k(GenArgs a, const typename vec2<T>::type *U, typename vec2<T>::type *out, const typename vec2<T>::type *chirp, int m, int kend,
               long long line0, long long total) {
  using V2 = typename vec2<T>::type;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int k = (int)(i % kend);
  const long long l = i / kend, L = line0 + l;
  const int c = (int)(L % a.ncols);
  const long long r = L / a.ncols;
  const int b1 = (int)(r % a.nb1), b2 = (int)(r / a.nb1);
  V2 x = U[l * m + k];
  if (chirp) { const V2 w = chirp[k]; V2 t; t.x = x.x * w.x - x.y * w.y; t.y = x.x * w.y + x.y * w.x; x = t; }
  V2 o;
  o.x = x.x * (T)a.scale;
  o.y = (a.conj ? -x.y : x.y) * (T)a.scale;
  V2 *dst = out + (long long)b1 * a.out_b1 + (long long)b2 * a.out_b2 + (long long)c * a.out_col;
  dst[split_off(k, a.out_split, a.out_nfloor, a.out_blk, a.out_axis, a.out_tab)] = o;
}

int long_pass(const offt_pass_desc *d, const void *in, void *out, void *stream, const LongTab &lt);

}  // namespace

extern "C" {

const char *offt_hipk_last_error(void) { return g_err; }

int offt_hipk_has_fast_path(int n, int precision) {
  return find_variant(n, precision, true, true, -1) != nullptr;
}

/* 1 if lines of n points run as a four-step decomposition (no single launch takes them): complex input only */
int offt_hipk_is_four_step(int n, int precision) {
  const size_t esz = precision == OFFT_PREC_F64 ? sizeof(double2) : sizeof(float2);
  return four_lookup(n, precision, nullptr) && !find_variant(n, precision, true, true, -1) && 2 * (size_t)n * esz > (size_t)160 * 1024;
}

int offt_hipk_variant_count(int n, int precision) {
  std::call_once(g_reg_once, build_registry);
  int c = 0;
  for (auto &v : registry())
    if (v.n == n && v.prec == precision && v.inc && v.outc && !v.r2c && v.id < VARIANT_ANYSPLIT) c = v.id + 1 > c ? v.id + 1 : c;
  return c;
}

const char *offt_hipk_variant_name(int n, int precision, int variant) {
  Variant *v = find_variant(n, precision, true, true, variant);
  return v ? v->name.c_str() : "mixed-radix any-length";
}

int offt_hipk_variant_info(int n, int precision, int variant, int *elems_per_thread, int *cols) {
  Variant *v = find_variant(n, precision, true, true, variant);
  if (!v || (variant >= 0 && v->id != variant)) return -1;
  if (elems_per_thread) *elems_per_thread = v->e;
  if (cols) *cols = v->cols;
  return v->id;
}

const char *offt_hipk_kernel_name(const offt_pass_desc *d) {
  const Variant *v = pick_variant(d);
  BlueTab bt;
  if (!v) return (!d->real_input && blue_lookup(d->n, d->precision, &bt) && find_blue(bt.m, d->precision, d->in_contig != 0, d->out_contig != 0)) ? "fft_bluestein_k" : "fft_mixed_k";
  return v->mixed ? "fft_panelx_k" : (v->prec == OFFT_PREC_F32_PAIR ? "fft_panel_k<pairs>" : "fft_panel_k");
}

int offt_hipk_prepare(int n, int precision) {
  if (n < 1) { snprintf(g_err, sizeof g_err, "offt_hipk_prepare: bad n=%d", n); return -1; }
  Tables tb;
  if (get_tables(n, precision, tb, true)) return -1;
  // a 31-smooth length of 256 .. 4096 points without a precompiled panel kernel gets one now (best effort)
  if (rtc_enabled() && n >= 256 && n <= 4096 && smooth13(n) && !find_variant(n, precision, true, true, -1)) (void)rtc_build(n, precision);
  // a length without any register kernel (a prime factor > 31, or outside 256 .. 4096 and not precompiled) runs as a
  // Bluestein convolution on a power-of-two panel kernel when 2n - 1 <= 4096
  // (lengths built from primes <= 13 stay on the any-length kernel, whose small-radix stages are whole butterflies)
  int maxp = 1;
  { int m = n; for (int p = 2; p * p <= m; ++p) while (m % p == 0) { maxp = p > maxp ? p : maxp; m /= p; } if (m > 1 && m > maxp) maxp = m; }
  if (blue_enabled() && n >= 32 && n <= 2048 && maxp > 13 && !find_variant(n, precision, true, true, -1) && find_blue(blue_m(n), precision, true, true)) {
    std::lock_guard<std::mutex> lk(g_blue_mu);
    if (!g_blue.count(std::make_pair(n, precision))) {
      BlueTab bt;
      const int rc = precision == OFFT_PREC_F64 ? blue_build<double>(n, precision, bt) : blue_build<float>(n, precision, bt);
      if (rc) return -1;
      g_blue[std::make_pair(n, precision)] = bt;
    }
  }
  // a length no single launch takes (no panel kernel, and two images of a line do not fit the LDS of the any-length
  // kernel) is split n = n1 n2 for the four-step path above; 8192 has a register kernel, but with one column per
  // workgroup: its strided flavours go the four-step way as well.  Only a length without such a split -- a prime, or a
  // prime factor too large itself -- is refused, HERE, at plan time: offt_3d_init returns NULL instead of every execute failing
  const size_t esz = precision == OFFT_PREC_F64 ? sizeof(double2) : sizeof(float2);
  const bool no_direct = !find_variant(n, precision, true, true, -1) && 2 * (size_t)n * esz > (size_t)160 * 1024;
  static const bool four_on = !(getenv("OFFT_FOURSTEP") && atoi(getenv("OFFT_FOURSTEP")) == 0);
  // ... and so does a length that would otherwise run on the any-length kernel -- above the plan-time kernels' range (4800,
  // 5000; in single precision up to 10240 points: 10000 at 15 % of the roofline there) or above the Bluestein panel kernel's with
  // a large prime factor (4076 = 4 x 1019) -- if it has a FUSED split (score < 0 below)
  BlueTab bt_any;
  const bool any_only = !no_direct && n > 2048 && !find_variant(n, precision, true, true, -1) && !blue_lookup(n, precision, &bt_any);
  if ((no_direct || n == 8192 || any_only) && four_on && !four_lookup(n, precision, nullptr)) {
    int best1 = 0;
    double best_score = 1e30;
    const int forced_n1 = getenv("OFFT_FOURSTEP_N1") ? atoi(getenv("OFFT_FOURSTEP_N1")) : 0;
    for (int n1 = 2; (long long)n1 * n1 <= n; ++n1) {
      if (n % n1) continue;
      const int n2 = n / n1;
      // both factors need a kernel of their own: prefer precompiled register kernels and a balanced split (candidates are
      // only looked up here, not prepared -- preparing may compile a plan-time kernel, seconds each)
      const bool f1 = find_variant(n1, precision, true, true, -1) != nullptr, f2 = find_variant(n2, precision, true, true, -1) != nullptr;
      if ((!f1 && 2 * (size_t)n1 * esz > (size_t)160 * 1024) || (!f2 && 2 * (size_t)n2 * esz > (size_t)160 * 1024)) continue;
      double score = std::log((double)n2 / (double)n1);
      if (!f1) score += 4.0;
      if (!f2) score += 4.0;
      // ... except that a SHORT first sub-pass with the twiddles on its stores wins when the other factor still has a register
      // kernel: few points per line, one thread per line on 64 unit-stride columns (8 columns at 64 points).  Order of
      // preference, from sweeps over 8192 / 16384 / 32768 / 6000 / 10000 / 12000 points (profiles/r03_four_step.txt):
      //   double  64 (8192 = 64 x 128: 37.3 % of the roofline, 36.4 % for 32 x 256), then 16, 8, 4 (6000 = 16 x 375: 31.5 %, 10000 =
      //           16 x 625: 31.6 %, both 17-18 % as balanced unfused splits; 12000 = 16 x 750: 26.7 %, 20.1 % for 32 x 375), 32, 2
      //   single  32 (8192 = 32 x 256: 37.5 %, 31.9 % for 64 x 128), then 16, 8, 4, 2
      static const int pref64[] = {64, 16, 8, 4, 32, 2, 0}, pref32[] = {32, 16, 8, 4, 2, 0};
      const int *pref = precision == OFFT_PREC_F64 ? pref64 : pref32;
      // (second choice, behind every split whose long factor is precompiled: a long factor that gets its kernel compiled NOW --
      //  seconds of plan time for 10000 = 16 x 625 at 32 % instead of 100 x 100 at 18 %)
      const bool rtc2 = !f2 && rtc_enabled() && n2 >= 256 && n2 <= 4096 && smooth13(n2);
      // (third: a long factor with a prime factor > 13 that runs on the Bluestein panel kernel -- 4076 = 4 x 1019)
      int maxp2 = 1;
      { int m = n2; for (int p = 2; p * p <= m; ++p) while (m % p == 0) { maxp2 = p > maxp2 ? p : maxp2; m /= p; } if (m > maxp2) maxp2 = m; }
      const bool blue2 = !f2 && !rtc2 && blue_enabled() && n2 >= 32 && n2 <= 2048 && maxp2 > 13 && find_blue(blue_m(n2), precision, true, false) != nullptr &&
                         find_blue(blue_m(n2), precision, false, false) != nullptr;
      for (int r = 0; pref[r]; ++r)
        if (n1 == pref[r] && (f2 || rtc2 || blue2) && find_variant(n1, precision, false, false, -1, false, false, true))
          score = (f2 ? -10.0 : rtc2 ? -5.0 : -3.0) + 0.1 * r;
      // a factor that would itself go through scratch lines (or the any-length kernel with a radix of hundreds) is a last resort
      { int mp1 = 1, m = n1; for (int p = 2; p * p <= m; ++p) while (m % p == 0) { mp1 = p > mp1 ? p : mp1; m /= p; } if (m > mp1) mp1 = m;
        if (!f1 && mp1 > 61 && !(n1 <= 2048 && blue_enabled())) score += 8.0; }
      if (!f2 && !rtc2 && !blue2 && maxp2 > 61 && !(n2 <= 2048 && blue_enabled())) score += 8.0;
      if (forced_n1 == n1) score = -100.0;  // (OFFT_FOURSTEP_N1, for sweeps: only among the splits that are possible at all)
      if (score < best_score) { best_score = score; best1 = n1; }
    }
    if (any_only && best_score >= 0.0) best1 = 0;  // (unfused, three sweeps: the any-length kernel is no worse)
    if (best1 && (offt_hipk_prepare(best1, precision) || offt_hipk_prepare(n / best1, precision) ||
                  !direct_ok(best1, precision) || !direct_ok(n / best1, precision)))
      best1 = 0;
    if (best1) {
      FourStep fs; fs.n1 = best1; fs.n2 = n / best1;
      fs.all = !find_variant(n, precision, true, true, -1);
      HIPK_CHECK(hipMalloc(&fs.t4, (size_t)n * esz));
      (void)hipGetLastError();
      if (precision == OFFT_PREC_F64) hipLaunchKernelGGL(four_table_k<double2>, dim3((n + 255) / 256), dim3(256), 0, nullptr, (double2 *)fs.t4, (const double2 *)tb.full, n, fs.n2);
      else hipLaunchKernelGGL(four_table_k<float2>, dim3((n + 255) / 256), dim3(256), 0, nullptr, (float2 *)fs.t4, (const float2 *)tb.full, n, fs.n2);
      HIPK_CHECK(hipGetLastError());
      HIPK_CHECK(hipStreamSynchronize(nullptr));
      std::lock_guard<std::mutex> lk(g_four_mu);
      g_four[std::make_pair(n, precision)] = fs;
    }
  }
  // ... and a length without a split becomes a Bluestein convolution on lines of M = 2^k >= 2n - 1 points through scratch
  static const bool long_on = !(getenv("OFFT_BLUESTEIN_LONG") && atoi(getenv("OFFT_BLUESTEIN_LONG")) == 0);
  // (also a length the any-length kernel COULD take, but only with a radix of hundreds -- r multiply-adds per output: 3057 =
  //  3 x 1019 points ran at 0.6 % of the roofline there)
  if ((no_direct || (any_only && maxp > 61)) && !four_lookup(n, precision, nullptr) && long_on && four_on /* (its M-point lines need the four-step path) */ && blue_enabled() && n < (1 << 24) && !long_lookup(n, precision, nullptr)) {
    BlueTab bt;
    static const bool short_m = !(getenv("OFFT_BLUESTEIN_LONG_POW2") && atoi(getenv("OFFT_BLUESTEIN_LONG_POW2")) != 0);
    const int mlong = short_m ? blue_m_long(n, precision) : 0;
    const int rc = precision == OFFT_PREC_F64 ? blue_build<double>(n, precision, bt, mlong) : blue_build<float>(n, precision, bt, mlong);
    if (!rc) {
      std::lock_guard<std::mutex> lk(g_long_mu);
      LongTab lt; lt.chirp = bt.chirp; lt.bhat = bt.bhat; lt.m = bt.m;
      g_long[std::make_pair(n, precision)] = lt;
    }
  }
  if (no_direct && !four_lookup(n, precision, nullptr) && !long_lookup(n, precision, nullptr)) {
    snprintf(g_err, sizeof g_err, "no kernel for lines of %d %s points: no register kernel, the any-length kernel holds at most %zu, and %d has no "
             "factorisation n1 n2 into lengths that have one", n, precision == OFFT_PREC_F64 ? "double-complex" : "single-complex",
             (size_t)160 * 1024 / (2 * esz), n);
    return -1;
  }
  return 0;
}

int offt_hipk_fft_pass(const offt_pass_desc *d, const void *in, void *out, void *stream) {
  hipStream_t st = (hipStream_t)stream;
  if (d->n < 1 || d->ncols < 1 || d->nb1 < 1 || d->nb2 < 1) return 0;  // empty batch: nothing to do
  if (d->n == 1 && d->scale == 1.0 && in == out && d->in_axis_stride == d->out_axis_stride &&
      d->in_col_stride == d->out_col_stride && d->in_b1_stride == d->out_b1_stride &&
      d->in_b2_stride == d->out_b2_stride)
    return 0;
  Tables tb;
  if (get_tables(d->n, d->precision, tb, false)) return -1;
  {
    FourStep fs;
    if (four_lookup(d->n, d->precision, &fs)) {
      const size_t esz4 = d->precision == OFFT_PREC_F64 ? sizeof(double2) : sizeof(float2);
      const bool no_direct = !find_variant(d->n, d->precision, true, true, -1) && 2 * (size_t)d->n * esz4 > (size_t)160 * 1024;
      // a length the any-length kernel could take as well stays there for what the decomposition does not follow: real
      // input, per-peer blocks that are not whole runs of n2 inputs / n1 outputs
      const bool follows = !d->real_input && !(d->in_split && (d->in_split_nfloor || d->in_split % fs.n2)) &&
                           !(d->out_split && (d->out_split_nfloor || d->out_split % fs.n1));
      if (no_direct || (fs.all ? follows : !(d->in_contig && d->out_contig))) return four_pass(d, in, out, stream, fs);
    }
    LongTab lt;
    if (long_lookup(d->n, d->precision, &lt)) return long_pass(d, in, out, stream, lt);
  }
  Variant *v = pick_variant(d);
  if (v && v->prec == OFFT_PREC_F32_PAIR &&
      ((!d->in_contig && ((uintptr_t)in & 15)) || (!d->out_contig && ((uintptr_t)out & 15))))
    v = pick_variant(d, false);  // a strided side off the 16-B grid: the one-column kernels
  BlueTab bt;
  BlueVariant *bv = nullptr;
  if (!v && !d->real_input && blue_lookup(d->n, d->precision, &bt)) bv = find_blue(bt.m, d->precision, d->in_contig != 0, d->out_contig != 0);
  // a REAL-input line of a Bluestein length: gathered as complex lines into scratch, transformed there by the Bluestein
  // panel kernel, the first n/2 + 1 outputs scattered -- three sweeps, against a radix of the size of its largest prime
  // factor on the any-length kernel (OFFT_R2C_BLUESTEIN=0: that kernel, as in rounds 1-2)
  static const bool r2c_blue = !(getenv("OFFT_R2C_BLUESTEIN") && atoi(getenv("OFFT_R2C_BLUESTEIN")) == 0);
  if (!v && d->real_input && r2c_blue && blue_lookup(d->n, d->precision, &bt) && find_blue(bt.m, d->precision, true, true)) {
    LongTab plain;
    return long_pass(d, in, out, stream, plain);
  }
  if (v || bv) {
    PassArgs a;
    const int cols = v ? v->cols : bv->cols;
    a.in_axis = d->in_axis_stride; a.in_col = d->in_col_stride; a.in_b1 = d->in_b1_stride; a.in_b2 = d->in_b2_stride;
    a.out_axis = d->out_axis_stride; a.out_col = d->out_col_stride; a.out_b1 = d->out_b1_stride; a.out_b2 = d->out_b2_stride;
    a.in_blk = d->in_block_stride; a.out_blk = d->out_block_stride;
    a.in_shift = d->in_split ? log2i(d->in_split) : 31;
    a.out_shift = d->out_split ? log2i(d->out_split) : 31;
    a.in_split = d->in_split; a.out_split = d->out_split;
    a.in_inv = d->in_split ? 1.0f / (float)d->in_split : 0.0f;
    a.out_inv = d->out_split ? 1.0f / (float)d->out_split : 0.0f;
    a.in_nfloor = d->in_split_nfloor; a.out_nfloor = d->out_split_nfloor;
    a.in_lim = d->in_split_nfloor ? d->in_split * d->in_split_nfloor : d->n;   // even split: every index below lim
    a.out_lim = d->out_split_nfloor ? d->out_split * d->out_split_nfloor : d->n;
    a.in_inv1 = 1.0f / (float)(d->in_split + 1);
    a.out_inv1 = 1.0f / (float)(d->out_split + 1);
    a.ncols = d->ncols;
    a.ncp = (d->ncols + cols - 1) / cols;
    a.nb1 = d->nb1;
    a.conj = d->direction > 0;
    a.scale = d->scale;
    a.in_tab = d->in_split ? d->in_block_tab : nullptr;
    a.out_tab = d->out_split ? d->out_block_tab : nullptr;
    a.tw4 = d->tw4; a.tw4_b1 = d->tw4_b1; a.tw4_n2 = d->tw4_n2;
    if (d->tw4 && !(v && v->tw4)) { snprintf(g_err, sizeof g_err, "offt_hipk_fft_pass: no kernel with four-step twiddles for n=%d", d->n); return -1; }
    long long nblk = (long long)a.ncp * d->nb1 * d->nb2;
    if (nblk > 0x7fffffffLL) { snprintf(g_err, sizeof g_err, "offt_hipk_fft_pass: grid too large"); return -1; }
    xcd_order(nblk, &a.xcd_lim, &a.xcd_gshift);
    if (bv) {  // Bluestein: the M-point panel machinery on a line of n points (offt_bluestein.hpp)
      Tables tm;
      if (get_tables(bt.m, d->precision, tm, false)) return -1;
      int nlen = d->n;
      void *args[] = {(void *)&a, (void *)&in, (void *)&out, (void *)&tm.full, (void *)&bt.chirp, (void *)&bt.bhat, (void *)&nlen};
      if (!bv->attr_set) {
        if (bv->lds > 48 * 1024)
          HIPK_CHECK(hipFuncSetAttribute(bv->fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bv->lds));
        bv->attr_set = true;
      }
      HIPK_CHECK(hipLaunchKernel(bv->fn, dim3((unsigned)nblk), dim3(bv->threads), args, bv->lds, st));
      return 0;
    }
    // every panel kernel stages its twiddles from the exact full-wave table w^m, m < n (the quarter- and half-wave
    // tables they keep in LDS are prefixes of it)
    void *args[] = {(void *)&a, (void *)&in, (void *)&out, (void *)&tb.full};
    if (v->modfn) {  // plan-time instance: a module function
      HIPK_CHECK(hipModuleLaunchKernel((hipFunction_t)v->modfn, (unsigned)nblk, 1, 1, v->threads, 1, 1, (unsigned)v->lds, st, args, nullptr));
      return 0;
    }
    if (!v->attr_set) {
      if (v->lds > 48 * 1024)
        HIPK_CHECK(hipFuncSetAttribute(v->fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)v->lds));
      v->attr_set = true;
    }
    HIPK_CHECK(hipLaunchKernel(v->fn, dim3((unsigned)nblk), dim3(v->threads), args, v->lds, st));
    return 0;
  }
  // any-length path
  GenArgs g;
  g.in_axis = d->in_axis_stride; g.in_col = d->in_col_stride; g.in_b1 = d->in_b1_stride; g.in_b2 = d->in_b2_stride;
  g.out_axis = d->out_axis_stride; g.out_col = d->out_col_stride; g.out_b1 = d->out_b1_stride; g.out_b2 = d->out_b2_stride;
  g.in_blk = d->in_block_stride; g.out_blk = d->out_block_stride;
  g.in_split = d->in_split; g.in_nfloor = d->in_split_nfloor;
  g.out_split = d->out_split; g.out_nfloor = d->out_split_nfloor;
  g.n = d->n; g.ncols = d->ncols; g.nb1 = d->nb1;
  g.in_contig = d->in_contig; g.out_contig = d->out_contig;
  g.conj = d->direction > 0;
  g.real_in = d->real_input;
  g.scale = d->scale;
  g.in_tab = d->in_split ? d->in_block_tab : nullptr;
  g.out_tab = d->out_split ? d->out_block_tab : nullptr;
  // radices: prime factors, pairs of 2 merged into 4 (fewer LDS round trips at equal cost)
  g.nfac = 0;
  {
    int m = d->n, twos = 0;
    while (m % 2 == 0) { twos++; m /= 2; }
    for (; twos >= 2; twos -= 2) g.fac[g.nfac++] = 4;
    if (twos) g.fac[g.nfac++] = 2;
    for (int f = 3; f * f <= m; f += 2)
      while (m % f == 0) {
        if (g.nfac >= OFFT_MIX_MAXFAC) { snprintf(g_err, sizeof g_err, "offt_hipk_fft_pass: n=%d has too many factors", d->n); return -1; }
        g.fac[g.nfac++] = f; m /= f;
      }
    if (m > 1) g.fac[g.nfac++] = m;
    if (d->n == 1) { g.nfac = 0; }
    if (g.nfac > OFFT_MIX_MAXFAC) { snprintf(g_err, sizeof g_err, "offt_hipk_fft_pass: n=%d has too many factors", d->n); return -1; }
  }
  const size_t esz = d->precision == OFFT_PREC_F64 ? sizeof(double2) : sizeof(float2);
  const size_t lds_cap = 160 * 1024;
  if (2 * (size_t)d->n * esz > lds_cap) {
    snprintf(g_err, sizeof g_err, "offt_hipk_fft_pass: n=%d too long for the any-length kernel", d->n);
    return -1;
  }
  int cols = 8;
  while (cols > 1 && (2 * (size_t)cols + 1) * d->n * esz > lds_cap) cols >>= 1;
  g.tw_in_lds = (2 * (size_t)cols + 1) * d->n * esz <= lds_cap;
  g.cols = cols;
  g.ncp = (d->ncols + cols - 1) / cols;
  size_t lds = (2 * (size_t)cols + (g.tw_in_lds ? 1 : 0)) * d->n * esz;
  long long nblk = (long long)g.ncp * d->nb1 * d->nb2;
  if (nblk > 0x7fffffffLL) { snprintf(g_err, sizeof g_err, "offt_hipk_fft_pass: grid too large"); return -1; }
  xcd_order(nblk, &g.xcd_lim, &g.xcd_gshift);
  // the panel's LDS footprint allows one or two workgroups per CU: size the workgroup so that the CU still
  // holds 8-16 waves to cover the LDS round trips between stages
  static const int mix_nt_env = getenv("OFFT_MIX_THREADS") ? atoi(getenv("OFFT_MIX_THREADS")) : 0;
  unsigned nt = lds > 80 * 1024 ? 1024 : lds > 40 * 1024 ? 512 : 256;
  while (nt > 64 && (long long)nt * 2 > (long long)cols * d->n) nt >>= 1;  // at least two elements per thread
  if (mix_nt_env >= 64 && mix_nt_env <= 1024) nt = (unsigned)mix_nt_env;
  (void)hipGetLastError();  // start from a clean slate: the check below must see only this launch
  if (d->precision == OFFT_PREC_F64) {
    static bool set64 = false;
    if (lds > 48 * 1024 && !set64) {
      HIPK_CHECK(hipFuncSetAttribute((const void *)fft_mixed_k<double>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_cap));
      set64 = true;
    }
    hipLaunchKernelGGL(fft_mixed_k<double>, dim3((unsigned)nblk), dim3(nt), lds, st, g, (const double2 *)in,
                       (double2 *)out, (const double2 *)tb.full);
  } else {
    static bool set32 = false;
    if (lds > 48 * 1024 && !set32) {
      HIPK_CHECK(hipFuncSetAttribute((const void *)fft_mixed_k<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_cap));
      set32 = true;
    }
    hipLaunchKernelGGL(fft_mixed_k<float>, dim3((unsigned)nblk), dim3(nt), lds, st, g, (const float2 *)in,
                       (float2 *)out, (const float2 *)tb.full);
  }
  HIPK_CHECK(hipGetLastError());
  return 0;
}

}  // extern "C"

namespace {
int four_pass(const offt_pass_desc *d, const void *in, void *out, void *stream, const FourStep &fs) {
  const int N = d->n, N1 = fs.n1, N2 = fs.n2;
  const size_t esz = d->precision == OFFT_PREC_F64 ? sizeof(double2) : sizeof(float2);
  if (d->real_input) { LongTab plain; return long_pass(d, in, out, stream, plain); }  // gathered as complex lines, first n/2 + 1 outputs scattered
  // a per-peer split must cut the axis where the decomposition can follow it: whole runs of n2 inputs / n1 outputs.  One that
  // does not (uneven blocks) sends the lines through scratch: gathered with the split, transformed as contiguous lines
  if ((d->in_split && (d->in_split_nfloor || d->in_split % N2)) || (d->out_split && (d->out_split_nfloor || d->out_split % N1))) {
    LongTab plain;
    return long_pass(d, in, out, stream, plain);
  }
  Tables tb;
  if (get_tables(N, d->precision, tb, false)) return -1;
  const bool inL = d->in_contig != 0, outL = d->out_contig != 0;
  // columns per chunk: whole rows of the caller's column dimension, as many b1 entries as fit 256 MiB of scratch
  static const int chunk_mib = getenv("OFFT_FOURSTEP_CHUNK_MIB") && atoi(getenv("OFFT_FOURSTEP_CHUNK_MIB")) > 0 ? atoi(getenv("OFFT_FOURSTEP_CHUNK_MIB")) : 192;
  const size_t cap = ((size_t)chunk_mib << 20) / esz;
  int cc = d->ncols, cb1 = d->nb1;
  if ((size_t)cc * N > cap) { cc = (int)(cap / N); if (cc < 1) cc = 1; cb1 = 1; }
  else { const size_t fit = cap / ((size_t)cc * N); if ((size_t)cb1 > fit) cb1 = (int)(fit < 1 ? 1 : fit); }
  const size_t sbytes = (size_t)cc * cb1 * N * esz;
  const DepthGuard depth;
  if (depth.d >= SCRATCH_DEPTHS) { snprintf(g_err, sizeof g_err, "four-step path: decomposition nested too deep for n=%d", N); return -1; }
  char *S = (char *)four_scratch(stream, 3 * depth.d + 0, sbytes);
  char *Sp = inL ? nullptr : (char *)four_scratch(stream, 3 * depth.d + 1, sbytes);
  if (!S || (!inL && !Sp)) { snprintf(g_err, sizeof g_err, "four-step path: cannot allocate %zu bytes of scratch", sbytes); return -1; }
  for (int b2 = 0; b2 < d->nb2; ++b2)
    for (int b10 = 0; b10 < d->nb1; b10 += cb1)
      for (int c0 = 0; c0 < d->ncols; c0 += cc) {
        const int nb = d->nb1 - b10 < cb1 ? d->nb1 - b10 : cb1, nc = d->ncols - c0 < cc ? d->ncols - c0 : cc;
        const char *pin = (const char *)in + ((long long)b2 * d->in_b2_stride + (long long)b10 * d->in_b1_stride + (long long)c0 * d->in_col_stride) * (long long)esz;
        char *pout = (char *)out + ((long long)b2 * d->out_b2_stride + (long long)b10 * d->out_b1_stride + (long long)c0 * d->out_col_stride) * (long long)esz;
        // ---- A: n1-point transforms over j1 (input index j1 n2 + j2) ----
        offt_pass_desc a;
        memset(&a, 0, sizeof a);
        a.n = N1; a.precision = d->precision; a.direction = d->direction; a.variant = -1; a.scale = 1.0; a.no_pairs = d->no_pairs;
        a.in_axis_stride = (long long)N2 * d->in_axis_stride;
        if (d->in_split) { a.in_split = d->in_split / N2; a.in_block_stride = d->in_block_stride; a.in_block_tab = d->in_block_tab; }
        a.in_contig = 0; a.out_contig = 0;
        // twiddles fused into A's stores when the n1-point kernel has such a twin: two sweeps instead of three.  A strided-in
        // pass then leaves S'[k1][j2][c] as it is and C reads it with the caller's columns as its columns -- which needs a
        // strided-out pass as well; strided-in / contig-out keeps the transposing twiddle sweep.
        static const bool fuse_on = !(getenv("OFFT_FOURSTEP_FUSE") && atoi(getenv("OFFT_FOURSTEP_FUSE")) == 0);
        const bool fused = fuse_on && (inL || !outL) && (!d->in_split || is_pow2(d->in_split / N2)) &&
                           find_variant(N1, d->precision, false, false, -1, false, false, true) != nullptr;
        if (fused) { a.tw4 = fs.t4; a.tw4_b1 = inL ? 0 : 1; a.tw4_n2 = N2; }
        if (inL) {  // the axis is the unit-stride dimension: j2 becomes the column dimension
          a.ncols = N2; a.in_col_stride = d->in_axis_stride;
          a.nb1 = nc; a.in_b1_stride = d->in_col_stride;
          a.nb2 = nb; a.in_b2_stride = d->in_b1_stride;
          a.out_axis_stride = N2; a.out_col_stride = 1; a.out_b1_stride = N; a.out_b2_stride = (long long)N * nc;  // S[q][k1][j2]
          if (offt_hipk_fft_pass(&a, pin, S, stream)) return -1;
        } else {    // the caller's columns are the unit-stride dimension and stay the columns: S'[b][k1][j2][c]
          a.ncols = nc; a.in_col_stride = d->in_col_stride;
          a.nb1 = N2; a.in_b1_stride = d->in_axis_stride;
          a.nb2 = nb; a.in_b2_stride = d->in_b1_stride;
          a.out_axis_stride = (long long)N2 * nc; a.out_col_stride = 1; a.out_b1_stride = nc; a.out_b2_stride = (long long)N * nc;
          if (offt_hipk_fft_pass(&a, pin, fused ? S : Sp, stream)) return -1;
        }
        // ---- T: twiddles w_n^(j2 k1) (and the transposition S' -> S) ----
        (void)hipGetLastError();
        const int conj = d->direction > 0;
        if (fused) {
          /* nothing: the twiddles rode on A's stores */
        } else if (inL) {
          const long long total = (long long)nc * nb * N;
          const unsigned blocks = (unsigned)((total + 255) / 256);
          if (d->precision == OFFT_PREC_F64)
            hipLaunchKernelGGL(four_twiddle_k<double2>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (double2 *)S, (const double2 *)tb.full, total, N1, N2, conj);
          else
            hipLaunchKernelGGL(four_twiddle_k<float2>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (float2 *)S, (const float2 *)tb.full, total, N1, N2, conj);
        } else {
          const dim3 grid((unsigned)((nc + 31) / 32), (unsigned)((N + 31) / 32), (unsigned)nb);
          if (d->precision == OFFT_PREC_F64)
            hipLaunchKernelGGL(four_twiddle_t_k<double2>, grid, dim3(256), 0, (hipStream_t)stream, (const double2 *)Sp, (double2 *)S, (const double2 *)tb.full, N, nc, N2, conj);
          else
            hipLaunchKernelGGL(four_twiddle_t_k<float2>, grid, dim3(256), 0, (hipStream_t)stream, (const float2 *)Sp, (float2 *)S, (const float2 *)tb.full, N, nc, N2, conj);
        }
        HIPK_CHECK(hipGetLastError());
        // ---- C: n2-point transforms over j2, output index k1 + n1 k2 ----
        offt_pass_desc c;
        memset(&c, 0, sizeof c);
        c.n = N2; c.precision = d->precision; c.direction = d->direction; c.variant = -1; c.scale = d->scale; c.no_pairs = d->no_pairs;
        c.out_keep = d->out_keep;
        c.in_axis_stride = 1; c.in_contig = 1; c.out_contig = 0;
        c.out_axis_stride = (long long)N1 * d->out_axis_stride;
        if (d->out_split) { c.out_split = d->out_split / N1; c.out_block_stride = d->out_block_stride; c.out_block_tab = d->out_block_tab; }
        if (fused && !inL) {  // S'[b][k1][j2][c] straight from A: the caller's columns are the columns on both sides
          c.in_contig = 0; c.in_axis_stride = nc;
          c.ncols = nc; c.in_col_stride = 1; c.out_col_stride = d->out_col_stride;
          c.nb1 = N1; c.in_b1_stride = (long long)N2 * nc; c.out_b1_stride = d->out_axis_stride;
          c.nb2 = nb; c.in_b2_stride = (long long)N * nc; c.out_b2_stride = d->out_b1_stride;
        } else if (outL) {  // k1 becomes the column dimension of the output
          c.ncols = N1; c.in_col_stride = N2; c.out_col_stride = d->out_axis_stride;
          c.nb1 = nc; c.in_b1_stride = N; c.out_b1_stride = d->out_col_stride;
          c.nb2 = nb; c.in_b2_stride = (long long)N * nc; c.out_b2_stride = d->out_b1_stride;
        } else {     // the caller's columns stay the columns
          c.ncols = nc; c.in_col_stride = N; c.out_col_stride = d->out_col_stride;
          c.nb1 = N1; c.in_b1_stride = N2; c.out_b1_stride = d->out_axis_stride;
          c.nb2 = nb; c.in_b2_stride = (long long)N * nc; c.out_b2_stride = d->out_b1_stride;
        }
        if (offt_hipk_fft_pass(&c, S, pout, stream)) return -1;
      }
  return 0;
}

int long_pass(const offt_pass_desc *d, const void *in, void *out, void *stream, const LongTab &lt) {
  const int N = d->n, M = lt.m ? lt.m : N;   // (lt.m == 0: plain complex lines of N points, the long real-input case)
  const size_t esz = d->precision == OFFT_PREC_F64 ? sizeof(double2) : sizeof(float2);
  const hipStream_t st = (hipStream_t)stream;
  GenArgs g;
  memset(&g, 0, sizeof g);
  g.in_axis = d->in_axis_stride; g.in_col = d->in_col_stride; g.in_b1 = d->in_b1_stride; g.in_b2 = d->in_b2_stride;
  g.out_axis = d->out_axis_stride; g.out_col = d->out_col_stride; g.out_b1 = d->out_b1_stride; g.out_b2 = d->out_b2_stride;
  g.in_blk = d->in_block_stride; g.out_blk = d->out_block_stride;
  g.in_split = d->in_split; g.in_nfloor = d->in_split_nfloor;
  g.out_split = d->out_split; g.out_nfloor = d->out_split_nfloor;
  g.n = N; g.ncols = d->ncols; g.nb1 = d->nb1;
  g.conj = d->direction > 0;
  g.real_in = d->real_input;
  g.scale = d->scale;
  g.in_tab = d->in_split ? d->in_block_tab : nullptr;
  g.out_tab = d->out_split ? d->out_block_tab : nullptr;
  const int kend = d->real_input ? N / 2 + 1 : N;
  const long long nlines = (long long)d->ncols * d->nb1 * d->nb2;
  long long per = (long long)(((size_t)256 << 20) / ((size_t)M * esz));
  if (per < 1) per = 1;
  if (per > nlines) per = nlines;
  const DepthGuard depth;
  if (depth.d >= SCRATCH_DEPTHS) { snprintf(g_err, sizeof g_err, "lines of %d points through scratch: decomposition nested too deep", N); return -1; }
  char *U = (char *)four_scratch(stream, 3 * depth.d + 2, (size_t)per * M * esz);
  if (!U) { snprintf(g_err, sizeof g_err, "lines of %d points through scratch: cannot allocate %zu bytes", N, (size_t)per * M * esz); return -1; }
  // the M-point transforms of the scratch lines: contiguous lines, in place
  offt_pass_desc f;
  memset(&f, 0, sizeof f);
  f.n = M; f.precision = d->precision; f.variant = -1; f.scale = 1.0; f.no_pairs = d->no_pairs;
  f.in_axis_stride = f.out_axis_stride = 1; f.in_contig = f.out_contig = 1;
  f.in_col_stride = f.out_col_stride = M; f.nb1 = f.nb2 = 1;
  for (long long l0 = 0; l0 < nlines; l0 += per) {
    const long long nl = nlines - l0 < per ? nlines - l0 : per;
    const long long tot = nl * M, tote = nl * kend;
    if (tot > 0x7fffffffLL * 256) { snprintf(g_err, sizeof g_err, "lines of %d points through scratch: chunk too large", N); return -1; }
    const unsigned blocks = (unsigned)((tot + 255) / 256), blockse = (unsigned)((tote + 255) / 256);
    f.ncols = (int)nl;
    (void)hipGetLastError();
    if (d->precision == OFFT_PREC_F64)
      hipLaunchKernelGGL(long_gather_k<double>, dim3(blocks), dim3(256), 0, st, g, (const double2 *)in, (double2 *)U, (const double2 *)lt.chirp, M, l0, tot);
    else
      hipLaunchKernelGGL(long_gather_k<float>, dim3(blocks), dim3(256), 0, st, g, (const float2 *)in, (float2 *)U, (const float2 *)lt.chirp, M, l0, tot);
    HIPK_CHECK(hipGetLastError());
    f.direction = -1;
    if (offt_hipk_fft_pass(&f, U, U, stream)) return -1;
    if (lt.m) {
      if (d->precision == OFFT_PREC_F64)
        hipLaunchKernelGGL(long_mul_k<double2>, dim3(blocks), dim3(256), 0, st, (double2 *)U, (const double2 *)lt.bhat, M, tot);
      else
        hipLaunchKernelGGL(long_mul_k<float2>, dim3(blocks), dim3(256), 0, st, (float2 *)U, (const float2 *)lt.bhat, M, tot);
      HIPK_CHECK(hipGetLastError());
      f.direction = +1;
      if (offt_hipk_fft_pass(&f, U, U, stream)) return -1;
    }
    if (d->precision == OFFT_PREC_F64)
      hipLaunchKernelGGL(long_scatter_k<double>, dim3(blockse), dim3(256), 0, st, g, (const double2 *)U, (double2 *)out, (const double2 *)lt.chirp, M, kend, l0, tote);
    else
      hipLaunchKernelGGL(long_scatter_k<float>, dim3(blockse), dim3(256), 0, st, g, (const float2 *)U, (float2 *)out, (const float2 *)lt.chirp, M, kend, l0, tote);
    HIPK_CHECK(hipGetLastError());
  }
  return 0;
}
}  // namespace

extern "C" {

// ---- flags of the direct-store exchange (offt_hipk.h) ----
namespace {
struct FlagArgs {
  unsigned long long *addr[OFFT_HIPK_MAX_FLAGS];
  int n;
  unsigned long long value;
  unsigned long long *status;
  long long timeout_ticks;  // of the 100 MHz constant clock (wall_clock64)
};
__global__ void __launch_bounds__(64) flag_signal_k(FlagArgs a) {
  const int i = threadIdx.x;
  // system scope: the word lives in a peer's memory (another GPU over xGMI, or another process's buffer on this one)
  if (i < a.n) __hip_atomic_store(a.addr[i], a.value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ void __launch_bounds__(64) flag_wait_k(FlagArgs a) {
  const int i = threadIdx.x;
  if (i < a.n) {
    const long long t0 = wall_clock64();
    // every lane polls its own word; the loop ends for every lane: the value arrives or the clock runs out
    while (__hip_atomic_load(a.addr[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < a.value) {
      if (wall_clock64() - t0 > a.timeout_ticks) {
        if (a.status) __hip_atomic_store(a.status, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        break;
      }
      __builtin_amdgcn_s_sleep(64);
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");  // system scope; the next kernel's start-of-kernel acquire does the rest
}
int flag_launch(bool wait, int n, unsigned long long *const *addr, unsigned long long value, unsigned long long *status, double timeout_s,
                void *stream) {
  if (n < 0 || n > OFFT_HIPK_MAX_FLAGS) { snprintf(g_err, sizeof g_err, "offt_hipk_flag_%s: %d words (at most %d)", wait ? "wait" : "signal", n, OFFT_HIPK_MAX_FLAGS); return -1; }
  if (n == 0) return 0;
  FlagArgs a;
  for (int i = 0; i < OFFT_HIPK_MAX_FLAGS; i++) a.addr[i] = i < n ? addr[i] : nullptr;
  a.n = n; a.value = value; a.status = status;
  a.timeout_ticks = (long long)(timeout_s * 1e8);
  (void)hipGetLastError();
  if (wait) hipLaunchKernelGGL(flag_wait_k, dim3(1), dim3(64), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(flag_signal_k, dim3(1), dim3(64), 0, (hipStream_t)stream, a);
  HIPK_CHECK(hipGetLastError());
  return 0;
}
}  // namespace

namespace {
__global__ void __launch_bounds__(64) delay_k(long long ticks) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
}
}  // namespace
// one wave that holds `stream` for `ms` milliseconds (100 MHz constant clock; at most 1 s): test builds put it in front of
// their passes to make the device lag behind the host (offt_host.c, OFFT_TEST_SLOW_PASS_MS)
int offt_hipk_delay(double ms, void *stream) {
  if (ms <= 0) return 0;
  if (ms > 1000.0) ms = 1000.0;
  (void)hipGetLastError();
  hipLaunchKernelGGL(delay_k, dim3(1), dim3(64), 0, (hipStream_t)stream, (long long)(ms * 1e5));
  HIPK_CHECK(hipGetLastError());
  return 0;
}

int offt_hipk_flag_signal(int n, unsigned long long *const *addr, unsigned long long value, void *stream) {
  return flag_launch(false, n, addr, value, nullptr, 0.0, stream);
}
int offt_hipk_flag_wait(int n, unsigned long long *const *addr, unsigned long long value, unsigned long long *status, double timeout_s,
                        void *stream) {
  return flag_launch(true, n, addr, value, status, timeout_s, stream);
}

int offt_hipk_copy3d(const void *in, void *out, int precision, int n0, int n1, int n2, long long is0,
                     long long is1, long long is2, long long os0, long long os1, long long os2, void *stream) {
  long long total = (long long)n0 * n1 * n2;
  if (total <= 0) return 0;
  long long nb = (total + 255) / 256;
  if (nb > 256 * 64) nb = 256 * 64;
  if (precision == OFFT_PREC_F64)
    hipLaunchKernelGGL(copy3d_k<double2>, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, (const double2 *)in,
                       (double2 *)out, n0, n1, n2, is0, is1, is2, os0, os1, os2);
  else
    hipLaunchKernelGGL(copy3d_k<float2>, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, (const float2 *)in,
                       (float2 *)out, n0, n1, n2, is0, is1, is2, os0, os1, os2);
  HIPK_CHECK(hipGetLastError());
  return 0;
}

int offt_hipk_fill(void *buf, int precision, int kind, int n0, int n1, int n2, int s0, int s1, int s2,
                   long long st0, long long st1, long long st2, void *stream) {
  long long total = (long long)n0 * n1 * n2;
  if (total <= 0) return 0;
  long long nb = (total + 255) / 256;
  if (nb > 256 * 64) nb = 256 * 64;
  if (precision & 0x100) {  // real-valued field (r2c input), strides in scalars
    if ((precision & 0xff) == OFFT_PREC_F64)
      hipLaunchKernelGGL(fill_real_k<double>, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, (double *)buf, kind,
                         n0, n1, n2, s0, s1, s2, st0, st1, st2);
    else
      hipLaunchKernelGGL(fill_real_k<float>, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, (float *)buf, kind,
                         n0, n1, n2, s0, s1, s2, st0, st1, st2);
    HIPK_CHECK(hipGetLastError());
    return 0;
  }
  if (precision == OFFT_PREC_F64)
    hipLaunchKernelGGL(fill_k<double2>, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, (double2 *)buf, kind,
                       n0, n1, n2, s0, s1, s2, st0, st1, st2);
  else
    hipLaunchKernelGGL(fill_k<float2>, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, (float2 *)buf, kind, n0,
                       n1, n2, s0, s1, s2, st0, st1, st2);
  HIPK_CHECK(hipGetLastError());
  return 0;
}

}  // extern "C"
