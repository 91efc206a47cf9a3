// offt_kernels.hip -- hand-written CDNA4 (gfx950) kernels for the OFFT hot path.
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
#include "offt_hipk.h"
#include "offt_w32_consts.h"

namespace {

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

template <typename T> struct vec2;
template <> struct vec2<double> { using type = double2; };
template <> struct vec2<float> { using type = float2; };

template <typename T> struct cx { T x, y; };

// global memory access helpers: 16-B (f64) / 8-B (f32) per lane.  Every element is
// touched exactly once per pass, so loads and stores are non-temporal (streaming): A/B on
// 1024^3 (profiles/r01_sweep.txt): -6 % transform time vs default cache policy.
// -DOFFT_NO_NT_LOAD / -DOFFT_NO_NT_STORE restore the default policy for A/B builds.
template <typename V2>
__device__ __forceinline__ V2 gload(const V2 *p) {
#ifndef OFFT_NO_NT_LOAD
  using E = decltype(p->x);
  typedef E vt __attribute__((ext_vector_type(2)));
  vt r = __builtin_nontemporal_load(reinterpret_cast<const vt *>(p));
  V2 o; o.x = r.x; o.y = r.y; return o;
#else
  return *p;
#endif
}
template <typename V2>
__device__ __forceinline__ void gstore(V2 *p, V2 v) {
#ifndef OFFT_NO_NT_STORE
  using E = decltype(p->x);
  typedef E vt __attribute__((ext_vector_type(2)));
  vt r; r.x = v.x; r.y = v.y;
  __builtin_nontemporal_store(r, reinterpret_cast<vt *>(p));
#else
  *p = v;
#endif
}

template <int B, int E_, class F>
__device__ __forceinline__ void static_for(F &&f) {
  if constexpr (B < E_) {
    f(std::integral_constant<int, B>{});
    static_for<B + 1, E_>(f);
  }
}

constexpr int ilog2(int n) { return n <= 1 ? 0 : 1 + ilog2(n >> 1); }
constexpr int bitrev(int v, int bits) {
  int r = 0;
  for (int i = 0; i < bits; ++i) r |= ((v >> i) & 1) << (bits - 1 - i);
  return r;
}

constexpr double W32C[32] = OFFT_W32_COS;
constexpr double W32S[32] = OFFT_W32_SIN;

// d * w32^K, w32 = exp(-2 pi i / 32)
template <typename T, int K>
__device__ __forceinline__ cx<T> mulw32(cx<T> d) {
  constexpr int k = K & 31;
  if constexpr (k == 0) return d;
  else if constexpr (k == 8) return cx<T>{d.y, -d.x};
  else if constexpr (k == 16) return cx<T>{-d.x, -d.y};
  else if constexpr (k == 24) return cx<T>{-d.y, d.x};
  else if constexpr (k == 4) {
    constexpr T s = (T)W32C[4];
    return cx<T>{(d.x + d.y) * s, (d.y - d.x) * s};
  } else if constexpr (k == 12) {
    constexpr T s = (T)W32C[4];
    return cx<T>{(d.y - d.x) * s, -(d.x + d.y) * s};
  } else {
    constexpr T c = (T)W32C[k], s = (T)W32S[k];
    return cx<T>{d.x * c + d.y * s, d.y * c - d.x * s};
  }
}

// In-register radix-R DFT (R = 2..32), radix-2 decimation in frequency with
// compile-time twiddles.  Result is left in bit-reversed order:
// X[k] = v[bitrev(k)].
template <typename T, int R>
__device__ __forceinline__ void dft_reg(cx<T> *v) {
  static_for<0, ilog2(R)>([&](auto st) {
    constexpr int h = R >> (decltype(st)::value + 1);
    static_for<0, R / 2>([&](auto bi) {
      constexpr int b = (decltype(bi)::value / h) * 2 * h;
      constexpr int i = decltype(bi)::value % h;
      cx<T> p = v[b + i], q = v[b + i + h];
      v[b + i] = cx<T>{p.x + q.x, p.y + q.y};
      cx<T> d{p.x - q.x, p.y - q.y};
      v[b + i + h] = mulw32<T, i * (16 / h)>(d);
    });
  });
}

struct PassArgs {
  long long in_axis, in_col, in_b1, in_b2, in_blk;
  long long out_axis, out_col, out_b1, out_b2, out_blk;
  int in_shift, out_shift;  // log2(split) or 31 for "no split"
  int ncols, ncp, nb1;      // ncp = column panels per batch entry
  int conj;                 // 1: inverse transform via conj-in / conj-out
  double scale;
};

template <int N, int E, int R0, int R1, int R2, int COLS, bool SPLIT, typename T>
struct PanelCfg {
  static constexpr int TPL = N / E;
  static constexpr int NT = TPL * COLS;
  static constexpr int NSTAGE = (R2 > 1) ? 3 : ((R1 > 1) ? 2 : 1);
  // LDS image of one column: Stockham stage s writes index (q-k)*R + k + t*Ns with lanes
  // along q and reads index q' + t'*(N/R') with lanes along q'.  Stage 0 (Ns = 1) is a
  // stride-R0 write: all lanes of a ds_write group would hit one bank.
  //  * R0 >= 16: XOR swizzle  i -> i ^ ((i >> log2 R0) & 15).  The strided writes spread
  //    over 16 bank pairs, and the unit-stride accesses are only permuted inside aligned
  //    16-element runs, so they stay conflict-free (a padded image misaligns them: the
  //    PMC pass showed SQ_LDS_BANK_CONFLICT = 50 % of SQ_LDS_IDX_ACTIVE with padding).
  //  * R0 < 16 (small N): pad one element every R0.
  static constexpr bool SWZ = (R0 >= 16);
  static constexpr int PADSHIFT = ilog2(R0) < 3 ? 3 : ilog2(R0);
  static constexpr int SWZSHIFT = ilog2(R0);
  static constexpr int NPAD = SWZ ? N : N + (N >> PADSHIFT);
  // column pitch == 4 (mod 32) elements: the 8 columns x 4 rows of one 32-lane ds_read_b64
  // group of a strided-store flavour land in 32 distinct bank pairs
  static constexpr int LSTRIDE = SWZ ? ((NPAD + 31) / 32) * 32 + 4 : ((NPAD + 13) / 16) * 16 + 2;
  static constexpr int QT = (N >= 4) ? N / 4 + 1 : 1;
  static constexpr size_t EX_BYTES =
      NSTAGE > 1 ? (size_t)COLS * LSTRIDE * sizeof(T) * (SPLIT ? 1 : 2) : 0;
  static constexpr size_t TW_OFF = (EX_BYTES + 15) / 16 * 16;
  static constexpr size_t LDS_BYTES = NSTAGE > 1 ? TW_OFF + (size_t)QT * 2 * sizeof(T) : 0;
  // occupancy target handed to __launch_bounds__ (2nd argument = waves per
  // SIMD): as many workgroups per CU as the 160 KiB LDS admits, at most 4
  // waves per SIMD -- enough to overlap one group's butterflies with another
  // group's HBM traffic without starving the register allocator.
  static constexpr int WG_PER_CU_LDS = LDS_BYTES ? (int)(160 * 1024 / LDS_BYTES) : 8;
  static constexpr int WPS_RAW = (WG_PER_CU_LDS * NT + 255) / 256;
  static constexpr int WPS = WPS_RAW < 1 ? 1 : (WPS_RAW > 4 ? 4 : WPS_RAW);
  // register budget: an E-point thread keeps E*sizeof(T)/2 data VGPRs; it needs
  // roughly twice that (butterfly temporaries, addresses, exchange staging)
  static constexpr int DATA_VGPR = E * (int)sizeof(T) / 2;
  // (a radix-32 butterfly alone keeps ~40 temporaries alive: never ask for more than 2 waves/SIMD)
  static constexpr int WPS_REG = (DATA_VGPR >= 128 || R0 >= 32 || R1 >= 32 || R2 >= 32) ? 2 : (DATA_VGPR >= 64 ? 3 : 4);
  static constexpr int WPS_MIN = (NT + 255) / 256;  // one workgroup must fit on a CU
  static constexpr int WPS_E = WPS < WPS_REG ? WPS : (WPS_REG < WPS_MIN ? WPS_MIN : WPS_REG);
};

template <bool SWZ, int SHIFT>
__device__ __forceinline__ int padidx(int i) {
  if constexpr (SWZ) return i ^ ((i >> SHIFT) & 15);
  else return i + (i >> SHIFT);
}

template <typename T, int N, int E, int R0, int R1, int R2, int COLS, bool INC, bool OUTC, bool SPLIT, bool R2C = false>
__global__ void __launch_bounds__((N / E) * COLS, (PanelCfg<N, E, R0, R1, R2, COLS, SPLIT, T>::WPS_E))
fft_panel_k(PassArgs a, const typename vec2<T>::type *in, typename vec2<T>::type *out,
            const typename vec2<T>::type *twq) {
  using V2 = typename vec2<T>::type;
  using Cfg = PanelCfg<N, E, R0, R1, R2, COLS, SPLIT, T>;
  constexpr int TPL = Cfg::TPL, NT = Cfg::NT, NSTAGE = Cfg::NSTAGE;
  constexpr int LSTRIDE = Cfg::LSTRIDE;
  constexpr bool SWZ = Cfg::SWZ;
  constexpr int PS = SWZ ? Cfg::SWZSHIFT : Cfg::PADSHIFT;
  static_assert(R0 * R1 * R2 == N, "radices must multiply to N");
  static_assert(E % R0 == 0 && E % R1 == 0 && E % R2 == 0 && N % E == 0, "bad E");

  extern __shared__ __align__(16) unsigned char smem[];
  T *exs = reinterpret_cast<T *>(smem);
  V2 *exv = reinterpret_cast<V2 *>(smem);
  V2 *tw = reinterpret_cast<V2 *>(smem + Cfg::TW_OFF);

  const int tid = threadIdx.x;
  if constexpr (NSTAGE > 1) {
    for (int i = tid; i < Cfg::QT; i += NT) tw[i] = twq[i];
  }

  // panel -> (column panel, b1, b2)
  const unsigned bid = blockIdx.x;
  const int cp = bid % (unsigned)a.ncp;
  const unsigned rest = bid / (unsigned)a.ncp;
  const int b1 = rest % (unsigned)a.nb1;
  const int b2 = rest / (unsigned)a.nb1;
  const int c0 = cp * COLS;

  cx<T> v[E];

  // ---------------- stage 0: global load -------------------------------------
  int c, j;
  if constexpr (INC) { j = tid % TPL; c = tid / TPL; }
  else               { c = tid % COLS; j = tid / COLS; }
  {
    const bool valid = (c0 + c) < a.ncols;
    const V2 *src = in + (long long)b1 * a.in_b1 + (long long)b2 * a.in_b2 + (long long)(c0 + c) * a.in_col;
    const int mask = (int)((1u << a.in_shift) - 1u);
    static_for<0, E>([&](auto ii) {
      constexpr int u = decltype(ii)::value / R0, t = decltype(ii)::value % R0;
      const int n = j + u * TPL + t * (N / R0);
      V2 val;
      val.x = 0; val.y = 0;
      if constexpr (R2C) {
        // n real values at the head of the row: element n is the n-th T of the row
        if (valid) val.x = reinterpret_cast<const T *>(src)[n];
        v[decltype(ii)::value] = cx<T>{val.x, (T)0};
      } else {
        if (valid) val = gload(&src[(long long)(n >> a.in_shift) * a.in_blk + (long long)(n & mask) * a.in_axis]);
        v[decltype(ii)::value] = cx<T>{val.x, a.conj ? -val.y : val.y};
      }
    });
  }

  // ---------------- stages ---------------------------------------------------
  static_for<0, NSTAGE>([&](auto sidx) {
    constexpr int s = decltype(sidx)::value;
    constexpr int R = (s == 0) ? R0 : ((s == 1) ? R1 : R2);
    constexpr int Ns = (s == 0) ? 1 : ((s == 1) ? R0 : R0 * R1);
    constexpr int NB = E / R;           // butterflies per thread
    constexpr int LR = ilog2(R);

    if constexpr (s > 0) {
      // inter-stage twiddles w_N^(k * t * N/(Ns*R)), k = q mod Ns
      constexpr int M = N / (Ns * R);
      static_for<0, NB>([&](auto uu) {
        constexpr int u = decltype(uu)::value;
        const int q = j + u * TPL;
        const int km = (q & (Ns - 1)) * M;
        static_for<1, R>([&](auto tt) {
          constexpr int t = decltype(tt)::value;
          const int e = km * t;
          const int qd = e / (N / 4);
          const int r = e & (N / 4 - 1);
          V2 w = tw[r];
          T wr = w.x, wi = w.y;
          // multiply by (-i)^qd
          T cr = (qd & 1) ? wi : wr;
          T ci = (qd & 1) ? -wr : wi;
          if (qd & 2) { cr = -cr; ci = -ci; }
          cx<T> x = v[u * R + t];
          v[u * R + t] = cx<T>{x.x * cr - x.y * ci, x.x * ci + x.y * cr};
        });
      });
    }

    static_for<0, NB>([&](auto uu) { dft_reg<T, R>(&v[decltype(uu)::value * R]); });

    if constexpr (s < NSTAGE - 1) {
      // ---- exchange through LDS: write Stockham-ordered, read strided --------
      constexpr int Rn = (s == 0) ? R1 : R2;      // next radix
      constexpr bool next_last = (s + 1 == NSTAGE - 1);
      int cn, jn;                                  // reader mapping
      if constexpr (next_last && !OUTC) { cn = tid % COLS; jn = tid / COLS; }
      else                              { jn = tid % TPL; cn = tid / TPL; }

      auto wr_idx = [&](int u, int t) {
        const int q = j + u * TPL;
        const int k = q & (Ns - 1);
        return c * LSTRIDE + padidx<SWZ, PS>((q - k) * R + k + t * Ns);
      };
      auto rd_idx = [&](int u, int t) {
        return cn * LSTRIDE + padidx<SWZ, PS>(jn + u * TPL + t * (N / Rn));
      };

      // (the twiddle table written at kernel entry becomes visible at the first
      //  barrier below, before any stage-1 lookup)
      if constexpr (s > 0) __syncthreads();  // previous exchange's reads done
      if constexpr (SPLIT) {
        static_for<0, E>([&](auto ii) {
          constexpr int u = decltype(ii)::value / R, t = decltype(ii)::value % R;
          exs[wr_idx(u, t)] = v[u * R + bitrev(t, LR)].x;
        });
        __syncthreads();
        T re[E];
        static_for<0, E>([&](auto ii) {
          constexpr int u = decltype(ii)::value / Rn, t = decltype(ii)::value % Rn;
          re[decltype(ii)::value] = exs[rd_idx(u, t)];
        });
        __syncthreads();
        static_for<0, E>([&](auto ii) {
          constexpr int u = decltype(ii)::value / R, t = decltype(ii)::value % R;
          exs[wr_idx(u, t)] = v[u * R + bitrev(t, LR)].y;
        });
        __syncthreads();
        static_for<0, E>([&](auto ii) {
          constexpr int u = decltype(ii)::value / Rn, t = decltype(ii)::value % Rn;
          v[decltype(ii)::value] = cx<T>{re[decltype(ii)::value], exs[rd_idx(u, t)]};
        });
      } else {
        static_for<0, E>([&](auto ii) {
          constexpr int u = decltype(ii)::value / R, t = decltype(ii)::value % R;
          cx<T> x = v[u * R + bitrev(t, LR)];
          V2 w; w.x = x.x; w.y = x.y;
          exv[wr_idx(u, t)] = w;
        });
        __syncthreads();
        static_for<0, E>([&](auto ii) {
          constexpr int u = decltype(ii)::value / Rn, t = decltype(ii)::value % Rn;
          V2 w = exv[rd_idx(u, t)];
          v[decltype(ii)::value] = cx<T>{w.x, w.y};
        });
      }
      c = cn; j = jn;
    } else {
      // ---------------- last stage: global store ------------------------------
      const bool valid = (c0 + c) < a.ncols;
      V2 *dst = out + (long long)b1 * a.out_b1 + (long long)b2 * a.out_b2 + (long long)(c0 + c) * a.out_col;
      const int mask = (int)((1u << a.out_shift) - 1u);
      const T sc = (T)a.scale;
      static_for<0, E>([&](auto ii) {
        constexpr int u = decltype(ii)::value / R, t = decltype(ii)::value % R;
        const int n = j + u * TPL + t * (N / R);
        cx<T> x = v[u * R + bitrev(t, LR)];
        V2 w;
        w.x = x.x * sc;
        w.y = (a.conj ? -x.y : x.y) * sc;
        if (valid && (!R2C || n <= N / 2))
          gstore(&dst[(long long)(n >> a.out_shift) * a.out_blk + (long long)(n & mask) * a.out_axis], w);
      });
    }
  });
}

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
// i.e. r complex multiply-adds with exact table twiddles w_N^m (m = 0..N-1,
// staged in LDS when it fits).  Cost N * sum(r_s) per line instead of N^2; a
// prime N degenerates to one stage of radix N (the plain DFT).
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
  int nfac;
  int fac[OFFT_MIX_MAXFAC];
  double scale;
};

__device__ __forceinline__ long long split_off(int k, int split, int nfloor, long long blk, long long axis) {
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
  return (long long)a * blk + (long long)r * axis;
}

template <typename T>
__global__ void __launch_bounds__(256)
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
  const unsigned bid = blockIdx.x;
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
  for (int i = tid; i < nc * N; i += NT) {
    int c, n;
    if (a.in_contig) { n = i % N; c = i / N; } else { c = i % nc; n = i / nc; }
    const V2 *src = in + ibase + (long long)(c0 + c) * a.in_col;
    V2 x;
    if (a.real_in) { x.x = reinterpret_cast<const T *>(src)[n]; x.y = 0; }
    else x = src[split_off(n, a.in_split, a.in_nfloor, a.in_blk, a.in_axis)];
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
    __syncthreads();
    V2 *tmp = x; x = y; y = tmp;
    Ns *= r;
  }

  const int kend = a.real_in ? N / 2 + 1 : N;
  for (int i = tid; i < nc * kend; i += NT) {
    int c, k;
    if (a.out_contig) { k = i % kend; c = i / kend; } else { c = i % nc; k = i / nc; }
    V2 v = x[c * N + k];
    V2 w;
    w.x = v.x * (T)a.scale;
    w.y = (a.conj ? -v.y : v.y) * (T)a.scale;
    V2 *dst = out + obase + (long long)(c0 + c) * a.out_col;
    dst[split_off(k, a.out_split, a.out_nfloor, a.out_blk, a.out_axis)] = w;
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
// host side: kernel registry, twiddle tables, launchers
// ---------------------------------------------------------------------------
struct Variant {
  int n, prec;
  bool inc, outc;
  int id;
  bool is_default;  // default for this (n, prec, inc, outc) flavour
  bool r2c;         // real-input z-pass instantiation
  int cols, threads, e;
  size_t lds;
  const void *fn;
  std::string name;
  bool attr_set;
};

std::vector<Variant> &registry() {
  static std::vector<Variant> r;
  return r;
}

// flavour bits for `defmask`: which (in_contig, out_contig) kernels use this variant by default
enum { F_CC = 1, F_SS = 2, F_CS = 4, F_SC = 8, F_ALL = 15 };

template <typename T, int N, int E, int R0, int R1, int R2, int COLS, bool SPLIT>
void reg_variant(int id, int defmask = -1) {
  if (defmask < 0) defmask = id == 0 ? F_ALL : 0;
  using Cfg = PanelCfg<N, E, R0, R1, R2, COLS, SPLIT, T>;
  const int prec = std::is_same<T, double>::value ? OFFT_PREC_F64 : OFFT_PREC_F32;
  char nm[160];
  snprintf(nm, sizeof nm, "%s N=%d E=%d radix=%dx%dx%d cols=%d %s lds=%zuB", prec ? "f32" : "f64", N, E, R0,
           R1, R2, COLS, SPLIT ? "split-re/im" : "packed", (size_t)Cfg::LDS_BYTES);
  auto add = [&](bool inc, bool outc, int bit, const void *fn, bool r2c = false) {
    registry().push_back(Variant{N, prec, inc, outc, id, (defmask & bit) != 0, r2c, COLS, Cfg::NT, E, Cfg::LDS_BYTES, fn, nm, false});
  };
  add(true, true, F_CC, (const void *)fft_panel_k<T, N, E, R0, R1, R2, COLS, true, true, SPLIT>);
  add(false, false, F_SS, (const void *)fft_panel_k<T, N, E, R0, R1, R2, COLS, false, false, SPLIT>);
  add(true, false, F_CS, (const void *)fft_panel_k<T, N, E, R0, R1, R2, COLS, true, false, SPLIT>);
  add(false, true, F_SC, (const void *)fft_panel_k<T, N, E, R0, R1, R2, COLS, false, true, SPLIT>);
  // real-input z pass: only the contiguous-read flavours of the default variant need it
  if (defmask & F_CC) add(true, true, F_CC, (const void *)fft_panel_k<T, N, E, R0, R1, R2, COLS, true, true, SPLIT, true>, true);
  if (defmask & F_CS) add(true, false, F_CS, (const void *)fft_panel_k<T, N, E, R0, R1, R2, COLS, true, false, SPLIT, true>, true);
}

std::once_flag g_reg_once;
void build_registry() {
  // variant 0 of every length is the default; higher ids are the static sweep
  // (LDS tile width x radix order x re/im split), see DESIGN.md section 5.
#ifdef OFFT_DEV_ONLY_1024  /* developer switch: compile just the 1024 kernels for quick iteration */
  reg_variant<double, 1024, 32, 32, 32, 1, 8, true>(0, 0);
  reg_variant<double, 1024, 16, 16, 16, 4, 8, true>(1, F_ALL);
#ifdef OFFT_DEV_F32
  reg_variant<float, 1024, 32, 32, 32, 1, 16, true>(0, F_SS | F_CS | F_SC);
  reg_variant<float, 1024, 32, 32, 32, 1, 8, false>(1, F_CC);
  reg_variant<float, 1024, 32, 32, 32, 1, 16, false>(2, 0);
  reg_variant<float, 1024, 16, 16, 16, 4, 16, true>(3, 0);
  reg_variant<float, 1024, 16, 16, 16, 4, 16, false>(4, 0);
  reg_variant<float, 2048, 32, 32, 32, 2, 8, true>(0);
  reg_variant<float, 2048, 64, 32, 32, 2, 16, true>(1, 0);
  reg_variant<float, 2048, 32, 32, 32, 2, 4, false>(2, 0);
  reg_variant<float, 2048, 32, 32, 32, 2, 8, false>(3, 0);
  reg_variant<float, 2048, 32, 16, 16, 8, 4, false>(4, 0);
#endif
#ifdef OFFT_DEV_EXTRA
  reg_variant<double, 1024, 16, 4, 16, 16, 8, true>(2, 0);
  reg_variant<double, 1024, 16, 16, 4, 16, 8, true>(3, 0);
  reg_variant<double, 1024, 16, 16, 16, 4, 16, true>(4, 0);
  reg_variant<double, 1024, 16, 16, 16, 4, 4, true>(5, 0);
  reg_variant<double, 1024, 16, 16, 16, 4, 8, false>(6, 0);
  reg_variant<double, 1024, 32, 32, 32, 1, 16, true>(7, 0);
#endif
#else
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
  // static sweep result (profiles/r01_sweep.txt): E=16 (radix 16x16x4, 4 waves/SIMD, no
  // spills) beats E=32 (radix 32x32, one exchange fewer but 256 VGPRs and 2 waves/SIMD)
  // on every flavour at 1024^3, so it is the default; E=32 stays selectable as variant 0.
  reg_variant<double, 1024, 32, 32, 32, 1, 8, true>(0, 0);
  reg_variant<double, 1024, 16, 16, 16, 4, 8, true>(1, F_ALL);
  reg_variant<double, 1024, 32, 32, 32, 1, 4, false>(2, 0);
  reg_variant<double, 2048, 32, 32, 32, 2, 8, true>(0);
  reg_variant<double, 4096, 32, 32, 32, 4, 4, true>(0);
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
  reg_variant<float, 1024, 32, 32, 32, 1, 16, true>(0, F_SS | F_CS | F_SC);
  reg_variant<float, 1024, 32, 32, 32, 1, 8, false>(1, F_CC);
  // 2048 f32 (profiles/r01_sweep.txt): wide 16-column panels need E=64 to stay within 512 threads;
  // the contiguous/contiguous flavour again prefers a narrow packed panel
  reg_variant<float, 2048, 32, 32, 32, 2, 8, true>(0, F_SS);
  reg_variant<float, 2048, 64, 32, 32, 2, 16, true>(1, F_CS | F_SC);
  reg_variant<float, 2048, 32, 32, 32, 2, 4, false>(2, F_CC);
  reg_variant<float, 4096, 32, 32, 32, 4, 4, true>(0);
#endif
}

Variant *find_variant(int n, int prec, bool inc, bool outc, int id, bool r2c = false) {
  std::call_once(g_reg_once, build_registry);
  Variant *def = nullptr;
  for (auto &v : registry()) {
    if (v.n == n && v.prec == prec && v.inc == inc && v.outc == outc && v.r2c == r2c) {
      if (v.id == id) return &v;
      if (v.is_default) def = &v;
    }
  }
  return def;
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

bool fast_ok(const offt_pass_desc *d) {
  if (d->real_input && (!d->in_contig || d->in_axis_stride != 1 || d->in_split || d->direction > 0)) return false;
  if (!find_variant(d->n, d->precision, d->in_contig != 0, d->out_contig != 0, -1, d->real_input != 0)) return false;
  if (d->in_split_nfloor > 0 || d->out_split_nfloor > 0) return false;
  if (d->in_split && !is_pow2(d->in_split)) return false;
  if (d->out_split && !is_pow2(d->out_split)) return false;
  return true;
}

int log2i(int n) { int l = 0; while ((1 << l) < n) ++l; return l; }

}  // namespace

extern "C" {

const char *offt_hipk_last_error(void) { return g_err; }

int offt_hipk_has_fast_path(int n, int precision) {
  return find_variant(n, precision, true, true, -1) != nullptr;
}

int offt_hipk_variant_count(int n, int precision) {
  std::call_once(g_reg_once, build_registry);
  int c = 0;
  for (auto &v : registry())
    if (v.n == n && v.prec == precision && v.inc && v.outc && !v.r2c) c = v.id + 1 > c ? v.id + 1 : c;
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
  if (!fast_ok(d)) return "fft_mixed_k";
  return "fft_panel_k";
}

int offt_hipk_prepare(int n, int precision) {
  if (n < 1) { snprintf(g_err, sizeof g_err, "offt_hipk_prepare: bad n=%d", n); return -1; }
  Tables tb;
  return get_tables(n, precision, tb, true);
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
  if (fast_ok(d)) {
    Variant *v = find_variant(d->n, d->precision, d->in_contig != 0, d->out_contig != 0, d->real_input ? -1 : d->variant, d->real_input != 0);
    PassArgs a;
    a.in_axis = d->in_axis_stride; a.in_col = d->in_col_stride; a.in_b1 = d->in_b1_stride; a.in_b2 = d->in_b2_stride;
    a.out_axis = d->out_axis_stride; a.out_col = d->out_col_stride; a.out_b1 = d->out_b1_stride; a.out_b2 = d->out_b2_stride;
    a.in_blk = d->in_block_stride; a.out_blk = d->out_block_stride;
    a.in_shift = d->in_split ? log2i(d->in_split) : 31;
    a.out_shift = d->out_split ? log2i(d->out_split) : 31;
    a.ncols = d->ncols;
    a.ncp = (d->ncols + v->cols - 1) / v->cols;
    a.nb1 = d->nb1;
    a.conj = d->direction > 0;
    a.scale = d->scale;
    long long nblk = (long long)a.ncp * d->nb1 * d->nb2;
    if (nblk > 0x7fffffffLL) { snprintf(g_err, sizeof g_err, "offt_hipk_fft_pass: grid too large"); return -1; }
    if (!v->attr_set) {
      if (v->lds > 48 * 1024)
        HIPK_CHECK(hipFuncSetAttribute(v->fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)v->lds));
      v->attr_set = true;
    }
    void *args[] = {(void *)&a, (void *)&in, (void *)&out, (void *)&tb.quarter};
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
  (void)hipGetLastError();  // start from a clean slate: the check below must see only this launch
  if (d->precision == OFFT_PREC_F64) {
    static bool set64 = false;
    if (lds > 48 * 1024 && !set64) {
      HIPK_CHECK(hipFuncSetAttribute((const void *)fft_mixed_k<double>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_cap));
      set64 = true;
    }
    hipLaunchKernelGGL(fft_mixed_k<double>, dim3((unsigned)nblk), dim3(256), lds, st, g, (const double2 *)in,
                       (double2 *)out, (const double2 *)tb.full);
  } else {
    static bool set32 = false;
    if (lds > 48 * 1024 && !set32) {
      HIPK_CHECK(hipFuncSetAttribute((const void *)fft_mixed_k<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_cap));
      set32 = true;
    }
    hipLaunchKernelGGL(fft_mixed_k<float>, dim3((unsigned)nblk), dim3(256), lds, st, g, (const float2 *)in,
                       (float2 *)out, (const float2 *)tb.full);
  }
  HIPK_CHECK(hipGetLastError());
  return 0;
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
