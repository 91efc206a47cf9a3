// offt_panel.hpp -- hand-written CDNA4 (gfx950) kernels for the OFFT hot path.
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

//
// This header holds the device templates and the variant registry helpers; the
// instantiations live in offt_reg_*.hip (one translation unit per group so that
// the build runs in parallel), the C ABI in offt_kernels.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include <vector>
#include <string>
#include <cstdio>
#include "offt_hipk.h"
#include "offt_w32_consts.h"
#include "offt_wr_consts.h"

namespace offtk {

template <typename T> struct vec2;
template <> struct vec2<double> { using type = double2; };
template <> struct vec2<float> { using type = float2; };

template <typename T> struct cx { T x, y; };

// COLUMN PAIRS (single precision).  T = f32x2 runs the same kernel with two adjacent columns per lane: every register
// value, butterfly operation and LDS word carries the pair (v_pk_* arithmetic, 8-B LDS words -- the double-precision
// kernel's instruction count for twice the elements), a strided side moves 16 B per lane (one 128-B segment = 8 lanes, as in
// double precision), twiddles and addresses are computed once per pair.  A contiguous side still moves 8 B per lane, one
// access per column.  Needs an even column count and, on a strided side, a unit column stride with even other strides
// (16-B alignment): pair_ok() in offt_kernels.hip; anything else runs on the one-column kernels.
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <> struct vec2<f32x2> { using type = float2; };  // memory and twiddle tables hold plain complex floats
template <typename T> struct lanes { using scalar = T; static constexpr int n = 1; };
template <> struct lanes<f32x2> { using scalar = float; static constexpr int n = 2; };

// global memory access helpers: 16-B (f64) / 8-B (f32) per lane.  Every element is
// touched exactly once per pass, so loads and stores are non-temporal (streaming): A/B on
// 1024^3 (profiles/r01_sweep.txt): -6 % transform time vs default cache policy.
template <typename V2>
__device__ __forceinline__ V2 gload(const V2 *p) {
  using E = decltype(p->x);
  typedef E vt __attribute__((ext_vector_type(2)));
  vt r = __builtin_nontemporal_load(reinterpret_cast<const vt *>(p));
  V2 o; o.x = r.x; o.y = r.y; return o;
}
template <typename V2>
__device__ __forceinline__ void gstore(V2 *p, V2 v) {
  using E = decltype(p->x);
  typedef E vt __attribute__((ext_vector_type(2)));
  vt r; r.x = v.x; r.y = v.y;
  __builtin_nontemporal_store(r, reinterpret_cast<vt *>(p));
}

template <bool KEEP, typename V2>
__device__ __forceinline__ void gstore_p(V2 *p, V2 v) {
  if constexpr (KEEP) *p = v;
  else gstore(p, v);
}

// v with its sign bit XORed by m (m = 0 or 0x80000000, uniform): conjugation and half-wave twiddle signs cost one
// 32-bit XOR instead of a select + negate
__device__ __forceinline__ double xor_sign(double v, unsigned m) { return __hiloint2double(__double2hiint(v) ^ (int)m, __double2loint(v)); }
__device__ __forceinline__ float xor_sign(float v, unsigned m) { return __int_as_float(__float_as_int(v) ^ (int)m); }
__device__ __forceinline__ f32x2 xor_sign(f32x2 v, unsigned m) { f32x2 r; r.x = xor_sign((float)v.x, m); r.y = xor_sign((float)v.y, m); return r; }

// i / d for 0 <= i < 2^22 with inv = 1.0f / d: float estimate, one correction step each way
__device__ __forceinline__ int fdiv(int i, int d, float inv) {
  int q = (int)((float)i * inv);
  int r = i - q * d;
  if (r < 0) q--;
  else if (r >= d) q++;
  return q;
}

// Workgroups are dealt round-robin to the 8 XCDs (each with its own L2 and address translation cache), so in
// launch order neighbouring panels land on different XCDs.  Renumbering gives every XCD runs of G = 2^gshift
// panels that are neighbours in memory: block b (XCD b % 8, its r-th block) takes panel
// (r / G) * 8G + (b % 8) * G + r % G.  Blocks at or above lim (the tail that does not fill 8 G) keep their index.
// 1024^3 f64: 17.65 -> 17.0 ms per transform (profiles/r01_sweep.txt).
__device__ __forceinline__ unsigned panel_of_block(unsigned bid, unsigned lim, unsigned gshift) {
  if (bid >= lim) return bid;
  const unsigned x = bid & 7u, r = bid >> 3;
  return ((r >> gshift) << (gshift + 3u)) + (x << gshift) + (r & ((1u << gshift) - 1u));
}

// element offset of axis index n under a per-peer split (SPLIT = false: none): indices below lim = split * nfloor
// sit in blocks of `split`, the others in blocks of split + 1 -- the reference's uneven A2AV partition
// (offt-compute.c:132-144); an even split is the case lim >= N.
// `tab` (offt_pass_desc::in_block_tab / out_block_tab): per-block element offsets replacing blk * blk_stride, or nullptr.
template <bool SPLIT>
__device__ __forceinline__ long long split_offset(int n, int split, float inv, int nfloor, int lim, float inv1, long long blk_stride,
                                                  long long axis_stride, const long long *tab = nullptr) {
  if constexpr (!SPLIT) return (long long)n * axis_stride;
  else {
    int blk, rem;
    if (n < lim) { blk = fdiv(n, split, inv); rem = n - blk * split; }
    else { const int m = n - lim, b = fdiv(m, split + 1, inv1); blk = nfloor + b; rem = m - b * (split + 1); }
    return (tab ? tab[blk] : (long long)blk * blk_stride) + (long long)rem * axis_stride;
  }
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
constexpr double WRC[33][32] = OFFT_WR_COS;
constexpr double WRS[33][32] = OFFT_WR_SIN;

// d * w32^K, w32 = exp(-2 pi i / 32)
template <typename T, int K>
__device__ __forceinline__ cx<T> mulw32(cx<T> d) {
  using S = typename lanes<T>::scalar;
  constexpr int k = K & 31;
  if constexpr (k == 0) return d;
  else if constexpr (k == 8) return cx<T>{d.y, -d.x};
  else if constexpr (k == 16) return cx<T>{-d.x, -d.y};
  else if constexpr (k == 24) return cx<T>{-d.y, d.x};
  else if constexpr (k == 4) {
    constexpr S s = (S)W32C[4];
    return cx<T>{(d.x + d.y) * s, (d.y - d.x) * s};
  } else if constexpr (k == 12) {
    constexpr S s = (S)W32C[4];
    return cx<T>{(d.y - d.x) * s, -(d.x + d.y) * s};
  } else {
    constexpr S c = (S)W32C[k], s = (S)W32S[k];
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
  int in_shift, out_shift;  // log2(split) or 31 for "no split"            (fft_panel_k)
  int in_split, out_split;  // split length, any value, 0 for "no split"   (fft_panelx_k)
  float in_inv, out_inv;    // 1 / split
  int in_nfloor, out_nfloor;  // uneven A2AV partition (offt-compute.c:132-144): the first nfloor blocks hold `split`
  int in_lim, out_lim;        //   indices (axis indices below lim = split * nfloor), the others split + 1;
  float in_inv1, out_inv1;    //   inv1 = 1 / (split + 1).  Even split: lim >= n.
  int ncols, ncp, nb1;      // ncp = column panels per batch entry
  int conj;                 // 1: inverse transform via conj-in / conj-out
  unsigned xcd_lim;         // XCD-aware panel order for blocks below this index (see panel_of_block), 0 = off
  unsigned xcd_gshift;      // log2 G, G = run of neighbouring panels one XCD takes
  double scale;
  const long long *in_tab, *out_tab;  // per-block element offsets (offt_pass_desc::in_block_tab / out_block_tab) or nullptr
  const void *tw4;          // TW4 kernels (first sub-pass of a four-step line): w^(k1 j2) of the LONG length as a table [k1][j2];
  int tw4_b1;               //   output index k1 of column j2 (tw4_b1 = 0) or of batch entry j2 = b1 (tw4_b1 = 1) times w^(k1 j2)
  int tw4_n2;               //   row length of that table
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
  static constexpr size_t EX_BYTES =
      NSTAGE > 1 ? (size_t)COLS * LSTRIDE * sizeof(T) * (SPLIT ? 1 : 2) : 0;
  static constexpr size_t TW_OFF = (EX_BYTES + 15) / 16 * 16;
  // Inter-stage twiddles, staged in LDS from the exact full-wave table in global memory:
  //  * one table for all stages: the QUARTER wave (N/4+1 entries; the other three quadrants by swap / sign, ~10 integer
  //    instructions per twiddle) or the HALF wave (N/2 entries; w^(e + N/2) = -w^e is one XOR) -- the half wave whenever it
  //    does not cost a workgroup per CU;
  //  * a COMPACT table for stage 1, tw1[t-1][k] = w^(k t N/(R0 R1)), k < R0, t < R1: the 16 lanes of an LDS read group
  //    then read 16 consecutive entries instead of entries 16 t N/(R0 R1) bytes apart (the r02 PMC pass showed
  //    SQ_LDS_BANK_CONFLICT = 55-60 % of SQ_LDS_IDX_ACTIVE on the 2048-point kernels, from exactly these reads) -- again
  //    only when it does not cost a workgroup per CU.
  static constexpr int QTQ = (N >= 4) ? N / 4 + 1 : 1, QTH = N / 2, T1N = NSTAGE > 1 ? R0 * (R1 - 1) : 0;
  static constexpr size_t lds_with(int shared_entries, int t1_entries) {
    return NSTAGE > 1 ? TW_OFF + ((size_t)shared_entries + (size_t)t1_entries) * 2 * sizeof(typename lanes<T>::scalar) : 0;
  }
  static constexpr int wg_for(size_t lds) { return lds ? (int)(160 * 1024 / lds) : 8; }
  static constexpr bool USE_T1 = NSTAGE > 1 && wg_for(lds_with(QTQ, T1N)) == wg_for(lds_with(QTQ, 0));
  static constexpr bool USE_HALF = NSTAGE > 1 && N >= 16 && wg_for(lds_with(QTH, USE_T1 ? T1N : 0)) == wg_for(lds_with(QTQ, 0));
  static constexpr int QT = USE_HALF ? QTH : QTQ;
  static constexpr size_t T1_OFF = TW_OFF + (size_t)QT * 2 * sizeof(typename lanes<T>::scalar);
  static constexpr size_t LDS_BYTES = lds_with(QT, USE_T1 ? T1N : 0);
  // occupancy target handed to __launch_bounds__ (2nd argument = waves per
  // SIMD): as many workgroups per CU as the 160 KiB LDS admits, at most 4
  // waves per SIMD -- enough to overlap one group's butterflies with another
  // group's HBM traffic without starving the register allocator.
  static constexpr int WG_PER_CU_LDS = wg_for(LDS_BYTES);
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

template <typename T, int N, int E, int R0, int R1, int R2, int COLS, bool INC, bool OUTC, bool SPLIT, bool R2C = false,
          bool KEEP = false /* stores with the default cache policy: the next launch re-reads the output (out_keep) */,
          bool TW4 = false /* four-step lines (offt_kernels.hip): the twiddles w_n^(j2 k1) of the long length ride on the stores */>
__global__ void __launch_bounds__((N / E) * COLS, (PanelCfg<N, E, R0, R1, R2, COLS, SPLIT, T>::WPS_E))
fft_panel_k(PassArgs a, const typename vec2<T>::type *in, typename vec2<T>::type *out,
            const typename vec2<T>::type *twq) {
  using V2 = typename vec2<T>::type;
  using S = typename lanes<T>::scalar;
  constexpr int NL = lanes<T>::n;   // memory columns per lane (2: column pairs)
  constexpr bool PAIR = NL == 2;
  using XV = std::conditional_t<PAIR, f32x4, V2>;  // packed exchange word: (re, im) of the lane's column(s)
  using Cfg = PanelCfg<N, E, R0, R1, R2, COLS, SPLIT, T>;
  constexpr int TPL = Cfg::TPL, NT = Cfg::NT, NSTAGE = Cfg::NSTAGE;
  constexpr int LSTRIDE = Cfg::LSTRIDE;
  constexpr bool SWZ = Cfg::SWZ;
  constexpr int PS = SWZ ? Cfg::SWZSHIFT : Cfg::PADSHIFT;
  static_assert(R0 * R1 * R2 == N, "radices must multiply to N");
  static_assert(E % R0 == 0 && E % R1 == 0 && E % R2 == 0 && N % E == 0, "bad E");
  static_assert(!(PAIR && R2C), "column pairs: complex input only");
  static_assert(!(PAIR && TW4), "four-step twiddles: one column per lane");

  extern __shared__ __align__(16) unsigned char smem[];
  T *exs = reinterpret_cast<T *>(smem);
  XV *exv = reinterpret_cast<XV *>(smem);
  V2 *tw = reinterpret_cast<V2 *>(smem + Cfg::TW_OFF);
  V2 *tw1 = reinterpret_cast<V2 *>(smem + Cfg::T1_OFF);

  const int tid = threadIdx.x;
  if constexpr (NSTAGE > 1) {
    // twq = the exact full-wave table w^m, m < N, in global memory (L2-resident: every workgroup reads it)
    for (int i = tid; i < Cfg::QT; i += NT) tw[i] = twq[i];
    if constexpr (Cfg::USE_T1) {
      constexpr int M1 = N / (R0 * R1);
      for (int i = tid; i < Cfg::T1N; i += NT) {
        const int t = i / R0 + 1, k = i - (t - 1) * R0;
        tw1[i] = twq[k * M1 * t];  // k t M1 <= (R0-1)(R1-1) M1 < N
      }
    }
  }

  // panel -> (column panel, b1, b2)
  const unsigned bid = panel_of_block(blockIdx.x, a.xcd_lim, a.xcd_gshift);
  const int cp = bid % (unsigned)a.ncp;
  const unsigned rest = bid / (unsigned)a.ncp;
  const int b1 = rest % (unsigned)a.nb1;
  const int b2 = rest / (unsigned)a.nb1;
  const int c0 = cp * COLS * NL;
  const unsigned conj_mask = a.conj ? 0x80000000u : 0u;  // inverse transform = conj-in / conj-out: one XOR per element

  cx<T> v[E];

  // ---------------- stage 0: global load -------------------------------------
  // Element (u, t) of this thread is axis index n = j + cn with j < TPL and cn = u TPL + t N/R0, a compile-time multiple
  // of TPL.  A per-peer split has a power-of-two length F = 2^shift (any other split runs on fft_panelx_k): for F >= TPL
  // cn's offset bits plus j stay below F, for F < TPL cn has no offset bits at all -- either way block index and offset
  // of n are the SUMS of those of j and cn.  So the address is (a per-lane base for j) + (a wave-uniform offset for cn):
  // the offsets are scalar-unit work and an element costs a 64-bit add instead of two 64-bit multiply-adds (the ISA of the
  // r01 kernels had 130 v_mad_u64_u32 per thread).
  int c, j;
  if constexpr (INC) { j = tid % TPL; c = tid / TPL; }
  else               { c = tid % COLS; j = tid / COLS; }
  {
    const bool valid = (c0 + c * NL) < a.ncols;  // (pairs: the host sends even column counts only)
    // pairs: lanes past the last column re-read the panel's first pair instead of being predicated off (a predicated
    // load whose components are then regrouped into register pairs compiled to one branch and one full wait PER LOAD);
    // what they compute is never stored
    const int cl = (PAIR && !valid) ? 0 : c;
    const V2 *src = in + (long long)b1 * a.in_b1 + (long long)b2 * a.in_b2 + (long long)(c0 + cl * NL) * a.in_col;
    const int mask = (int)((1u << a.in_shift) - 1u);
    // TAB: the blocks of the split come from a table of element offsets (in_block_tab) instead of blk * in_blk; block
    // index and offset of n = j + cn are still the sums of those of j and cn, so the lookup index is (j >> shift) + (cn >> shift)
    auto load_all = [&](auto tabbed) {
      constexpr bool TAB = decltype(tabbed)::value;
      const int jb = j >> a.in_shift;
      const V2 *p0 = src + (TAB ? 0LL : (long long)jb * a.in_blk) + (long long)(j & mask) * a.in_axis;
      static_for<0, E>([&](auto ii) {
        constexpr int u = decltype(ii)::value / R0, t = decltype(ii)::value % R0;
        constexpr int cn = u * TPL + t * (N / R0);
        const int n = j + cn;
        V2 val;
        val.x = 0; val.y = 0;
        if constexpr (R2C) {
          // n real values at the head of the row: element n is the n-th T of the row
          if constexpr (!PAIR) {
            if (valid) val.x = reinterpret_cast<const T *>(src)[n];
            v[decltype(ii)::value] = cx<T>{val.x, (T)0};
          }
        } else {
          long long off = (long long)(cn & mask) * a.in_axis;  // uniform
          if constexpr (TAB) off += a.in_tab[jb + (cn >> a.in_shift)];
          else off += (long long)(cn >> a.in_shift) * a.in_blk;  // uniform
          if constexpr (PAIR) {
            T re, im;
            if constexpr (INC) {  // lanes along the line: one 8-B access per column
              const V2 w0 = gload(p0 + off), w1 = gload(p0 + off + a.in_col);
              re.x = w0.x; re.y = w1.x; im.x = w0.y; im.y = w1.y;
            } else {              // lanes across columns: the pair is 16 contiguous bytes
              const f32x4 q = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(p0 + off));
              re.x = q.x; im.x = q.y; re.y = q.z; im.y = q.w;
            }
            v[decltype(ii)::value] = cx<T>{re, xor_sign(im, conj_mask)};
          } else {
            if (valid) val = gload(p0 + off);
            v[decltype(ii)::value] = cx<T>{val.x, xor_sign(val.y, conj_mask)};
          }
        }
      });
    };
    if (a.in_tab) load_all(std::true_type{});
    else load_all(std::false_type{});
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
      // the twiddle tables staged at kernel entry must be visible: exchange 0 had a workgroup barrier unless it was
      // wave-private
      if constexpr (s == 1 && (64 % TPL == 0) && (NT % 64 == 0) && INC && !(NSTAGE == 2 && !OUTC)) __syncthreads();
      static_for<0, NB>([&](auto uu) {
        constexpr int u = decltype(uu)::value;
        const int q = j + u * TPL;
        const int km = (q & (Ns - 1)) * M;
        static_for<1, R>([&](auto tt) {
          constexpr int t = decltype(tt)::value;
          S cr, ci;
          if constexpr (s == 1 && Cfg::USE_T1) {
            const V2 w = tw1[(t - 1) * R0 + (q & (R0 - 1))];  // consecutive lanes, consecutive entries
            cr = w.x; ci = w.y;
          } else if constexpr (Cfg::USE_HALF) {
            const int e = km * t;                              // < N
            const V2 w = tw[e & (N / 2 - 1)];
            const unsigned sm = ((unsigned)e << (32 - ilog2(N))) & 0x80000000u;  // bit log2(N)-1 of e: w^(e) = -w^(e - N/2)
            cr = xor_sign(w.x, sm); ci = xor_sign(w.y, sm);
          } else {
            const int e = km * t;
            const int qd = e / (N / 4);
            const int r = e & (N / 4 - 1);
            V2 w = tw[r];
            S wr = w.x, wi = w.y;
            // multiply by (-i)^qd
            cr = (qd & 1) ? wi : wr;
            ci = (qd & 1) ? -wr : wi;
            if (qd & 2) { cr = -cr; ci = -ci; }
          }
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

      // Synchronisation.  With lanes along the line (j = tid % TPL) and TPL a divisor of 64, a wave owns whole columns: it
      // writes and reads only its own columns' LDS image, DS operations of one wave execute in order, and no workgroup
      // barrier is needed -- waves drift apart and one wave's butterflies overlap another's LDS traffic.  Only an exchange
      // whose writer (stage 0 of a strided-in pass) or reader (last stage of a strided-out pass) runs lanes across
      // columns needs s_barrier.
      constexpr bool WAVE_COLS = (64 % TPL == 0) && (NT % 64 == 0);
      constexpr bool PRIV_W = WAVE_COLS && (s > 0 || INC);                 // writer mapping is wave-private
      constexpr bool PRIV = PRIV_W && !(next_last && !OUTC);              // ... and so is the reader's
      auto xsync = [&](auto priv) {
        if constexpr (decltype(priv)::value) {
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
          __builtin_amdgcn_wave_barrier();
        } else {
          __syncthreads();
        }
      };
      // previous exchange's reads done (its reader mapping is this exchange's writer mapping)
      if constexpr (s > 0) xsync(std::integral_constant<bool, PRIV_W>{});
      constexpr std::integral_constant<bool, PRIV> priv{};
      if constexpr (SPLIT) {
        static_for<0, E>([&](auto ii) {
          constexpr int u = decltype(ii)::value / R, t = decltype(ii)::value % R;
          exs[wr_idx(u, t)] = v[u * R + bitrev(t, LR)].x;
        });
        xsync(priv);
        T re[E];
        static_for<0, E>([&](auto ii) {
          constexpr int u = decltype(ii)::value / Rn, t = decltype(ii)::value % Rn;
          re[decltype(ii)::value] = exs[rd_idx(u, t)];
        });
        xsync(priv);
        static_for<0, E>([&](auto ii) {
          constexpr int u = decltype(ii)::value / R, t = decltype(ii)::value % R;
          exs[wr_idx(u, t)] = v[u * R + bitrev(t, LR)].y;
        });
        xsync(priv);
        static_for<0, E>([&](auto ii) {
          constexpr int u = decltype(ii)::value / Rn, t = decltype(ii)::value % Rn;
          v[decltype(ii)::value] = cx<T>{re[decltype(ii)::value], exs[rd_idx(u, t)]};
        });
      } else {
        static_for<0, E>([&](auto ii) {
          constexpr int u = decltype(ii)::value / R, t = decltype(ii)::value % R;
          cx<T> x = v[u * R + bitrev(t, LR)];
          XV w;
          if constexpr (PAIR) { w.x = x.x.x; w.y = x.x.y; w.z = x.y.x; w.w = x.y.y; }
          else { w.x = x.x; w.y = x.y; }
          exv[wr_idx(u, t)] = w;
        });
        xsync(priv);
        static_for<0, E>([&](auto ii) {
          constexpr int u = decltype(ii)::value / Rn, t = decltype(ii)::value % Rn;
          XV w = exv[rd_idx(u, t)];
          if constexpr (PAIR) { T re = {w.x, w.y}, im = {w.z, w.w}; v[decltype(ii)::value] = cx<T>{re, im}; }
          else v[decltype(ii)::value] = cx<T>{w.x, w.y};
        });
      }
      c = cn; j = jn;
    } else {
      // ---------------- last stage: global store ------------------------------
      const bool valid = (c0 + c * NL) < a.ncols;
      V2 *dst = out + (long long)b1 * a.out_b1 + (long long)b2 * a.out_b2 + (long long)(c0 + c * NL) * a.out_col;
      const int mask = (int)((1u << a.out_shift) - 1u);
      const S sc = (S)a.scale;
      const S scy = a.conj ? -sc : sc;  // conj-out rides on the scale
      auto store_all = [&](auto tabbed) {
        constexpr bool TAB = decltype(tabbed)::value;  // blocks from out_block_tab (see the load side)
        const int jb = j >> a.out_shift;
        V2 *p0 = dst + (TAB ? 0LL : (long long)jb * a.out_blk) + (long long)(j & mask) * a.out_axis;
        static_for<0, E>([&](auto ii) {
          constexpr int u = decltype(ii)::value / R, t = decltype(ii)::value % R;
          constexpr int cn = u * TPL + t * (N / R);
          const int n = j + cn;
          cx<T> x = v[u * R + bitrev(t, LR)];
          if constexpr (TW4) {
            // times w^(k1 j2) of the long line, before the conj-out / scale below (for the inverse conj(v w) = conj(v) conj(w))
            const long long jj = a.tw4_b1 ? b1 : (c0 + c);
            const V2 w = reinterpret_cast<const V2 *>(a.tw4)[(long long)n * a.tw4_n2 + jj];
            x = cx<T>{x.x * w.x - x.y * w.y, x.x * w.y + x.y * w.x};
          }
          long long off = (long long)(cn & mask) * a.out_axis;  // uniform
          if constexpr (TAB) off += a.out_tab[jb + (cn >> a.out_shift)];
          else off += (long long)(cn >> a.out_shift) * a.out_blk;  // uniform
          if constexpr (PAIR) {
            const T wx = x.x * sc, wy = x.y * scy;
            if constexpr (OUTC) {
              V2 w0, w1;
              w0.x = wx.x; w0.y = wy.x; w1.x = wx.y; w1.y = wy.y;
              if (valid) gstore_p<KEEP>(p0 + off, w0);
              if (valid) gstore_p<KEEP>(p0 + off + a.out_col, w1);
            } else {
              f32x4 q;
              q.x = wx.x; q.y = wy.x; q.z = wx.y; q.w = wy.y;
              if constexpr (KEEP) { if (valid) *reinterpret_cast<f32x4 *>(p0 + off) = q; }
              else { if (valid) __builtin_nontemporal_store(q, reinterpret_cast<f32x4 *>(p0 + off)); }
            }
          } else {
            V2 w;
            w.x = x.x * sc;
            w.y = x.y * scy;
            if (valid && (!R2C || n <= N / 2)) gstore_p<KEEP>(p0 + off, w);
          }
        });
      };
      if (a.out_tab) store_all(std::true_type{});
      else store_all(std::false_type{});
    }
  });
}

// ---------------------------------------------------------------------------
// Mixed-radix panel kernel: the same three-stage register/LDS Stockham scheme for
// lengths N = R0 * R1 * R2 whose radices are products of small primes (2, 3, 5 with hand-written
// butterflies, 7, 11, 13 as direct DFTs): 768 = 12 x 8 x 8, 1000 = 10 x 10 x 10, 896 = 8 x 8 x 14, ...  Differences from fft_panel_k:
//  * the register butterfly is a mixed-radix decimation in frequency (prime steps with
//    compile-time twiddles w_R^k); its output order is the digit-reversal perm_mixed;
//  * TPL threads share a line and a thread owns ceil((N/R)/TPL) butterflies of a stage, the
//    last one predicated when TPL does not divide N/R, so the register count may differ from
//    stage to stage (the exchange goes through LDS anyway);
//  * index arithmetic uses division by compile-time constants instead of masks; per-peer
//    splits of any length (N/p is rarely a power of two here) use a float-reciprocal divide;
//  * LDS image: the XOR swizzle when R0 is a multiple of 16, else one pad element every R0 when
//    R0 is even (stage-0 write stride becomes odd), none when R0 is odd.
// ---------------------------------------------------------------------------
constexpr int first_factor(int r) {
  return r % 2 == 0 ? 2 : (r % 3 == 0 ? 3 : (r % 5 == 0 ? 5 : (r % 7 == 0 ? 7 : (r % 11 == 0 ? 11 : (r % 13 == 0 ? 13 : r)))));
}
// register radices (<= 32): products of the primes 2 .. 13, or one of the primes 17 .. 31 (a radix <= 32 with such a
// factor is that prime itself; first_factor returns it)
constexpr bool smooth235(int r) { return r <= 1 ? true : (first_factor(r) <= 31 && smooth235(r / first_factor(r))); }
// X[k] of dft_mixed<R> is left in v[perm_mixed(R, k)]
constexpr int perm_mixed(int r, int k) {
  if (r <= 1) return 0;
  const int p = first_factor(r), m = r / p;
  return m * (k % p) + perm_mixed(m, k / p);
}

// d * w_R^K, w_R = exp(-2 pi i / R)
template <typename T, int R, int K>
__device__ __forceinline__ cx<T> mulwr(cx<T> d) {
  constexpr int k = ((K % R) + R) % R;
  if constexpr (k == 0) return d;
  else if constexpr (4 * k == R) return cx<T>{d.y, -d.x};
  else if constexpr (2 * k == R) return cx<T>{-d.x, -d.y};
  else if constexpr (4 * k == 3 * R) return cx<T>{-d.y, d.x};
  else if constexpr (8 * k == R) {
    constexpr T s = (T)W32C[4];
    return cx<T>{(d.x + d.y) * s, (d.y - d.x) * s};
  } else if constexpr (8 * k == 3 * R) {
    constexpr T s = (T)W32C[4];
    return cx<T>{(d.y - d.x) * s, -(d.x + d.y) * s};
  } else {
    constexpr T c = (T)WRC[R][k], s = (T)WRS[R][k];
    return cx<T>{d.x * c + d.y * s, d.y * c - d.x * s};
  }
}

// In-register DFT of R = 2^a 3^b 5^c points, decimation in frequency by the smallest prime p
// (R = p m):  y_d[b] = w_R^(b d) * sum_a x[m a + b] w_p^(a d)  stored at v[m d + b], then a
// DFT of length m on every block d.  X[d + p c] ends in v[m d + perm_mixed(m, c)].
template <typename T, int R>
__device__ __forceinline__ void dft_mixed(cx<T> *v) {
  if constexpr (R > 1) {
    constexpr int p = first_factor(R), m = R / p;
    static_assert(p == 2 || p == 3 || p == 5 || p == 7 || p == 11 || p == 13 || p == 17 || p == 19 || p == 23 || p == 29 || p == 31,
                  "register radix must be a product of primes <= 13 or a prime <= 31");
    static_for<0, m>([&](auto bb) {
      constexpr int b = decltype(bb)::value;
      if constexpr (p == 2) {
        const cx<T> x0 = v[b], x1 = v[m + b];
        v[b] = cx<T>{x0.x + x1.x, x0.y + x1.y};
        v[m + b] = mulwr<T, R, b>(cx<T>{x0.x - x1.x, x0.y - x1.y});
      } else if constexpr (p == 3) {
        constexpr T S3 = (T)WRS[3][1];
        const cx<T> x0 = v[b], x1 = v[m + b], x2 = v[2 * m + b];
        const cx<T> sm{x1.x + x2.x, x1.y + x2.y}, df{x1.x - x2.x, x1.y - x2.y};
        const cx<T> t{x0.x - (T)0.5 * sm.x, x0.y - (T)0.5 * sm.y};
        const cx<T> e{S3 * df.y, -S3 * df.x};  // -i sin(2 pi/3) (x1 - x2)
        v[b] = cx<T>{x0.x + sm.x, x0.y + sm.y};
        v[m + b] = mulwr<T, R, b>(cx<T>{t.x + e.x, t.y + e.y});
        v[2 * m + b] = mulwr<T, R, 2 * b>(cx<T>{t.x - e.x, t.y - e.y});
      } else if constexpr (p == 5) {
        constexpr T C1 = (T)WRC[5][1], C2 = (T)WRC[5][2], S1 = (T)WRS[5][1], S2 = (T)WRS[5][2];
        const cx<T> x0 = v[b], x1 = v[m + b], x2 = v[2 * m + b], x3 = v[3 * m + b], x4 = v[4 * m + b];
        const cx<T> s1{x1.x + x4.x, x1.y + x4.y}, s2{x2.x + x3.x, x2.y + x3.y};
        const cx<T> d1{x1.x - x4.x, x1.y - x4.y}, d2{x2.x - x3.x, x2.y - x3.y};
        const cx<T> p1{x0.x + C1 * s1.x + C2 * s2.x, x0.y + C1 * s1.y + C2 * s2.y};
        const cx<T> p2{x0.x + C2 * s1.x + C1 * s2.x, x0.y + C2 * s1.y + C1 * s2.y};
        const cx<T> q1{S1 * d1.x + S2 * d2.x, S1 * d1.y + S2 * d2.y};
        const cx<T> q2{S2 * d1.x - S1 * d2.x, S2 * d1.y - S1 * d2.y};
        v[b] = cx<T>{x0.x + s1.x + s2.x, x0.y + s1.y + s2.y};
        v[m + b] = mulwr<T, R, b>(cx<T>{p1.x + q1.y, p1.y - q1.x});          // p1 - i q1
        v[2 * m + b] = mulwr<T, R, 2 * b>(cx<T>{p2.x + q2.y, p2.y - q2.x});  // p2 - i q2
        v[3 * m + b] = mulwr<T, R, 3 * b>(cx<T>{p2.x - q2.y, p2.y + q2.x});  // p2 + i q2
        v[4 * m + b] = mulwr<T, R, 4 * b>(cx<T>{p1.x - q1.y, p1.y + q1.x});  // p1 + i q1
      } else {
        // any odd prime p (7, 11, 13; 17 .. 31 in plan-time instances): X_k = x0 + sum_j cos(2 pi j k/p) s_j - i sum_j sin(2 pi j k/p) d_j with
        // s_j = x_j + x_(p-j), d_j = x_j - x_(p-j), j = 1..(p-1)/2; X_(p-k) is the same with + i.  The path is
        // HBM-bound, so the h^2 multiply-adds are not worth a Winograd factorisation.
        constexpr int h = (p - 1) / 2;
        const cx<T> x0 = v[b];
        cx<T> sj[h], dj[h];
        static_for<0, h>([&](auto jj) {
          constexpr int j = decltype(jj)::value + 1;
          const cx<T> xa = v[j * m + b], xb = v[(p - j) * m + b];
          sj[j - 1] = cx<T>{xa.x + xb.x, xa.y + xb.y};
          dj[j - 1] = cx<T>{xa.x - xb.x, xa.y - xb.y};
        });
        cx<T> sum = x0;
        static_for<0, h>([&](auto jj) { sum.x += sj[decltype(jj)::value].x; sum.y += sj[decltype(jj)::value].y; });
        v[b] = sum;
        static_for<0, h>([&](auto kk) {
          constexpr int k = decltype(kk)::value + 1;
          cx<T> pc = x0, qs{(T)0, (T)0};
          static_for<0, h>([&](auto jj) {
            constexpr int j = decltype(jj)::value + 1;
            constexpr T c = (T)WRC[p][(j * k) % p], sn = (T)WRS[p][(j * k) % p];
            pc.x += c * sj[j - 1].x; pc.y += c * sj[j - 1].y;
            qs.x += sn * dj[j - 1].x; qs.y += sn * dj[j - 1].y;
          });
          v[k * m + b] = mulwr<T, R, k * b>(cx<T>{pc.x + qs.y, pc.y - qs.x});              // pc - i qs
          v[(p - k) * m + b] = mulwr<T, R, (p - k) * b>(cx<T>{pc.x - qs.y, pc.y + qs.x});  // pc + i qs
        });
      }
    });
    static_for<0, p>([&](auto dd) { dft_mixed<T, m>(v + decltype(dd)::value * m); });
  }
}

constexpr int cdiv(int a, int b) { return (a + b - 1) / b; }
constexpr int cmax(int a, int b) { return a > b ? a : b; }

template <int N, int TPL, int R0, int R1, int R2, int COLS, bool SPLIT, typename T>
struct PanelXCfg {
  static constexpr int NT = TPL * COLS;
  static constexpr int NSTAGE = (R2 > 1) ? 3 : ((R1 > 1) ? 2 : 1);
  static constexpr int NB0 = cdiv(N / R0, TPL), NB1 = cdiv(N / R1, TPL), NB2 = cdiv(N / R2, TPL);
  static constexpr int EMAX = cmax(NB0 * R0, cmax(R1 > 1 ? NB1 * R1 : 0, R2 > 1 ? NB2 * R2 : 0));
  // LDS image of one column (see PanelCfg): R0 a multiple of 16 -> the XOR swizzle of fft_panel_k (it only
  // permutes inside aligned 16-element runs, so any N that is a multiple of 16 works); otherwise one pad
  // element every R0 when R0 is even (stage-0 write stride becomes odd), nothing when R0 is odd.
  static constexpr bool SWZ = (R0 % 16 == 0) && (N % 16 == 0);
  static constexpr int PADDIV = (!SWZ && R0 % 2 == 0) ? R0 : 0;
  static constexpr int NPAD = PADDIV ? N + N / PADDIV : N;
  static constexpr int LSTRIDE = ((NPAD + 31) / 32) * 32 + 4;
  static constexpr bool QUARTER = (N % 4 == 0);
  static constexpr int QT = QUARTER ? N / 4 + 1 : N;  // twiddle table entries staged in LDS
  static constexpr size_t EX_BYTES = NSTAGE > 1 ? (size_t)COLS * LSTRIDE * sizeof(T) * (SPLIT ? 1 : 2) : 0;
  static constexpr size_t TW_OFF = (EX_BYTES + 15) / 16 * 16;
  static constexpr size_t LDS_BYTES = NSTAGE > 1 ? TW_OFF + (size_t)QT * 2 * sizeof(T) : 0;
  static constexpr int WG_PER_CU_LDS = LDS_BYTES ? (int)(160 * 1024 / LDS_BYTES) : 8;
  static constexpr int WAVES = (NT + 63) / 64;
  static constexpr int WPS_RAW = (WG_PER_CU_LDS * WAVES + 3) / 4;
  static constexpr int WPS = WPS_RAW < 1 ? 1 : (WPS_RAW > 4 ? 4 : WPS_RAW);
  static constexpr int DATA_VGPR = EMAX * (int)sizeof(T) / 2;
  static constexpr int WPS_REG = DATA_VGPR >= 128 ? 2 : (DATA_VGPR >= 64 ? 3 : 4);
  static constexpr int WPS_MIN = (WAVES + 3) / 4;  // one workgroup must fit on a CU
  static constexpr int WPS_E = WPS < WPS_REG ? (WPS < WPS_MIN ? WPS_MIN : WPS) : (WPS_REG < WPS_MIN ? WPS_MIN : WPS_REG);
};

template <typename T, int N, int TPL, int R0, int R1, int R2, int COLS, bool INC, bool OUTC, bool SPLIT, bool R2C = false>
__global__ void __launch_bounds__(TPL * COLS, (PanelXCfg<N, TPL, R0, R1, R2, COLS, SPLIT, T>::WPS_E))
fft_panelx_k(PassArgs a, const typename vec2<T>::type *in, typename vec2<T>::type *out,
             const typename vec2<T>::type *twt) {
  using V2 = typename vec2<T>::type;
  using Cfg = PanelXCfg<N, TPL, R0, R1, R2, COLS, SPLIT, T>;
  constexpr int NT = Cfg::NT, NSTAGE = Cfg::NSTAGE, LSTRIDE = Cfg::LSTRIDE, EMAX = Cfg::EMAX;
  constexpr int PADDIV = Cfg::PADDIV;
  static_assert(R0 * R1 * R2 == N, "radices must multiply to N");
  static_assert(smooth235(R0) && smooth235(R1) && smooth235(R2), "radices must be products of primes <= 13, or primes <= 31");
  static_assert(R0 <= 32 && R1 <= 32 && R2 <= 32, "register radix <= 32");

  extern __shared__ __align__(16) unsigned char smem[];
  T *exs = reinterpret_cast<T *>(smem);
  V2 *exv = reinterpret_cast<V2 *>(smem);
  V2 *tw = reinterpret_cast<V2 *>(smem + Cfg::TW_OFF);

  const int tid = threadIdx.x;
  if constexpr (NSTAGE > 1) {
    for (int i = tid; i < Cfg::QT; i += NT) tw[i] = twt[i];
  }
  auto pad = [](int i) {
    if constexpr (Cfg::SWZ) return i ^ ((i / R0) & 15);
    else if constexpr (PADDIV > 0) return i + i / PADDIV;
    else return i;
  };

  const unsigned bid = panel_of_block(blockIdx.x, a.xcd_lim, a.xcd_gshift);
  const int cp = bid % (unsigned)a.ncp;
  const unsigned rest = bid / (unsigned)a.ncp;
  const int b1 = rest % (unsigned)a.nb1;
  const int b2 = rest / (unsigned)a.nb1;
  const int c0 = cp * COLS;

  cx<T> v[EMAX];

  // ---------------- stage 0: global load -------------------------------------
  int c, j;
  if constexpr (INC) { j = tid % TPL; c = tid / TPL; }
  else               { c = tid % COLS; j = tid / COLS; }
  {
    const bool valid = (c0 + c) < a.ncols;
    const V2 *src = in + (long long)b1 * a.in_b1 + (long long)b2 * a.in_b2 + (long long)(c0 + c) * a.in_col;
    constexpr int NBF = N / R0;
    // (the split test is hoisted out of the unrolled loop: the loads of one thread stay back to back)
    auto load_all = [&](auto has_split) {
      static_for<0, Cfg::NB0 * R0>([&](auto ii) {
        constexpr int u = decltype(ii)::value / R0, t = decltype(ii)::value % R0;
        const int q = j + u * TPL;
        const int n = q + t * NBF;
        const bool live = valid && ((u + 1) * TPL <= NBF || q < NBF);
        V2 val;
        val.x = 0; val.y = 0;
        if constexpr (R2C) {
          if (live) val.x = reinterpret_cast<const T *>(src)[n];
          v[decltype(ii)::value] = cx<T>{val.x, (T)0};
        } else {
          if (live)
            val = gload(&src[split_offset<decltype(has_split)::value>(n, a.in_split, a.in_inv, a.in_nfloor, a.in_lim, a.in_inv1, a.in_blk, a.in_axis, a.in_tab)]);
          v[decltype(ii)::value] = cx<T>{val.x, a.conj ? -val.y : val.y};
        }
      });
    };
    if (a.in_split || a.in_nfloor) load_all(std::true_type{});
    else load_all(std::false_type{});
  }

  // ---------------- stages ---------------------------------------------------
  static_for<0, NSTAGE>([&](auto sidx) {
    constexpr int s = decltype(sidx)::value;
    constexpr int R = (s == 0) ? R0 : ((s == 1) ? R1 : R2);
    constexpr int Ns = (s == 0) ? 1 : ((s == 1) ? R0 : R0 * R1);
    constexpr int NBF = N / R;             // butterflies per line
    constexpr int NB = cdiv(NBF, TPL);     // butterflies per thread (the last one predicated)

    if constexpr (s > 0) {
      // inter-stage twiddles w_N^(k t M), k = q mod Ns, M = N / (Ns R)
      constexpr int M = N / (Ns * R);
      static_for<0, NB>([&](auto uu) {
        constexpr int u = decltype(uu)::value;
        const int q = j + u * TPL;
        const int km = (q % Ns) * M;
        static_for<1, R>([&](auto tt) {
          constexpr int t = decltype(tt)::value;
          const int e = km * t;            // <= (Ns-1)(R-1)M < N, also for a predicated-off butterfly
          T cr, ci;
          if constexpr (Cfg::QUARTER) {
            const int qd = e / (N / 4);
            const V2 w = tw[e - qd * (N / 4)];
            cr = (qd & 1) ? w.y : w.x;     // times (-i)^qd
            ci = (qd & 1) ? -w.x : w.y;
            if (qd & 2) { cr = -cr; ci = -ci; }
          } else {
            const V2 w = tw[e];
            cr = w.x; ci = w.y;
          }
          const cx<T> x = v[u * R + t];
          v[u * R + t] = cx<T>{x.x * cr - x.y * ci, x.x * ci + x.y * cr};
        });
      });
    }

    static_for<0, NB>([&](auto uu) { dft_mixed<T, R>(&v[decltype(uu)::value * R]); });

    if constexpr (s < NSTAGE - 1) {
      // ---- exchange through LDS: write Stockham-ordered, read strided --------
      constexpr int Rn = (s == 0) ? R1 : R2;
      constexpr int NBFn = N / Rn, NBn = cdiv(NBFn, TPL);
      constexpr bool next_last = (s + 1 == NSTAGE - 1);
      int cn, jn;
      if constexpr (next_last && !OUTC) { cn = tid % COLS; jn = tid / COLS; }
      else                              { jn = tid % TPL; cn = tid / TPL; }

      auto wr_idx = [&](int u, int t) {
        const int q = j + u * TPL;
        const int k = q % Ns;
        return c * LSTRIDE + pad((q - k) * R + k + t * Ns);
      };
      auto wr_live = [&](int u) { return (u + 1) * TPL <= NBF || j + u * TPL < NBF; };
      auto rd_idx = [&](int u, int t) { return cn * LSTRIDE + pad(jn + u * TPL + t * NBFn); };
      auto rd_live = [&](int u) { return (u + 1) * TPL <= NBFn || jn + u * TPL < NBFn; };

      if constexpr (s > 0) __syncthreads();  // previous exchange's reads done
      if constexpr (SPLIT) {
        static_for<0, NB * R>([&](auto ii) {
          constexpr int u = decltype(ii)::value / R, t = decltype(ii)::value % R;
          constexpr int src = u * R + perm_mixed(R, t);
          if (wr_live(u)) exs[wr_idx(u, t)] = v[src].x;
        });
        __syncthreads();
        T re[NBn * Rn];
        static_for<0, NBn * Rn>([&](auto ii) {
          constexpr int u = decltype(ii)::value / Rn, t = decltype(ii)::value % Rn;
          re[decltype(ii)::value] = rd_live(u) ? exs[rd_idx(u, t)] : (T)0;
        });
        __syncthreads();
        static_for<0, NB * R>([&](auto ii) {
          constexpr int u = decltype(ii)::value / R, t = decltype(ii)::value % R;
          constexpr int src = u * R + perm_mixed(R, t);
          if (wr_live(u)) exs[wr_idx(u, t)] = v[src].y;
        });
        __syncthreads();
        static_for<0, NBn * Rn>([&](auto ii) {
          constexpr int u = decltype(ii)::value / Rn, t = decltype(ii)::value % Rn;
          v[decltype(ii)::value] = cx<T>{re[decltype(ii)::value], rd_live(u) ? exs[rd_idx(u, t)] : (T)0};
        });
      } else {
        static_for<0, NB * R>([&](auto ii) {
          constexpr int u = decltype(ii)::value / R, t = decltype(ii)::value % R;
          constexpr int src = u * R + perm_mixed(R, t);
          const cx<T> x = v[src];
          V2 w; w.x = x.x; w.y = x.y;
          if (wr_live(u)) exv[wr_idx(u, t)] = w;
        });
        __syncthreads();
        static_for<0, NBn * Rn>([&](auto ii) {
          constexpr int u = decltype(ii)::value / Rn, t = decltype(ii)::value % Rn;
          V2 w; w.x = 0; w.y = 0;
          if (rd_live(u)) w = exv[rd_idx(u, t)];
          v[decltype(ii)::value] = cx<T>{w.x, w.y};
        });
      }
      c = cn; j = jn;
    } else {
      // ---------------- last stage: global store ------------------------------
      const bool valid = (c0 + c) < a.ncols;
      V2 *dst = out + (long long)b1 * a.out_b1 + (long long)b2 * a.out_b2 + (long long)(c0 + c) * a.out_col;
      const T sc = (T)a.scale;
      auto store_all = [&](auto has_split) {
        static_for<0, NB * R>([&](auto ii) {
          constexpr int u = decltype(ii)::value / R, t = decltype(ii)::value % R;
          const int q = j + u * TPL;
          const int n = q + t * NBF;
          constexpr int src = u * R + perm_mixed(R, t);
          const cx<T> x = v[src];
          V2 w;
          w.x = x.x * sc;
          w.y = (a.conj ? -x.y : x.y) * sc;
          const bool live = valid && ((u + 1) * TPL <= NBF || q < NBF);
          if (live && (!R2C || n <= N / 2))
            gstore(&dst[split_offset<decltype(has_split)::value>(n, a.out_split, a.out_inv, a.out_nfloor, a.out_lim, a.out_inv1, a.out_blk, a.out_axis, a.out_tab)], w);
        });
      };
      if (a.out_split || a.out_nfloor) store_all(std::true_type{});
      else store_all(std::false_type{});
    }
  });
}

// ---------------------------------------------------------------------------
// variant registry: every instantiation registers itself under (n, precision, flavour, id)
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
  bool mixed;       // fft_panelx_k (any split length incl. uneven, quarter or full twiddle table)
  bool full_table;
  void *modfn;      // hipFunction_t of an instance compiled at plan time (hipRTC), launched instead of fn
  bool keep = false;  // KEEP instantiation (offt_pass_desc::out_keep): default-policy stores
  bool tw4 = false;   // TW4 instantiation (offt_pass_desc::tw4): four-step twiddles on the stores
};
// id of the fft_panelx_k instance a power-of-two length keeps for per-peer splits fft_panel_k cannot address
// (uneven, or not a power of two: grids split over 3, 6, ... ranks)
enum { VARIANT_ANYSPLIT = 100 };

std::vector<Variant> &registry();  // defined in offt_kernels.hip

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
    registry().push_back(Variant{N, prec, inc, outc, id, (defmask & bit) != 0, r2c, COLS, Cfg::NT, E, Cfg::LDS_BYTES, fn, nm, false, false, false, nullptr});
  };
  add(true, true, F_CC, (const void *)fft_panel_k<T, N, E, R0, R1, R2, COLS, true, true, SPLIT>);
  add(false, false, F_SS, (const void *)fft_panel_k<T, N, E, R0, R1, R2, COLS, false, false, SPLIT>);
  add(true, false, F_CS, (const void *)fft_panel_k<T, N, E, R0, R1, R2, COLS, true, false, SPLIT>);
  add(false, true, F_SC, (const void *)fft_panel_k<T, N, E, R0, R1, R2, COLS, false, true, SPLIT>);
  // real-input z pass: only the contiguous-read flavours of the default variant need it
  if (defmask & F_CC) add(true, true, F_CC, (const void *)fft_panel_k<T, N, E, R0, R1, R2, COLS, true, true, SPLIT, true>, true);
  if (defmask & F_CS) add(true, false, F_CS, (const void *)fft_panel_k<T, N, E, R0, R1, R2, COLS, true, false, SPLIT, true>, true);
  // cache-keeping stores (out_keep): the contig-in / strided-out default, i.e. the y pass of the z-y-x schedules ...
  if (defmask & F_CS) {
    add(true, false, F_CS, (const void *)fft_panel_k<T, N, E, R0, R1, R2, COLS, true, false, SPLIT, false, true>);
    registry().back().keep = true;
  }
  // ... and the contig / contig default: the x pass of the INVERSE z-y-x transform, whose planes the y pass re-reads
  if (defmask & F_CC) {
    add(true, true, F_CC, (const void *)fft_panel_k<T, N, E, R0, R1, R2, COLS, true, true, SPLIT, false, true>);
    registry().back().keep = true;
  }
}

// strided / strided instance with the four-step twiddles on its stores (the first sub-pass of a long line, offt_kernels.hip)
#ifndef OFFT_TW4_KEEP
#define OFFT_TW4_KEEP 1
#endif
constexpr bool TW4_KEEP = OFFT_TW4_KEEP != 0;  // its stores go to the scratch the second sub-pass reads right away: default cache policy
template <typename T, int N, int E, int R0, int R1, int R2, int COLS, bool SPLIT>
void reg_variant_tw4(int id) {
  using Cfg = PanelCfg<N, E, R0, R1, R2, COLS, SPLIT, T>;
  const int prec = std::is_same<T, double>::value ? OFFT_PREC_F64 : OFFT_PREC_F32;
  char nm[160];
  snprintf(nm, sizeof nm, "%s N=%d E=%d radix=%dx%dx%d cols=%d four-step twiddles on the stores lds=%zuB", prec ? "f32" : "f64", N, E, R0, R1, R2,
           COLS, (size_t)Cfg::LDS_BYTES);
  registry().push_back(Variant{N, prec, false, false, id, true, false, COLS, Cfg::NT, E, Cfg::LDS_BYTES,
                               (const void *)fft_panel_k<T, N, E, R0, R1, R2, COLS, false, false, SPLIT, false, TW4_KEEP, true>, nm, false, false, false, nullptr});
  registry().back().tw4 = true;
}

// column-pair instances of fft_panel_k (T = f32x2): kept under their own precision key so that the one-column variants and
// their ids stay what they were; `defmask` says for which flavours an eligible descriptor prefers the pair kernel.
enum { OFFT_PREC_F32_PAIR = 3, VARIANT_PAIR0 = 200 };  // descriptor variant 200 + id forces pair variant `id`
template <int N, int E, int R0, int R1, int R2, int COLS, bool SPLIT>
void reg_variant_pair(int id, int defmask) {
  using T = f32x2;
  using Cfg = PanelCfg<N, E, R0, R1, R2, COLS, SPLIT, T>;
  char nm[160];
  snprintf(nm, sizeof nm, "f32 N=%d E=%d radix=%dx%dx%d cols=%d (column pairs) %s lds=%zuB", N, E, R0, R1, R2, 2 * COLS,
           SPLIT ? "split-re/im" : "packed", (size_t)Cfg::LDS_BYTES);
  auto add = [&](bool inc, bool outc, int bit, const void *fn) {
    registry().push_back(Variant{N, OFFT_PREC_F32_PAIR, inc, outc, id, (defmask & bit) != 0, false, 2 * COLS, Cfg::NT, E, Cfg::LDS_BYTES, fn, nm, false, false, false, nullptr});
  };
  add(true, true, F_CC, (const void *)fft_panel_k<T, N, E, R0, R1, R2, COLS, true, true, SPLIT>);
  add(false, false, F_SS, (const void *)fft_panel_k<T, N, E, R0, R1, R2, COLS, false, false, SPLIT>);
  add(true, false, F_CS, (const void *)fft_panel_k<T, N, E, R0, R1, R2, COLS, true, false, SPLIT>);
  add(false, true, F_SC, (const void *)fft_panel_k<T, N, E, R0, R1, R2, COLS, false, true, SPLIT>);
  if (defmask & F_CS) {
    add(true, false, F_CS, (const void *)fft_panel_k<T, N, E, R0, R1, R2, COLS, true, false, SPLIT, false, true>);
    registry().back().keep = true;
  }
  if (defmask & F_CC) {
    add(true, true, F_CC, (const void *)fft_panel_k<T, N, E, R0, R1, R2, COLS, true, true, SPLIT, false, true>);
    registry().back().keep = true;
  }
}

// mixed-radix (2^a 3^b 5^c) panel kernel: TPL threads per line instead of elements per thread.
// FLAV limits which (in_contig, out_contig) flavours are instantiated at all (compile time), defmask says for
// which of them this variant is the default.
template <typename T, int N, int TPL, int R0, int R1, int R2, int COLS, bool SPLIT, int FLAV = F_ALL>
void reg_variantx(int id, int defmask = -1) {
  if (defmask < 0) defmask = id == 0 ? F_ALL : 0;
  defmask &= FLAV;
  using Cfg = PanelXCfg<N, TPL, R0, R1, R2, COLS, SPLIT, T>;
  const int prec = std::is_same<T, double>::value ? OFFT_PREC_F64 : OFFT_PREC_F32;
  char nm[160];
  snprintf(nm, sizeof nm, "%s N=%d mixed radix=%dx%dx%d threads/line=%d (<=%d elems/thread) cols=%d %s lds=%zuB", prec ? "f32" : "f64", N,
           R0, R1, R2, TPL, Cfg::EMAX, COLS, SPLIT ? "split-re/im" : "packed", (size_t)Cfg::LDS_BYTES);
  auto add = [&](bool inc, bool outc, int bit, const void *fn, bool r2c = false) {
    registry().push_back(Variant{N, prec, inc, outc, id, (defmask & bit) != 0, r2c, COLS, Cfg::NT, Cfg::EMAX, Cfg::LDS_BYTES, fn, nm, false, true, !Cfg::QUARTER, nullptr});
  };
  if constexpr ((FLAV & F_CC) != 0) {
    add(true, true, F_CC, (const void *)fft_panelx_k<T, N, TPL, R0, R1, R2, COLS, true, true, SPLIT>);
    if (defmask & F_CC) add(true, true, F_CC, (const void *)fft_panelx_k<T, N, TPL, R0, R1, R2, COLS, true, true, SPLIT, true>, true);
  }
  if constexpr ((FLAV & F_SS) != 0) add(false, false, F_SS, (const void *)fft_panelx_k<T, N, TPL, R0, R1, R2, COLS, false, false, SPLIT>);
  if constexpr ((FLAV & F_CS) != 0) {
    add(true, false, F_CS, (const void *)fft_panelx_k<T, N, TPL, R0, R1, R2, COLS, true, false, SPLIT>);
    if (defmask & F_CS) add(true, false, F_CS, (const void *)fft_panelx_k<T, N, TPL, R0, R1, R2, COLS, true, false, SPLIT, true>, true);
  }
  if constexpr ((FLAV & F_SC) != 0) add(false, true, F_SC, (const void *)fft_panelx_k<T, N, TPL, R0, R1, R2, COLS, false, true, SPLIT>);
}

// instantiation groups (offt_reg_*.hip)
void reg_pow2_f64();
void reg_pow2_f64_1024();
void reg_pow2_f64_anysplit();
void reg_pow2_f32();
void reg_pow2_f32_big();
void reg_pow2_f32_anysplit();
void reg_pow2_f32_pair();
void reg_pow2_tw4();
void reg_mixed_f64_a();
void reg_mixed_f64_b();
void reg_mixed_f64_c();
void reg_mixed_f64_d();
void reg_mixed_f64_e();
void reg_mixed_f32_a();
void reg_mixed_f32_b();
void reg_dev();

}  // namespace offtk
