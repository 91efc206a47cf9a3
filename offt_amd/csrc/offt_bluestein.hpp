// offt_bluestein.hpp -- lines of ANY length on the power-of-two panel machinery (Bluestein's chirp-z identity).
//
// The reference takes any N because FFTW does (offt-compute.c:335-341 plans whatever Nx, Ny, Nz it is given).  Lengths
// whose prime factors are <= 31 have register kernels here (fft_panel_k / fft_panelx_k); a length with a larger prime
// factor r used to fall to the any-length kernel, whose radix-r stage costs r multiply-adds per output (1016 = 8 * 127:
// 4 % of the HBM roofline).  With  n k = (n^2 + k^2 - (k - n)^2) / 2  the DFT of length N is a convolution,
//     X[k] = a[k] * sum_n (x[n] a[n]) b[k - n],   a[n] = exp(-i pi n^2 / N),  b[m] = exp(+i pi m^2 / N),
// and a cyclic convolution of length M >= 2N - 1, M a power of two, is two M-point FFTs and a pointwise product:
//     conv = IFFT_M( FFT_M(x a, zero-padded) * FFT_M(b wrapped) ).
// One workgroup does ALL of it for a panel [N x COLS] in one HBM round trip: load + chirp, M-point FFT in registers/LDS,
// multiply by the precomputed spectrum B^ = FFT_M(b) / M, second M-point FFT (the inverse one, by conjugation), chirp,
// store the first N points.  The two FFTs are the three-stage register/LDS Stockham scheme of fft_panel_k with radices
// RA x RB x RA: the last radix equals the first, so the outputs of the first FFT sit in exactly the registers the second
// FFT's first stage reads (a static register permutation, no extra exchange).
#pragma once
#include "offt_panel.hpp"

namespace offtk {

template <typename T, int M, int E, int RA, int RB, int RC, int COLS, bool INC, bool OUTC>
__global__ void __launch_bounds__((M / E) * COLS, (PanelCfg<M, E, RA, RB, RC, COLS, true, T>::WPS_E))
fft_bluestein_k(PassArgs a, const typename vec2<T>::type *in, typename vec2<T>::type *out,
                const typename vec2<T>::type *twq, const typename vec2<T>::type *chirp,
                const typename vec2<T>::type *bhat, int N) {
  using V2 = typename vec2<T>::type;
  using Cfg = PanelCfg<M, E, RA, RB, RC, COLS, true, T>;
  constexpr int TPL = Cfg::TPL, NT = Cfg::NT, NSTAGE = Cfg::NSTAGE;
  constexpr int LSTRIDE = Cfg::LSTRIDE;
  constexpr bool SWZ = Cfg::SWZ;
  constexpr int PS = SWZ ? Cfg::SWZSHIFT : Cfg::PADSHIFT;
  constexpr int RL = (NSTAGE == 3) ? RC : RB;  // last radix
  static_assert(NSTAGE >= 2, "Bluestein panels have at least two register stages");
  static_assert(RL == RA, "the last radix must equal the first: the first FFT's outputs are the second one's inputs");
  static_assert(RA * RB * RC == M, "radices must multiply to M");
  constexpr int LRA = ilog2(RA);

  extern __shared__ __align__(16) unsigned char smem[];
  T *exs = reinterpret_cast<T *>(smem);
  V2 *tw = reinterpret_cast<V2 *>(smem + Cfg::TW_OFF);
  V2 *tw1 = reinterpret_cast<V2 *>(smem + Cfg::T1_OFF);

  const int tid = threadIdx.x;
  for (int i = tid; i < Cfg::QT; i += NT) tw[i] = twq[i];
  if constexpr (Cfg::USE_T1) {
    constexpr int M1 = M / (RA * RB);
    for (int i = tid; i < Cfg::T1N; i += NT) {
      const int t = i / RA + 1, k = i - (t - 1) * RA;
      tw1[i] = twq[k * M1 * t];
    }
  }
  __syncthreads();  // tables visible (the first exchange may run without a workgroup barrier)

  const unsigned bid = panel_of_block(blockIdx.x, a.xcd_lim, a.xcd_gshift);
  const int cp = bid % (unsigned)a.ncp;
  const unsigned rest = bid / (unsigned)a.ncp;
  const int b1 = rest % (unsigned)a.nb1;
  const int b2 = rest / (unsigned)a.nb1;
  const int c0 = cp * COLS;
  const unsigned conj_mask = a.conj ? 0x80000000u : 0u;

  cx<T> v[E];
  int c, j;
  if constexpr (INC) { j = tid % TPL; c = tid / TPL; }
  else               { c = tid % COLS; j = tid / COLS; }

  // ---------------- load x[n] a[n], n < N; zero above ---------------------------------------------
  {
    const bool valid = (c0 + c) < a.ncols;
    const V2 *src = in + (long long)b1 * a.in_b1 + (long long)b2 * a.in_b2 + (long long)(c0 + c) * a.in_col;
    auto load_all = [&](auto has_split) {
      static_for<0, E>([&](auto ii) {
        constexpr int u = decltype(ii)::value / RA, t = decltype(ii)::value % RA;
        const int n = j + u * TPL + t * (M / RA);
        cx<T> x{(T)0, (T)0};
        if (valid && n < N) {
          const V2 val = gload(&src[split_offset<decltype(has_split)::value>(n, a.in_split, a.in_inv, a.in_nfloor, a.in_lim, a.in_inv1, a.in_blk, a.in_axis, a.in_tab)]);
          const V2 w = chirp[n];
          const T xi = xor_sign(val.y, conj_mask);
          x = cx<T>{val.x * w.x - xi * w.y, val.x * w.y + xi * w.x};
        }
        v[decltype(ii)::value] = x;
      });
    };
    if (a.in_split || a.in_nfloor) load_all(std::true_type{});
    else load_all(std::false_type{});
  }

  // ---------------- one M-point FFT: X[k], k = j + u TPL + t M/RA, ends in v[u RA + bitrev(t)] ----
  // first_inc: the stage-0 lanes run along the line (wave-private columns); last_across: the last stage's lanes run
  // across columns (strided store side)
  auto run_fft = [&](auto first_inc_, auto last_across_) {
    constexpr bool FIRST_INC = decltype(first_inc_)::value, LAST_ACROSS = decltype(last_across_)::value;
    static_for<0, NSTAGE>([&](auto sidx) {
      constexpr int s = decltype(sidx)::value;
      constexpr int R = (s == 0) ? RA : ((s == 1) ? RB : RC);
      constexpr int Ns = (s == 0) ? 1 : ((s == 1) ? RA : RA * RB);
      constexpr int NB = E / R;
      constexpr int LR = ilog2(R);
      if constexpr (s > 0) {
        constexpr int Mt = M / (Ns * R);
        static_for<0, NB>([&](auto uu) {
          constexpr int u = decltype(uu)::value;
          const int q = j + u * TPL;
          const int km = (q & (Ns - 1)) * Mt;
          static_for<1, R>([&](auto tt) {
            constexpr int t = decltype(tt)::value;
            T cr, ci;
            if constexpr (s == 1 && Cfg::USE_T1) {
              const V2 w = tw1[(t - 1) * RA + (q & (RA - 1))];
              cr = w.x; ci = w.y;
            } else if constexpr (Cfg::USE_HALF) {
              const int e = km * t;
              const V2 w = tw[e & (M / 2 - 1)];
              const unsigned sm = ((unsigned)e << (32 - ilog2(M))) & 0x80000000u;
              cr = xor_sign(w.x, sm); ci = xor_sign(w.y, sm);
            } else {
              const int e = km * t;
              const int qd = e / (M / 4);
              const V2 w = tw[e & (M / 4 - 1)];
              cr = (qd & 1) ? w.y : w.x;
              ci = (qd & 1) ? -w.x : w.y;
              if (qd & 2) { cr = -cr; ci = -ci; }
            }
            const cx<T> x = v[u * R + t];
            v[u * R + t] = cx<T>{x.x * cr - x.y * ci, x.x * ci + x.y * cr};
          });
        });
      }
      static_for<0, NB>([&](auto uu) { dft_reg<T, R>(&v[decltype(uu)::value * R]); });

      if constexpr (s < NSTAGE - 1) {
        constexpr int Rn = (s == 0) ? RB : RC;
        constexpr bool next_last = (s + 1 == NSTAGE - 1);
        int cn, jn;
        if constexpr (next_last && LAST_ACROSS) { cn = tid % COLS; jn = tid / COLS; }
        else                                    { jn = tid % TPL; cn = tid / TPL; }
        auto wr_idx = [&](int u, int t) {
          const int q = j + u * TPL;
          const int k = q & (Ns - 1);
          return c * LSTRIDE + padidx<SWZ, PS>((q - k) * R + k + t * Ns);
        };
        auto rd_idx = [&](int u, int t) { return cn * LSTRIDE + padidx<SWZ, PS>(jn + u * TPL + t * (M / Rn)); };
        constexpr bool WAVE_COLS = (64 % TPL == 0) && (NT % 64 == 0);
        constexpr bool PRIV_W = WAVE_COLS && (s > 0 || FIRST_INC);
        constexpr bool PRIV = PRIV_W && !(next_last && LAST_ACROSS);
        auto xsync = [&](auto priv) {
          if constexpr (decltype(priv)::value) {
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
          } else {
            __syncthreads();
          }
        };
        // the image was last read by the previous exchange (or the previous FFT's last exchange): wave-private only if
        // both that reader and this writer are
        if constexpr (s > 0) xsync(std::integral_constant<bool, PRIV_W>{});
        else xsync(std::integral_constant<bool, PRIV_W && FIRST_INC>{});
        constexpr std::integral_constant<bool, PRIV> priv{};
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
        c = cn; j = jn;
      }
    });
  };

  // ---------------- FFT_M(x a) ------------------------------------------------------------------------
  run_fft(std::integral_constant<bool, INC>{}, std::false_type{});

  // ---------------- times B^[k], conjugate (the second FFT is the inverse one), back into stage-0 order ---
  {
    // register u RA + s holds X[k(u, bitrev(s))] and must end up holding w[k(u, s)]: swap in pairs (s, bitrev(s))
    auto conv = [&](cx<T> x, int k) {
      const V2 b = bhat[k];
      return cx<T>{x.x * b.x - x.y * b.y, -(x.x * b.y + x.y * b.x)};
    };
    static_for<0, E>([&](auto ii) {
      constexpr int u = decltype(ii)::value / RA, s = decltype(ii)::value % RA;
      constexpr int br = bitrev(s, LRA);
      if constexpr (s <= br) {
        const int ks = j + u * TPL + s * (M / RA), kb = j + u * TPL + br * (M / RA);
        const cx<T> xs = v[u * RA + br], xb = v[u * RA + s];  // X[k(u, s)], X[k(u, br)]
        v[u * RA + s] = conv(xs, ks);
        if constexpr (s != br) v[u * RA + br] = conv(xb, kb);
      }
    });
  }

  // ---------------- second FFT; its first stage is fed from registers, lanes along the line -----------
  run_fft(std::true_type{}, std::integral_constant<bool, !OUTC>{});

  // ---------------- X[n] = a[n] conj(Z[n]), n < N; conj-out and scale; store ---------------------------
  {
    const bool valid = (c0 + c) < a.ncols;
    V2 *dst = out + (long long)b1 * a.out_b1 + (long long)b2 * a.out_b2 + (long long)(c0 + c) * a.out_col;
    const T sc = (T)a.scale;
    const T scy = a.conj ? -sc : sc;
    auto store_all = [&](auto has_split) {
      static_for<0, E>([&](auto ii) {
        constexpr int u = decltype(ii)::value / RA, t = decltype(ii)::value % RA;
        const int n = j + u * TPL + t * (M / RA);
        if (valid && n < N) {
          const cx<T> z = v[u * RA + bitrev(t, LRA)];
          const V2 w = chirp[n];
          // a[n] * conj(z) = (w.x + i w.y)(z.x - i z.y)
          V2 o;
          o.x = (w.x * z.x + w.y * z.y) * sc;
          o.y = (w.y * z.x - w.x * z.y) * scy;
          gstore(&dst[split_offset<decltype(has_split)::value>(n, a.out_split, a.out_inv, a.out_nfloor, a.out_lim, a.out_inv1, a.out_blk, a.out_axis, a.out_tab)], o);
        }
      });
    };
    if (a.out_split || a.out_nfloor) store_all(std::true_type{});
    else store_all(std::false_type{});
  }
}

// registry of the Bluestein instances (offt_reg_bluestein.hip): one panel shape per (M, precision), four flavours
struct BlueVariant {
  int m, prec;
  bool inc, outc;
  int cols, threads;
  size_t lds;
  const void *fn;
  bool attr_set;
};
std::vector<BlueVariant> &blue_registry();  // defined in offt_kernels.hip

template <typename T, int M, int E, int RA, int RB, int RC, int COLS>
void reg_bluestein() {
  using Cfg = PanelCfg<M, E, RA, RB, RC, COLS, true, T>;
  const int prec = std::is_same<T, double>::value ? OFFT_PREC_F64 : OFFT_PREC_F32;
  auto add = [&](bool inc, bool outc, const void *fn) {
    blue_registry().push_back(BlueVariant{M, prec, inc, outc, COLS, Cfg::NT, Cfg::LDS_BYTES, fn, false});
  };
  add(true, true, (const void *)fft_bluestein_k<T, M, E, RA, RB, RC, COLS, true, true>);
  add(false, false, (const void *)fft_bluestein_k<T, M, E, RA, RB, RC, COLS, false, false>);
  add(true, false, (const void *)fft_bluestein_k<T, M, E, RA, RB, RC, COLS, true, false>);
  add(false, true, (const void *)fft_bluestein_k<T, M, E, RA, RB, RC, COLS, false, true>);
}

void reg_bluestein_all();

}  // namespace offtk
