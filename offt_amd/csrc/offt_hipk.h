/*
 * offt_hipk.h -- thin C ABI between the C host library (offt_host.c) and the
 * hand-written HIP kernels (offt_kernels.hip, offt_panel.hpp).  Plain pointers and sizes only.
 *
 * One "pass" = a batch of 1-D FFTs of length n along one axis of a strided
 * complex array, reading a panel [n x COLS] per workgroup, with independent
 * input and output addressing so that the local transposes and the pack /
 * unpack copies of the reference (offt-compute.c:905-2993, 523-653) are folded
 * into the load / store side of the butterfly kernel instead of being separate
 * sweeps over memory.
 */
#ifndef OFFT_HIPK_H
#define OFFT_HIPK_H

#ifdef __cplusplus
extern "C" {
#endif

#define OFFT_PREC_F64 0
#define OFFT_PREC_F32 1

typedef struct offt_pass_desc {
  int n;          /* FFT length along the axis                                   */
  int precision;  /* OFFT_PREC_F64 / OFFT_PREC_F32                               */
  int direction;  /* -1: forward exp(-2 pi i nk/N) (the reference's only mode),  */
                  /* +1: inverse, unnormalised                                   */
  int ncols;      /* number of columns (independent lines) per batch entry       */
  int nb1, nb2;   /* two outer batch dimensions                                  */
  /* all strides in complex elements */
  long long in_axis_stride, in_col_stride, in_b1_stride, in_b2_stride;
  long long out_axis_stride, out_col_stride, out_b1_stride, out_b2_stride;
  /* optional split of the axis index into per-peer blocks (fused unpack on the
   * load side / fused pack on the store side):
   *   idx k -> (k / split) * block_stride + (k % split) * axis_stride
   * split == 0 means "no split".  split_nfloor > 0 selects the reference's
   * uneven A2AV partition (offt-compute.c:132-144): the first split_nfloor
   * blocks hold `split` indices, the remaining ones `split + 1`.               */
  int in_split, in_split_nfloor;
  int out_split, out_split_nfloor;
  long long in_block_stride, out_block_stride;
  /* per-block base table on a split side (device memory; NULL = block b sits at b * block_stride): tab[b] is the
   * ELEMENT offset of axis block b relative to the launch's `in` / `out` pointer, for every block the axis has
   * (ceil(n / split) of them; the number of peers for an uneven split).  The reference packs peer a's share into block
   * a of ONE send buffer (offt-compute.c:1084-1109) which MPI then copies (835-881); with a table the blocks of one pass
   * may live in different allocations -- the self block where the next pass reads it, a peer's block in that peer's
   * receive volume (mapped through hipIpc) -- so the pack IS the exchange.  Offsets may be negative.                  */
  const long long *in_block_tab, *out_block_tab;
  /* coalescing hints: 1 = the FFT axis is the unit-stride dimension,
   *                   0 = the column dimension is the unit-stride dimension    */
  int in_contig, out_contig;
  int variant;    /* static-sweep variant id, -1 = default for this n           */
  double scale;   /* multiplied into the output (1.0 = unnormalised)            */
  /* real-to-complex z pass (fftw_plan_dft_r2c_1d, offt-compute.c:334-336, 960-961):
   * the input line holds n REAL values (unit stride, in_contig = 1, no split) at the
   * start of a row of n/2+1 complex slots; only output indices 0..n/2 are stored. */
  int real_input;
  /* cache hint: 1 = the output is read again right away by the next launch (the x pass over the group of z-planes the
   * y pass has just written): store with the default cache policy so that it stays in L2 / the memory-side Infinity
   * Cache, instead of the streaming (non-temporal) stores every other pass uses */
  int out_keep;
  /* single precision: 1 = do not use the column-pair kernels for this pass (plan option OFFT_HIP_OPT_F32_PAIRS) */
  int no_pairs;
  /* first sub-pass of a four-step line (set by the launcher itself, offt_kernels.hip): multiply output index k1 of column
   * j2 (tw4_b1 = 0) or of batch entry b1 = j2 (tw4_b1 = 1) by tw4[k1 * tw4_n2 + j2] = w_n^(k1 j2), a table of the long
   * length n = n1 n2 laid out [k1][j2] (the lanes of a wave are neighbouring columns j2: one 128-B line per 8 lanes, where
   * indexing the full-wave table by k1 j2 would touch a line per lane).  NULL otherwise.  Needs a kernel with the twiddles
   * on its stores (strided / strided, 32 ... 256 points). */
  const void *tw4;
  int tw4_b1;
  int tw4_n2;
} offt_pass_desc;

/* Build device twiddle tables etc. for length n; call at plan time (allocates). */
int offt_hipk_prepare(int n, int precision);
/* Launch one pass on `stream` (a hipStream_t).  No allocation, no sync.        */
int offt_hipk_fft_pass(const offt_pass_desc *d, const void *in, void *out, void *stream);
/* 1 if the pass, with out_keep set, runs on a kernel whose stores stay cached (otherwise out_keep is ignored)          */
int offt_hipk_keeps_output(const offt_pass_desc *d);
/* 1 if a register/LDS Stockham panel kernel exists for (n, precision): powers of two up to 4096
 * and the swept 2^a 3^b 5^c lengths; 0 if the pass will run on the any-length kernel.        */
int offt_hipk_has_fast_path(int n, int precision);
/* 1 if lines of n points have no single-launch kernel and run as a four-step decomposition n = n1 n2 (two sub-passes and a
 * twiddle sweep through scratch; complex input only).  Valid after offt_hipk_prepare(n, precision).                    */
int offt_hipk_is_four_step(int n, int precision);
/* number of sweep variants registered for (n, precision, in_contig, out_contig) */
int offt_hipk_variant_count(int n, int precision);
/* human-readable description of a variant, for sweep logs                      */
const char *offt_hipk_variant_name(int n, int precision, int variant);
/* panel shape of a variant (variant = -1: the default): elements per thread and columns
 * per workgroup; returns the variant id or -1 if (n, precision, variant) does not exist  */
int offt_hipk_variant_info(int n, int precision, int variant, int *elems_per_thread, int *cols);
/* name of the kernel symbol a descriptor resolves to (for rocprof matching)    */
const char *offt_hipk_kernel_name(const offt_pass_desc *d);
/* strided complex copy / permutation (used for layouts no FFT pass can fold)   */
int offt_hipk_copy3d(const void *in, void *out, int precision,
                     int n0, int n1, int n2,
                     long long is0, long long is1, long long is2,
                     long long os0, long long os1, long long os2, void *stream);
/* fill a local block with the seeded position hash / the harness ramp
 * (run-fft.c:46-61); kind 0 = ramp, 1 = hash.                                   */
int offt_hipk_fill(void *buf, int precision, int kind,
                   int n0, int n1, int n2, int s0, int s1, int s2,
                   long long st0, long long st1, long long st2, void *stream);
/* ---- flags of the direct-store exchange (offt_host.c, p2p mode) ----------------------------------------------------
 * A rank that has stored its blocks straight into its peers' receive volumes tells them so by writing a monotonically
 * growing value into one 64-bit word per peer (the role MPI_Wait plays behind MPI_Ialltoall, offt-compute.c:883-890);
 * the peers wait for the words of all their senders before the next pass reads.  Both are one-wave launches on the
 * stream: the signal is ordered behind the kernel that stored (whose end-of-kernel release has written its data back),
 * the wait holds the stream until every word has reached `value` or `timeout_s` seconds have passed -- then it writes 1
 * into *status (host-mapped or device memory) and gives up, so that a dead peer ends as an error, never as a hung GPU. */
#define OFFT_HIPK_MAX_FLAGS 16
int offt_hipk_flag_signal(int n, unsigned long long *const *addr, unsigned long long value, void *stream);
int offt_hipk_flag_wait(int n, unsigned long long *const *addr, unsigned long long value, unsigned long long *status,
                        double timeout_s, void *stream);
/* one wave that holds `stream` for `ms` milliseconds (used by test builds to make the device lag behind the host) */
int offt_hipk_delay(double ms, void *stream);
const char *offt_hipk_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
