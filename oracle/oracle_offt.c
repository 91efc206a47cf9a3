/*
 * oracle_offt.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * PARITY UNPINNED.  The reference (rchyena/offt) holds no tests, golden vectors or fixtures, and it cannot be built in
 * this image (offt.h:63-64 includes fftw3.h / fftw3-mpi.h, which are absent; stand-in headers are not a reference
 * build).  This restatement is therefore checked only against (a) numpy.fft.fftn on seeded fields, (b) the closed form of
 * the harness ramp, (c) values the survey stage recorded from a run of the reference linked against stand-in FFTW
 * headers + MKL (tests/golden/ref_n18_*.npz, survey_recorded.json) -- corroboration, NOT a pin.  What IS independently
 * checked: the FFT values (numpy), and by reading, the decomposition / default-parameter / pack-unpack formulas cited
 * function by function below.
 *
 * CPU restatement of the reference's hot path, rchyena/offt offt-compute.c, in
 * its Hopper build (-DA2AV -DSTRIDE, Makefile:27-29): the p ranks of
 * MPI_COMM_WORLD are simulated inside one process and an all-to-all is a set
 * of block copies between the ranks' send and receive buffers.  Each function
 * cites the reference lines it follows.  The MPI_Test frequencies (F*), the
 * window sizes (W*) and the cache sub-tile sizes (P*, U*) only steer overlap
 * and loop blocking in the reference; they cannot change results and are
 * carried here as parameters without effect.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "oracle.h"

enum { P1_ = 0, T1_ = 1, W1_ = 2, Px1_ = 3, Py1_ = 4, Fz_ = 5, FP1_ = 6, Ux1_ = 7, Uz1_ = 8, FU1_ = 9, Fy1_ = 10,
       Ry_ = 11, T2_ = 12, W2_ = 13, Pz2_ = 14, Px2_ = 15, Fy2_ = 16, FP2_ = 17, Uz2_ = 18, Uy2_ = 19, FU2_ = 20,
       Fx_ = 21, V_ = 22, S_ = 23 };

static int imin(int a, int b) { return a < b ? a : b; }
static int imax(int a, int b) { return a > b ? a : b; }

double orc_hash_val(int x, int y, int z, int c) { /* SURVEY.md Appendix D */
  unsigned h = (unsigned)x * 73856093u ^ (unsigned)y * 19349663u ^ (unsigned)z * 83492791u ^ (unsigned)c * 2654435761u;
  h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15;
  return (double)(h & 0xffffffu) / 8388608.0 - 1.0;
}

/* ---------------------------------------------------------------------------
 * offt_comm_malloc, offt-compute.c:57-315 (A2AV branch 127-144, 246-251, 267-274;
 * output strides 282-300)
 * ------------------------------------------------------------------------- */
void orc_comm_build(orc_comm *c, int Nx, int Ny, int Nz, int p, int rank, int p1, int is_r2c, int is_equalxy, int S) {
  int Nzn = is_r2c ? Nz / 2 + 1 : Nz;
  int p2 = p / p1;
  memset(c, 0, sizeof *c);
  c->p1 = p1; c->p2 = p2;
  int rx = c->rank_x = rank / p2, ry = c->rank_y = rank % p2; /* :75-76 */
  c->M1 = (Nx + p1 - 1) / p1; c->M2 = (Ny + p2 - 1) / p2; c->M3 = (Nzn + p2 - 1) / p2; c->M4 = (Ny + p1 - 1) / p1;
  c->F1 = Nx / p1; c->F2 = Ny / p2; c->F3 = Nzn / p2; c->F4 = Ny / p1;
  c->b1 = Nx % p1; c->b2 = Ny % p2; c->b3 = Nzn % p2; c->b4 = Ny % p1;
  c->m1 = (rx < p1 - c->b1) ? c->F1 : c->F1 + 1;
  c->m2 = (ry < p2 - c->b2) ? c->F2 : c->F2 + 1;
  c->m3 = (ry < p2 - c->b3) ? c->F3 : c->F3 + 1;
  c->m4 = (rx < p1 - c->b4) ? c->F4 : c->F4 + 1;
  c->istart[0] = (rx < p1 - c->b1) ? rx * c->F1 : (p1 - c->b1) * c->F1 + (rx - (p1 - c->b1)) * (c->F1 + 1);
  c->istart[1] = (ry < p2 - c->b2) ? ry * c->F2 : (p2 - c->b2) * c->F2 + (ry - (p2 - c->b2)) * (c->F2 + 1);
  c->istart[2] = 0;
  c->isize[0] = c->m1; c->isize[1] = c->m2; c->isize[2] = Nz;
  c->istride[0] = (c->M2 * p2 > c->M4 * p1) ? c->M2 * c->M3 * p2 : c->M4 * p1 * c->M3; /* :260-261 */
  c->istride[1] = c->M3 * p2;
  c->istride[2] = 1;
  c->ostart[0] = 0;
  c->ostart[1] = (rx < p1 - c->b4) ? rx * c->F4 : (p1 - c->b4) * c->F4 + (rx - (p1 - c->b4)) * (c->F4 + 1);
  c->ostart[2] = (ry < p2 - c->b3) ? ry * c->F3 : (p2 - c->b3) * c->F3 + (ry - (p2 - c->b3)) * (c->F3 + 1);
  c->osize[0] = Nx; c->osize[1] = c->m4; c->osize[2] = c->m3;
  if (S) { c->ostride[0] = c->M3 * c->M4; c->ostride[1] = c->M3; c->ostride[2] = 1; }
  else if (is_equalxy && c->M1 == c->M4) { c->ostride[0] = 1; c->ostride[1] = c->M1 * p1 * c->M3; c->ostride[2] = c->M1 * p1; }
  else { c->ostride[0] = 1; c->ostride[1] = c->M1 * p1; c->ostride[2] = c->M1 * p1 * c->M4; }
}

/* ---------------------------------------------------------------------------
 * params_range_setup / grid_value_floor / params_set_default,
 * offt-compute.c:2998-3225.  The lattice is materialised exactly as the
 * reference does, then searched from the top.
 * ------------------------------------------------------------------------- */
static int floor_log(int n) { if (n == 0) return -1; int c = -1; while (n > 0) { c++; n >>= 1; } return c; }
static int inv_log(int n) { return n == -1 ? 0 : (1 << n); }

static int lattice(int i, int Nx, int Ny, int Nzn, int p, int *list) { /* :2998-3093 */
  int c = 0;
  if (i == P1_) {
    int p_u = imin(imin(Nx, Ny), p), p_l = imax(imax(p / Nzn, p / Ny), 1);
    for (int d = p_l; d <= p_u; d++) if (p % d == 0) list[c++] = d;
    return c;
  }
  if (i == W1_ || i == W2_ || i == Ry_) { for (c = 0; c < 11; c++) list[c] = c; return 11; }
  if (i == V_) { for (c = 0; c < 4; c++) list[c] = c; return 4; }
  if (i == S_) { list[0] = 0; list[1] = 1; return 2; }
  int v_max = -1, zero = 0;
  switch (i) {
    case T1_: case Px1_: case Ux1_: case Px2_: v_max = Nx; break;
    case Py1_: case Uy2_: v_max = Ny; break;
    case Uz1_: case T2_: case Pz2_: case Uz2_: v_max = Nzn; break;
    case Fz_: case FP1_: v_max = Nx * Ny; zero = 1; break;
    case Fy1_: case FU1_: case Fy2_: case FP2_: v_max = Nx * Nzn; zero = 1; break;
    case FU2_: case Fx_: v_max = Ny * Nzn; zero = 1; break;
  }
  int l = floor_log(v_max);
  if (zero) list[c++] = 0;
  for (int cc = 0; cc < l + 1; cc++) list[c++] = inv_log(cc);
  if (inv_log(l) < v_max) list[c++] = v_max;
  return c;
}

static int grid_floor(int i, int raw, int Nx, int Ny, int Nzn, int p) { /* :3096-3109 */
  int list[256];
  int n = lattice(i, Nx, Ny, Nzn, p, list);
  for (int j = n - 1; j >= 0; j--) if (list[j] <= raw) return list[j];
  return raw;
}

void orc_params_default(int Nx, int Ny, int Nz, int p, int is_r2c, int is_W0, int is_notest, int *v) {
  int Nzn = is_r2c ? Nz / 2 + 1 : Nz;
#define GF(i) v[i] = grid_floor(i, v[i], Nx, Ny, Nzn, p)
  v[P1_] = (int)sqrt((double)p); GF(P1_);
  int p2 = p / v[P1_];
  int M1 = (Nx + v[P1_] - 1) / v[P1_], M2 = (Ny + p2 - 1) / p2, M3 = (Nzn + p2 - 1) / p2, M4 = (Ny + v[P1_] - 1) / v[P1_];
  v[T1_] = imax(M1 / 16, 1); GF(T1_);
  v[W1_] = imin(imax(2, 0), (M1 + v[T1_] - 1) / v[T1_]); GF(W1_);
  int P1_xy = 8192 / Nzn;
  v[Px1_] = imin(imax((int)sqrt((double)P1_xy), 1), v[T1_]); GF(Px1_);
  v[Py1_] = imin(imax(P1_xy / v[Px1_], 1), M2); GF(Py1_);
  v[Fz_] = imin(imax(p2 / 2, 0), v[T1_] * M2); GF(Fz_);
  v[FP1_] = imin(imax(v[Fz_], 0), v[T1_] / v[Px1_] * M2 / v[Py1_]); GF(FP1_);
  int U1_xz = 8192 / Ny;
  v[Ux1_] = imin(imax((int)sqrt((double)U1_xz), 1), v[T1_]); GF(Ux1_);
  v[Uz1_] = imin(imax(U1_xz / v[Ux1_], 1), M3); GF(Uz1_);
  v[FU1_] = imin(imax(v[Fz_], 0), v[T1_] / v[Ux1_] * M3 / v[Uz1_]); GF(FU1_);
  v[Fy1_] = imin(imax(v[Fz_], 0), v[T1_] * M3); GF(Fy1_);
  v[Ry_] = 5;
  v[T2_] = imax(M3 / 16, 1); GF(T2_);
  v[W2_] = imin(imax(2, 0), (M3 + v[T2_] - 1) / v[T2_]); GF(W2_);
  int P2_xz = 8192 / Ny;
  v[Pz2_] = imin(imax((int)sqrt((double)P2_xz), 1), v[T2_]); GF(Pz2_);
  v[Px2_] = imin(imax(P2_xz / v[Pz2_], 1), M1); GF(Px2_);
  v[Fy2_] = imin(imax(v[P1_] / 2, 0), v[T2_] * M1); GF(Fy2_);
  v[FP2_] = imin(imax(v[Fy2_], 0), M1 / v[Px2_] * v[T2_] / v[Pz2_]); GF(FP2_);
  int U2_yz = 8192 / Nx;
  v[Uz2_] = imin(imax((int)sqrt((double)U2_yz), 1), v[T2_]); GF(Uz2_);
  v[Uy2_] = imin(imax(U2_yz / v[Uz2_], 1), M4); GF(Uy2_);
  v[FU2_] = imin(imax(v[FP2_], 0), M4 / v[Uy2_] * v[T2_] / v[Uz2_]); GF(FU2_);
  v[Fx_] = imin(imax(v[FP2_], 0), v[T2_] * M4); GF(Fx_);
#undef GF
  v[V_] = 0; v[S_] = 0;
  static const int f[8] = {Fz_, FP1_, FU1_, Fy1_, Fy2_, FP2_, FU2_, Fx_};
  if (is_W0) { v[W1_] = v[W2_] = 0; for (int i = 0; i < 8; i++) v[f[i]] = 0; }
  if (is_notest) for (int i = 0; i < 8; i++) v[f[i]] = 0;
}

/* ---------------------------------------------------------------------------
 * world
 * ------------------------------------------------------------------------- */
typedef struct orank {
  orc_comm c;
  double *out;          /* in == out array, 2*local_elems doubles */
  double *a2as, *a2ar;  /* one send/receive pair (set_buffer, offt-compute.c:710-746) */
} orank;

struct orc_world {
  int Nx, Ny, Nz, Nzn, p, is_r2c, is_oned, is_equalxy;
  int v[ORC_PARAM_COUNT];
  long local_elems, buf_elems;
  orank *r;
  orc_fft_plan *px, *py, *pz;
};

orc_world *orc_world_create(int Nx, int Ny, int Nz, int p, int is_r2c, int is_oned, int is_equalxy, const int *custom_v) {
  orc_world *w = (orc_world *)calloc(1, sizeof *w);
  w->Nx = Nx; w->Ny = Ny; w->Nz = Nz; w->Nzn = is_r2c ? Nz / 2 + 1 : Nz; w->p = p;
  w->is_r2c = is_r2c; w->is_oned = is_oned; w->is_equalxy = is_equalxy;
  orc_params_default(Nx, Ny, Nz, p, is_r2c, 0, 0, w->v);
  if (custom_v) for (int i = 0; i < ORC_PARAM_COUNT; i++) if (custom_v[i] >= 0) w->v[i] = custom_v[i]; /* :3227-3234 */
  w->r = (orank *)calloc(p, sizeof(orank));
  for (int k = 0; k < p; k++) orc_comm_build(&w->r[k].c, Nx, Ny, Nz, p, k, w->v[P1_], is_r2c, is_equalxy, w->v[S_]);
  const orc_comm *c = &w->r[0].c;
  w->local_elems = (c->M2 * c->p2 > c->M4 * c->p1) ? (long)c->M1 * c->M2 * c->M3 * c->p2 : (long)c->M1 * c->M3 * c->M4 * c->p1;
  /* one tile pair per phase: max(T1*M2*M3*p2, M1*M4*p1*T2), offt-compute.c:697-699, 722-723 */
  long b1 = (long)w->v[T1_] * c->M2 * c->M3 * c->p2, b2 = (long)c->M1 * c->M4 * c->p1 * w->v[T2_];
  w->buf_elems = b1 > b2 ? b1 : b2;
  for (int k = 0; k < p; k++) {
    w->r[k].out = (double *)calloc(2 * (size_t)w->local_elems, sizeof(double));
    w->r[k].a2as = (double *)calloc(2 * (size_t)w->buf_elems, sizeof(double));
    w->r[k].a2ar = (double *)calloc(2 * (size_t)w->buf_elems, sizeof(double));
  }
  w->px = orc_fft_plan_create(Nx); w->py = orc_fft_plan_create(Ny); w->pz = orc_fft_plan_create(Nz);
  return w;
}

void orc_world_destroy(orc_world *w) {
  if (!w) return;
  for (int k = 0; k < w->p; k++) { free(w->r[k].out); free(w->r[k].a2as); free(w->r[k].a2ar); }
  free(w->r);
  orc_fft_plan_destroy(w->px); orc_fft_plan_destroy(w->py); orc_fft_plan_destroy(w->pz);
  free(w);
}

const orc_comm *orc_world_comm(const orc_world *w, int rank) { return &w->r[rank].c; }
const int *orc_world_params(const orc_world *w) { return w->v; }
long orc_world_local_elems(const orc_world *w) { return w->local_elems; }
double *orc_world_buffer(orc_world *w, int rank) { return w->r[rank].out; }

void orc_world_fill(orc_world *w, int kind) { /* run-fft.c:46-61 */
  for (int k = 0; k < w->p; k++) {
    const orc_comm *c = &w->r[k].c;
    double *in = w->r[k].out;
    memset(in, 0, sizeof(double) * 2 * (size_t)w->local_elems);
    for (int x = 0; x < c->isize[0]; x++)
      for (int y = 0; y < c->isize[1]; y++)
        for (int z = 0; z < c->isize[2]; z++) {
          int gx = x + c->istart[0], gy = y + c->istart[1], gz = z + c->istart[2];
          if (w->is_r2c) {
            long o = z + 2L * c->istride[1] * y + 2L * c->istride[0] * x;
            in[o] = kind == 0 ? (double)(gz + 10 * gy + 100 * gx) : orc_hash_val(gx, gy, gz, 0);
          } else {
            long o = 2 * (z + (long)c->istride[1] * y + (long)c->istride[0] * x);
            in[o] = kind == 0 ? (double)(gz + 10 * gy + 100 * gx) : orc_hash_val(gx, gy, gz, 0);
            in[o + 1] = kind == 0 ? 0.0 : orc_hash_val(gx, gy, gz, 1);
          }
        }
  }
}

/* the A2AV block arithmetic shared by every pack/unpack site
 * (e.g. offt-compute.c:1000-1028): owner a of global index k, its offset, and
 * the block base for the equal-block (V bit clear) and the true alltoallv case */
typedef struct { int a, off; } own_t;
static own_t owner(int k, int F, int b, int p) {
  own_t o;
  if (F * (p - b) <= k) { o.a = (k - F * (p - b)) / (F + 1) + (p - b); o.off = (p - b) * F + (o.a - (p - b)) * (F + 1); }
  else { o.a = k / F; o.off = o.a * F; }
  return o;
}

static void fft_z_line(const orc_world *w, double *ptr, double *scr) { /* :959-963 */
  if (w->is_r2c) {
    /* fftw_plan_dft_r2c_1d in place: Nz reals in, Nz/2+1 complex out */
    int n = w->Nz;
    double *t = scr + 4 * (size_t)n;
    for (int k = 0; k < n; k++) { t[2 * k] = ptr[k]; t[2 * k + 1] = 0.0; }
    orc_fft_execute(w->pz, t, 1, 0, 1, scr);
    memcpy(ptr, t, sizeof(double) * 2 * (size_t)w->Nzn);
  } else {
    orc_fft_execute(w->pz, ptr, 1, 0, 1, scr);
  }
}

/* compute_fftz_pack1, offt-compute.c:905-1206 */
static void fftz_pack1(const orc_world *w, orank *R, int tile, int myT, double *scr) {
  const orc_comm *c = &R->c;
  const int T = w->v[T1_], S = w->v[S_], is_a2av = w->v[V_] & 2;
  const int from_x = tile * T, to_x = from_x + myT;
  const int F3 = c->F3, b3 = c->b3, m2 = c->m2, p2 = c->p2, M2 = c->M2, M3 = c->M3;
  double *out = R->out;
  for (int x = from_x; x < to_x; x++)
    for (int y = 0; y < m2; y++)
      fft_z_line(w, out + 2L * c->istride[1] * y + 2L * c->istride[0] * x, scr);
  /* the owner of a z index does not depend on (x, y): look it up once per call (the reference
   * recomputes it per element, offt-compute.c:1000-1028; same values) */
  own_t *own = (own_t *)malloc(sizeof(own_t) * (size_t)w->Nzn);
  for (int z = 0; z < w->Nzn; z++) own[z] = owner(z, F3, b3, p2);
#define PACK1_ELEM(x, y, z)                                                                                        \
  do {                                                                                                             \
    const own_t o = own[z];                                                                                        \
    long B, dst;                                                                                                   \
    const int wide = (F3 * (p2 - b3) <= (z));                                                                      \
    if (S) { /* block [x][y][z-run], :1000-1032 */                                                                 \
      int Sy, Sx;                                                                                                  \
      if (is_a2av) {                                                                                               \
        B = wide ? (long)(p2 - b3) * (myT * m2 * F3) + (long)(o.a - (p2 - b3)) * (myT * m2 * (F3 + 1)) : (long)o.a * (myT * m2 * F3); \
        Sy = wide ? F3 + 1 : F3; Sx = m2 * Sy;                                                                     \
      } else { B = (long)o.a * (myT * M2 * M3); Sy = M3; Sx = M2 * M3; }                                           \
      dst = B + ((z) - o.off) + (long)(y) * Sy + (long)((x) - from_x) * Sx;                                        \
    } else { /* block [x][z][y], :1070-1109 */                                                                     \
      int Sz, Sx;                                                                                                  \
      if (is_a2av) {                                                                                               \
        B = wide ? (long)(p2 - b3) * (myT * m2 * F3) + (long)(o.a - (p2 - b3)) * (myT * m2 * (F3 + 1)) : (long)o.a * (myT * m2 * F3); \
        Sz = m2; Sx = wide ? m2 * (F3 + 1) : m2 * F3;                                                              \
      } else { B = (long)o.a * (myT * M2 * M3); Sz = M2; Sx = M2 * M3; }                                           \
      dst = B + (y) + (long)((z) - o.off) * Sz + (long)((x) - from_x) * Sx;                                        \
    }                                                                                                              \
    const double *e_ = out + 2L * (z) + 2L * c->istride[1] * (y) + 2L * c->istride[0] * (x);                       \
    R->a2as[2 * dst] = e_[0]; R->a2as[2 * dst + 1] = e_[1];                                                        \
  } while (0)
  /* loop order = destination order, as in the reference (:1038-1045 xyz, :1061-1118 xzy) */
  if (S) {
    for (int x = from_x; x < to_x; x++)
      for (int y = 0; y < m2; y++)
        for (int z = 0; z < w->Nzn; z++) PACK1_ELEM(x, y, z);
  } else {
    for (int x = from_x; x < to_x; x++)
      for (int z = 0; z < w->Nzn; z++)
        for (int y = 0; y < m2; y++) PACK1_ELEM(x, y, z);
  }
#undef PACK1_ELEM
  free(own);
}

/* communicate_a2a / communicate_a2av, offt-compute.c:835-881, counts from 3512-3526 */
static void a2a_phase1(orc_world *w, int rank_x, int myT) {
  const orc_comm *c0 = &w->r[rank_x * w->r[0].c.p2].c;
  const int p2 = c0->p2, is_a2av = w->v[V_] & 2;
  for (int j = 0; j < p2; j++) {       /* receiver */
    orank *Rj = &w->r[rank_x * p2 + j];
    long rdis = 0;
    for (int i = 0; i < p2; i++) {     /* sender */
      orank *Ri = &w->r[rank_x * p2 + i];
      long cnt, sdis = 0;
      if (is_a2av) {
        /* sender i's displacement of block j, and count = 2*myT*m2(i)*(F3 [+1]) */
        for (int k = 0; k < j; k++) sdis += 2L * myT * Ri->c.m2 * (c0->F3 + (k >= p2 - c0->b3));
        cnt = 2L * myT * Ri->c.m2 * (c0->F3 + (j >= p2 - c0->b3));
      } else { cnt = 2L * myT * c0->M2 * c0->M3; sdis = j * cnt; rdis = i * cnt; }
      memcpy(Rj->a2ar + rdis, Ri->a2as + sdis, sizeof(double) * (size_t)cnt);
      if (is_a2av) rdis += cnt;
    }
  }
}

/* compute_unpack1_ffty, offt-compute.c:1208-1520 */
static void unpack1_ffty(const orc_world *w, orank *R, int tile, int myT, double *scr) {
  const orc_comm *c = &R->c;
  const int T = w->v[T1_], S = w->v[S_], is_a2av = w->v[V_] & 2, Ry = w->v[Ry_];
  const int from_x = tile * T, to_x = from_x + myT;
  const int F2 = c->F2, b2 = c->b2, M2 = c->M2, M3 = c->M3, M4 = c->M4, m3 = c->m3, p1 = c->p1, p2 = c->p2;
  const int ignore_Ry = (w->is_oned && p1 == 1); /* :1240 */
  double *out = R->out;
  own_t *own = (own_t *)malloc(sizeof(own_t) * (size_t)w->Ny);
  for (int y = 0; y < w->Ny; y++) own[y] = owner(y, F2, b2, p2);
#define UNPACK1_ELEM(x, y, z)                                                                                      \
  do {                                                                                                             \
    const own_t o = own[y];                                                                                        \
    const int wide = (F2 * (p2 - b2) <= (y));                                                                      \
    long B, src, dst;                                                                                              \
    if (S) { /* :1278-1311 */                                                                                      \
      int Sy, Sx;                                                                                                  \
      if (is_a2av) {                                                                                               \
        B = wide ? (long)(p2 - b2) * (myT * F2 * m3) + (long)(o.a - (p2 - b2)) * (myT * (F2 + 1) * m3) : (long)o.a * (myT * F2 * m3); \
        Sy = m3; Sx = wide ? (F2 + 1) * m3 : F2 * m3;                                                              \
      } else { B = (long)o.a * (myT * M2 * M3); Sy = M3; Sx = M2 * M3; }                                           \
      src = B + (z) + (long)((y) - o.off) * Sy + (long)((x) - from_x) * Sx;                                        \
      dst = (z) + (long)M3 * (y) + (long)M3 * M4 * p1 * (x);                                                       \
    } else { /* :1353-1385 */                                                                                      \
      int Sz, Sx;                                                                                                  \
      if (is_a2av) {                                                                                               \
        B = wide ? (long)(p2 - b2) * (myT * F2 * m3) + (long)(o.a - (p2 - b2)) * (myT * (F2 + 1) * m3) : (long)o.a * (myT * F2 * m3); \
        Sz = wide ? F2 + 1 : F2; Sx = Sz * m3;                                                                     \
      } else { B = (long)o.a * (myT * M2 * M3); Sz = M2; Sx = M2 * M3; }                                           \
      src = B + ((y) - o.off) + (long)(z) * Sz + (long)((x) - from_x) * Sx;                                        \
      dst = (y) + (long)M4 * p1 * ((z) + (long)M3 * (x));                                                          \
    }                                                                                                              \
    out[2 * dst] = R->a2ar[2 * src]; out[2 * dst + 1] = R->a2ar[2 * src + 1];                                      \
  } while (0)
  /* loop order = destination order (:1269-1271 xyz, :1341-1346 xzy) */
  if (S) {
    for (int x = from_x; x < to_x; x++)
      for (int y = 0; y < w->Ny; y++)
        for (int z = 0; z < m3; z++) UNPACK1_ELEM(x, y, z);
  } else {
    for (int x = from_x; x < to_x; x++)
      for (int z = 0; z < m3; z++)
        for (int y = 0; y < w->Ny; y++) UNPACK1_ELEM(x, y, z);
  }
#undef UNPACK1_ELEM
  free(own);
  /* FFTy share of phase 1, :1479-1495 */
  for (int x = from_x; x < to_x; x++)
    for (int z = 0; z < m3; z++)
      if (ignore_Ry || x % 10 < Ry) {
        if (S) orc_fft_execute(w->py, out + 2 * (z + (long)M4 * p1 * M3 * x), M3, 0, 1, scr);
        else orc_fft_execute(w->py, out + 2L * M4 * p1 * (z + (long)M3 * x), 1, 0, 1, scr);
      }
}

/* fftw rank-0 guru plan = pure permutation (setup_transpose, offt-compute.c:523-653) */
static void permute3(double *buf, long total_elems, const int n[3], const long is[3], const long os[3]) {
  double *tmp = (double *)malloc(sizeof(double) * 2 * (size_t)total_elems);
  memcpy(tmp, buf, sizeof(double) * 2 * (size_t)total_elems);
  for (int a = 0; a < n[0]; a++)
    for (int b = 0; b < n[1]; b++)
      for (int c = 0; c < n[2]; c++) {
        long si = a * is[0] + b * is[1] + c * is[2], di = a * os[0] + b * os[1] + c * os[2];
        buf[2 * di] = tmp[2 * si]; buf[2 * di + 1] = tmp[2 * si + 1];
      }
  free(tmp);
}

static void transpose_local(const orc_world *w, orank *R) {
  const orc_comm *c = &R->c;
  const int eq = (w->is_equalxy && c->M1 == c->M4);
  int n[3]; long is[3], os[3];
  if (w->is_oned && c->p1 == 1) {
    if (eq) { /* xzy -> yzx, :563-573 */
      n[0] = c->M1; is[0] = (long)c->M4 * c->M3; os[0] = 1;
      n[1] = c->M3; is[1] = c->M4; os[1] = c->M1;
      n[2] = c->M4; is[2] = 1; os[2] = (long)c->M1 * c->M3;
    } else { /* xzy -> zyx, :575-584 */
      n[0] = c->M1; is[0] = (long)c->M4 * c->M3; os[0] = 1;
      n[1] = c->M3; is[1] = c->M4; os[1] = (long)c->M1 * c->M4;
      n[2] = c->M4; is[2] = 1; os[2] = c->M1;
    }
  } else if (w->is_oned && c->p1 == w->p) {
    if (eq) { /* xyz -> xzy, :587-598 */
      n[0] = c->M1; is[0] = (long)c->M4 * c->p1 * c->M3; os[0] = (long)c->M4 * c->p1 * c->M3;
      n[1] = c->M4 * c->p1; is[1] = c->M3; os[1] = 1;
      n[2] = c->M3; is[2] = 1; os[2] = (long)c->M4 * c->p1;
    } else { /* xyz -> zxy, :600-610 */
      n[0] = c->M1; is[0] = (long)c->M4 * c->p1 * c->M3; os[0] = (long)c->M4 * c->p1;
      n[1] = c->M4 * c->p1; is[1] = c->M3; os[1] = 1;
      n[2] = c->M3; is[2] = 1; os[2] = (long)c->M4 * c->p1 * c->M1;
    }
  } else {
    if (eq) return; /* :613-623 */
    /* xzy -> zxy, :625-634 */
    n[0] = c->M1; is[0] = (long)c->M4 * c->p1 * c->M3; os[0] = (long)c->M4 * c->p1;
    n[1] = c->M3; is[1] = (long)c->M4 * c->p1; os[1] = (long)c->M4 * c->p1 * c->M1;
    n[2] = c->M4 * c->p1; is[2] = 1; os[2] = 1;
  }
  permute3(R->out, w->local_elems, n, is, os);
}

/* compute_ffty_pack2, offt-compute.c:1636-2345 */
static void ffty_pack2(const orc_world *w, orank *R, int tile, int myT, double *scr) {
  const orc_comm *c = &R->c;
  const int T = w->v[T2_], S = w->v[S_], is_a2av = w->v[V_] & 1, Ry = w->v[Ry_];
  const int from_z = tile * T, to_z = from_z + myT;
  const int F4 = c->F4, b4 = c->b4, M1 = c->M1, M3 = c->M3, M4 = c->M4, m1 = c->m1, p1 = c->p1;
  const int eq = (w->is_equalxy && M1 == M4);
  const int ignore_Ry = (w->is_oned && p1 == w->p); /* :1690 */
  double *out = R->out;
  /* FFTy share of phase 2 */
  for (int x = 0; x < m1; x++)
    for (int z = from_z; z < to_z; z++)
      if (ignore_Ry || x % 10 >= Ry) {
        if (S) orc_fft_execute(w->py, out + 2L * z + 2L * M4 * p1 * ((long)M3 * x), M3, 0, 1, scr);          /* :1709 */
        else if (eq) orc_fft_execute(w->py, out + 2L * M4 * p1 * (z + (long)M3 * x), 1, 0, 1, scr);          /* :1842 */
        else orc_fft_execute(w->py, out + 2L * M4 * p1 * (x + (long)M1 * z), 1, 0, 1, scr);                  /* :1989 */
      }
  own_t *own = (own_t *)malloc(sizeof(own_t) * (size_t)w->Ny);
  for (int y = 0; y < w->Ny; y++) own[y] = owner(y, F4, b4, p1);
#define PACK2_ELEM(x, y, z)                                                                                        \
  do {                                                                                                             \
    const own_t o = own[y];                                                                                        \
    const int wide = (F4 * (p1 - b4) <= (y));                                                                      \
    const long Bv = wide ? (long)(p1 - b4) * (m1 * F4 * myT) + (long)(o.a - (p1 - b4)) * (m1 * (F4 + 1) * myT) : (long)o.a * (m1 * F4 * myT); \
    const long B = is_a2av ? Bv : (long)o.a * (M1 * M4 * myT);                                                     \
    long src, dst;                                                                                                 \
    if (S) { /* :1744-1776 */                                                                                      \
      const int Sy = myT, Sx = is_a2av ? myT * (wide ? F4 + 1 : F4) : myT * M4;                                    \
      dst = B + ((z) - from_z) + (long)((y) - o.off) * Sy + (long)(x) * Sx;                                        \
      src = (z) + (long)M3 * (y) + (long)M3 * M4 * p1 * (x);                                                       \
    } else if (eq) { /* block [y][z][x], :1879-1911 */                                                             \
      const int Sz = is_a2av ? m1 : M1, Sy = Sz * myT;                                                             \
      dst = B + (x) + (long)((z) - from_z) * Sz + (long)((y) - o.off) * Sy;                                        \
      src = (y) + (long)w->Ny * (z) + (long)M4 * p1 * M3 * (x); /* note Ny*z, SURVEY.md Appendix F */              \
    } else { /* block [z][y][x], :2026-2058 */                                                                     \
      const int Sy = is_a2av ? m1 : M1, Sz = is_a2av ? m1 * (wide ? F4 + 1 : F4) : M1 * M4;                        \
      dst = B + (x) + (long)((y) - o.off) * Sy + (long)((z) - from_z) * Sz;                                        \
      src = (y) + (long)M4 * p1 * (x) + (long)M4 * p1 * M1 * (z);                                                  \
    }                                                                                                              \
    R->a2as[2 * dst] = out[2 * src]; R->a2as[2 * dst + 1] = out[2 * src + 1];                                      \
  } while (0)
  /* loop order = destination order (:1735-1737 xyz, :1870-1872 yzx, :2016-2020 zyx) */
  if (S) {
    for (int x = 0; x < m1; x++)
      for (int y = 0; y < w->Ny; y++)
        for (int z = from_z; z < to_z; z++) PACK2_ELEM(x, y, z);
  } else if (eq) {
    for (int y = 0; y < w->Ny; y++)
      for (int z = from_z; z < to_z; z++)
        for (int x = 0; x < m1; x++) PACK2_ELEM(x, y, z);
  } else {
    for (int z = from_z; z < to_z; z++)
      for (int y = 0; y < w->Ny; y++)
        for (int x = 0; x < m1; x++) PACK2_ELEM(x, y, z);
  }
#undef PACK2_ELEM
  free(own);
}

static void a2a_phase2(orc_world *w, int rank_y, int myT) { /* counts :3693-3707 */
  const orc_comm *c0 = &w->r[rank_y].c;
  const int p1 = c0->p1, p2 = c0->p2, is_a2av = w->v[V_] & 1;
  for (int j = 0; j < p1; j++) {
    orank *Rj = &w->r[j * p2 + rank_y];
    long rdis = 0;
    for (int i = 0; i < p1; i++) {
      orank *Ri = &w->r[i * p2 + rank_y];
      long cnt, sdis = 0;
      if (is_a2av) {
        for (int k = 0; k < j; k++) sdis += 2L * myT * Ri->c.m1 * (c0->F4 + (k >= p1 - c0->b4));
        cnt = 2L * myT * Ri->c.m1 * (c0->F4 + (j >= p1 - c0->b4));
      } else { cnt = 2L * myT * c0->M1 * c0->M4; sdis = j * cnt; rdis = i * cnt; }
      memcpy(Rj->a2ar + rdis, Ri->a2as + sdis, sizeof(double) * (size_t)cnt);
      if (is_a2av) rdis += cnt;
    }
  }
}

/* compute_unpack2_fftx, offt-compute.c:2347-2993 */
static void unpack2_fftx(const orc_world *w, orank *R, int tile, int myT, double *scr) {
  const orc_comm *c = &R->c;
  const int T = w->v[T2_], S = w->v[S_], is_a2av = w->v[V_] & 1;
  const int from_z = tile * T, to_z = from_z + myT;
  const int F1 = c->F1, b1 = c->b1, M1 = c->M1, M3 = c->M3, M4 = c->M4, m4 = c->m4, p1 = c->p1;
  const int eq = (w->is_equalxy && M1 == M4);
  double *out = R->out;
  own_t *own = (own_t *)malloc(sizeof(own_t) * (size_t)w->Nx);
  for (int x = 0; x < w->Nx; x++) own[x] = owner(x, F1, b1, p1);
#define UNPACK2_ELEM(x, y, z)                                                                                      \
  do {                                                                                                             \
    const own_t o = own[x];                                                                                        \
    const int wide = (F1 * (p1 - b1) <= (x));                                                                      \
    const long Bv = wide ? (long)(p1 - b1) * (F1 * m4 * myT) + (long)(o.a - (p1 - b1)) * ((F1 + 1) * m4 * myT) : (long)o.a * (F1 * m4 * myT); \
    const long B = is_a2av ? Bv : (long)o.a * (M1 * M4 * myT);                                                     \
    long src, dst;                                                                                                 \
    if (S) { /* :2418-2450 */                                                                                      \
      const int Sy = myT, Sx = is_a2av ? myT * m4 : myT * M4;                                                      \
      src = B + ((z) - from_z) + (long)(y) * Sy + (long)((x) - o.off) * Sx;                                        \
      dst = (z) + (long)M3 * (y) + (long)M3 * M4 * (x);                                                            \
    } else if (eq) { /* :2538-2570 */                                                                              \
      const int Sz = is_a2av ? (wide ? F1 + 1 : F1) : M1, Sy = Sz * myT;                                           \
      src = B + ((x) - o.off) + (long)((z) - from_z) * Sz + (long)(y) * Sy;                                        \
      dst = (x) + (long)M1 * p1 * ((z) + (long)M3 * (y));                                                          \
    } else { /* :2655-2687 */                                                                                      \
      const int Sy = is_a2av ? (wide ? F1 + 1 : F1) : M1, Sz = is_a2av ? Sy * m4 : M1 * M4;                        \
      src = B + ((x) - o.off) + (long)(y) * Sy + (long)((z) - from_z) * Sz;                                        \
      dst = (x) + (long)M1 * p1 * ((y) + (long)M4 * (z));                                                          \
    }                                                                                                              \
    out[2 * dst] = R->a2ar[2 * src]; out[2 * dst + 1] = R->a2ar[2 * src + 1];                                      \
  } while (0)
  /* loop order = destination order (:2409-2411 xyz, :2528-2532 yzx, :2645-2649 zyx) */
  if (S) {
    for (int x = 0; x < w->Nx; x++)
      for (int y = 0; y < m4; y++)
        for (int z = from_z; z < to_z; z++) UNPACK2_ELEM(x, y, z);
  } else if (eq) {
    for (int y = 0; y < m4; y++)
      for (int z = from_z; z < to_z; z++)
        for (int x = 0; x < w->Nx; x++) UNPACK2_ELEM(x, y, z);
  } else {
    for (int z = from_z; z < to_z; z++)
      for (int y = 0; y < m4; y++)
        for (int x = 0; x < w->Nx; x++) UNPACK2_ELEM(x, y, z);
  }
#undef UNPACK2_ELEM
  free(own);
  for (int y = 0; y < m4; y++)
    for (int z = from_z; z < to_z; z++) {
      if (S) orc_fft_execute(w->px, out + 2L * z + 2L * M3 * y, (long)M3 * M4, 0, 1, scr);                 /* :2493-2495 */
      else if (eq) orc_fft_execute(w->px, out + 2L * (M1 * p1) * (z + (long)M3 * y), 1, 0, 1, scr);       /* :2612 */
      else orc_fft_execute(w->px, out + 2L * (M1 * p1) * (y + (long)M4 * z), 1, 0, 1, scr);               /* :2729 */
    }
}

static double *scratch_for(const orc_world *w) {
  int n = imax(imax(w->Nx, w->Ny), w->Nz);
  return (double *)malloc(sizeof(double) * 6 * (size_t)n + 64);
}

/* offt_3d_execute_phase1, offt-compute.c:3501-3680: tiles along x over comm1.
 * All row groups advance tile by tile together (as concurrently running MPI
 * ranks would); a rank whose group has fewer tiles simply sits a step out. */
static void phase1(orc_world *w, int nthreads) {
  const int p1 = w->r[0].c.p1, p2 = w->r[0].c.p2, T = w->v[T1_];
  int maxblocks = 0;
  for (int rx = 0; rx < p1; rx++) maxblocks = imax(maxblocks, (w->r[rx * p2].c.m1 + T - 1) / T);
  for (int i = 0; i < maxblocks; i++) {
#pragma omp parallel for num_threads(nthreads) schedule(dynamic, 1)
    for (int k = 0; k < w->p; k++) {
      const int m1 = w->r[k].c.m1, blocks = (m1 + T - 1) / T;
      if (i >= blocks) continue;
      const int myT = (i == blocks - 1) ? m1 - (blocks - 1) * T : T; /* :3551-3552 */
      double *scr = scratch_for(w); fftz_pack1(w, &w->r[k], i, myT, scr); free(scr);
    }
    for (int rx = 0; rx < p1; rx++) {
      const int m1 = w->r[rx * p2].c.m1, blocks = (m1 + T - 1) / T;
      if (i < blocks) a2a_phase1(w, rx, (i == blocks - 1) ? m1 - (blocks - 1) * T : T);
    }
#pragma omp parallel for num_threads(nthreads) schedule(dynamic, 1)
    for (int k = 0; k < w->p; k++) {
      const int m1 = w->r[k].c.m1, blocks = (m1 + T - 1) / T;
      if (i >= blocks) continue;
      const int myT = (i == blocks - 1) ? m1 - (blocks - 1) * T : T;
      double *scr = scratch_for(w); unpack1_ffty(w, &w->r[k], i, myT, scr); free(scr);
    }
  }
}

/* offt_3d_execute_phase2, offt-compute.c:3682-3862: tiles along z over comm2 */
static void phase2(orc_world *w, int nthreads) {
  const int p1 = w->r[0].c.p1, p2 = w->r[0].c.p2, T = w->v[T2_];
  int maxblocks = 0;
  for (int ry = 0; ry < p2; ry++) maxblocks = imax(maxblocks, (w->r[ry].c.m3 + T - 1) / T);
  for (int i = 0; i < maxblocks; i++) {
#pragma omp parallel for num_threads(nthreads) schedule(dynamic, 1)
    for (int k = 0; k < w->p; k++) {
      const int m3 = w->r[k].c.m3, blocks = (m3 + T - 1) / T;
      if (i >= blocks) continue;
      const int myT = (i == blocks - 1) ? m3 - (blocks - 1) * T : T;
      double *scr = scratch_for(w); ffty_pack2(w, &w->r[k], i, myT, scr); free(scr);
    }
    for (int ry = 0; ry < p2; ry++) {
      const int m3 = w->r[ry].c.m3, blocks = (m3 + T - 1) / T;
      if (i < blocks) a2a_phase2(w, ry, (i == blocks - 1) ? m3 - (blocks - 1) * T : T);
    }
#pragma omp parallel for num_threads(nthreads) schedule(dynamic, 1)
    for (int k = 0; k < w->p; k++) {
      const int m3 = w->r[k].c.m3, blocks = (m3 + T - 1) / T;
      if (i >= blocks) continue;
      const int myT = (i == blocks - 1) ? m3 - (blocks - 1) * T : T;
      double *scr = scratch_for(w); unpack2_fftx(w, &w->r[k], i, myT, scr); free(scr);
    }
  }
  (void)p1;
}

/* offt_3d_execute, offt-compute.c:3864-4048 */
void orc_world_execute(orc_world *w, int nthreads) {
  const int p1 = w->r[0].c.p1, S = w->v[S_];
  if (nthreads < 1) nthreads = 1;
  if (w->is_oned && p1 == 1) { /* mode A, :3896-3950 */
    phase1(w, nthreads);
    if (!S) for (int k = 0; k < w->p; k++) transpose_local(w, &w->r[k]);
#pragma omp parallel for num_threads(nthreads) schedule(static)
    for (int k = 0; k < w->p; k++) {
      orank *R = &w->r[k]; const orc_comm *c = &R->c;
      double *scr = scratch_for(w);
      if (S) orc_fft_execute(w->px, R->out, (long)c->M3 * c->M4, 1, c->M3 * c->M4, scr); /* plan_many, :400-403, 3932 */
      else
        for (int z = 0; z < c->m3; z++)
          for (int y = 0; y < w->Ny; y++)
            orc_fft_execute(w->px, R->out + 2L * (c->M1 * c->p1) * (y + (long)c->M4 * z), 1, 0, 1, scr); /* :3935-3939 */
      free(scr);
    }
  } else if (w->is_oned && p1 == w->p) { /* mode B, :3951-3998 */
#pragma omp parallel for num_threads(nthreads) schedule(static)
    for (int k = 0; k < w->p; k++) {
      orank *R = &w->r[k]; const orc_comm *c = &R->c;
      double *scr = scratch_for(w);
      for (int x = 0; x < c->m1; x++)
        for (int y = 0; y < w->Ny; y++)
          fft_z_line(w, R->out + 2L * c->istride[1] * y + 2L * c->istride[0] * x, scr); /* :3970-3979 */
      free(scr);
    }
    if (!S) for (int k = 0; k < w->p; k++) transpose_local(w, &w->r[k]);
    phase2(w, nthreads);
  } else { /* mode C, :3999-4036 */
    phase1(w, nthreads);
    if (!S) {
#pragma omp parallel for num_threads(nthreads) schedule(static)
      for (int k = 0; k < w->p; k++) transpose_local(w, &w->r[k]);
    }
    phase2(w, nthreads);
  }
}

void orc_world_gather(const orc_world *w, double *global) {
  for (int k = 0; k < w->p; k++) {
    const orc_comm *c = &w->r[k].c;
    const double *out = w->r[k].out;
    for (int x = 0; x < c->osize[0]; x++)
      for (int y = 0; y < c->osize[1]; y++)
        for (int z = 0; z < c->osize[2]; z++) {
          long o = 2 * ((long)x * c->ostride[0] + (long)y * c->ostride[1] + (long)z * c->ostride[2]);
          long g = 2 * (((long)(x + c->ostart[0]) * w->Ny + (y + c->ostart[1])) * w->Nzn + (z + c->ostart[2]));
          global[g] = out[o]; global[g + 1] = out[o + 1];
        }
  }
}
