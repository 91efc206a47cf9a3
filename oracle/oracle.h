/*
 * oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of the reference's hot path (rchyena/offt offt-compute.c) used
 * as the parity checker by tests/, __graft_entry__.smoke() and the cpu_baseline
 * leg of bench.py.  Nothing in offt_amd/ (the product) links, loads or calls it.
 *
 * Pinning status: PARITY UNPINNED.  The reference's arithmetic lives in FFTW3 (not vendored, no
 * version pin beyond Makefile:17-18 "fftw-3.3.2"), the reference has no tests, golden vectors or
 * fixtures, and it is unbuildable in this image (it needs fftw3.h / fftw3-mpi.h; writing stand-in
 * headers is not a reference build), so oracle/_ref is not built.  What the oracle is checked
 * against -- corroboration, not a pin: (1) numpy.fft (pocketfft) full grids and the closed form of
 * the harness ramp; (2) outputs the survey stage recorded from a run of the reference that was
 * linked against declarations-only FFTW headers + MKL (SURVEY.md 8c / BASELINE.md 2: full 18^3 grid
 * on 6 ranks, spot values at 128^3, the default-parameter line for N=128 p=2), kept under
 * tests/golden/ and labelled as such.
 */
#ifndef ORACLE_H
#define ORACLE_H
#ifdef __cplusplus
extern "C" {
#endif

#define ORC_PARAM_COUNT 24

/* ---- 1-D FFT (stands in for the FFTW calls, offt-compute.c:335-341, 416-425) ---- */
typedef struct orc_fft_plan orc_fft_plan;
orc_fft_plan *orc_fft_plan_create(int n);
void orc_fft_plan_destroy(orc_fft_plan *p);
/* in-place forward DFT of `howmany` lines: element k of line h at data[2*(h*dist + k*stride)] */
void orc_fft_execute(const orc_fft_plan *p, double *data, long stride, long dist, int howmany, double *scratch /* 4n doubles */);

/* ---- decomposition (offt_comm_malloc, offt-compute.c:57-315) ---- */
typedef struct orc_comm {
  int p1, p2, rank_x, rank_y;
  int M1, M2, M3, M4, F1, F2, F3, F4, m1, m2, m3, m4, b1, b2, b3, b4;
  int istart[3], isize[3], istride[3], ostart[3], osize[3], ostride[3];
} orc_comm;
void orc_comm_build(orc_comm *c, int Nx, int Ny, int Nz, int p, int rank, int p1, int is_r2c, int is_equalxy, int S);

/* ---- default parameters (params_set_default, offt-compute.c:3127-3225) ---- */
void orc_params_default(int Nx, int Ny, int Nz, int p, int is_r2c, int is_W0, int is_notest, int *v24);

/* ---- a simulated world of p ranks running offt_3d_execute (offt-compute.c:3864-4048) ---- */
typedef struct orc_world orc_world;
/* custom_v: 24 ints, <0 = keep default (set_params_custom, offt-compute.c:3227-3234); may be NULL */
orc_world *orc_world_create(int Nx, int Ny, int Nz, int p, int is_r2c, int is_oned, int is_equalxy, const int *custom_v);
void orc_world_destroy(orc_world *w);
const orc_comm *orc_world_comm(const orc_world *w, int rank);
const int *orc_world_params(const orc_world *w);
long orc_world_local_elems(const orc_world *w);      /* complex elements per rank, run-fft.c:294-300 */
double *orc_world_buffer(orc_world *w, int rank);     /* the rank's in==out array */
/* fill every rank's input block: kind 0 = harness ramp (run-fft.c:46-61), 1 = seeded hash */
void orc_world_fill(orc_world *w, int kind);
/* run the forward transform on all ranks; nthreads > 1 runs the ranks of a
 * row/column group concurrently (OpenMP), the way one MPI rank per core would */
void orc_world_execute(orc_world *w, int nthreads);
/* gather the distributed result into a natural-order [Nx][Ny][Nz'] complex array */
void orc_world_gather(const orc_world *w, double *global);
double orc_hash_val(int x, int y, int z, int c);

#ifdef __cplusplus
}
#endif
#endif
