/*
 * offt_host.c -- C host side of the MI355X-native OFFT: plan lifecycle,
 * decomposition, default parameters, tile pipeline, timers, printing.
 *
 * Mirrors the reference's host layer function by function (all citations are
 * rchyena/offt file:line) but drives hand-written HIP kernels through the thin
 * C ABI in offt_hipk.h and RCCL instead of FFTW and MPI:
 *
 *   offt_comm_malloc      offt-compute.c:57-315    -> comm_build()
 *   params_range_setup /
 *   grid_value_floor /
 *   params_set_default    offt-compute.c:2998-3225 -> params_default()
 *   set_params_custom     offt-compute.c:3227-3234 -> params_custom()
 *   print_params          offt-compute.c:3239-3272
 *   offt_print_time       offt-compute.c:3283-3294
 *   offt_3d_init / _fin   offt-compute.c:3299-3499
 *   offt_3d_execute       offt-compute.c:3864-4048 -> execute_single() / execute_pipeline()
 *   _execute_phase1/2     offt-compute.c:3501-3862 -> the tile loop in execute_pipeline()
 *   communicate_a2a/_wait offt-compute.c:835-900   -> a2a_tile() on comm streams + events
 *   set_buffer(_chunk)    offt-compute.c:672-746   -> ring of (W+1) device tile buffers
 *
 * There is no CPU compute path in this file or anywhere in the library: if the
 * HIP runtime, a GPU or (for size > 1) RCCL is missing, init fails loudly.
 */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <dlfcn.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

#include "offt_hip.h"
#include "offt_hipk.h"
#include "offt_backend.h"

/* ------------------------------------------------------------------------- */
/* errors                                                                     */
/* ------------------------------------------------------------------------- */
static char g_err[1024] = "";
const char *offt_hip_last_error(void) { return g_err; }
#define SET_ERR(...)                                 \
  do {                                               \
    snprintf(g_err, sizeof g_err, __VA_ARGS__);      \
    fprintf(stderr, "offt(hip): %s\n", g_err);       \
  } while (0)
#define HCHECK(call, fail)                                                           \
  do {                                                                               \
    hipError_t e_ = (call);                                                          \
    if (e_ != hipSuccess) {                                                          \
      SET_ERR("%s:%d %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e_));   \
      fail;                                                                          \
    }                                                                                \
  } while (0)

static double wall_seconds(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* ------------------------------------------------------------------------- */
/* RCCL, bound at run time (dlopen) so that the library shares whichever RCCL */
/* the hosting process already loaded (e.g. the one bundled with PyTorch)     */
/* ------------------------------------------------------------------------- */
typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[OFFT_HIP_UNIQUE_ID_BYTES]; } ncclUniqueId;
typedef int ncclResult_t;
#define NCCL_INT8 0
#define NCCL_INT32 2
#define NCCL_FLOAT64 8
#define NCCL_SUM 0
#define NCCL_MAX 2
#define NCCL_IN_PROGRESS 7
static struct {
  void *lib;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *);
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int);
  ncclResult_t (*CommSplit)(ncclComm_t, int, int, ncclComm_t *, void *);
  ncclResult_t (*CommDestroy)(ncclComm_t);
  ncclResult_t (*Send)(const void *, size_t, int, int, ncclComm_t, hipStream_t);
  ncclResult_t (*Recv)(void *, size_t, int, int, ncclComm_t, hipStream_t);
  ncclResult_t (*AllReduce)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t);
  ncclResult_t (*GroupStart)(void);
  ncclResult_t (*GroupEnd)(void);
  const char *(*GetErrorString)(ncclResult_t);
  /* optional (older RCCL builds lack them): asynchronous error state of a communicator, abort of its kernels */
  ncclResult_t (*CommGetAsyncError)(ncclComm_t, ncclResult_t *);
  ncclResult_t (*CommAbort)(ncclComm_t);
} R;

static int rccl_load(void) {
  if (R.lib) return 0;
  const char *cands[4] = {getenv("OFFT_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (int i = 0; i < 4 && !R.lib; i++)
    if (cands[i]) R.lib = dlopen(cands[i], RTLD_NOW | RTLD_GLOBAL);
  if (!R.lib) { SET_ERR("cannot dlopen RCCL (librccl.so.1): %s", dlerror()); return -1; }
#define RSYM(field, name)                                                   \
  do {                                                                      \
    *(void **)(&R.field) = dlsym(R.lib, name);                              \
    if (!R.field) { SET_ERR("RCCL symbol %s missing", name); return -1; }   \
  } while (0)
  RSYM(GetUniqueId, "ncclGetUniqueId");
  RSYM(CommInitRank, "ncclCommInitRank");
  RSYM(CommSplit, "ncclCommSplit");
  RSYM(CommDestroy, "ncclCommDestroy");
  RSYM(Send, "ncclSend");
  RSYM(Recv, "ncclRecv");
  RSYM(AllReduce, "ncclAllReduce");
  RSYM(GroupStart, "ncclGroupStart");
  RSYM(GroupEnd, "ncclGroupEnd");
  RSYM(GetErrorString, "ncclGetErrorString");
  *(void **)(&R.CommGetAsyncError) = dlsym(R.lib, "ncclCommGetAsyncError");
  *(void **)(&R.CommAbort) = dlsym(R.lib, "ncclCommAbort");
  return 0;
}
#define NCHECK(call, fail)                                                                   \
  do {                                                                                       \
    ncclResult_t r_ = (call);                                                                \
    if (r_ != 0) {                                                                           \
      SET_ERR("%s:%d %s -> %s", __FILE__, __LINE__, #call, R.GetErrorString(r_));            \
      fail;                                                                                  \
    }                                                                                        \
  } while (0)

/* ------------------------------------------------------------------------- */
/* world                                                                      */
/* ------------------------------------------------------------------------- */
typedef struct world_state {
  int rank, size, device, have_comm;
  ncclComm_t world;
  int comm_failed; /* an RCCL communicator reported an asynchronous error or a wait timed out: every later execute fails fast */
} world_state;
static world_state G_proc = {0, 1, 0, 0, NULL, 0};

static int is_device_ptr(const void *p);
static int world_max(struct _offt_plan *po, double *v);
static void static_sweep(struct _offt_plan *po, void *user_buf, const struct _offt_params *custom);
struct hip_state;
static void slab_teardown(struct hip_state *st);
static void inv_cache_drop(struct hip_state *st);
static int slab_setup(struct _offt_plan *po, struct hip_state *st);

#ifdef OFFT_TEST_SEAMS
/* Test-only seams (offt_backend.h), compiled ONLY into tests/liboffthip_test.so (make tests/liboffthip_test.so);
 * the product library offt_amd/liboffthip.so is built without this block and exports neither function.
 * The seam state is per thread, so that one test process can run several ranks as threads on one GPU. */
static _Thread_local world_state G_tls;
static _Thread_local int G_tls_on = 0;
#define G (*(G_tls_on ? &G_tls : &G_proc))
static _Thread_local const offt_backend *g_backend = NULL;
void offt_hip_test_set_backend(const offt_backend *b, int rank, int size) {
  g_backend = b;
  G_tls_on = b != NULL;
  if (b) { memset(&G_tls, 0, sizeof G_tls); G_tls.rank = rank; G_tls.size = size; }
}
/* mesh rows of the plan whose exchange is running: a test transport maps (group, member) to a world rank with it */
static _Thread_local int g_test_cur_p1 = 1;
int offt_hip_test_current_p1(void) { return g_test_cur_p1; }
#define TEST_NOTE_MESH(po) (g_test_cur_p1 = (po)->comm->p1)
#else
#define G G_proc
#define g_backend ((const offt_backend *)NULL)
#define TEST_NOTE_MESH(po) ((void)0)
#endif

int offt_hip_world_rank(void) { return G.rank; }
int offt_hip_world_size(void) { return G.size; }

int offt_hip_get_unique_id(void *id128) {
  if (rccl_load()) return -1;
  ncclUniqueId id;
  NCHECK(R.GetUniqueId(&id), return -1);
  memcpy(id128, &id, sizeof id);
  return 0;
}

int offt_hip_set_world(int rank, int size, const void *id128, int device) {
  if (size < 1 || rank < 0 || rank >= size) { SET_ERR("bad world rank %d size %d", rank, size); return -1; }
  HCHECK(hipSetDevice(device), return -1);
  if (G.have_comm) { /* a second call replaces the world: do not leak the previous communicator */
    (void)R.CommDestroy(G.world);
    G.have_comm = 0; G.world = NULL;
  }
  G.rank = rank; G.size = size; G.device = device; G.comm_failed = 0;
  if (size > 1 && !id128) { SET_ERR("offt_hip_set_world: size %d needs an RCCL unique id", size); return -1; }
  if (id128) { /* size 1 with an id: a one-rank communicator, used by the RCCL self-test */
    if (rccl_load()) return -1;
    ncclUniqueId id;
    memcpy(&id, id128, sizeof id);
    NCHECK(R.CommInitRank(&G.world, size, id, rank), return -1);
    G.have_comm = 1;
  }
  return 0;
}

int offt_hip_finalize_world(void) {
  if (G.have_comm) {
    /* after a communication failure the communicator's kernels may still be spinning: abort, do not wait */
    if (G.comm_failed) { if (R.CommAbort) (void)R.CommAbort(G.world); /* else: leaked, ncclCommDestroy could block for ever */ }
    else (void)R.CommDestroy(G.world);
    G.have_comm = 0; G.world = NULL;
  }
  G.rank = 0; G.size = 1; G.comm_failed = 0;
  return 0;
}

/* how many ranks answer on the world communicator: an ncclAllReduce(sum) of ones.  A launcher prints it next to
 * its results so that a run on fewer GPUs than intended cannot pass for the real thing.  -1 on failure. */
int offt_hip_world_count(void) {
  if (G.size == 1 && !G.have_comm) return 1;
  if (!G.have_comm) { SET_ERR("offt_hip_world_count: no communicator"); return -1; }
  int one = 1, sum = -1, *d = NULL;
  hipStream_t s = NULL;
  HCHECK(hipMalloc((void **)&d, sizeof(int)), return -1);
  HCHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking), { (void)hipFree(d); return -1; });
  int rc = -1;
  if (hipMemcpy(d, &one, sizeof one, hipMemcpyHostToDevice) == hipSuccess &&
      R.AllReduce(d, d, 1, NCCL_INT32, NCCL_SUM, G.world, s) == 0 && hipStreamSynchronize(s) == hipSuccess &&
      hipMemcpy(&sum, d, sizeof sum, hipMemcpyDeviceToHost) == hipSuccess)
    rc = sum;
  else SET_ERR("offt_hip_world_count: all-reduce failed");
  (void)hipStreamDestroy(s);
  (void)hipFree(d);
  return rc;
}

/* ------------------------------------------------------------------------- */
/* parameters (offt-compute.c:2998-3234)                                      */
/* ------------------------------------------------------------------------- */
static int ilog2_floor(int n) { int c = -1; while (n > 0) { c++; n >>= 1; } return c; }

/* largest value of the reference's power-of-two lattice {[0,] 1,2,4,..,2^k [, vmax]}
 * that does not exceed raw; raw itself when nothing fits (grid_value_floor,
 * offt-compute.c:3096-3109 over the lists of params_range_setup 3042-3079). */
static int lattice_floor(int raw, int vmax, int has_zero) {
  if (raw >= vmax) return vmax;
  if (raw >= 1) return 1 << ilog2_floor(raw);
  if (raw == 0 && has_zero) return 0;
  return raw;
}

static int p1_floor(int p, int Nx, int Ny, int Nzn, int raw) {
  int pu = p < (Nx < Ny ? Nx : Ny) ? p : (Nx < Ny ? Nx : Ny);
  int pl = p / Nzn > p / Ny ? p / Nzn : p / Ny;
  if (pl < 1) pl = 1;
  int best = raw; /* unchanged when no divisor <= raw is on the lattice */
  for (int d = pl; d <= pu; d++)
    if (p % d == 0 && d <= raw) best = d;
  return best;
}

static void params_default(struct _offt_plan *po) {
  int *v = po->params->v;
  int p = po->p, Nx = po->Nx, Ny = po->Ny;
  int Nzn = po->is_r2c ? po->Nz / 2 + 1 : po->Nz;
  po->params->is_converged = 1;
  po->params->is_infeasible = 0;
  po->params->is_in_database = 0;
  v[_P1_] = p1_floor(p, Nx, Ny, Nzn, (int)sqrt((double)p));
  int p2 = p / v[_P1_];
  int M1 = (Nx + v[_P1_] - 1) / v[_P1_], M2 = (Ny + p2 - 1) / p2;
  int M3 = (Nzn + p2 - 1) / p2, M4 = (Ny + v[_P1_] - 1) / v[_P1_];
  /* phase 1 */
  v[_T1_] = lattice_floor(max(M1 / 16, 1), Nx, 0);
  v[_W1_] = min(min(2, (M1 + v[_T1_] - 1) / v[_T1_]), 10);
  int P1_xy = 8192 / Nzn;
  v[_Px1_] = lattice_floor(min(max((int)sqrt((double)P1_xy), 1), v[_T1_]), Nx, 0);
  v[_Py1_] = lattice_floor(min(max(P1_xy / v[_Px1_], 1), M2), Ny, 0);
  v[_Fz_] = lattice_floor(min(max(p2 / 2, 0), v[_T1_] * M2), Nx * Ny, 1);
  v[_FP1_] = lattice_floor(min(max(v[_Fz_], 0), v[_T1_] / v[_Px1_] * M2 / v[_Py1_]), Nx * Ny, 1);
  int U1_xz = 8192 / Ny;
  v[_Ux1_] = lattice_floor(min(max((int)sqrt((double)U1_xz), 1), v[_T1_]), Nx, 0);
  v[_Uz1_] = lattice_floor(min(max(U1_xz / v[_Ux1_], 1), M3), Nzn, 0);
  v[_FU1_] = lattice_floor(min(max(v[_Fz_], 0), v[_T1_] / v[_Ux1_] * M3 / v[_Uz1_]), Nx * Nzn, 1);
  v[_Fy1_] = lattice_floor(min(max(v[_Fz_], 0), v[_T1_] * M3), Nx * Nzn, 1);
  v[_Ry_] = 5;
  /* phase 2 */
  v[_T2_] = lattice_floor(max(M3 / 16, 1), Nzn, 0);
  v[_W2_] = min(min(2, (M3 + v[_T2_] - 1) / v[_T2_]), 10);
  int P2_xz = 8192 / Ny;
  v[_Pz2_] = lattice_floor(min(max((int)sqrt((double)P2_xz), 1), v[_T2_]), Nzn, 0);
  v[_Px2_] = lattice_floor(min(max(P2_xz / v[_Pz2_], 1), M1), Nx, 0);
  v[_Fy2_] = lattice_floor(min(max(v[_P1_] / 2, 0), v[_T2_] * M1), Nx * Nzn, 1);
  v[_FP2_] = lattice_floor(min(max(v[_Fy2_], 0), M1 / v[_Px2_] * v[_T2_] / v[_Pz2_]), Nx * Nzn, 1);
  int U2_yz = 8192 / Nx;
  v[_Uz2_] = lattice_floor(min(max((int)sqrt((double)U2_yz), 1), v[_T2_]), Nzn, 0);
  v[_Uy2_] = lattice_floor(min(max(U2_yz / v[_Uz2_], 1), M4), Ny, 0);
  v[_FU2_] = lattice_floor(min(max(v[_FP2_], 0), M4 / v[_Uy2_] * v[_T2_] / v[_Uz2_]), Ny * Nzn, 1);
  v[_Fx_] = lattice_floor(min(max(v[_FP2_], 0), v[_T2_] * M4), Ny * Nzn, 1);
  v[_V_] = 0;
  v[_S_] = 0;
  static const int fidx[8] = {_Fz_, _FP1_, _FU1_, _Fy1_, _Fy2_, _FP2_, _FU2_, _Fx_};
  if (po->is_W0) {
    v[_W1_] = v[_W2_] = 0;
    for (int i = 0; i < 8; i++) v[fidx[i]] = 0;
  }
  if (po->is_notest)
    for (int i = 0; i < 8; i++) v[fidx[i]] = 0;
}

static void params_custom(struct _offt_plan *po, const struct _offt_params *c) {
  if (!c) return;
  for (int i = 0; i < PARAM_COUNT; i++)
    if (c->v[i] >= 0) po->params->v[i] = c->v[i];
}

static const char *const k_param_names[PARAM_COUNT] = {
    "P1", "T1", "W1", "Px1", "Py1", "Fz", "FP1", "Ux1", "Uz1", "FU1", "Fy1", "Ry",
    "T2", "W2", "Pz2", "Px2", "Fy2", "FP2", "Uz2", "Uy2", "FU2", "Fx", "V", "S"};

void print_params(int *v) {
  for (int i = 0; i < PARAM_COUNT; i++) {
    if (v[i] < 0) continue;
    printf("%s %d ", k_param_names[i], v[i]);
  }
  printf("\n");
}

void offt_print_time(double *t) {
  printf("%.5f  %.5f %.5f %.5f %.5f %.5f %.5f  %.5f  %.5f %.5f %.5f %.5f  %.5f %.5f %.5f %.5f\n",
         t[ALL], t[INIT1], t[WAIT1], t[TEST1], t[INIT2], t[WAIT2], t[TEST2], t[TRANSPOSE],
         t[PACK1], t[UNPACK1], t[PACK2], t[UNPACK2], t[FFTz], t[FFTy1], t[FFTy2], t[FFTx]);
}

/* ------------------------------------------------------------------------- */
/* decomposition (offt-compute.c:57-315, A2AV + STRIDE build)                 */
/* ------------------------------------------------------------------------- */
static int blk_start(int r, int F, int b, int p) {
  return (r < p - b) ? r * F : (p - b) * F + (r - (p - b)) * (F + 1);
}
static int blk_size(int r, int F, int b, int p) { return (r < p - b) ? F : F + 1; }

static struct _offt_comm *comm_build(const struct _offt_plan *po) {
  struct _offt_comm *c = (struct _offt_comm *)calloc(1, sizeof *c);
  int Nx = po->Nx, Ny = po->Ny, Nzn = po->is_r2c ? po->Nz / 2 + 1 : po->Nz;
  int p1 = c->p1 = po->params->v[_P1_];
  int p2 = c->p2 = po->p / p1;
  int rx = po->rank / p2, ry = po->rank % p2; /* offt-compute.c:75-76 */
  c->M1 = (Nx + p1 - 1) / p1; c->M2 = (Ny + p2 - 1) / p2;
  c->M3 = (Nzn + p2 - 1) / p2; c->M4 = (Ny + p1 - 1) / p1;
  c->F1 = Nx / p1; c->F2 = Ny / p2; c->F3 = Nzn / p2; c->F4 = Ny / p1;
  c->b1 = Nx % p1; c->b2 = Ny % p2; c->b3 = Nzn % p2; c->b4 = Ny % p1;
  c->m1 = blk_size(rx, c->F1, c->b1, p1);
  c->m2 = blk_size(ry, c->F2, c->b2, p2);
  c->m3 = blk_size(ry, c->F3, c->b3, p2);
  c->m4 = blk_size(rx, c->F4, c->b4, p1);
  c->istart[0] = blk_start(rx, c->F1, c->b1, p1);
  c->istart[1] = blk_start(ry, c->F2, c->b2, p2);
  c->istart[2] = 0;
  c->isize[0] = c->m1; c->isize[1] = c->m2; c->isize[2] = po->Nz;
  c->istride[0] = (c->M2 * p2 > c->M4 * p1) ? c->M2 * c->M3 * p2 : c->M4 * p1 * c->M3;
  c->istride[1] = c->M3 * p2;
  c->istride[2] = 1;
  c->ostart[0] = 0;
  c->ostart[1] = blk_start(rx, c->F4, c->b4, p1);
  c->ostart[2] = blk_start(ry, c->F3, c->b3, p2);
  c->osize[0] = Nx; c->osize[1] = c->m4; c->osize[2] = c->m3;
  if (po->params->v[_S_]) { /* x-y-z */
    c->ostride[0] = c->M3 * c->M4; c->ostride[1] = c->M3; c->ostride[2] = 1;
  } else if (po->is_equalxy && c->M1 == c->M4) { /* y-z-x */
    c->ostride[0] = 1; c->ostride[1] = c->M1 * p1 * c->M3; c->ostride[2] = c->M1 * p1;
  } else { /* z-y-x */
    c->ostride[0] = 1; c->ostride[1] = c->M1 * p1; c->ostride[2] = c->M1 * p1 * c->M4;
  }
  return c;
}

/* ------------------------------------------------------------------------- */
/* device state                                                               */
/* ------------------------------------------------------------------------- */
typedef struct hip_state {
  int prec;
  size_t esz;           /* bytes per complex element */
  int use_pipeline;     /* 0: single-rank direct 3-pass path, 1: tile pipeline */
  const offt_backend *be;
  void *s_compute; int own_stream;
  void *s_comm1, *s_comm2;
  void *ev0, *ev1, *evp[4];
  void *work; size_t work_elems; /* single path: transposed-output scratch */
  void *work2;                   /* ... x-y-z output (S = 1): the second rotation's scratch */
  /* pipeline */
  int T, ntiles, ring;
  int slab_zyx;          /* p1 == 1 and z-y-x output: single-exchange slab schedule, see execute_slab() */
  int t1_custom, t2_custom; /* the caller fixed T1 / T2 (run-fft -T / -t): use them as given */
  int sT, sTz, sNt, sH;  /* slab schedule: x-tile, z-chunk thickness, #tiles, #chunks */
  int slab_yc;           /* slab blocks laid out [chunk][x_t][z in chunk][y] (y contiguous), see execute_slab() */
  int slab_inplace;      /* K2 stores its y-transformed chunk straight into the caller's array and K3 runs in place: no R2 volume */
  size_t sblkS;          /* elements per (tile, peer) block of S1 / R1: [z_l][y][x_t] */
  size_t sBc;            /* y-contiguous layout: pitch of a (peer, chunk) block = all its tiles + a de-aliasing pad (slab_setup) */
  void *S1, *R1, *R2;    /* packed send volume, receive volume (same layout), y-transformed volume */
  void **ev_s1;          /* per x-tile: K1 done */
  void **ev_sa;          /* per z-chunk: every tile's share of the chunk has arrived */
  int x1, x2;            /* exchange 1 / 2 really happen (p2 > 1 / p1 > 1, or forced for self-tests) */
  size_t blk1, blk2;     /* pitch of a peer block in elements: ex1 tile block, ex2 full block (both with a de-aliasing pad) */
  size_t B2t;            /* pencil_yc: pitch of a (peer, x-tile) block of the exchange-2 volume = its H chunks + pad */
  int Tz2, H2;           /* pencil schedule, phase 2: z-chunk thickness (T2) and number of chunks of exchange 2 / FFTx */
  int pencil_yc;         /* pencil exchange volumes laid out y- / x-contiguous (two strided sides instead of four), see execute_pipeline() */
  void **ev_a2;          /* per z-chunk: the chunk's share of every x-tile has arrived (exchange 2) */
  void **ev_t2;          /* per x-tile (mirrored pencil schedule): the tile's mirrored exchange 2 has landed */
  int uses_rccl;         /* this plan exchanges over RCCL communicators (watched while waiting) */
  int skip_mask;         /* diagnostics (offt_hip_set_debug_skip): 1 = no FFT passes, 2 = no exchanges */
  void **send1, **recv1; /* ring */
  void **ev_k1, **ev_a1, **ev_k2;
  void *send2, *recv2;
  /* per-block base tables of the packing passes (offt_pass_desc::out_block_tab, device memory): the block this rank keeps
   * for itself is stored straight into the receive side, where the next pass reads it, and the exchange skips this rank */
  int self_bypass;
  long long *tab_s1;     /* slab schedule: K1's blocks into S1, the self block into R1 */
  long long **tab_r1;    /* pencil schedule, per ring slot: K1's blocks into send1[r], the self block into recv1[r] */
  long long *tab_x2;     /* pencil schedule: K2's blocks into send2, the self block into recv2 */
  /* direct-store exchange ("p2p", offt_hip_set_exchange / OFFT_EXCHANGE=p2p): the packing passes store every block
   * straight into the receive volume of the rank it is for (peer memory mapped at plan time), flags say when */
  int want_p2p;          /* asked for */
  int p2p;               /* in use by this mesh (every rank could map its peers) */
  struct p2p_group { int which, n, self, nslots; unsigned long long *flags, **pflags; } g1, g2;
  void **peer_r1, **peer_x2; /* the group members' receive volumes: exchange 1 (R1 / the recv1 ring), exchange 2 (recv2) */
  void *recv1_base;      /* pencil schedule: the recv1 ring as ONE allocation (one mapping per peer) */
  long long *tab_p1, **tab_pr, *tab_p2; /* per-block base tables into the peers' volumes: slab K1, pencil K1 per ring slot, pencil K2 */
  unsigned long long epoch, *use1, tiles2, bar1, bar2; /* flag values: transforms done, uses of each ring slot, tiles sent in exchange 2, barriers */
  unsigned long long *p2p_status; /* host-visible: a wait kernel gave up */
  ncclComm_t comm1, comm2; int have_comm1, have_comm2;
  void *stage; size_t stage_bytes;
  int variant[3];
  double out_scale;
  int yx_fused;        /* the last single-rank execute alternated launches i and i+1 over groups of planes: i + 1 (0: none) */
  void *s_aux, *ev_aux[4]; /* ... with the x launches on this second stream, ordered behind their y launch by these events */
  /* plan-level options (offt_hip_set_option).  The environment variables of the same meaning are read ONCE, in
   * offt_3d_init, as defaults: nothing below depends on the process environment after the plan exists */
  struct {
    int zgroup_mib;      /* single rank: group size of the alternating launches in MiB; 0 = off; -1 = the library's rule */
    int zgroup_streams;  /* ... 2 = consumer launches on a second stream */
    int slab_chunk_mib;  /* slab schedule: largest z-chunk of the y-transformed volume (Infinity-Cache reuse K2 -> K3) */
    int comm_streams;    /* pencil schedule: 2 = row and column exchanges on separate streams */
    long long min_msg;   /* tiles are merged upwards until a per-peer message has this many bytes */
    int f32_pairs;       /* single precision: 0 = never use the column-pair kernels */
    int block_pad;       /* exchange volumes: per-peer / per-chunk blocks padded against HBM channel aliasing */
    double exec_timeout_s, p2p_timeout_s;
  } opt;
  double *agree_d, *agree_h; /* world_max(): 2 p doubles on the device and on the host, made once at plan time so that the
                                agreement itself never allocates (it runs right after a rank may have run out of memory) */
  int k1_streams;      /* slab schedule: 2 = the K1 launches of consecutive x-tiles alternate between two streams, so that one tile's
                          last workgroups and the next tile's first ones share the chip (a launch tail is ~7 % of a 190-us launch) */
  void *s_k1b, *ev_fork;
  int wpad, wrow;      /* scratch volume W: extra elements per x-plane / per y-line (de-aliasing pads) */
  int async;
  int timed;            /* this call records timing events (synchronous call, or host-staged) */
  double last_dev_s, pass_s[3];
  int pass_slot[3];
  int warned_in;
  struct step_list *rec; /* non-NULL: the schedule is being recorded, not run (multi-rank inverse) */
  int rec_tag;           /* ... tag of the steps being recorded (see step) */
  struct step_list *inv_cache; /* the recorded schedule of the last multi-rank inverse, kept while the plan's buffers, kernels and */
  const void *inv_data;        /* ... the caller's array stay what they were (inv_cache_drop: re-setup, options, variants) */
} hip_state;

/* ---- default backend: HIP + RCCL ----------------------------------------- */
static void *hb_malloc(size_t bytes) {
  void *p = NULL;
  if (bytes == 0) bytes = 16;
  HCHECK(hipMalloc(&p, bytes), return NULL);
  return p;
}
static void hb_free(void *p) { if (p) (void)hipFree(p); }
static int hb_pass(const offt_pass_desc *d, const void *in, void *out, void *stream) {
  /* OFFT_LOG_PASSES=1: one line per launch (kernel family, length, batch, addressing) -- profiling tools match it, in
   * dispatch order, with the kernel trace of rocprofv3 to turn durations into bytes per second */
  static int log_passes = -1;
  if (log_passes < 0) log_passes = getenv("OFFT_LOG_PASSES") && atoi(getenv("OFFT_LOG_PASSES"));
  if (log_passes && d->n > 0 && d->ncols > 0 && d->nb1 > 0 && d->nb2 > 0)
    fprintf(stderr, "offt-pass %s n %d ncols %d nb1 %d nb2 %d prec %d in_contig %d out_contig %d in_split %d out_split %d elems %lld\n",
            offt_hipk_kernel_name(d), d->n, d->ncols, d->nb1, d->nb2, d->precision, d->in_contig, d->out_contig, d->in_split,
            d->out_split, (long long)d->n * d->ncols * d->nb1 * d->nb2);
#ifdef OFFT_TEST_SEAMS
  /* test build: OFFT_TEST_SLOW_PASS_MS=<ms> holds the stream that long in front of every pass, so that the device lags behind
   * the host and an exchange that does not wait for the kernel that packs its data really reads too early
   * (tools/async_negative_control.sh) */
  static double slow_ms = -1.0;
  static int slow_odd = 0; /* OFFT_TEST_SLOW_RANKS=odd: only the odd ranks are slow, so that the even ones run ahead of them */
  if (slow_ms < 0) {
    slow_odd = getenv("OFFT_TEST_SLOW_RANKS") && !strcmp(getenv("OFFT_TEST_SLOW_RANKS"), "odd");
    slow_ms = getenv("OFFT_TEST_SLOW_PASS_MS") ? atof(getenv("OFFT_TEST_SLOW_PASS_MS")) : 0.0;
  }
  if (slow_ms > 0 && (!slow_odd || (G.rank & 1)) && d->n > 0 && d->ncols > 0 && d->nb1 > 0 && d->nb2 > 0) (void)offt_hipk_delay(slow_ms, stream);
#endif
  int rc = offt_hipk_fft_pass(d, in, out, stream);
  if (rc) SET_ERR("pass n=%d failed: %s", d->n, offt_hipk_last_error());
  return rc;
}
static int hb_prepare(int n, int prec) { return offt_hipk_prepare(n, prec); }
static void *hb_stream_create(void) {
  hipStream_t s;
  HCHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking), return NULL);
  return s;
}
static void hb_stream_destroy(void *s) { if (s) (void)hipStreamDestroy((hipStream_t)s); }
static void *hb_event_create(void) {
  hipEvent_t e;
  HCHECK(hipEventCreate(&e), return NULL);
  return e;
}
static void hb_event_destroy(void *e) { if (e) (void)hipEventDestroy((hipEvent_t)e); }
static int hb_event_record(void *e, void *s) { HCHECK(hipEventRecord((hipEvent_t)e, (hipStream_t)s), return -1); return 0; }
static int hb_stream_wait(void *s, void *e) { HCHECK(hipStreamWaitEvent((hipStream_t)s, (hipEvent_t)e, 0), return -1); return 0; }
static int hb_stream_sync(void *s) { HCHECK(hipStreamSynchronize((hipStream_t)s), return -1); return 0; }
static double hb_event_ms(void *a, void *b) {
  float ms = 0;
  if (hipEventElapsedTime(&ms, (hipEvent_t)a, (hipEvent_t)b) != hipSuccess) {
    (void)hipGetLastError(); /* an event of a pass that did not run: not an error, clear the sticky code */
    return 0;
  }
  return ms;
}
#ifdef OFFT_TEST_SEAMS
/* test-only transport override (offt_backend.h): the HIP kernels, streams and events stay,
 * only the exchange goes through a callback.  Lets several ranks share ONE GPU (RCCL refuses
 * duplicate devices) so the multi-rank schedules run on real device memory in -m gpu tests. */
static _Thread_local offt_test_transport_fn g_test_transport = NULL;
void offt_hip_test_set_transport(offt_test_transport_fn fn, int rank, int size) {
  g_test_transport = fn;
  G_tls_on = fn != NULL;
  if (fn) { memset(&G_tls, 0, sizeof G_tls); G_tls.rank = rank; G_tls.size = size; }
}
static _Thread_local offt_test_transport_async_fn g_test_transport_async = NULL;
void offt_hip_test_set_transport_async(offt_test_transport_async_fn fn) { g_test_transport_async = fn; }
#else
#define g_test_transport ((offt_test_transport_fn)NULL)
#define g_test_transport_async ((offt_test_transport_async_fn)NULL)
#endif

/* all-to-all of one tile inside a row/column group (communicate_a2a(v),
 * offt-compute.c:835-881): grouped ncclSend/ncclRecv over xGMI */
static int hb_a2a(void *ctx, int which, int npeers, const int *peer_rank_in_comm, const void *const *sendp,
                  const size_t *sendbytes, void *const *recvp, const size_t *recvbytes, void *stream) {
  hip_state *st = (hip_state *)ctx;
  if (g_test_transport && g_test_transport_async) /* the test enqueues the copies on `stream` itself: nothing is drained */
    return g_test_transport_async(which, npeers, peer_rank_in_comm, sendp, sendbytes, recvp, recvbytes, stream);
  if (g_test_transport) { /* synchronous, host-staged by the test */
    /* (a rehearsal transport that moves nothing may skip the host synchronisation: OFFT_TEST_TRANSPORT_NOSYNC=1) */
    static int nosync = -1;
    if (nosync < 0) nosync = getenv("OFFT_TEST_TRANSPORT_NOSYNC") && atoi(getenv("OFFT_TEST_TRANSPORT_NOSYNC"));
    if (!nosync) HCHECK(hipStreamSynchronize((hipStream_t)stream), return -1);
    return g_test_transport(which, npeers, peer_rank_in_comm, sendp, sendbytes, recvp, recvbytes);
  }
  ncclComm_t cm = which == 1 ? st->comm1 : (which == 2 ? st->comm2 : G.world);
  if (!cm) { SET_ERR("exchange %d without a communicator", which); return -1; }
  NCHECK(R.GroupStart(), return -1);
  int rc = 0;
  for (int a = 0; a < npeers && !rc; a++) {
    if (sendbytes[a]) NCHECK(R.Send(sendp[a], sendbytes[a], NCCL_INT8, peer_rank_in_comm[a], cm, (hipStream_t)stream), rc = -1);
    if (!rc && recvbytes[a]) NCHECK(R.Recv(recvp[a], recvbytes[a], NCCL_INT8, peer_rank_in_comm[a], cm, (hipStream_t)stream), rc = -1);
  }
  if (rc) { (void)R.GroupEnd(); return -1; } /* never leave a group open behind a failed call */
  NCHECK(R.GroupEnd(), return -1);
  return 0;
}
static int hb_memcpy_dd(void *dst, const void *src, size_t bytes, void *s) {
  HCHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, (hipStream_t)s), return -1);
  return 0;
}
static int hb_upload(void *dst, const void *src, size_t bytes) {
  HCHECK(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice), return -1);
  return 0;
}
#ifdef OFFT_TEST_SEAMS
static _Thread_local offt_test_peer_open_fn g_test_peer_open = NULL;
static _Thread_local offt_test_hook_fn g_test_p2p_hook = NULL;
void offt_hip_test_set_p2p(offt_test_peer_open_fn fn, offt_test_hook_fn hook) { g_test_peer_open = fn; g_test_p2p_hook = hook; }
#define P2P_HOOK() do { if (g_test_p2p_hook) g_test_p2p_hook(); } while (0)
#else
#define g_test_peer_open ((offt_test_peer_open_fn)NULL)
#define P2P_HOOK() ((void)0)
#endif

/* map the group members' allocations (offt_backend::peer_open): hipIpc handles, gathered with the group's own exchange
 * primitive (64 bytes to and from every member), opened with peer access enabled on first use.  What the reference gets
 * from MPI for free -- the library moves the packed blocks -- becomes an address the packing kernel can store to. */
static int hb_peer_open(void *ctx, int which, int n, int self, void *local, size_t bytes, void **peers) {
  hip_state *st = (hip_state *)ctx;
  if (g_test_peer_open) return g_test_peer_open(which, n, self, local, bytes, peers);
  (void)bytes;
  const size_t HB = sizeof(hipIpcMemHandle_t);
  hipIpcMemHandle_t mine;
  char *d = NULL, *h = (char *)malloc(2 * HB * (size_t)n);
  int rc = -1;
  if (!h) return -1;
  if (hipIpcGetMemHandle(&mine, local) != hipSuccess) { (void)hipGetLastError(); goto out; }
  if (hipMalloc((void **)&d, 2 * HB * (size_t)n) != hipSuccess) { (void)hipGetLastError(); d = NULL; goto out; }
  for (int a = 0; a < n; a++) memcpy(h + (size_t)a * HB, &mine, HB);
  memset(h + HB * (size_t)n, 0, HB * (size_t)n);
  if (hipMemcpy(d, h, 2 * HB * (size_t)n, hipMemcpyHostToDevice) != hipSuccess) goto out;
  {
    const void *sp[n]; void *rp[n]; size_t sb[n], rb[n]; int pr[n];
    for (int a = 0; a < n; a++) { pr[a] = a; sp[a] = d + (size_t)a * HB; rp[a] = d + (size_t)(n + a) * HB; sb[a] = rb[a] = HB; }
    if (hb_a2a(st, which, n, pr, sp, sb, rp, rb, st->s_compute)) goto out;
    if (hipStreamSynchronize((hipStream_t)st->s_compute) != hipSuccess) goto out;
  }
  if (hipMemcpy(h, d, 2 * HB * (size_t)n, hipMemcpyDeviceToHost) != hipSuccess) goto out;
  rc = 0;
  for (int a = 0; a < n; a++) peers[a] = NULL;
  for (int a = 0; a < n; a++) {
    if (a == self) { peers[a] = local; continue; }
    hipIpcMemHandle_t hm;
    memcpy(&hm, h + (size_t)(n + a) * HB, HB);
    if (hipIpcOpenMemHandle(&peers[a], hm, hipIpcMemLazyEnablePeerAccess) != hipSuccess) {
      SET_ERR("hipIpcOpenMemHandle for group %d member %d failed: %s", which, a, hipGetErrorString(hipGetLastError()));
      peers[a] = NULL; rc = -1;
    }
  }
  if (rc) for (int a = 0; a < n; a++) if (a != self && peers[a]) { (void)hipIpcCloseMemHandle(peers[a]); peers[a] = NULL; }
out:
  if (d) (void)hipFree(d);
  free(h);
  return rc;
}
static void hb_peer_close(void *ctx, int n, int self, void **peers) {
  (void)ctx;
  if (g_test_peer_open || !peers) return;
  for (int a = 0; a < n; a++) if (a != self && peers[a]) (void)hipIpcCloseMemHandle(peers[a]);
}
static void *hb_flag_alloc(size_t bytes, int host_visible) {
  void *p = NULL;
  if (host_visible) {
    HCHECK(hipHostMalloc(&p, bytes, hipHostMallocMapped), return NULL);
    memset(p, 0, bytes);
    return p;
  }
  HCHECK(hipMalloc(&p, bytes), return NULL);
  HCHECK(hipMemset(p, 0, bytes), { (void)hipFree(p); return NULL; });
  HCHECK(hipDeviceSynchronize(), { (void)hipFree(p); return NULL; });
  return p;
}
static void hb_flag_free(void *p, int host_visible) {
  if (!p) return;
  if (host_visible) (void)hipHostFree(p); else (void)hipFree(p);
}
static int hb_flag_signal(int n, unsigned long long *const *addr, unsigned long long value, void *stream) {
  const int rc = offt_hipk_flag_signal(n, addr, value, stream);
  if (rc) SET_ERR("flag signal failed: %s", offt_hipk_last_error());
  return rc;
}
static int hb_flag_wait(int n, unsigned long long *const *addr, unsigned long long value, unsigned long long *status, double timeout_s, void *stream) {
  const int rc = offt_hipk_flag_wait(n, addr, value, status, timeout_s, stream);
  if (rc) SET_ERR("flag wait failed: %s", offt_hipk_last_error());
  return rc;
}
static const offt_backend k_hip_backend = {
    hb_malloc, hb_free, hb_prepare, hb_pass, hb_stream_create, hb_stream_destroy, hb_event_create,
    hb_event_destroy, hb_event_record, hb_stream_wait, hb_stream_sync, hb_event_ms, hb_a2a, hb_memcpy_dd, hb_upload,
    hb_peer_open, hb_peer_close, hb_flag_alloc, hb_flag_free, hb_flag_signal, hb_flag_wait};

/* ------------------------------------------------------------------------- */
/* helpers                                                                    */
/* ------------------------------------------------------------------------- */
static size_t local_elems(const struct _offt_comm *c) {
  /* run-fft.c:294-300 */
  return (c->M2 * c->p2 > c->M4 * c->p1) ? (size_t)c->M1 * c->M2 * c->M3 * c->p2
                                         : (size_t)c->M1 * c->M3 * c->M4 * c->p1;
}

long long offt_hip_local_bytes(const struct _offt_plan *po) {
  const hip_state *st = (const hip_state *)po->hip_state;
  return (long long)(local_elems(po->comm) * st->esz);
}

static int is_device_ptr(const void *p) {
  hipPointerAttribute_t at;
  memset(&at, 0, sizeof at);
  hipError_t e = hipPointerGetAttributes(&at, p);
  if (e != hipSuccess) { (void)hipGetLastError(); return 0; }
  return at.type == hipMemoryTypeDevice || at.type == hipMemoryTypeManaged;
}

static void desc_init(offt_pass_desc *d, const hip_state *st, int n, int dir, int axis) {
  memset(d, 0, sizeof *d);
  d->n = n;
  d->precision = st->prec;
  d->direction = dir;
  d->nb1 = d->nb2 = 1;
  d->variant = st->variant[axis];
  d->scale = 1.0;
  d->no_pairs = !st->opt.f32_pairs;
}

/* device copy of a per-block base table (offt_pass_desc::in_block_tab / out_block_tab): block b of the split axis sits
 * b * stride elements behind the launch pointer, except the blocks of member `self` (nper consecutive blocks per group
 * member), which sit `delta` elements further on -- in the receive buffer, where the next pass reads them.  This is the
 * reference's pack into block a of a2as (offt-compute.c:1084-1109) with the copy to self that MPI_Ialltoall makes of
 * block `rank` (835-881) folded into the pack. */
static long long *tab_self(const hip_state *st, int members, int nper, long long stride, int self, long long delta) {
  const int n = members * nper;
  long long *h = (long long *)malloc(sizeof(long long) * (size_t)n), *d = (long long *)st->be->dmalloc(sizeof(long long) * (size_t)n);
  if (!h || !d) { free(h); st->be->dfree(d); return NULL; }
  for (int b = 0; b < n; b++) h[b] = (long long)b * stride + (b / nper == self ? delta : 0);
  if (st->be->upload(d, h, sizeof(long long) * (size_t)n)) { st->be->dfree(d); d = NULL; }
  free(h);
  return d;
}
/* distance between two buffers in elements; 0 with *ok = 0 when it is not a whole (for single precision: even) number */
static long long elem_delta(const hip_state *st, const void *to, const void *from, int *ok) {
  const long long db = (long long)((const char *)to - (const char *)from);
  if (db % 16) { *ok = 0; return 0; }
  return db / (long long)st->esz;
}

/* ------------------------------------------------------------------------- */
/* direct-store exchange ("p2p"): groups, flags, tables                        */
/*                                                                           */
/* The reference packs peer a's share into block a of a send buffer           */
/* (offt-compute.c:1084-1109) and MPI_Ialltoall copies block a into peer a's  */
/* receive buffer, block `rank` (835-881).  Here the packing kernel stores     */
/* block a straight into peer a's receive volume at sender slot `self`        */
/* (per-block base tables, offt_hipk.h): no send volume, no copy kernel, the  */
/* bytes cross xGMI once as 128-B stores.  What is left of the exchange is     */
/* its completion: one 64-bit flag word per (slot, sender) in every rank's    */
/* memory, written by the sender behind its kernel (flag_signal) and polled    */
/* by a one-wave wait on the receiver's stream (flag_wait):                    */
/*   READY  "my blocks of this tile / transform are in your volume"           */
/*   FREE   "I have consumed what you stored: overwrite it"                   */
/* Values only grow, so nothing is ever reset.                                 */
/* ------------------------------------------------------------------------- */
typedef struct p2p_group p2p_group;
static void p2p_group_close(hip_state *st, p2p_group *g) {
  const offt_backend *be = st->be;
  if (g->pflags) { be->peer_close(st, g->n, g->self, (void **)g->pflags); free(g->pflags); }
  be->flag_free(g->flags, 0);
  memset(g, 0, sizeof *g);
}
static int p2p_group_open(hip_state *st, p2p_group *g, int which, int n, int self, int nslots) {
  const offt_backend *be = st->be;
  memset(g, 0, sizeof *g);
  g->which = which; g->n = n; g->self = self; g->nslots = nslots;
  const size_t bytes = sizeof(unsigned long long) * (size_t)nslots * (size_t)n;
  g->flags = (unsigned long long *)be->flag_alloc(bytes, 0);
  g->pflags = (unsigned long long **)calloc((size_t)n, sizeof(void *));
  if (!g->flags || !g->pflags) return -1;
  return be->peer_open(st, which, n, self, g->flags, bytes, (void **)g->pflags);
}
/* flag words: slot-major, one word per sender */
static int p2p_signal(hip_state *st, const p2p_group *g, int slot, unsigned long long value, void *stream) {
  unsigned long long *addr[OFFT_HIPK_MAX_FLAGS];
  for (int a = 0; a < g->n; a++) addr[a] = g->pflags[a] + (size_t)slot * g->n + g->self;
  return st->be->flag_signal(g->n, addr, value, stream);
}
static int p2p_wait(hip_state *st, const p2p_group *g, int slot, unsigned long long value, void *stream) {
  unsigned long long *addr[OFFT_HIPK_MAX_FLAGS];
  if (value == 0) return 0;
  for (int a = 0; a < g->n; a++) addr[a] = g->flags + (size_t)slot * g->n + a;
  const double limit = st->opt.p2p_timeout_s;
  P2P_HOOK(); /* (thread worlds of the tests: every signal this wait depends on is enqueued by now) */
  return st->be->flag_wait(g->n, addr, value, st->p2p_status, limit, stream);
}
/* per-block base table of a packing pass in p2p mode: block b (nper blocks per group member) goes into member b / nper's
 * volume at sender slot `self`, i.e. (peer - mine) elements from the launch pointer (which is based on this rank's OWN
 * receive volume) plus (self * nper + b % nper) * stride */
static long long *tab_peers(const hip_state *st, int members, int nper, long long stride, int self, void *const *peer, const void *mine) {
  const int n = members * nper;
  long long *h = (long long *)malloc(sizeof(long long) * (size_t)n), *d = (long long *)st->be->dmalloc(sizeof(long long) * (size_t)n);
  int ok = h && d;
  for (int b = 0; b < n && ok; b++) {
    const long long delta = elem_delta(st, peer[b / nper], mine, &ok);
    h[b] = delta + ((long long)self * nper + b % nper) * stride;
  }
  if (ok && st->be->upload(d, h, sizeof(long long) * (size_t)n)) ok = 0;
  free(h);
  if (!ok) { st->be->dfree(d); return NULL; }
  return d;
}
static void p2p_teardown(hip_state *st) {
  const offt_backend *be = st->be;
  if (st->peer_r1) { be->peer_close(st, st->g1.n, st->g1.self, st->peer_r1); free(st->peer_r1); st->peer_r1 = NULL; }
  if (st->peer_x2) { be->peer_close(st, st->g2.n, st->g2.self, st->peer_x2); free(st->peer_x2); st->peer_x2 = NULL; }
  if (st->g1.flags || st->g1.pflags) p2p_group_close(st, &st->g1);
  if (st->g2.flags || st->g2.pflags) p2p_group_close(st, &st->g2);
  be->dfree(st->tab_p1); st->tab_p1 = NULL;
  be->dfree(st->tab_p2); st->tab_p2 = NULL;
  if (st->tab_pr) { for (int r = 0; r < st->ring; r++) be->dfree(st->tab_pr[r]); free(st->tab_pr); st->tab_pr = NULL; }
  free(st->use1); st->use1 = NULL;
  be->flag_free(st->p2p_status, 1); st->p2p_status = NULL;
  st->p2p = 0;
}

/* ------------------------------------------------------------------------- */
/* plan                                                                       */
/* ------------------------------------------------------------------------- */
/* ------------------------------------------------------------------------- */
/* pencil schedule buffers (set_buffer, offt-compute.c:710-746):               */
/*   phase 1: ring of (W1+1) send/receive pairs of T1 x-planes                 */
/*   phase 2: one send and one receive volume in z-chunks of T2 planes,        */
/*            per peer [chunk][x][y_l][z in chunk]                             */
/* Rebuilt when the static sweep tries another T1/W1/T2 or another mesh.       */
/* ------------------------------------------------------------------------- */
static void ring_teardown(hip_state *st) {
  const offt_backend *be = st->be;
  inv_cache_drop(st);
  for (int r = 0; r < st->ring; r++) {
    if (st->send1 && !st->recv1_base) be->dfree(st->send1[r]);
    if (st->recv1 && st->recv1 != st->send1 && !st->recv1_base) be->dfree(st->recv1[r]);
    if (st->ev_k1) be->event_destroy(st->ev_k1[r]);
    if (st->ev_a1) be->event_destroy(st->ev_a1[r]);
    if (st->ev_k2) be->event_destroy(st->ev_k2[r]);
  }
  be->dfree(st->recv1_base); st->recv1_base = NULL;
  if (st->recv1 != st->send1) free(st->recv1);
  free(st->send1);
  free(st->ev_k1); free(st->ev_a1); free(st->ev_k2);
  for (int r = 0; r < st->ring && st->tab_r1; r++) be->dfree(st->tab_r1[r]);
  free(st->tab_r1); st->tab_r1 = NULL;
  be->dfree(st->tab_x2); st->tab_x2 = NULL;
  st->send1 = st->recv1 = NULL; st->ev_k1 = st->ev_a1 = st->ev_k2 = NULL;
  st->ring = 0;
  for (int h = 0; h < st->H2 && st->ev_a2; h++) be->event_destroy(st->ev_a2[h]);
  free(st->ev_a2); st->ev_a2 = NULL;
  for (int i = 0; i < st->ntiles && st->ev_t2; i++) be->event_destroy(st->ev_t2[i]);
  free(st->ev_t2); st->ev_t2 = NULL;
  st->H2 = 0;
  if (st->send2 != st->recv2) be->dfree(st->send2);
  be->dfree(st->recv2);
  st->send2 = st->recv2 = NULL;
}

static int ring_setup(struct _offt_plan *po, hip_state *st) {
  const offt_backend *be = st->be;
  const struct _offt_comm *c = po->comm;
  const size_t min_msg = (size_t)st->opt.min_msg;
  /* phase 1: x-tiles of T1 planes, W1 + 1 ring slots.  Unless the caller fixed T1 the tile is merged upwards until a
   * per-peer message of exchange 1 is at least 4 MiB (the reference's M1/16 was sized for CPU caches and MPI eager limits) */
  st->T = po->params->v[_T1_];
  if (st->T < 1) st->T = 1;
  if (!st->t1_custom) {
    /* at most 8 tiles: the reference's M1/16 gives 54-us launches at 1024^3 on 8 ranks (two rounds of workgroups per
     * launch, 61 % instead of 75 % of the roofline; profiles/r02_rehearse_f64_1024_2x4_first.txt) */
    /* (... and at most 4 since round 3, like the slab schedule: 2.46 -> 2.28 ms of kernels per rank at 1024^3 on a 2 x 4 mesh,
     * profiles/r03_rehearse_touch.txt) */
    const int t8 = (c->M1 + 3) / 4;
    if (st->T < t8) st->T = t8;
    while (st->T < c->M1 && (size_t)st->T * c->M2 * c->M3 * st->esz < min_msg) st->T *= 2;
  }
  if (st->T > c->M1) st->T = c->M1;
  st->ntiles = (c->M1 + st->T - 1) / st->T;
  int W = po->params->v[_W1_];
  if (W < 0) W = 0;
  st->ring = W + 1;
  if (st->ring > st->ntiles) st->ring = st->ntiles;
  /* (block pitches carry a pad of nine 128-B lines: peer blocks a power of two apart alias onto the same HBM channels, see slab_setup) */
  const size_t padE = st->opt.block_pad ? 1152 / st->esz : 0;
  st->blk1 = (size_t)st->T * c->M2 * c->M3 + padE;
  st->send1 = (void **)calloc(st->ring, sizeof(void *));
  st->recv1 = (st->x1 && !st->p2p) ? (void **)calloc(st->ring, sizeof(void *)) : st->send1;
  st->ev_k1 = (void **)calloc(st->ring, sizeof(void *));
  st->ev_a1 = (void **)calloc(st->ring, sizeof(void *));
  st->ev_k2 = (void **)calloc(st->ring, sizeof(void *));
  if (st->p2p) {
    /* direct-store exchange: no send side; the receive ring is ONE allocation (one mapping per peer), slot r at r * slot bytes */
    const size_t slot = ((st->blk1 * c->p2 * st->esz + 255) / 256) * 256;
    st->recv1_base = be->dmalloc(slot * (size_t)st->ring);
    if (!st->recv1_base) return -1;
    for (int r = 0; r < st->ring; r++) st->send1[r] = (char *)st->recv1_base + (size_t)r * slot;
  }
  for (int r = 0; r < st->ring; r++) {
    if (!st->p2p) st->send1[r] = be->dmalloc(st->blk1 * c->p2 * st->esz);
    if (st->x1 && !st->p2p) st->recv1[r] = be->dmalloc(st->blk1 * c->p2 * st->esz);
    st->ev_k1[r] = be->event_create(); st->ev_a1[r] = be->event_create(); st->ev_k2[r] = be->event_create();
    if (!st->send1[r] || !st->recv1[r]) return -1;
  }
  /* phase 2: z-chunks of T2 planes (the reference's z-tiles, offt-compute.c:3682-3862).  Unless the caller fixed T2 there
   * are at most 8 chunks, each a whole number of 8-column panels, and a per-peer message (one x-tile of one chunk) is at
   * least 4 MiB */
  int Tz = po->params->v[_T2_];
  if (Tz < 1) Tz = 1;
  if (!st->t2_custom) {
    const int z8 = (c->M3 + 7) / 8;
    if (Tz < z8) Tz = z8;
    while (Tz < c->M3 && (size_t)st->T * c->M4 * Tz * st->esz < min_msg) Tz *= 2;
    if (Tz < c->M3) Tz = (Tz + 7) / 8 * 8;
  }
  if (Tz > c->M3) Tz = c->M3;
  st->Tz2 = Tz;
  st->H2 = (c->M3 + Tz - 1) / Tz;
  /* contiguous-line layouts of both exchange volumes need even x and z blocks made of whole tiles / chunks */
  st->pencil_yc = c->b1 == 0 && c->b3 == 0 && c->M1 % st->T == 0 && c->M3 % Tz == 0 &&
                  !(getenv("OFFT_PENCIL_ZC_LAYOUT") && atoi(getenv("OFFT_PENCIL_ZC_LAYOUT")));
  st->B2t = (size_t)st->H2 * Tz * c->M4 * st->T + padE;
  st->blk2 = st->pencil_yc ? (size_t)st->ntiles * st->B2t : (size_t)c->M1 * c->M4 * c->M3 + padE;
  st->recv2 = be->dmalloc(st->blk2 * c->p1 * st->esz);
  st->send2 = (st->x2 && !st->p2p) ? be->dmalloc(st->blk2 * c->p1 * st->esz) : st->recv2;
  if (!st->recv2 || !st->send2) return -1;
  st->ev_a2 = (void **)calloc(st->H2, sizeof(void *));
  for (int h = 0; h < st->H2; h++) st->ev_a2[h] = be->event_create();
  st->ev_t2 = (void **)calloc(st->ntiles, sizeof(void *));
  for (int i = 0; i < st->ntiles; i++) st->ev_t2[i] = be->event_create();
  /* self blocks bypass the exchanges (see tab_self): K1 stores its own z-block into recv1[r], K2 its own y-block into recv2 */
  if (st->self_bypass && st->x1 && c->p2 > 1 && !st->p2p) {
    int ok = 1;
    st->tab_r1 = (long long **)calloc(st->ring, sizeof(long long *));
    for (int r = 0; r < st->ring && ok; r++) {
      const long long delta = elem_delta(st, st->recv1[r], st->send1[r], &ok);
      if (ok) st->tab_r1[r] = tab_self(st, c->p2, 1, (long long)st->blk1, po->rank % c->p2, delta);
      if (!st->tab_r1[r]) ok = 0;
    }
    if (!ok) { for (int r = 0; r < st->ring; r++) be->dfree(st->tab_r1[r]); free(st->tab_r1); st->tab_r1 = NULL; }
  }
  if (st->self_bypass && st->x2 && c->p1 > 1 && !st->p2p) {
    int ok = 1;
    const long long delta = elem_delta(st, st->recv2, st->send2, &ok);
    const long long stride = (long long)st->blk2;
    if (ok) st->tab_x2 = tab_self(st, c->p1, 1, stride, po->rank / c->p2, delta);
  }
  return 0;
}

/* ------------------------------------------------------------------------- */
/* everything that depends on the mesh shape p1 x p2: which exchanges exist,   */
/* the schedule (slab / pencil), its buffers, and the row / column             */
/* communicators (offt_comm_malloc, offt-compute.c:78-125).  The static sweep  */
/* rebuilds it per P1 candidate like the tuner does (offt-tuning.c:929).       */
/* ------------------------------------------------------------------------- */
static void mesh_teardown(hip_state *st) {
  p2p_teardown(st); /* (before the buffers it maps) */
  ring_teardown(st);
  slab_teardown(st);
  /* (after a communication failure without ncclCommAbort the communicators are leaked, not destroyed: see comm_fail) */
  if (st->have_comm1) { if (!G.comm_failed) (void)R.CommDestroy(st->comm1); st->have_comm1 = 0; }
  if (st->have_comm2) { if (!G.comm_failed) (void)R.CommDestroy(st->comm2); st->have_comm2 = 0; }
  st->comm1 = st->comm2 = NULL;
  st->uses_rccl = 0;
}

static int p2p_open(struct _offt_plan *po, hip_state *st);
static int mesh_setup(struct _offt_plan *po, hip_state *st) {
  const struct _offt_comm *c = po->comm;
  const int p1 = c->p1, p2 = c->p2;
  const int force = getenv("OFFT_FORCE_A2A") && atoi(getenv("OFFT_FORCE_A2A"));
  st->x1 = (p2 > 1) || force;
  st->x2 = (p1 > 1) || force;
  /* direct-store exchange: asked for, a real world, groups the flag kernels can address, complex input (the r2c z pass
   * has no split side to put a table on when p2 == 1, and is not worth a special case) */
  st->p2p = st->want_p2p && po->p > 1 && p1 <= OFFT_HIPK_MAX_FLAGS && p2 <= OFFT_HIPK_MAX_FLAGS && !force;
  st->slab_zyx = (p1 == 1) && !po->params->v[_S_] && !(po->is_equalxy && c->M1 == c->M4) &&
                 !(getenv("OFFT_NO_SLAB_LAYOUT") && atoi(getenv("OFFT_NO_SLAB_LAYOUT")));
  st->blk2 = (size_t)c->M1 * c->M4 * c->M3;
  /* the local part (buffers) may fail on ONE rank (out of memory); the collective part below -- ncclCommSplit has no
   * time-out -- is therefore run by every rank whatever its local result, which is returned afterwards: the callers
   * (offt_3d_init_ex, the static sweep) then agree on it across the world before anybody enters an exchange */
  int rc_local = 0;
  if (st->slab_zyx) {
    st->x2 = 0; /* p1 == 1: there is no second exchange */
    rc_local = slab_setup(po, st);
  } else rc_local = ring_setup(po, st);
  if (!g_backend && !g_test_transport && (po->p > 1 || force)) {
    /* comm1: ranks sharing rank_x (contiguous), comm2: ranks sharing rank_y
     * (stride p2) -- offt-compute.c:78-125 */
    if (!G.have_comm) { SET_ERR("offt_3d_init: world of %d ranks but no RCCL communicator (offt_hip_set_world)", po->p); return -1; }
    if (G.comm_failed) { SET_ERR("offt_3d_init: the world communicator failed earlier"); return -1; }
    const int rx = po->rank / p2, ry = po->rank % p2;
    /* a group that spans the whole world (slab shapes 1 x p, p x 1) simply uses the world
     * communicator: peer index == world rank.  Real sub-groups are split off; every rank
     * makes the same sequence of collective ncclCommSplit calls. */
    if (st->x1) {
      if (p2 == po->p) st->comm1 = G.world;
      else { NCHECK(R.CommSplit(G.world, rx, ry, &st->comm1, NULL), return -1); st->have_comm1 = 1; }
    }
    if (st->x2) {
      if (p1 == po->p && !(st->x1 && p2 == po->p)) st->comm2 = G.world; /* never share one comm between two exchanges */
      else { NCHECK(R.CommSplit(G.world, ry, rx, &st->comm2, NULL), return -1); st->have_comm2 = 1; }
    }
    st->uses_rccl = st->x1 || st->x2;
  }
  if (st->p2p) {
    /* map the peers' volumes (collective).  Every rank first learns whether every rank has its buffers; afterwards
     * whether every rank could map every peer -- if not, ALL ranks fall back to the staged exchange together. */
    double bad = rc_local ? 1.0 : 0.0;
    if (world_max(po, &bad) || bad > 0.0) return -1;
    bad = p2p_open(po, st) ? 1.0 : 0.0;
    if (world_max(po, &bad)) return -1;
    if (bad > 0.0) {
      if (!po->rank) fprintf(stderr, "offt(hip): direct-store exchange not available here (%s); using the staged RCCL exchange\n", g_err);
      const int want = st->want_p2p;
      mesh_teardown(st);
      st->want_p2p = 0;
      const int rc = mesh_setup(po, st);
      st->want_p2p = want;
      return rc;
    }
    st->uses_rccl = 0; /* the communicators were only needed to hand the mappings round */
  }
  return rc_local;
}

/* map what the packing passes store into and build their per-block tables (see the p2p block above) */
static int p2p_open(struct _offt_plan *po, hip_state *st) {
  const offt_backend *be = st->be;
  const struct _offt_comm *c = po->comm;
  const int p1 = c->p1, p2 = c->p2, rx = po->rank / p2, ry = po->rank % p2;
  st->p2p_status = (unsigned long long *)be->flag_alloc(sizeof(unsigned long long), 1);
  if (!st->p2p_status) return -1;
  st->epoch = st->tiles2 = st->bar1 = st->bar2 = 0;
  if (st->x1) {
    /* group 1 = the row group (p2 members; the whole world on a 1 x p mesh).  Slots: READY and FREE per ring slot (the slab
     * schedule has one "slot", its receive volume), then one barrier slot for the mirrored inverse */
    const int ring = st->slab_zyx ? 1 : st->ring;
    if (p2p_group_open(st, &st->g1, 1, p2, ry, 2 * ring + 1)) return -1;
    st->use1 = (unsigned long long *)calloc((size_t)ring, sizeof(unsigned long long));
    st->peer_r1 = (void **)calloc((size_t)p2, sizeof(void *));
    if (!st->use1 || !st->peer_r1) return -1;
    if (st->slab_zyx) {
      const size_t bytes = (st->slab_yc ? st->sBc * st->sH : st->sblkS * st->sNt) * p2 * st->esz;
      if (be->peer_open(st, 1, p2, ry, st->R1, bytes, st->peer_r1)) return -1;
      st->tab_p1 = st->slab_yc ? tab_peers(st, p2, st->sH, (long long)st->sBc, ry, st->peer_r1, st->R1)
                               : tab_peers(st, p2, 1, (long long)st->sblkS, ry, st->peer_r1, st->R1);
      if (!st->tab_p1) return -1;
    } else {
      const size_t slot = ((st->blk1 * p2 * st->esz + 255) / 256) * 256;
      if (be->peer_open(st, 1, p2, ry, st->recv1_base, slot * (size_t)st->ring, st->peer_r1)) return -1;
      st->tab_pr = (long long **)calloc((size_t)st->ring, sizeof(long long *));
      if (!st->tab_pr) return -1;
      for (int r = 0; r < st->ring; r++) {
        void *pr[p2];
        for (int a = 0; a < p2; a++) pr[a] = (char *)st->peer_r1[a] + (size_t)r * slot;
        st->tab_pr[r] = tab_peers(st, p2, 1, (long long)st->blk1, ry, pr, st->recv1[r]);
        if (!st->tab_pr[r]) return -1;
      }
    }
  }
  if (st->x2) {
    /* group 2 = the column group (p1 members).  Slots: READY (tiles sent so far), FREE (transforms consumed), barrier */
    if (p2p_group_open(st, &st->g2, 2, p1, rx, 3)) return -1;
    st->peer_x2 = (void **)calloc((size_t)p1, sizeof(void *));
    if (!st->peer_x2) return -1;
    if (be->peer_open(st, 2, p1, rx, st->recv2, st->blk2 * p1 * st->esz, st->peer_x2)) return -1;
    const long long stride = (long long)st->blk2;
    st->tab_p2 = tab_peers(st, p1, 1, stride, rx, st->peer_x2, st->recv2);
    if (!st->tab_p2) return -1;
  }
  return 0;
}

static void state_free(hip_state *st) {
  if (!st) return;
  const offt_backend *be = st->be;
  be->dfree(st->work);
  be->dfree(st->work2);
  be->dfree(st->agree_d); free(st->agree_h);
  mesh_teardown(st);
  be->dfree(st->stage);
  be->event_destroy(st->ev0); be->event_destroy(st->ev1);
  for (int i = 0; i < 4; i++) be->event_destroy(st->evp[i]);
  if (st->own_stream) be->stream_destroy(st->s_compute);
  if (st->s_aux) be->stream_destroy(st->s_aux);
  if (st->s_k1b) be->stream_destroy(st->s_k1b);
  if (st->ev_fork) be->event_destroy(st->ev_fork);
  for (int i = 0; i < 4; i++) if (st->ev_aux[i]) be->event_destroy(st->ev_aux[i]);
  if (st->s_comm2 != st->s_comm1) be->stream_destroy(st->s_comm2);
  be->stream_destroy(st->s_comm1);
  free(st);
}

struct _offt_plan *offt_3d_init_ex(int Nx, int Ny, int Nz, void *in, void *out, int is_r2c, int fftw_flag,
                                   int is_oned, int is_a2a, int is_equalxy, int is_notest, int ah_strategy,
                                   int max_loop, int tuning_mode, int is_W0, int extrapolation_window,
                                   struct _offt_params *custom_params, int precision) {
  (void)in;
  double t0 = wall_seconds();
  if (Nx < 1 || Ny < 1 || Nz < 1) { SET_ERR("offt_3d_init: bad grid %d %d %d", Nx, Ny, Nz); return NULL; }
  if (precision != OFFT_HIP_F64 && precision != OFFT_HIP_F32) { SET_ERR("bad precision %d", precision); return NULL; }
  struct _offt_plan *po = (struct _offt_plan *)calloc(1, sizeof *po);
  po->Nx = Nx; po->Ny = Ny; po->Nz = Nz;
  po->p = G.size; po->rank = G.rank; /* offt-compute.c:3315-3316 */
  po->is_r2c = is_r2c; po->fftw_flag = fftw_flag; po->is_oned = is_oned; po->is_a2a = is_a2a;
  po->is_equalxy = is_equalxy; po->is_notest = is_notest; po->ah_strategy = ah_strategy;
  po->max_loop = max_loop; po->tuning_mode = tuning_mode; po->is_W0 = is_W0;
  po->extrapolation_window = extrapolation_window;
  po->params = (struct _offt_params *)calloc(1, sizeof *po->params);
  params_default(po);
  if (!po->rank) print_params(po->params->v); /* offt-compute.c:3416 */
  if (max_loop > 0 && !po->rank)
    printf("offt(hip): max_loop=%d: Active-Harmony search replaced by a static sweep of at most %d points\n", max_loop, max_loop);
  params_custom(po, custom_params);
  {
    int p1 = po->params->v[_P1_];
    if (p1 < 1 || po->p % p1 != 0) {
      SET_ERR("offt_3d_init: P1=%d does not divide p=%d", p1, po->p);
      free(po->params); free(po); return NULL;
    }
    if (po->params->v[_T1_] < 1) po->params->v[_T1_] = 1;
  }
  po->comm = comm_build(po);

  hip_state *st = (hip_state *)calloc(1, sizeof *st);
  po->hip_state = st;
  st->prec = precision;
  st->esz = precision == OFFT_HIP_F64 ? 16 : 8;
  st->be = g_backend ? g_backend : &k_hip_backend;
  st->variant[0] = st->variant[1] = st->variant[2] = -1;
  st->out_scale = 1.0;
  st->self_bypass = !(getenv("OFFT_SELF_BYPASS") && atoi(getenv("OFFT_SELF_BYPASS")) == 0);
  st->k1_streams = getenv("OFFT_K1_STREAMS") ? atoi(getenv("OFFT_K1_STREAMS")) : 1;
  st->opt.zgroup_mib = getenv("OFFT_ZGROUP_MIB") ? atoi(getenv("OFFT_ZGROUP_MIB")) : -1;
  st->opt.zgroup_streams = getenv("OFFT_ZGROUP_STREAMS") ? atoi(getenv("OFFT_ZGROUP_STREAMS")) : 1;
  st->opt.slab_chunk_mib = getenv("OFFT_SLAB_CHUNK_MIB") ? atoi(getenv("OFFT_SLAB_CHUNK_MIB")) : 256;
  st->opt.comm_streams = getenv("OFFT_COMM_STREAMS") ? atoi(getenv("OFFT_COMM_STREAMS")) : 1;
  st->opt.min_msg = getenv("OFFT_MIN_MSG") ? atoll(getenv("OFFT_MIN_MSG")) : 4LL << 20;
  st->opt.f32_pairs = !(getenv("OFFT_F32_PAIRS") && atoi(getenv("OFFT_F32_PAIRS")) == 0);
  /* (off by default: measured, it buys nothing -- the power-of-two block pitches are NOT what holds K1 / K2 back,
   * profiles/r03_rehearse_block_pad_ab.txt; OFFT_BLOCK_PAD=1 turns it on) */
  st->opt.block_pad = getenv("OFFT_BLOCK_PAD") && atoi(getenv("OFFT_BLOCK_PAD")) != 0;
  st->opt.exec_timeout_s = getenv("OFFT_EXEC_TIMEOUT") ? atof(getenv("OFFT_EXEC_TIMEOUT")) : 120.0;
  st->opt.p2p_timeout_s = getenv("OFFT_P2P_TIMEOUT") ? atof(getenv("OFFT_P2P_TIMEOUT")) : 30.0;
  st->want_p2p = getenv("OFFT_EXCHANGE") && !strcmp(getenv("OFFT_EXCHANGE"), "p2p");
  /* scratch planes are offset by an odd number of 128-B lines (9 = 1152 B) so that the x-planes a
   * y-pass panel reads do not alias onto the same HBM channels (sweep: profiles/r01_sweep.txt).  In
   * elements that is 72 for double and 144 for single precision: the 72 single-precision elements of r01
   * put every odd plane 64 B off the lines, and the z pass's 128-B store segments straddled two of them
   * (profiles/r02_wpad_f32.txt: 3-8 % on the z pass). */
  st->wpad = getenv("OFFT_WPAD") ? atoi(getenv("OFFT_WPAD")) : (precision == OFFT_HIP_F64 ? 72 : 144);
  /* ... and its y-lines can be given a pad too, so that the z pass's strided stores (one 128-B segment per line, lines
   * Ny elements apart) do not all fall on the same HBM channels when Ny * 16 B is a large power of two */
  /* (measured: +7 % on the z pass of 256 x 2048 x 2048 f64 -- a 32 KiB pitch --, nothing or a slight loss at 16 KiB,
   * profiles/r02_wrowpad_ab.txt: on by default, as nine 128-B lines, from 32 KiB up) */
  st->wrow = getenv("OFFT_WROWPAD") ? atoi(getenv("OFFT_WROWPAD"))
                                    : (((size_t)Ny * st->esz >= 32768 && (Ny & (Ny - 1)) == 0) ? (int)(1152 / st->esz) : 0);
  const offt_backend *be = st->be;
  double tb0 = wall_seconds();
  if (!g_backend) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) {
      SET_ERR("offt_3d_init: no HIP device visible -- this library has no CPU path");
      goto fail;
    }
  }
  /* a caller-supplied point (run-fft -P <E> -p <cols>, or a line of the sweep's point database)
   * selects the panel kernel whose shape it names; anything else keeps the registry default */
  if (custom_params && custom_params->v[_Px1_] > 0 && custom_params->v[_Py1_] > 0 && !g_backend) {
    const int dims[3] = {Nx, Ny, Nz};
    for (int ax = 0; ax < 3; ax++) {
      const int nv = offt_hipk_variant_count(dims[ax], precision);
      for (int var = 0; var < nv; var++) {
        int e = 0, cols = 0;
        if (offt_hipk_variant_info(dims[ax], precision, var, &e, &cols) == var && e == custom_params->v[_Px1_] &&
            cols == custom_params->v[_Py1_])
          st->variant[ax] = var;
      }
    }
  }
  st->use_pipeline = (po->p > 1) || (getenv("OFFT_FORCE_PIPELINE") && atoi(getenv("OFFT_FORCE_PIPELINE")));
  if (be->prepare(Nx, precision) || be->prepare(Ny, precision) || be->prepare(Nz, precision)) {
    if (!g_backend) SET_ERR("no kernel for this grid: %s", offt_hipk_last_error());
    goto fail;
  }
  st->s_compute = be->stream_create(); st->own_stream = 1;
  st->ev0 = be->event_create(); st->ev1 = be->event_create();
  for (int i = 0; i < 4; i++) st->evp[i] = be->event_create();
  if (!st->s_compute || !st->ev0 || !st->ev1) goto fail;

  if (!st->use_pipeline) {
    /* layouts whose x pass would walk memory plane by plane (x-y-z; y-z-x) rotate through a second scratch volume instead
     * when an x-plane is a whole number of MiB (see execute_single); OFFT_ROTATE=1 / 0 forces either (OFFT_S1_INPLACE=1 is
     * the older name of OFFT_ROTATE=0) */
    const char *renv = getenv("OFFT_ROTATE") ? getenv("OFFT_ROTATE") : (getenv("OFFT_S1_INPLACE") ? (atoi(getenv("OFFT_S1_INPLACE")) ? "0" : "1") : NULL);
    const int rotate = renv ? atoi(renv) != 0 : ((size_t)Ny * (size_t)(is_r2c ? Nz / 2 + 1 : Nz) * st->esz) % ((size_t)1 << 20) == 0;
    if (!po->params->v[_S_]) { /* transposed output layouts need one scratch volume */
      st->work_elems = (size_t)Nx * ((size_t)(Ny + st->wrow) * (is_r2c ? Nz / 2 + 1 : Nz) + (size_t)st->wpad);
      st->work = be->dmalloc(st->work_elems * st->esz);
      if (!st->work) goto fail;
      if (rotate && is_equalxy && po->comm->M1 == po->comm->M4) /* y-z-x: V[z][y][x]; without it the older schedule runs */
        st->work2 = be->dmalloc((size_t)(is_r2c ? Nz / 2 + 1 : Nz) * ((size_t)Nx * Ny + (size_t)st->wpad) * st->esz);
    } else if (rotate) {
      /* x-y-z output: two scratch volumes for the rotating schedule of execute_single, W[y][z][x] and V[z][x][y].  Used when
       * an x-plane is a whole number of MiB -- then the in-place x pass, whose lines step through memory plane by plane,
       * keeps hitting the same HBM channels (1024^3 f64: 7.8 ms = 55 % against 5.9 ms; 512^3: 0.95 against 0.72 ms), while
       * at other pitches (768^3: 9 MiB planes) the in-place passes are as fast and need no scratch
       * (profiles/r03_layouts.txt).  OFFT_ROTATE=0 / 1 forces either.  Without the volumes (allocation failed) the
       * three passes run in place. */
      const size_t nzc = (size_t)(is_r2c ? Nz / 2 + 1 : Nz);
      st->work = be->dmalloc((size_t)Ny * (nzc * Nx + (size_t)st->wpad) * st->esz);
      st->work2 = be->dmalloc(nzc * ((size_t)Nx * Ny + (size_t)st->wpad) * st->esz);
      if (!st->work || !st->work2) { be->dfree(st->work); be->dfree(st->work2); st->work = st->work2 = NULL; }
    }
  } else {
    st->t1_custom = custom_params && custom_params->v[_T1_] >= 0;
    st->t2_custom = custom_params && custom_params->v[_T2_] >= 0;
    /* Exchange 1 (row group) and exchange 2 (column group) run on different communicators.  Two communicators driven
     * concurrently from two streams is the one RCCL usage pattern this library cannot rehearse on a single GPU, so by
     * default both exchanges are issued on ONE comm stream, in the same order on every rank (no cross-communicator wait
     * cycle is possible); OFFT_COMM_STREAMS=2 gives each exchange its own stream (they then overlap on the wire). */
    st->s_comm1 = be->stream_create();
    st->s_comm2 = st->opt.comm_streams >= 2 ? be->stream_create() : st->s_comm1;
    if (!st->s_comm1 || !st->s_comm2) goto fail;
    if (po->p > 1) {
      st->agree_d = (double *)be->dmalloc(sizeof(double) * 2 * (size_t)po->p);
      st->agree_h = (double *)malloc(sizeof(double) * 2 * (size_t)po->p);
      if (!st->agree_d || !st->agree_h) goto fail;
    }
    {
      /* every rank learns whether EVERY rank could build the plan (a max over the world of the local result): a rank
       * that ran out of memory must not leave the others waiting in the first exchange */
      double bad = mesh_setup(po, st) ? 1.0 : 0.0;
      const int mine = bad > 0.0;
      if (world_max(po, &bad) || bad > 0.0) {
        if (!mine) SET_ERR("offt_3d_init: another rank could not build the plan");
        goto fail;
      }
    }
  }
  po->t_init[INIT_BUFFER] = wall_seconds() - tb0;
  if (st->use_pipeline && !st->slab_zyx && custom_params && custom_params->v[_W2_] > 0 && !po->rank) {
    /* the reference keeps W2 tiles of phase 2 in flight (offt-compute.c:3682-3862); here exchange 2 streams behind
     * exchange 1 into a full receive volume and FFTx follows chunk by chunk, so only W2 == 0 (wait for the whole
     * exchange, the reference's blocking mode) differs from any other value */
    printf("offt(hip): W2=%d: only W2 == 0 (blocking) vs W2 != 0 (FFTx overlapped with exchange 2) matters here; the window depth has no effect\n",
           custom_params->v[_W2_]);
  }
  if (max_loop > 0) static_sweep(po, out, custom_params);
  po->t_init[INIT_ALL] = wall_seconds() - t0;
  if (!po->rank) { /* offt-compute.c:3469-3471 */
    const struct _offt_comm *c = po->comm;
    printf("M1 %d M2 %d M3 %d M4 %d m1 %d m2 %d m3 %d m4 %d\n", c->M1, c->M2, c->M3, c->M4, c->m1, c->m2, c->m3, c->m4);
  }
  return po;
fail:
  state_free(st);
  free(po->comm); free(po->params); free(po);
  return NULL;
}

/* ------------------------------------------------------------------------- */
/* static sweep: what replaces ah_tuning() (offt-tuning.c:744-1023).          */
/*                                                                           */
/* offt_3d_init(..., max_loop > 0, ...) in the reference starts an Active-    */
/* Harmony search over the 24-D lattice, timing offt_3d_execute(is_tuning=1)  */
/* per point and logging "perf v0 .. v23" lines to a point database           */
/* (offt-tuning.c:231-277).  Here the search space is the handful of knobs    */
/* that matter on the GPU and it is enumerated, not searched:                 */
/*   single rank : LDS panel shape / radix order = kernel variant, recorded   */
/*                 in the point as Px1 = elements per thread, Py1 = columns   */
/*   p1 x p2     : tile thickness T1 in {T/2, T, 2T} x window W1 in {1, 2, 3} */
/* at most max_loop points, each timed like the tuner does (one warm-up, then */
/* TUNING_REPS executes, device time, max over ranks).  Points are appended   */
/* to the same text format, and a point already in the file is not re-timed   */
/* (is_in_database_point, offt-tuning.c:231-263).                             */
/* ------------------------------------------------------------------------- */
static int db_lookup(const char *path, const int *v, double *perf) {
  FILE *f = fopen(path, "r");
  if (!f) return 0;
  double pr;
  int y[PARAM_COUNT], found = 0;
  while (fscanf(f, "%lf", &pr) == 1) {
    int ok = 1;
    for (int j = 0; j < PARAM_COUNT; j++) {
      if (fscanf(f, "%d", &y[j]) != 1) { ok = 0; break; }
      if (y[j] != v[j]) ok = 0;
    }
    if (ok) { *perf = pr; found = 1; break; }
  }
  fclose(f);
  return found;
}

static void db_append(const char *path, const int *v, double perf) {
  FILE *f = fopen(path, "a");
  if (!f) return;
  fprintf(f, "%.5f ", perf);
  for (int j = 0; j < PARAM_COUNT; j++) fprintf(f, "%d ", v[j]);
  fprintf(f, "\n");
  fclose(f);
}

/* max over all ranks of one double -- collective: EVERY rank of the world calls it, also after a local failure (it
 * then contributes a large value), so that no rank is left alone inside an exchange.  It is an all-to-all of 8 bytes on
 * the world group through the backend's own exchange (which = 0: peer index == world rank), so it works over RCCL,
 * over the shared-GPU test transport and on the CPU test backend alike.  0 on success. */
static int wait_compute_ex(hip_state *st, int watch);
static int world_max(struct _offt_plan *po, double *v) {
  hip_state *st = (hip_state *)po->hip_state;
  const offt_backend *be = st->be;
  const int p = po->p;
  if (p == 1) return 0;
  TEST_NOTE_MESH(po);
  double *d = st->agree_d, *h = st->agree_h;
  if (!d || !h) { *v = 99999999.0; return -1; }
  for (int a = 0; a < p; a++) { h[a] = *v; h[p + a] = 0; }
  const void *sp[p]; void *rp[p]; size_t sb[p], rb[p]; int pr[p];
  for (int a = 0; a < p; a++) { pr[a] = a; sp[a] = d + a; rp[a] = d + p + a; sb[a] = rb[a] = sizeof(double); }
  int rc = 0;
  if (g_backend) memcpy(d, h, sizeof(double) * 2 * p);
  else if (hipMemcpy(d, h, sizeof(double) * 2 * p, hipMemcpyHostToDevice) != hipSuccess) rc = -1;
  if (!rc) rc = be->a2a(st, 0, p, pr, sp, sb, rp, rb, st->s_compute);
  /* bounded and error-polling like the end of an execute: a rank that never joins must not hang the others for ever */
  if (!rc) rc = wait_compute_ex(st, !g_backend && !g_test_transport && G.have_comm);
  if (!rc) {
    if (g_backend) memcpy(h, d, sizeof(double) * 2 * p);
    else if (hipMemcpy(h, d, sizeof(double) * 2 * p, hipMemcpyDeviceToHost) != hipSuccess) rc = -1;
  }
  if (!rc) for (int a = 0; a < p; a++) if (h[p + a] > *v) *v = h[p + a];
  if (rc) *v = 99999999.0;
  return rc;
}

/* time one point like the tuner does (offt-tuning.c:955-975: one warm-up, TUNING_REPS executes), max over ranks.
 * `setup_rc` is this rank's result of building the point: all ranks first agree whether everybody could build it. */
static double sweep_time_point(struct _offt_plan *po, void *buf, int setup_rc) {
  hip_state *st = (hip_state *)po->hip_state;
  double bad = setup_rc ? 1.0 : 0.0;
  if (world_max(po, &bad) || bad > 0.0) return 99999999.0; /* somebody could not build this point: nobody runs it */
  double best = 1e30;
  offt_3d_execute_dir(po, buf, buf, -1); /* warm-up */
  int failed = po->t[ALL] >= 99999999.0;
  for (int r = 0; r < TUNING_REPS && !failed; r++) {
    offt_3d_execute_dir(po, buf, buf, -1);
    if (po->t[ALL] >= 99999999.0) failed = 1;
    /* device time; the CPU test backend has no device clock, there the wall time of the call ranks the points */
    const double tpt = g_backend ? po->t[ALL] : st->last_dev_s;
    if (tpt < best) best = tpt;
  }
  if (failed) best = 99999999.0;
  (void)world_max(po, &best); /* every rank must rank the points identically */
  return best;
}

/* the P1 lattice of the reference: divisors of p within [max(p/Nz', p/Ny, 1), min(Nx, Ny, p)]
 * (params_range_setup, offt-compute.c:3002-3023) */
static int p1_feasible(const struct _offt_plan *po, int d) {
  const int Nzn = po->is_r2c ? po->Nz / 2 + 1 : po->Nz;
  int pu = po->p < (po->Nx < po->Ny ? po->Nx : po->Ny) ? po->p : (po->Nx < po->Ny ? po->Nx : po->Ny);
  int pl = po->p / Nzn > po->p / po->Ny ? po->p / Nzn : po->p / po->Ny;
  if (pl < 1) pl = 1;
  return d >= pl && d <= pu && po->p % d == 0;
}

/* switch the plan to mesh p1 x (p / p1): new decomposition, buffers and row / column communicators.  Collective. */
static int mesh_rebuild(struct _offt_plan *po, int p1) {
  hip_state *st = (hip_state *)po->hip_state;
  mesh_teardown(st);
  po->params->v[_P1_] = p1;
  free(po->comm);
  po->comm = comm_build(po);
  return mesh_setup(po, st);
}

static void sweep_report(struct _offt_plan *po, double perf) {
  if (po->rank) return;
  db_append(po->point_database_file, po->params->v, perf);
  printf("@ SWEEP %.5f ", perf);
  print_params(po->params->v);
}

static void static_sweep(struct _offt_plan *po, void *user_buf, const struct _offt_params *custom) {
  hip_state *st = (hip_state *)po->hip_state;
  int *v = po->params->v;
  const double t0 = wall_seconds();
  const char *envdb = getenv("OFFT_SWEEP_DB");
  if (envdb) snprintf(po->point_database_file, sizeof po->point_database_file, "%s", envdb);
  else snprintf(po->point_database_file, sizeof po->point_database_file, "./tmp-db-%08d", (int)(getpid() % 100000000));
  if (!po->rank) printf("point_database_file %s\n", po->point_database_file);
  int best_v[PARAM_COUNT], points = 0, best_variant = -1;
  double best = 1e30;
  void *buf = NULL;
  int own = 0;
  if (!st->use_pipeline) {
    if (g_backend) return; /* kernel variants exist on the GPU only */
    buf = user_buf;
    if (!buf || !is_device_ptr(buf)) { /* tune on scratch: never stage a host array per point */
      buf = st->be->dmalloc(local_elems(po->comm) * st->esz);
      own = 1;
      if (!buf) return;
      (void)hipMemset(buf, 0, local_elems(po->comm) * st->esz);
    }
    memcpy(best_v, v, sizeof best_v);
    int nv = offt_hipk_variant_count(po->Nx, st->prec);
    const int ny = offt_hipk_variant_count(po->Ny, st->prec), nz = offt_hipk_variant_count(po->Nz, st->prec);
    if (ny > nv) nv = ny;
    if (nz > nv) nv = nz;
    /* point 0 = the registry defaults (a different shape per pass flavour, column pairs in single precision): a uniform
     * variant has to BEAT them to be taken.  It is recorded under the reference's default Px1 / Py1, which name no
     * kernel shape, so feeding that line back (-P/-p) selects the defaults again. */
    if (points < po->max_loop) {
      for (int ax = 0; ax < 3; ax++) st->variant[ax] = -1;
      double perf;
      if (!db_lookup(po->point_database_file, v, &perf)) {
        perf = sweep_time_point(po, buf, 0);
        db_append(po->point_database_file, v, perf);
      }
      printf("@ SWEEP %.5f ", perf); print_params(v);
      best = perf;
      points++;
    }
    for (int var = 0; var < nv && points < po->max_loop; var++, points++) {
      int e = 0, cols = 0;
      if (offt_hipk_variant_info(po->Nx, st->prec, var, &e, &cols) < 0 &&
          offt_hipk_variant_info(po->Ny, st->prec, var, &e, &cols) < 0 &&
          offt_hipk_variant_info(po->Nz, st->prec, var, &e, &cols) < 0)
        continue;
      v[_Px1_] = e; v[_Py1_] = cols;
      for (int ax = 0; ax < 3; ax++) st->variant[ax] = var;
      double perf;
      if (!db_lookup(po->point_database_file, v, &perf)) {
        perf = sweep_time_point(po, buf, 0);
        db_append(po->point_database_file, v, perf);
      }
      printf("@ SWEEP %.5f ", perf); print_params(v);
      /* (2 % hysteresis: box noise must not trade the per-flavour defaults for a uniform variant) */
      if (perf < 0.98 * best) { best = perf; best_variant = var; memcpy(best_v, v, sizeof best_v); }
    }
    for (int ax = 0; ax < 3; ax++) st->variant[ax] = best_variant;
    memcpy(v, best_v, sizeof best_v);
  } else {
    /* ---- stage A: mesh shape.  Every P1 of the reference's lattice -- the divisors of p within
     * [max(p/Nz', p/Ny, 1), min(Nx, Ny, p)], offt-compute.c:3002-3023 -- is a candidate, as for the tuner
     * (offt-tuning.c:879-992 rebuilds communicator, buffers and plans per point like mesh_rebuild); the order is the
     * default, the two slab shapes 1 and p, then the remaining divisors, so that a small max_loop still sees the
     * shapes that differ most ---- */
    enum { MAXC = 64 };
    int cand[MAXC], nc = 0;
    const int p1_now = v[_P1_];
    if (custom && custom->v[_P1_] >= 0) cand[nc++] = p1_now; /* the caller fixed the mesh (-d) */
    else {
      int want[MAXC], nw = 0;
      want[nw++] = p1_now; want[nw++] = 1; want[nw++] = po->p;
      for (int d = 2; d < po->p && nw < MAXC; d++) if (po->p % d == 0) want[nw++] = d;
      for (int i = 0; i < nw && nc < MAXC; i++) {
        int dup = 0;
        for (int j = 0; j < nc; j++) dup |= cand[j] == want[i];
        if (!dup && p1_feasible(po, want[i])) cand[nc++] = want[i];
      }
      if (!nc) cand[nc++] = p1_now;
    }
    /* the tiling every candidate starts from: the caller's / default T1 and T2, not the merged values of the mesh
     * tried before it (each mesh merges them upwards for its own message sizes) */
    const int T1_0 = v[_T1_], T2_0 = v[_T2_];
    int candT[MAXC], candTz[MAXC];
    /* the scratch array must hold the local volume of every mesh tried (run-fft.c:270-288 sizes its array the same way) */
    size_t need = local_elems(po->comm);
    for (int i = 0; i < nc; i++) {
      struct _offt_plan tmp = *po;
      struct _offt_params tp = *po->params;
      tmp.params = &tp; tp.v[_P1_] = cand[i];
      struct _offt_comm *tc = comm_build(&tmp);
      if (local_elems(tc) > need) need = local_elems(tc);
      free(tc);
    }
    buf = st->be->dmalloc(need * st->esz);
    own = 1;
    int rc_buf = buf ? 0 : -1;
    if (buf && !g_backend) (void)hipMemset(buf, 0, need * st->esz);
    int best_p1 = p1_now, best_i = 0;
    for (int i = 0; i < nc && points < po->max_loop; i++, points++) {
      int rc = rc_buf;
      v[_T1_] = T1_0; v[_T2_] = T2_0;
      if (cand[i] != v[_P1_] && mesh_rebuild(po, cand[i])) rc = -1;
      if (st->slab_zyx) { v[_T1_] = st->sT; v[_T2_] = st->sTz; } else { v[_T1_] = st->T; v[_T2_] = st->Tz2; }
      candT[i] = v[_T1_]; candTz[i] = v[_T2_]; /* the tiling this mesh was timed with */
      const double perf = sweep_time_point(po, buf, rc);
      sweep_report(po, perf);
      if (perf < best) { best = perf; best_p1 = cand[i]; best_i = i; }
    }
    /* back to the winner, with exactly the tiling it was timed with (as fixed values: `best` belongs to that point) */
    int rc_mesh = 0;
    st->t1_custom = st->t2_custom = 1;
    v[_T1_] = candT[best_i]; v[_T2_] = candTz[best_i];
    if (best_p1 != v[_P1_]) rc_mesh = mesh_rebuild(po, best_p1);
    else if (st->slab_zyx) { slab_teardown(st); rc_mesh = slab_setup(po, st); }
    else { ring_teardown(st); rc_mesh = ring_setup(po, st); }
    /* ---- stage B: tiling at that mesh.  slab: x-tile thickness T1 (message granularity of the exchange) x z-chunk
     * thickness T2 (granularity of the overlapped FFTy/FFTx work); pencil: T1 x W1, then T2 -- around the winner ---- */
    const int M1 = po->comm->M1, M3 = po->comm->M3;
    if (st->slab_zyx) { v[_T1_] = st->sT; v[_T2_] = st->sTz; } else { v[_T1_] = st->T; v[_T2_] = st->Tz2; }
    memcpy(best_v, v, sizeof best_v);
    const int T0 = v[_T1_], Z0 = v[_T2_];
    const int Tc[3] = {T0, T0 / 2 > 0 ? T0 / 2 : 1, 2 * T0 <= M1 ? 2 * T0 : M1};
    const int Zc[3] = {Z0, Z0 / 2 > 0 ? Z0 / 2 : 1, 2 * Z0 <= M3 ? 2 * Z0 : M3};
    const int Wc[3] = {best_v[_W1_], best_v[_W1_] == 1 ? 2 : 1, 3};
    for (int ti = 0; ti < 3 && points < po->max_loop; ti++) {
      if ((ti > 0 && Tc[ti] == Tc[0]) || (ti > 1 && Tc[ti] == Tc[1])) continue;
      for (int zi = 0; zi < 3 && points < po->max_loop; zi++) {
        if (ti == 0 && zi == 0) continue; /* the stage-A point */
        if (st->slab_zyx) {
          if ((zi > 0 && Zc[zi] == Zc[0]) || (zi > 1 && Zc[zi] == Zc[1])) continue;
          v[_T1_] = Tc[ti]; v[_T2_] = Zc[zi];
          slab_teardown(st);
          const int rc = slab_setup(po, st) | rc_buf | rc_mesh;
          const double perf = sweep_time_point(po, buf, rc);
          points++;
          sweep_report(po, perf);
          if (perf < best) { best = perf; memcpy(best_v, v, sizeof best_v); }
        } else {
          if ((zi > 0 && Wc[zi] == Wc[0]) || (zi > 1 && Wc[zi] == Wc[1])) continue;
          v[_T1_] = Tc[ti]; v[_W1_] = Wc[zi];
          ring_teardown(st);
          const int rc = ring_setup(po, st) | rc_buf | rc_mesh;
          const double perf = sweep_time_point(po, buf, rc);
          points++;
          sweep_report(po, perf);
          if (perf < best) { best = perf; memcpy(best_v, v, sizeof best_v); }
        }
      }
    }
    if (!st->slab_zyx) { /* pencil: phase-2 chunk thickness at the best (T1, W1) */
      for (int zi = 1; zi < 3 && points < po->max_loop; zi++) {
        if (Zc[zi] == Zc[0] || (zi > 1 && Zc[zi] == Zc[1])) continue;
        memcpy(v, best_v, sizeof best_v);
        v[_T2_] = Zc[zi];
        ring_teardown(st);
        const int rc = ring_setup(po, st) | rc_buf | rc_mesh;
        const double perf = sweep_time_point(po, buf, rc);
        points++;
        sweep_report(po, perf);
        if (perf < best) { best = perf; memcpy(best_v, v, sizeof best_v); }
      }
    }
    memcpy(v, best_v, sizeof best_v);
    if (st->slab_zyx) { slab_teardown(st); (void)slab_setup(po, st); }
    else { ring_teardown(st); (void)ring_setup(po, st); }
  }
  po->params->is_converged = 1;
  if (own) st->be->dfree(buf);
  po->t_init[INIT_AH] = wall_seconds() - t0;
}

struct _offt_plan *offt_3d_init(int Nx, int Ny, int Nz, double *in, double *out, int is_r2c, int fftw_flag,
                                int is_oned, int is_a2a, int is_equalxy, int is_notest, int ah_strategy,
                                int max_loop, int tuning_mode, int is_W0, int extrapolation_window,
                                struct _offt_params *custom_params) {
  return offt_3d_init_ex(Nx, Ny, Nz, in, out, is_r2c, fftw_flag, is_oned, is_a2a, is_equalxy, is_notest,
                         ah_strategy, max_loop, tuning_mode, is_W0, extrapolation_window, custom_params,
                         OFFT_HIP_F64);
}

void offt_3d_fin(struct _offt_plan *po) {
  if (!po) return;
  state_free((hip_state *)po->hip_state);
  free(po->comm);
  free(po->params);
  free(po);
}

void offt_hip_set_stream(struct _offt_plan *po, void *stream) {
  hip_state *st = (hip_state *)po->hip_state;
  if (st->own_stream && st->s_compute) st->be->stream_destroy(st->s_compute);
  if (stream) { st->s_compute = stream; st->own_stream = 0; }
  else { st->s_compute = st->be->stream_create(); st->own_stream = 1; }
}
/* exchange of a multi-rank plan: OFFT_HIP_EXCHANGE_STAGED (packed send volume + grouped RCCL send/recv, the default) or
 * OFFT_HIP_EXCHANGE_DIRECT (the packing passes store straight into the peers' receive volumes).  Collective: every rank
 * of the world calls it with the same mode; buffers and mappings are rebuilt.  Returns the mode now in use (direct falls
 * back to staged, on all ranks together, where peer memory cannot be mapped), or -1. */
int offt_hip_set_exchange(struct _offt_plan *po, int mode) {
  hip_state *st = (hip_state *)po->hip_state;
  if (!st->use_pipeline) return OFFT_HIP_EXCHANGE_STAGED;
  if (st->be->stream_sync(st->s_compute)) return -1;
  mesh_teardown(st);
  st->want_p2p = mode == OFFT_HIP_EXCHANGE_DIRECT;
  double bad = mesh_setup(po, st) ? 1.0 : 0.0;
  if (world_max(po, &bad) || bad > 0.0) { SET_ERR("offt_hip_set_exchange: a rank could not rebuild its buffers"); return -1; }
  return st->p2p ? OFFT_HIP_EXCHANGE_DIRECT : OFFT_HIP_EXCHANGE_STAGED;
}
/* plan-level options (offt_hip.h).  Options that shape buffers (slab chunk, comm streams, minimum message, self bypass)
 * rebuild the mesh like offt_hip_set_exchange and are therefore collective. */
int offt_hip_set_option(struct _offt_plan *po, int option, long long value) {
  hip_state *st = (hip_state *)po->hip_state;
  int rebuild = 0;
  inv_cache_drop(st);
  switch (option) {
    case OFFT_HIP_OPT_ZGROUP_MIB: st->opt.zgroup_mib = (int)value; break;
    case OFFT_HIP_OPT_ZGROUP_STREAMS: st->opt.zgroup_streams = (int)value; break;
    case OFFT_HIP_OPT_F32_PAIRS: st->opt.f32_pairs = value != 0; break;
    case OFFT_HIP_OPT_K1_STREAMS: st->k1_streams = (int)value; break;
    case OFFT_HIP_OPT_EXEC_TIMEOUT_S: st->opt.exec_timeout_s = (double)value; break;
    case OFFT_HIP_OPT_P2P_TIMEOUT_S: st->opt.p2p_timeout_s = (double)value; break;
    case OFFT_HIP_OPT_SLAB_CHUNK_MIB: st->opt.slab_chunk_mib = (int)value; rebuild = 1; break;
    case OFFT_HIP_OPT_MIN_MSG: st->opt.min_msg = value; rebuild = 1; break;
    case OFFT_HIP_OPT_SELF_BYPASS: st->self_bypass = value != 0; rebuild = 1; break;
    case OFFT_HIP_OPT_COMM_STREAMS:
      if (st->use_pipeline && (value >= 2) != (st->opt.comm_streams >= 2)) {
        if (st->be->stream_sync(st->s_compute)) return -1;
        if (st->s_comm2 != st->s_comm1) st->be->stream_destroy(st->s_comm2);
        st->s_comm2 = value >= 2 ? st->be->stream_create() : st->s_comm1;
        if (!st->s_comm2) { st->s_comm2 = st->s_comm1; return -1; }
      }
      st->opt.comm_streams = (int)value;
      break;
    default: SET_ERR("offt_hip_set_option: unknown option %d", option); return -1;
  }
  if (rebuild && st->use_pipeline) {
    if (st->be->stream_sync(st->s_compute)) return -1;
    mesh_teardown(st);
    double bad = mesh_setup(po, st) ? 1.0 : 0.0;
    if (world_max(po, &bad) || bad > 0.0) { SET_ERR("offt_hip_set_option: a rank could not rebuild its buffers"); return -1; }
  }
  return 0;
}
long long offt_hip_get_option(const struct _offt_plan *po, int option) {
  const hip_state *st = (const hip_state *)po->hip_state;
  switch (option) {
    case OFFT_HIP_OPT_ZGROUP_MIB: return st->opt.zgroup_mib;
    case OFFT_HIP_OPT_ZGROUP_STREAMS: return st->opt.zgroup_streams;
    case OFFT_HIP_OPT_F32_PAIRS: return st->opt.f32_pairs;
    case OFFT_HIP_OPT_K1_STREAMS: return st->k1_streams;
    case OFFT_HIP_OPT_EXEC_TIMEOUT_S: return (long long)st->opt.exec_timeout_s;
    case OFFT_HIP_OPT_P2P_TIMEOUT_S: return (long long)st->opt.p2p_timeout_s;
    case OFFT_HIP_OPT_SLAB_CHUNK_MIB: return st->opt.slab_chunk_mib;
    case OFFT_HIP_OPT_MIN_MSG: return st->opt.min_msg;
    case OFFT_HIP_OPT_SELF_BYPASS: return st->self_bypass;
    case OFFT_HIP_OPT_COMM_STREAMS: return st->opt.comm_streams;
    default: return -1;
  }
}
int offt_hip_get_exchange(const struct _offt_plan *po) { return ((const hip_state *)po->hip_state)->p2p ? OFFT_HIP_EXCHANGE_DIRECT : OFFT_HIP_EXCHANGE_STAGED; }
void offt_hip_set_output_scale(struct _offt_plan *po, double scale) { ((hip_state *)po->hip_state)->out_scale = scale; }
void offt_hip_set_async(struct _offt_plan *po, int async) { ((hip_state *)po->hip_state)->async = async; }
#if defined(OFFT_TEST_SEAMS) || defined(OFFT_BENCH_DIAGNOSTICS)
/* diagnostics (compiled into the test build and into builds made with -DOFFT_BENCH_DIAGNOSTICS only): leave out the FFT
 * passes (mask 1) or the exchanges (mask 2) of the multi-rank schedules -- results are meaningless while a mask is set */
void offt_hip_set_debug_skip(struct _offt_plan *po, int mask) { inv_cache_drop((hip_state *)po->hip_state); ((hip_state *)po->hip_state)->skip_mask = mask; }
#endif
void offt_hip_set_variant(struct _offt_plan *po, int axis, int variant) {
  if (axis >= 0 && axis < 3) ((hip_state *)po->hip_state)->variant[axis] = variant;
  inv_cache_drop((hip_state *)po->hip_state);
}
double offt_hip_last_device_seconds(const struct _offt_plan *po) { return ((const hip_state *)po->hip_state)->last_dev_s; }
/* 0, or the two timer slots (bit 0 = z, 1 = y, 2 = x) whose launches alternated and were measured as a pair */
int offt_hip_last_passes_paired(const struct _offt_plan *po) {
  const hip_state *st = (const hip_state *)po->hip_state;
  if (!st->yx_fused) return 0;
  return (1 << st->pass_slot[st->yx_fused - 1]) | (1 << st->pass_slot[st->yx_fused]);
}
void offt_hip_last_pass_seconds(const struct _offt_plan *po, double t[3]) {
  const hip_state *st = (const hip_state *)po->hip_state;
  t[0] = st->pass_s[0]; t[1] = st->pass_s[1]; t[2] = st->pass_s[2];
}

/* ------------------------------------------------------------------------- */
/* single-rank direct path: three panel passes, transposes folded into the    */
/* x pass (replaces phase1 + setup_transpose + phase2 at p = 1)               */
/* ------------------------------------------------------------------------- */
static int execute_single(struct _offt_plan *po, void *data, int dir) {
  hip_state *st = (hip_state *)po->hip_state;
  const offt_backend *be = st->be;
  const struct _offt_comm *c = po->comm;
  /* Nzf = length of the z transform; Nz = number of z values kept afterwards
   * (Nz/2+1 for real-to-complex, offt-compute.c:63) -- every extent and stride below uses Nz */
  const int Nx = po->Nx, Ny = po->Ny, Nzf = po->Nz, Nz = po->is_r2c ? po->Nz / 2 + 1 : po->Nz;
  const long long is0 = c->istride[0], is1 = c->istride[1];
  const long long os0 = c->ostride[0], os1 = c->ostride[1], os2 = c->ostride[2];
  void *s = st->s_compute;
  void *W = st->work;
  if (po->is_r2c && dir > 0) { SET_ERR("complex-to-real inverse is not built (the reference has no inverse at all)"); return -1; }
  offt_pass_desc d[3];
  const void *src[3];
  void *dst[3];
  int slot[3]; /* which timer slot (0 = z, 1 = y, 2 = x) each launch feeds */
  const int S = po->params->v[_S_] != 0;
  const int zyx = !S && !(po->is_equalxy && c->M1 == c->M4);

  int s1_rot = 0, yzx_rot = 0;
  if (S && st->work && st->work2) {
    /* x-y-z output == input layout, x outermost on both sides: an FFT along x that touched this layout directly would walk
     * memory at a plane-sized stride (16 MiB at 1024^3: 56 % of the roofline, profiles/r02_layouts_zgroup.txt).  Instead
     * every pass reads whole contiguous lines and rotates on its stores (128-B column segments at a pitch of one line),
     * like the z-y-x schedule, with the x pass in the MIDDLE, between two scratch volumes:
     *   P1  in[x][y][z] --FFTz--> W[y][z][x]    columns = 8 x-planes: eight lines gathered from eight planes
     *   P2  W[y][z][x]  --FFTx--> V[z][x][y]    columns = 8 y
     *   P3  V[z][x][y]  --FFTy--> out[x][y][z]  columns = 8 z
     * (the inverse runs y, x, z with loads and stores swapped).  Three contig-in / strided-out passes. */
    const long long wy = (long long)Nz * Nx + st->wpad;  /* W: y-plane pitch */
    const long long vz = (long long)Nx * Ny + st->wpad;  /* V: z-plane pitch */
    void *V = st->work2;
    s1_rot = 1;
    desc_init(&d[0], st, Nzf, dir, 2);
    d[0].real_input = po->is_r2c;
    desc_init(&d[1], st, Nx, dir, 0);
    desc_init(&d[2], st, Ny, dir, 1);
    d[0].ncols = Nx; d[0].nb1 = Ny;
    d[1].ncols = Ny; d[1].nb1 = Nz;
    d[2].ncols = Nz; d[2].nb1 = Nx;
    if (dir < 0) {
      d[0].in_axis_stride = 1; d[0].in_col_stride = is0; d[0].in_b1_stride = is1; d[0].in_contig = 1;
      d[0].out_axis_stride = Nx; d[0].out_col_stride = 1; d[0].out_b1_stride = wy; d[0].out_contig = 0;
      d[1].in_axis_stride = 1; d[1].in_col_stride = wy; d[1].in_b1_stride = Nx; d[1].in_contig = 1;
      d[1].out_axis_stride = Ny; d[1].out_col_stride = 1; d[1].out_b1_stride = vz; d[1].out_contig = 0;
      d[2].in_axis_stride = 1; d[2].in_col_stride = vz; d[2].in_b1_stride = Ny; d[2].in_contig = 1;
      d[2].out_axis_stride = os1; d[2].out_col_stride = os2; d[2].out_b1_stride = os0; d[2].out_contig = 0;
      src[0] = data; dst[0] = W; src[1] = W; dst[1] = V; src[2] = V; dst[2] = data;
      slot[0] = 0; slot[1] = 2; slot[2] = 1;
    } else {
      offt_pass_desc t;
      d[2].in_axis_stride = os1; d[2].in_col_stride = os2; d[2].in_b1_stride = os0; d[2].in_contig = 0;
      d[2].out_axis_stride = 1; d[2].out_col_stride = vz; d[2].out_b1_stride = Ny; d[2].out_contig = 1;
      d[1].in_axis_stride = Ny; d[1].in_col_stride = 1; d[1].in_b1_stride = vz; d[1].in_contig = 0;
      d[1].out_axis_stride = 1; d[1].out_col_stride = wy; d[1].out_b1_stride = Nx; d[1].out_contig = 1;
      d[0].in_axis_stride = Nx; d[0].in_col_stride = 1; d[0].in_b1_stride = wy; d[0].in_contig = 0;
      d[0].out_axis_stride = 1; d[0].out_col_stride = is0; d[0].out_b1_stride = is1; d[0].out_contig = 1;
      t = d[0]; d[0] = d[2]; d[2] = t; /* launch order y, x, z */
      src[0] = data; dst[0] = V; src[1] = V; dst[1] = W; src[2] = W; dst[2] = data;
      slot[0] = 1; slot[1] = 2; slot[2] = 0;
    }
  } else if (S) {
    /* x-y-z output == input layout without scratch: three in-place passes, strided along y and x */
    desc_init(&d[0], st, Nzf, dir, 2);
    d[0].real_input = po->is_r2c;
    d[0].ncols = Ny; d[0].nb1 = Nx;
    d[0].in_axis_stride = d[0].out_axis_stride = 1;
    d[0].in_col_stride = d[0].out_col_stride = is1;
    d[0].in_b1_stride = d[0].out_b1_stride = is0;
    d[0].in_contig = d[0].out_contig = 1;
    desc_init(&d[1], st, Ny, dir, 1);
    d[1].ncols = Nz; d[1].nb1 = Nx;
    d[1].in_axis_stride = d[1].out_axis_stride = is1;
    d[1].in_col_stride = d[1].out_col_stride = 1;
    d[1].in_b1_stride = d[1].out_b1_stride = is0;
    desc_init(&d[2], st, Nx, dir, 0);
    if (is0 == (long long)Ny * is1 && is1 == Nz) { d[2].ncols = Ny * Nz; d[2].nb1 = 1; }
    else { d[2].ncols = Nz; d[2].nb1 = Ny; }
    d[2].in_axis_stride = d[2].out_axis_stride = is0;
    d[2].in_col_stride = d[2].out_col_stride = 1;
    d[2].in_b1_stride = d[2].out_b1_stride = is1;
    for (int i = 0; i < 3; i++) { src[i] = data; dst[i] = data; slot[i] = i; }
  } else if (zyx) {
    /* default z-y-x output.  Every pass READS whole contiguous lines and the
     * axis rotation rides on the stores (128-B column segments, row pitch of
     * one line), so no pass walks memory at a plane-sized stride:
     *   P1  [x][y][z] --FFTz--> W[x][z][y]     (local transpose folded in)
     *   P2  W[x][z][y] --FFTy--> out[z][y][x]  (columns = 8 x-planes)
     *   P3  out[z][y][x] --FFTx--> in place
     * this replaces FFTz + pack/unpack + setup_transpose's xzy->zxy permutation
     * + FFTy + FFTx of the reference (offt-compute.c:625-634, 4019-4036). */
    const long long wy = Ny + st->wrow;                 /* W line pitch */
    const long long wx = (long long)Nz * wy + st->wpad; /* W plane (+ optional pad, elements) */
    desc_init(&d[0], st, Nzf, dir, 2);
    d[0].real_input = po->is_r2c;
    desc_init(&d[1], st, Ny, dir, 1);
    desc_init(&d[2], st, Nx, dir, 0);
    if (dir < 0) {
      d[0].ncols = Ny; d[0].nb1 = Nx;
      d[0].in_axis_stride = 1; d[0].in_col_stride = is1; d[0].in_b1_stride = is0; d[0].in_contig = 1;
      d[0].out_axis_stride = wy; d[0].out_col_stride = 1; d[0].out_b1_stride = wx; d[0].out_contig = 0;
      d[1].ncols = Nx; d[1].nb1 = Nz;
      d[1].in_axis_stride = 1; d[1].in_col_stride = wx; d[1].in_b1_stride = wy; d[1].in_contig = 1;
      d[1].out_axis_stride = os1; d[1].out_col_stride = os0; d[1].out_b1_stride = os2; d[1].out_contig = 0;
      d[2].ncols = Ny; d[2].nb1 = Nz;
      d[2].in_axis_stride = d[2].out_axis_stride = os0;
      d[2].in_col_stride = d[2].out_col_stride = os1;
      d[2].in_b1_stride = d[2].out_b1_stride = os2;
      d[2].in_contig = d[2].out_contig = 1;
      src[0] = data; dst[0] = W; src[1] = W; dst[1] = data; src[2] = data; dst[2] = data;
      slot[0] = 0; slot[1] = 1; slot[2] = 2;
    } else {
      /* inverse: the same three steps backwards (x in place, y -> W, z -> input layout) */
      offt_pass_desc t;
      d[2].ncols = Ny; d[2].nb1 = Nz;
      d[2].in_axis_stride = d[2].out_axis_stride = os0;
      d[2].in_col_stride = d[2].out_col_stride = os1;
      d[2].in_b1_stride = d[2].out_b1_stride = os2;
      d[2].in_contig = d[2].out_contig = 1;
      d[1].ncols = Nx; d[1].nb1 = Nz;
      d[1].in_axis_stride = os1; d[1].in_col_stride = os0; d[1].in_b1_stride = os2; d[1].in_contig = 0;
      d[1].out_axis_stride = 1; d[1].out_col_stride = wx; d[1].out_b1_stride = wy; d[1].out_contig = 1;
      d[0].ncols = Ny; d[0].nb1 = Nx;
      d[0].in_axis_stride = wy; d[0].in_col_stride = 1; d[0].in_b1_stride = wx; d[0].in_contig = 0;
      d[0].out_axis_stride = 1; d[0].out_col_stride = is1; d[0].out_b1_stride = is0; d[0].out_contig = 1;
      t = d[0]; d[0] = d[2]; d[2] = t; /* launch order x, y, z */
      src[0] = data; dst[0] = data; src[1] = data; dst[1] = W; src[2] = W; dst[2] = data;
      slot[0] = 2; slot[1] = 1; slot[2] = 0;
    }
  } else if (st->work2) {
    /* y-z-x output (is_equalxy) as a variation of the z-y-x schedule: the same two rotations, the second one into a second
     * scratch volume V[z][y][x], and the x pass -- whole contiguous lines on both sides -- puts every line where the
     * caller's layout wants it (line (y, z) at y * ostride[1] + z * ostride[2]):
     *   P1  in[x][y][z] --FFTz--> W[x][z][y]     P2  W --FFTy--> V[z][y][x]     P3  V --FFTx--> out[y][z][x]
     * (the older schedule below reads its x lines out of W[x][y][z] at a plane-sized stride: 62 % on that pass) */
    const long long wy = Ny + st->wrow, wx = (long long)Nz * wy + st->wpad;
    const long long vz = (long long)Nx * Ny + st->wpad;
    void *V = st->work2;
    yzx_rot = 1;
    desc_init(&d[0], st, Nzf, dir, 2);
    d[0].real_input = po->is_r2c;
    desc_init(&d[1], st, Ny, dir, 1);
    desc_init(&d[2], st, Nx, dir, 0);
    d[0].ncols = Ny; d[0].nb1 = Nx;
    d[1].ncols = Nx; d[1].nb1 = Nz;
    d[2].ncols = Ny; d[2].nb1 = Nz;
    if (dir < 0) {
      d[0].in_axis_stride = 1; d[0].in_col_stride = is1; d[0].in_b1_stride = is0; d[0].in_contig = 1;
      d[0].out_axis_stride = wy; d[0].out_col_stride = 1; d[0].out_b1_stride = wx; d[0].out_contig = 0;
      d[1].in_axis_stride = 1; d[1].in_col_stride = wx; d[1].in_b1_stride = wy; d[1].in_contig = 1;
      d[1].out_axis_stride = Nx; d[1].out_col_stride = 1; d[1].out_b1_stride = vz; d[1].out_contig = 0;
      d[2].in_axis_stride = 1; d[2].in_col_stride = Nx; d[2].in_b1_stride = vz; d[2].in_contig = 1;
      d[2].out_axis_stride = os0; d[2].out_col_stride = os1; d[2].out_b1_stride = os2; d[2].out_contig = 1;
      src[0] = data; dst[0] = W; src[1] = W; dst[1] = V; src[2] = V; dst[2] = data;
      slot[0] = 0; slot[1] = 1; slot[2] = 2;
    } else {
      offt_pass_desc t;
      d[2].in_axis_stride = os0; d[2].in_col_stride = os1; d[2].in_b1_stride = os2; d[2].in_contig = 1;
      d[2].out_axis_stride = 1; d[2].out_col_stride = Nx; d[2].out_b1_stride = vz; d[2].out_contig = 1;
      d[1].in_axis_stride = Nx; d[1].in_col_stride = 1; d[1].in_b1_stride = vz; d[1].in_contig = 0;
      d[1].out_axis_stride = 1; d[1].out_col_stride = wx; d[1].out_b1_stride = wy; d[1].out_contig = 1;
      d[0].in_axis_stride = wy; d[0].in_col_stride = 1; d[0].in_b1_stride = wx; d[0].in_contig = 0;
      d[0].out_axis_stride = 1; d[0].out_col_stride = is1; d[0].out_b1_stride = is0; d[0].out_contig = 1;
      t = d[0]; d[0] = d[2]; d[2] = t; /* launch order x, y, z */
      src[0] = data; dst[0] = V; src[1] = V; dst[1] = W; src[2] = W; dst[2] = data;
      slot[0] = 2; slot[1] = 1; slot[2] = 0;
    }
  } else {
    /* y-z-x output (is_equalxy), one scratch volume: z and y passes in the natural layout inside W,
     * the x pass transposes into the caller's layout */
    const long long w1 = Nz, w0 = (long long)Ny * Nz;
    desc_init(&d[0], st, Nzf, dir, 2);
    d[0].real_input = po->is_r2c;
    d[0].ncols = Ny; d[0].nb1 = Nx; d[0].in_axis_stride = d[0].out_axis_stride = 1;
    d[0].in_contig = d[0].out_contig = 1;
    desc_init(&d[1], st, Ny, dir, 1);
    d[1].ncols = Nz; d[1].nb1 = Nx;
    d[1].in_axis_stride = d[1].out_axis_stride = w1;
    d[1].in_col_stride = d[1].out_col_stride = 1;
    d[1].in_b1_stride = d[1].out_b1_stride = w0;
    desc_init(&d[2], st, Nx, dir, 0);
    d[2].ncols = Nz; d[2].nb1 = Ny;
    if (dir < 0) {
      d[0].in_col_stride = is1; d[0].in_b1_stride = is0; d[0].out_col_stride = w1; d[0].out_b1_stride = w0;
      d[2].in_axis_stride = w0; d[2].in_col_stride = 1; d[2].in_b1_stride = w1;
      d[2].out_axis_stride = os0; d[2].out_col_stride = os2; d[2].out_b1_stride = os1; d[2].out_contig = (os0 == 1);
      src[0] = data; dst[0] = W; src[1] = W; dst[1] = W; src[2] = W; dst[2] = data;
      slot[0] = 0; slot[1] = 1; slot[2] = 2;
    } else {
      offt_pass_desc t;
      d[2].in_axis_stride = os0; d[2].in_col_stride = os2; d[2].in_b1_stride = os1; d[2].in_contig = (os0 == 1);
      d[2].out_axis_stride = w0; d[2].out_col_stride = 1; d[2].out_b1_stride = w1;
      d[0].in_col_stride = w1; d[0].in_b1_stride = w0; d[0].out_col_stride = is1; d[0].out_b1_stride = is0;
      t = d[0]; d[0] = d[2]; d[2] = t;
      src[0] = data; dst[0] = W; src[1] = W; dst[1] = W; src[2] = W; dst[2] = data;
      slot[0] = 2; slot[1] = 1; slot[2] = 0;
    }
  }
  d[2].scale = st->out_scale; /* last launch */
  for (int i = 0; i < 3; i++) st->pass_slot[i] = slot[i];
  /* Forward z-y-x: the y pass writes out[z][y][x] plane by plane and the x pass transforms those planes in place, so
   * the two ALTERNATE over groups of z-planes small enough for the 256 MiB memory-side Infinity Cache: y(group) stores
   * with the default cache policy (out_keep), x(group) finds its input there instead of in HBM -- one of the six
   * read/write sweeps of the transform is served by the cache.  1024^3 f64: y + x 11.5 -> 10.7 ms, the transform
   * 17.3 -> 16.5 ms (tools/mall_probe.py, profiles/r02_mall_probe.txt, r02_zgroup*.txt).  OFFT_ZGROUP_MIB sets
   * the group size (0: off). */
  st->yx_fused = 0;
  /* (lines of up to 1024 points: the 2048-point kernels fill a CU with one workgroup, a group launch of theirs ends in
   * a long tail, and 2048^3 f32 came out 2 % slower -- profiles/r02_zgroup2.txt) */
  /* (... and only where the y pass has a kernel with cache-keeping stores: without them the groups are just more
   * launches -- the mixed-radix lengths lost 3-6 %, profiles/r02_size_table_final2.txt) */
  /* The inverse runs the same three steps backwards (x in place, y, z): there the x launch of a group is the producer
   * and the y launch the consumer.  The other two layouts have such a pair as well: their z and y passes both work
   * inside x-planes (x-y-z: both in place; y-z-x forward: z into the scratch volume, y in place there), so z(group of
   * x-planes) keeps and y(group) re-reads. */
  int ia = -1, cnt = 0;       /* producer launch (the consumer is the next one); planes the two share */
  double plane_elems = 0.0;
  int len_a = 0, len_b = 0;   /* their line lengths */
  /* (the rotating y-z-x schedule has the same y / x pair over z-planes as z-y-x) */
  if (zyx || yzx_rot) { ia = dir < 0 ? 1 : 0; cnt = Nz; plane_elems = (double)Nx * Ny; len_a = dir < 0 ? Ny : Nx; len_b = dir < 0 ? Nx : Ny; }
  /* (the rotating x-y-z schedule runs its three launches plainly: P1 and P2 do share y-planes of W, but alternating them
   * over groups of planes -- the consumer sliced along its columns -- came out SLOWER, 17.9 against 17.4 ms at 1024^3 f64,
   * profiles/r03_layouts.txt) */
  else if (s1_rot) ia = -1;
  else if (S || dir < 0) { ia = 0; cnt = Nx; plane_elems = (double)Ny * Nz; len_a = Nzf; len_b = Ny; }
  if (ia >= 0 && d[ia].nb1 == cnt && d[ia + 1].nb1 == cnt && !d[ia].real_input &&
      (st->opt.zgroup_mib >= 0 || (len_a <= 1024 && len_b <= 1024 && (g_backend || offt_hipk_keeps_output(&d[ia]))))) {
    const int ib = ia + 1;
    const int group_mib = st->opt.zgroup_mib >= 0 ? st->opt.zgroup_mib : 256;
    /* OFFT_ZGROUP_STREAMS=2: consumer launches on a second stream (then 128 MiB groups do as well as 256 MiB on one stream) */
    const int two_streams = st->opt.zgroup_streams >= 2;
    const double plane_mib = plane_elems * (double)st->esz / (1024.0 * 1024.0);
    int ng = group_mib > 0 ? (int)((double)group_mib / plane_mib) : 0;
    if (ng >= 1) {
      if (ng > cnt) ng = cnt;
      /* the consumer of a group runs on a second stream behind its producer, so that the next group's producer fills the
       * CUs its last workgroups leave idle: groups can be small (good for the cache) without paying a launch tail each */
      int aux = two_streams && ng < cnt;
      if (aux && !st->s_aux) {
        st->s_aux = be->stream_create();
        for (int i = 0; i < 4; i++) st->ev_aux[i] = be->event_create();
        if (!st->s_aux || !st->ev_aux[0] || !st->ev_aux[1] || !st->ev_aux[2] || !st->ev_aux[3]) aux = 0;
      }
      for (int i = 0; i < ia; i++) { /* the launch ahead of the pair */
        if (st->timed) be->event_record(st->evp[i], s);
        if (be->pass(&d[i], src[i], dst[i], s)) return -1;
      }
      if (st->timed) be->event_record(st->evp[ia], s);
      int k = 0;
      for (int z0 = 0; z0 < cnt; z0 += ng, k++) {
        const int g = cnt - z0 < ng ? cnt - z0 : ng;
        offt_pass_desc da = d[ia], db = d[ib];
        da.nb1 = g; db.nb1 = g;
        da.out_keep = 1;
        const char *sa = (const char *)src[ia] + (size_t)z0 * (size_t)da.in_b1_stride * st->esz;
        char *oa = (char *)dst[ia] + (size_t)z0 * (size_t)da.out_b1_stride * st->esz;
        const char *sb = (const char *)src[ib] + (size_t)z0 * (size_t)db.in_b1_stride * st->esz;
        char *ob = (char *)dst[ib] + (size_t)z0 * (size_t)db.out_b1_stride * st->esz;
        if (be->pass(&da, sa, oa, s)) return -1;
        if (aux) {
          be->event_record(st->ev_aux[k & 3], s);
          be->stream_wait(st->s_aux, st->ev_aux[k & 3]);
        }
        if (be->pass(&db, sb, ob, aux ? st->s_aux : s)) return -1;
      }
      if (aux) { /* the compute stream goes on behind the last consumer launch */
        be->event_record(st->ev_aux[k & 3], st->s_aux);
        be->stream_wait(s, st->ev_aux[k & 3]);
      }
      /* (the events around the pair span both launches; the boundary inside it is recorded at its end: the reader
       *  splits the pair's time evenly) */
      if (st->timed) { be->event_record(st->evp[ib], s); be->event_record(st->evp[ib + 1], s); }
      for (int i = ib + 1; i < 3; i++) { /* the launch behind the pair */
        if (be->pass(&d[i], src[i], dst[i], s)) return -1;
        if (st->timed) be->event_record(st->evp[i + 1], s);
      }
      st->yx_fused = ia + 1;
      return 0;
    }
  }
  /* per-pass timing events only when somebody will read them: in asynchronous mode the call returns before
   * the GPU has finished, and each record costs a barrier packet (~2 us) between launches -- 20 % of a
   * 128^3 transform */
  for (int i = 0; i < 3; i++) {
    if (st->timed) be->event_record(st->evp[i], s);
    if (be->pass(&d[i], src[i], dst[i], s)) return -1;
  }
  if (st->timed) be->event_record(st->evp[3], s);
  return 0;
}

/* ------------------------------------------------------------------------- */
/* step recorder.  The multi-rank schedules issue their passes and exchanges   */
/* through run_pass()/run_a2a().  Normally these launch at once; for the       */
/* inverse transform (an extension: the reference is forward-only) the forward */
/* schedule is recorded instead and then replayed BACKWARDS with every step    */
/* mirrored: a pass swaps its input and output descriptors and flips the sign  */
/* of the exponent, an exchange swaps its send and receive sides.  The forward */
/* schedule is hazard-free, so its exact reverse is too; the replay runs in    */
/* order on the compute stream (no overlap tuning for this extension).         */
/* ------------------------------------------------------------------------- */
typedef struct step {
  int kind;               /* 0 = pass, 1 = exchange, 2 = a flag operation of the direct-store exchange on group `which` */
  int first;              /* pass reads the caller's input layout (K1): it writes the inverse's result */
  offt_pass_desc d; const void *src; void *dst;
  int which, cnt; int *peer; const void **sp; size_t *sb; void **rp; size_t *rb;
  int tag;                /* slab schedule: the z-chunk the step belongs to (-1: the K1 phase) */
} step;
typedef struct step_list { step *v; int n, cap; } step_list;

static step *rec_new(step_list *l) {
  if (l->n == l->cap) { l->cap = l->cap ? 2 * l->cap : 64; l->v = (step *)realloc(l->v, (size_t)l->cap * sizeof(step)); }
  step *e = &l->v[l->n++];
  memset(e, 0, sizeof *e);
  return e;
}
static void rec_free(step_list *l) {
  for (int i = 0; i < l->n; i++) { free(l->v[i].peer); free(l->v[i].sp); free(l->v[i].sb); free(l->v[i].rp); free(l->v[i].rb); }
  free(l->v);
  l->v = NULL; l->n = l->cap = 0;
}

static int run_pass(hip_state *st, const offt_pass_desc *d, const void *src, void *dst, void *stream, int first) {
  if (st->skip_mask & 1) return 0; /* diagnostics: exchange-only timing */
  if (!st->rec) return st->be->pass(d, src, dst, stream);
  step *e = rec_new(st->rec);
  e->kind = 0; e->first = first; e->d = *d; e->src = src; e->dst = dst; e->tag = st->rec_tag;
  return 0;
}
static int run_a2a(hip_state *st, int which, int cnt, const int *peer, const void *const *sp, const size_t *sb,
                   void *const *rp, const size_t *rb, void *stream) {
  if (st->skip_mask & 2) return 0; /* diagnostics: compute-only timing */
  if (!st->rec) return st->be->a2a(st, which, cnt, peer, sp, sb, rp, rb, stream);
  step *e = rec_new(st->rec);
  e->kind = 1; e->which = which; e->cnt = cnt; e->tag = st->rec_tag;
  e->peer = (int *)malloc(sizeof(int) * cnt); e->sp = (const void **)malloc(sizeof(void *) * cnt);
  e->rp = (void **)malloc(sizeof(void *) * cnt); e->sb = (size_t *)malloc(sizeof(size_t) * cnt); e->rb = (size_t *)malloc(sizeof(size_t) * cnt);
  memcpy(e->peer, peer, sizeof(int) * cnt); memcpy(e->sp, sp, sizeof(void *) * cnt); memcpy(e->rp, rp, sizeof(void *) * cnt);
  memcpy(e->sb, sb, sizeof(size_t) * cnt); memcpy(e->rb, rb, sizeof(size_t) * cnt);
  return 0;
}

/* The event edges between compute and comm streams that the multi-rank schedules rest on, numbered.  In the test build
 * OFFT_TEST_DROP_EDGE=<id> leaves one class of them out: the negative control of the asynchronous-transport tests
 * (tools/async_negative_control.sh) -- with the edge gone the result must come out wrong, which shows both that the
 * edge is needed and that the tests would notice its absence.  The product build has no such switch. */
enum { EDGE_SLAB_K2_AFTER_EXCHANGE = 1, EDGE_SLAB_EXCHANGE_AFTER_K1 = 2, EDGE_PENCIL_EX1_AFTER_K1 = 3, EDGE_PENCIL_K2_AFTER_EX1 = 4,
       EDGE_PENCIL_EX2_AFTER_K2 = 5, EDGE_PENCIL_K3_AFTER_EX2 = 6, EDGE_INV_EXCHANGE_AFTER_PASSES = 7, EDGE_INV_K1_AFTER_EXCHANGES = 8,
       /* ... and the two kinds of flag wait of the direct-store exchange */
       EDGE_P2P_WAIT_READY = 9, EDGE_P2P_WAIT_FREE = 10,
       /* ... and the mirrored pencil schedule's */
       EDGE_INV_PENCIL_EX2_AFTER_K3 = 11, EDGE_INV_PENCIL_K2_AFTER_EX2 = 12, EDGE_INV_PENCIL_EX1_AFTER_K2 = 13, EDGE_INV_PENCIL_K1_AFTER_EX1 = 14 };
#define PTAG(stage, idx) (((stage) << 16) | (idx)) /* pencil schedule: tag of a recorded step = (stage 1 K1/exchange 1 of tile idx, 2 K2/exchange 2
                                                      of tile idx, 3 exchange 2 of the last tile's chunk idx, 4 K3 of chunk idx) */
#ifdef OFFT_TEST_SEAMS
static int edge_dropped(int id) {
  static int drop = -1;
  if (drop < 0) drop = getenv("OFFT_TEST_DROP_EDGE") ? atoi(getenv("OFFT_TEST_DROP_EDGE")) : 0;
  return drop == id;
}
#define SCHED_WAIT(id, stream, ev) do { if (!edge_dropped(id)) be->stream_wait((stream), (ev)); } while (0)
#else
#define SCHED_WAIT(id, stream, ev) be->stream_wait((stream), (ev))
#define edge_dropped(id) 0
#endif

/* flag operations of the direct-store exchange.  Recorded (multi-rank inverse) they become plain sync points: the
 * mirrored schedule puts a barrier of the group wherever the forward schedule signals or waits -- every dependency of the
 * forward schedule crosses at least one of these points, so its mirror image is ordered by the barriers. */
static int run_signal(hip_state *st, p2p_group *g, int slot, unsigned long long value, void *stream) {
  if (st->skip_mask & 2) return 0;
  if (st->rec) { step *e = rec_new(st->rec); e->kind = 2; e->which = g->which; return 0; }
  return p2p_signal(st, g, slot, value, stream);
}
static int run_wait(hip_state *st, p2p_group *g, int slot, unsigned long long value, void *stream, int edge) {
  (void)edge;
  if (st->skip_mask & 2) return 0;
  if (!st->rec && edge_dropped(edge)) return 0; /* (test build: negative control, every rank drops the same waits) */
  if (st->rec) { step *e = rec_new(st->rec); e->kind = 2; e->which = g->which; return 0; }
  return p2p_wait(st, g, slot, value, stream);
}
static int p2p_barrier(hip_state *st, int which, void *stream) {
  p2p_group *g = which == 2 ? &st->g2 : &st->g1;
  unsigned long long *bar = which == 2 ? &st->bar2 : &st->bar1;
  const unsigned long long v = ++*bar;
  if (p2p_signal(st, g, g->nslots - 1, v, stream)) return -1;
  return p2p_wait(st, g, g->nslots - 1, v, stream);
}

static int execute_slab(struct _offt_plan *po, void *data);
static int execute_pipeline(struct _offt_plan *po, void *data, int dir);

/* one recorded step, mirrored, on `sx` */
static int mirror_step(hip_state *st, const step_list *L, int i, void *sx) {
  const offt_backend *be = st->be;
  const step *e = &L->v[i];
  if (e->kind == 0) {
    offt_pass_desc d = e->d, f = e->d;
    d.direction = +1;
    d.in_axis_stride = f.out_axis_stride; d.in_col_stride = f.out_col_stride; d.in_b1_stride = f.out_b1_stride; d.in_b2_stride = f.out_b2_stride;
    d.out_axis_stride = f.in_axis_stride; d.out_col_stride = f.in_col_stride; d.out_b1_stride = f.in_b1_stride; d.out_b2_stride = f.in_b2_stride;
    d.in_split = f.out_split; d.in_split_nfloor = f.out_split_nfloor; d.in_block_stride = f.out_block_stride;
    d.out_split = f.in_split; d.out_split_nfloor = f.in_split_nfloor; d.out_block_stride = f.in_block_stride;
    d.in_block_tab = f.out_block_tab; d.out_block_tab = f.in_block_tab;
    d.in_contig = f.out_contig; d.out_contig = f.in_contig;
    d.scale = e->first ? st->out_scale : 1.0;
    /* cache hint: forward, a pass with out_keep is followed by the pass that re-reads its output; mirrored, that
     * follower is the producer and this one the consumer */
    d.out_keep = (i > 0 && L->v[i - 1].kind == 0 && L->v[i - 1].d.out_keep && !f.out_keep) ? 1 : 0;
    return be->pass(&d, e->dst, (void *)e->src, sx);
  }
  if (e->kind == 2)
    /* one barrier per recorded flag operation, never merged: ranks with an empty tile record no pass between two
     * operations where the others do, and every rank must run the same number of barriers */
    return p2p_barrier(st, e->which, sx);
  return be->a2a(st, e->which, e->cnt, e->peer, (const void *const *)e->rp, e->rb, (void *const *)e->sp, e->sb, sx);
}
/* the steps recorded under `tag` that are exchanges (want_exchange) or passes, mirrored, last recorded first; returns how
 * many ran, -1 on failure */
static int mirror_tagged(hip_state *st, const step_list *L, int tag, int want_exchange, void *sx) {
  int n = 0;
  for (int i = L->n - 1; i >= 0; i--) {
    if (L->v[i].tag != tag || (L->v[i].kind == 1) != (want_exchange != 0)) continue;
    if (mirror_step(st, L, i, sx)) return -1;
    n++;
  }
  return n;
}

static int execute_inverse_multi(struct _offt_plan *po, void *data) {
  hip_state *st = (hip_state *)po->hip_state;
  const offt_backend *be = st->be;
  void *s = st->s_compute;
  if (po->is_r2c) { SET_ERR("complex-to-real inverse is not built (the reference has no inverse at all)"); return -1; }
  /* the forward schedule, recorded: once per (plan state, caller's array) -- a second inverse on the same array replays the
   * kept list (recording costs the host some 0.1 ms during which the device waits) */
  int rc = 0;
  if (!st->inv_cache || st->inv_data != data) {
    inv_cache_drop(st);
    st->inv_cache = (step_list *)calloc(1, sizeof(step_list));
    if (!st->inv_cache) { SET_ERR("out of memory"); return -1; }
    st->rec = st->inv_cache;
    rc = st->slab_zyx ? execute_slab(po, data) : execute_pipeline(po, data, -1);
    st->rec = NULL;
    st->inv_data = data;
    if (rc) inv_cache_drop(st);
  }
  step_list Lempty = {NULL, 0, 0};
  const step_list L = st->inv_cache ? *st->inv_cache : Lempty;
  be->event_record(st->evp[0], s);
  const int overlap_slab = st->slab_zyx && st->x1 && !st->p2p && st->sH > 0 && st->ev_sa && st->ev_s1 && st->sNt > 0;
  const int overlap_pencil = !st->slab_zyx && !st->p2p && (st->x1 || st->x2) && st->ntiles > 0 && st->ev_t2 && st->ev_a2 &&
                             !(getenv("OFFT_INVERSE_IN_ORDER") && atoi(getenv("OFFT_INVERSE_IN_ORDER")));
  if (rc) {
  } else if (overlap_slab) {
    /* Slab schedule over the staged exchange: the mirror image keeps the forward's overlap.  Chunk by chunk, last chunk first:
     * FFTx^-1 and FFTy^-1 of chunk h on the compute stream, then the chunk's exchange (send and receive sides swapped) on the
     * comm stream while the next chunk's passes run; the FFTz^-1 launches wait for the last exchange -- what the forward
     * transform exposes at its start (K1) the inverse exposes at its end. */
    for (int h = st->sH - 1; h >= 0 && rc >= 0; h--) {
      rc = mirror_tagged(st, &L, h, 0, s);
      if (rc < 0) break;
      be->event_record(st->ev_sa[h], s);
      SCHED_WAIT(EDGE_INV_EXCHANGE_AFTER_PASSES, st->s_comm1, st->ev_sa[h]);
      rc = mirror_tagged(st, &L, h, 1, st->s_comm1);
      be->event_record(st->ev_s1[0], st->s_comm1); /* (in order on the comm stream: the last record covers all) */
    }
    if (rc >= 0) {
      SCHED_WAIT(EDGE_INV_K1_AFTER_EXCHANGES, s, st->ev_s1[0]); /* every exchange has landed */
      rc = mirror_tagged(st, &L, -1, 0, s);
    }
    rc = rc < 0 ? -1 : 0;
  } else if (overlap_pencil) {
    /* Pencil schedule over the staged exchanges, mirrored with the forward's overlap:
     *   K3'(h), h = H-1 .. 0, on the compute stream; behind each, the last x-tile's share of chunk h goes back over the
     *     column group (exchange 2, sides swapped) on comm stream 2 -- under K3'(h-1) ..;
     *   then the other tiles' exchange 2 (one message per peer holds every chunk of a tile, so it needs all K3'), tile by
     *     tile, last tile first, with an event behind each;
     *   K2'(k), k = last .. 0, as soon as its tile has landed; behind it exchange 1 of the tile (sides swapped) on comm
     *     stream 1; K1'(k) W1 tiles later -- "K2'(k); K1'(k + W1)" on one in-order stream is the forward's software
     *     pipeline run backwards, and the ring slot of tile k is free again when K2'(k - ring) is issued.             */
    const int nt = st->ntiles, H = st->H2, W = st->ring - 1;
    for (int g = H - 1; g >= 0 && rc >= 0; g--) {
      rc = mirror_tagged(st, &L, PTAG(4, g), 0, s);
      if (rc < 0 || !st->x2) continue;
      be->event_record(st->ev_a2[g], s);
      SCHED_WAIT(EDGE_INV_PENCIL_EX2_AFTER_K3, st->s_comm2, st->ev_a2[g]);
      rc = mirror_tagged(st, &L, PTAG(3, g), 1, st->s_comm2);
    }
    for (int k = nt - 1; k >= 0 && rc >= 0 && st->x2; k--) {
      if (k < nt - 1) rc = mirror_tagged(st, &L, PTAG(2, k), 1, st->s_comm2);
      be->event_record(st->ev_t2[k], st->s_comm2);
    }
    be->event_record(st->evp[1], s);
    be->event_record(st->evp[2], s);
    for (int j = 0; j < nt + W && rc >= 0; j++) {
      if (j < nt) {
        const int k = nt - 1 - j, r = k % st->ring;
        if (st->x2) SCHED_WAIT(EDGE_INV_PENCIL_K2_AFTER_EX2, s, st->ev_t2[k]);
        rc = mirror_tagged(st, &L, PTAG(2, k), 0, s);
        if (rc >= 0 && st->x1) {
          be->event_record(st->ev_k2[r], s);
          SCHED_WAIT(EDGE_INV_PENCIL_EX1_AFTER_K2, st->s_comm1, st->ev_k2[r]);
          rc = mirror_tagged(st, &L, PTAG(1, k), 1, st->s_comm1);
          be->event_record(st->ev_a1[r], st->s_comm1);
        }
      }
      const int jj = j - W;
      if (jj >= 0 && jj < nt && rc >= 0) {
        const int k = nt - 1 - jj, r = k % st->ring;
        if (st->x1) SCHED_WAIT(EDGE_INV_PENCIL_K1_AFTER_EX1, s, st->ev_a1[r]);
        rc = mirror_tagged(st, &L, PTAG(1, k), 0, s);
      }
    }
    rc = rc < 0 ? -1 : 0;
  } else {
    /* every other schedule (direct-store exchange: barriers) replays in order on the compute stream */
    for (int i = L.n - 1; i >= 0 && !rc; i--) rc = mirror_step(st, &L, i, s);
  }
  if (!overlap_pencil) { be->event_record(st->evp[1], s); be->event_record(st->evp[2], s); }
  be->event_record(st->evp[3], s);
  return rc;
}
static void inv_cache_drop(hip_state *st) {
  if (st->inv_cache) { rec_free(st->inv_cache); free(st->inv_cache); }
  st->inv_cache = NULL; st->inv_data = NULL;
}

/* ------------------------------------------------------------------------- */
/* slab schedule (p1 == 1, z-y-x output): ONE exchange, streamed in z-chunks   */
/*                                                                           */
/* With p1 == 1 every rank holds all x and all z of its y-block, there is no  */
/* second exchange, and the reference's phase 2 (z-tiles of T2 planes:        */
/* unpack2 + FFTx, offt-compute.c:3682-3862) only needs the planes it works   */
/* on.  So the exchange is ordered z-chunk-major:                            */
/*   K1(i)     FFTz of x-tile i, stored per peer as [z-chunk][x_t][z][y]  (all i) */
/*             (y contiguous: K1 stores 128-B runs of 8 y at a pitch of one      */
/*             y-line, K2 then READS whole y-lines -- both passes are the        */
/*             contig-in / strided-out flavour at kilobyte pitches.  The first   */
/*             version stored [z_l][y][x_t]: z rows 512 KiB apart at 1024^3 on   */
/*             8 ranks, 62 % of the roofline instead of 74 %; it remains the     */
/*             layout for uneven z blocks and ragged chunks)                     */
/*   a2a(h,i)  z-chunk h of tile i to every peer, chunk-major: chunk 0 tile   */
/*             by tile behind K1(i), later chunks as one grouped call each    */
/*   K2(h)     FFTy of chunk h from the receive volume into R2[z_l][y][x]     */
/*   K3(h)     FFTx of chunk h, contiguous lines, into the caller's z-y-x     */
/* The wire is busy from the end of K1(0) on and K2(h)/K3(h) run while later  */
/* chunks are in flight, so only K1(0) and (if the links are the bottleneck)  */
/* the last chunk's compute are exposed -- instead of a whole x pass after    */
/* the last tile.  Three HBM round trips; every kernel reads or writes whole  */
/* lines or 128-B runs.  (A tile-major order was tried: it delays chunk 0     */
/* until all earlier tiles are fully sent and loses whenever the exchange     */
/* takes longer than the K1 phase.)                                           */
/* T1 / T2 keep their reference meaning (x-tile thickness of the exchange,    */
/* z-thickness of the phase-2 work); unless the caller fixed them they are    */
/* merged upwards until a per-peer message is at least 4 MiB -- the           */
/* reference defaults (M/16) were sized for CPU caches and MPI eager limits.  */
/* ------------------------------------------------------------------------- */
static void slab_teardown(hip_state *st) {
  const offt_backend *be = st->be;
  inv_cache_drop(st);
  be->dfree(st->S1); be->dfree(st->R1); be->dfree(st->R2);
  st->S1 = st->R1 = st->R2 = NULL;
  be->dfree(st->tab_s1); st->tab_s1 = NULL;
  for (int i = 0; i < st->sNt && st->ev_s1; i++) be->event_destroy(st->ev_s1[i]);
  for (int h = 0; h < st->sH && st->ev_sa; h++) be->event_destroy(st->ev_sa[h]);
  free(st->ev_s1); st->ev_s1 = NULL;
  free(st->ev_sa); st->ev_sa = NULL;
  st->sNt = st->sH = 0;
}

static int slab_setup(struct _offt_plan *po, hip_state *st) {
  const offt_backend *be = st->be;
  const struct _offt_comm *c = po->comm;
  const size_t min_msg = (size_t)st->opt.min_msg;
  int T = po->params->v[_T1_], Tz = po->params->v[_T2_];
  if (T < 1) T = 1;
  if (Tz < 1) Tz = 1;
  if (!st->t1_custom) { int t4 = (c->M1 + 3) / 4; if (T < t4) T = t4; } /* 4 tiles: measured best kernel efficiency */
  if (T > c->M1) T = c->M1;
  if (!st->t2_custom) {
    int z8 = (c->M3 + 7) / 8; if (Tz < z8) Tz = z8;
    /* ... but no chunk of R2 larger than the Infinity Cache: K2(h) leaves its chunk there for K3(h) (out_keep).  Per-rank
     * kernels of 2048^3 f32 on 8 ranks: 9.54 ms with 1 GiB chunks, 9.03 ms with 256 MiB ones; 1024^3 f64 has 256 MiB
     * chunks either way (profiles/r02_rehearse_chunks.txt).  More chunks also means a finer exchange pipeline. */
    const int chunk_mib = st->opt.slab_chunk_mib;
    const double plane_mib = (double)c->M4 * (double)c->M1 * (double)st->esz / (1024.0 * 1024.0);
    const int tzc = chunk_mib > 0 ? (int)((double)chunk_mib / plane_mib) : 0;
    if (tzc >= 1 && Tz > tzc) { Tz = tzc; while (Tz > 1 && c->M3 % Tz) Tz--; }
    while (Tz < c->M3 && (size_t)T * c->M2 * Tz * st->esz < min_msg) Tz *= 2;
  }
  if (Tz > c->M3) Tz = c->M3;
  st->sT = T; st->sTz = Tz;
  st->sNt = (c->M1 + T - 1) / T;
  st->sH = (c->M3 + Tz - 1) / Tz;
  st->sblkS = (size_t)c->M3 * c->M2 * T;
  /* y-contiguous block layout (see execute_slab): needs even z blocks that are whole chunks */
  st->slab_yc = c->b3 == 0 && c->M3 % Tz == 0 && !(getenv("OFFT_SLAB_XC_LAYOUT") && atoi(getenv("OFFT_SLAB_XC_LAYOUT")));
  /* y-contiguous layout: a (peer, chunk) block holds its nt tiles, then -- optionally -- a pad of nine 128-B lines, so that
   * the p2 * H blocks one K1 workgroup stores into (and the p2 blocks a K2 line is read from) do not sit a power of two
   * apart (32 MiB at 1024^3 on 8 ranks).  For the x-planes of the single-rank scratch volume such a pad pays (wpad); here
   * it measured as nothing, so it is off by default (opt.block_pad) */
  st->sBc = (size_t)c->M2 * Tz * T * st->sNt + (st->opt.block_pad ? 1152 / st->esz : 0);
  const size_t vol = (st->slab_yc ? st->sBc * st->sH : st->sblkS * st->sNt) * c->p2;
  /* (direct-store exchange: no send volume -- K1 stores into the peers' R1, its own block into its own) */
  if (!(st->p2p && st->x1)) st->S1 = be->dmalloc(vol * st->esz);
  /* the y-transformed volume [z_l][y][x] has the strides of the caller's z-y-x output: K2(h) stores its chunk straight into
   * the caller's array (every FFTz has read its input by then -- the K1 phase precedes the chunks on the compute stream) and
   * K3(h) transforms the x-lines in place.  One volume less per rank: send + receive volumes and the caller's array, the
   * receive volume alone with the direct-store exchange.  (OFFT_SLAB_R2=1: the separate volume of rounds 1-2) */
  st->slab_inplace = c->ostride[0] == 1 && c->ostride[1] == c->M1 && c->ostride[2] == (long long)c->M4 * c->M1 &&
                     !(getenv("OFFT_SLAB_R2") && atoi(getenv("OFFT_SLAB_R2")));
  if (!st->slab_inplace) st->R2 = be->dmalloc((size_t)c->M3 * c->M4 * c->M1 * st->esz);
  if ((!st->S1 && !(st->p2p && st->x1)) || (!st->R2 && !st->slab_inplace)) return -1;
  if (st->x1) {
    st->R1 = be->dmalloc(vol * st->esz);
    if (!st->R1) return -1;
  }
  st->ev_s1 = (void **)calloc(st->sNt, sizeof(void *));
  st->ev_sa = (void **)calloc(st->sH, sizeof(void *));
  for (int i = 0; i < st->sNt; i++) st->ev_s1[i] = be->event_create();
  for (int h = 0; h < st->sH; h++) st->ev_sa[h] = be->event_create();
  /* the self block bypasses the exchange (see tab_self): K1 stores this rank's own z-block straight into R1.  In the
   * y-contiguous layout a peer's share is sH chunk blocks of the split, otherwise one block */
  if (st->self_bypass && st->x1 && c->p2 > 1 && !st->p2p) {
    int ok = 1;
    const long long delta = elem_delta(st, st->R1, st->S1, &ok);
    if (ok) st->tab_s1 = st->slab_yc ? tab_self(st, c->p2, st->sH, (long long)st->sBc, po->rank % c->p2, delta)
                                     : tab_self(st, c->p2, 1, (long long)st->sblkS, po->rank % c->p2, delta);
  }
  return 0;
}

static int execute_slab(struct _offt_plan *po, void *data) {
  hip_state *st = (hip_state *)po->hip_state;
  const offt_backend *be = st->be;
  const struct _offt_comm *c = po->comm;
  const int p2 = c->p2, T = st->sT, Tz = st->sTz, nt = st->sNt, H = st->sH;
  const size_t esz = st->esz;
  void *s = st->s_compute, *sc = st->s_comm1;
  int peers[p2 > 0 ? p2 : 1];
  for (int a = 0; a < p2; a++) peers[a] = a;
  const int nfull = c->m1 / T, tail = c->m1 - nfull * T; /* full x-tiles and the ragged last one */
  /* direct-store exchange: K1 stores every block into its owner's R1 (tab_p1), flags replace the exchange:
   *   wait FREE(e-1) | K1(all tiles) | signal READY(e) | wait READY(e) | K2(h) K3(h) ... | signal FREE(e)              */
  const int p2p = st->p2p && st->x1;
  /* (a schedule that is only being RECORDED for the mirrored inverse issues no flag operation and must not count one) */
  const unsigned long long ep = (p2p && !st->rec) ? ++st->epoch : 0;
  if (p2p && run_wait(st, &st->g1, 1, ep - 1, s, EDGE_P2P_WAIT_FREE)) return -1;

  /* ---- K1: FFTz + pack (offt-compute.c:905-1206), all x-tiles ---- */
  be->event_record(st->evp[0], s);
  int two = st->k1_streams >= 2 && nt > 1 && !st->rec;
  if (two && !st->s_k1b) {
    st->s_k1b = be->stream_create(); st->ev_fork = be->event_create();
    if (!st->s_k1b || !st->ev_fork) two = 0;
  }
  if (two) { be->event_record(st->ev_fork, s); be->stream_wait(st->s_k1b, st->ev_fork); }
  for (int i = 0; i < nt; i++) {
    void *sk = (two && (i & 1)) ? st->s_k1b : s; /* odd tiles on the second stream */
    const int x0 = i * T;
    int myT = c->m1 - x0; if (myT > T) myT = T; if (myT < 0) myT = 0;
    if (myT > 0 && c->m2 > 0) {
      offt_pass_desc d;
      desc_init(&d, st, po->Nz, -1, 2);
      d.real_input = po->is_r2c;
      d.in_axis_stride = 1; d.in_contig = 1; d.out_contig = 0;
      if (st->slab_yc) {
        /* columns = y: block (tile, peer) = [chunk][x_t][z in chunk][y]; z = (peer, chunk, z in chunk) and the peer blocks
         * are whole numbers of chunk blocks, so one split of length Tz addresses both levels */
        d.ncols = c->m2; d.nb1 = myT;
        d.in_col_stride = c->istride[1]; d.in_b1_stride = c->istride[0];
        d.out_axis_stride = c->M2; d.out_col_stride = 1; d.out_b1_stride = (long long)Tz * c->M2;
        d.out_split = Tz; d.out_block_stride = (long long)st->sBc; /* [peer][chunk][tile]: a (peer, chunk) run holds all tiles (+ pad) */
      } else {
        d.ncols = myT; d.nb1 = c->m2;
        d.in_col_stride = c->istride[0]; d.in_b1_stride = c->istride[1];
        d.out_axis_stride = (long long)c->M2 * T; d.out_col_stride = 1; d.out_b1_stride = T;
        if (p2 > 1) { d.out_split = c->F3; d.out_split_nfloor = c->b3 ? p2 - c->b3 : 0; d.out_block_stride = (long long)st->sblkS; }
      }
      d.out_block_tab = p2p ? st->tab_p1 : st->tab_s1; /* (relative to k1dst: the tile offset is the same in S1 and every R1) */
      char *k1dst = (char *)(p2p ? st->R1 : st->S1) + (st->slab_yc ? (size_t)i * c->M2 * Tz * T : (size_t)i * p2 * st->sblkS) * esz;
      if (run_pass(st, &d, (char *)data + (size_t)x0 * c->istride[0] * esz, k1dst, sk, 1)) return -1;
    }
    be->event_record(st->ev_s1[i], sk);
  }
  if (two) be->stream_wait(s, st->ev_s1[((nt - 1) & 1) ? nt - 1 : nt - 2]); /* the last odd tile: the second stream joins */
  be->event_record(st->evp[1], s);

  /* ---- exchange: communicate_a2a (offt-compute.c:862-881), z-chunk-major.  Chunk 0 goes tile by
   * tile right behind the K1 that produced the tile; every later chunk has all tiles ready and
   * goes out as ONE grouped call (nt * p2 pieces).  With t_K1 the K1 phase and C the time on the
   * wire, chunk h lands at about t_K1(0) + (h+1) C/H and is needed at t_K1 + h (t_K2+t_K3)/H:
   * the wire is never idle and compute only waits if the links are the bottleneck. ---- */
  if (p2p) {
    if (run_signal(st, &st->g1, 0, ep, s) || run_wait(st, &st->g1, 0, ep, s, EDGE_P2P_WAIT_READY)) return -1;
  } else if (st->x1) {
    for (int h = 0; h < H; h++) {
      st->rec_tag = h;
      const int z0 = h * Tz;
      int tz = c->M3 - z0; if (tz > Tz) tz = Tz;
      /* y-contiguous layout: the volume is [peer][chunk][tile], so chunk h > 0 is ONE contiguous message per peer (RCCL runs
       * several operations to the same peer of one group one after the other: profiles/r02_overlap_trace_*_first.txt) */
      const int merged = st->slab_yc && h > 0;
      const int groups = (h == 0) ? nt : 1, per = (h == 0 || merged) ? 1 : nt;
      for (int g = 0; g < groups; g++) {
        int cnt = 0;
        const void *sp[per * p2]; void *rp[per * p2]; size_t sb[per * p2], rb[per * p2]; int pr[per * p2];
        if (h == 0) SCHED_WAIT(EDGE_SLAB_EXCHANGE_AFTER_K1, sc, st->ev_s1[g]);
        for (int ii = 0; ii < per; ii++) {
          const int i = (h == 0) ? g : ii;
          for (int a = 0; a < p2; a++) {
            if (st->tab_s1 && a == po->rank % p2) continue; /* K1 stored this rank's own block straight into R1 */
            const int e = cnt++;
            const size_t B = (size_t)c->M2 * Tz * T;
            const size_t off = st->slab_yc ? (((size_t)a * H + h) * st->sBc + (merged ? 0 : (size_t)i) * B) * esz
                                           : (((size_t)i * p2 + a) * st->sblkS + (size_t)z0 * c->M2 * T) * esz;
            pr[e] = peers[a];
            sp[e] = (char *)st->S1 + off;
            rp[e] = (char *)st->R1 + off;
            sb[e] = rb[e] = (merged ? (size_t)nt : (size_t)1) * tz * c->M2 * T * esz;
          }
        }
        if (cnt && run_a2a(st, 1, cnt, pr, sp, sb, rp, rb, sc)) return -1;
      }
      be->event_record(st->ev_sa[h], sc);
    }
  }

  for (int h = 0; h < H; h++) {
    st->rec_tag = h;
    const int z0 = h * Tz;
    int tzh = c->M3 - z0; if (tzh > Tz) tzh = Tz;           /* planes of this chunk in a (padded) peer block */
    int nz = c->m3 - z0; if (nz > tzh) nz = tzh; if (nz < 0) nz = 0; /* ... of which this rank owns nz */
    if (st->x1 && !p2p) SCHED_WAIT(EDGE_SLAB_K2_AFTER_EXCHANGE, s, st->ev_sa[h]);
    /* ---- K2(h): unpack1 + FFTy (offt-compute.c:1208-1520) into R2[z_l][y][x] ---- */
    if (nz > 0) {
      const char *src = (const char *)(st->x1 ? st->R1 : st->S1) + (st->slab_yc ? (size_t)h * st->sBc : (size_t)z0 * c->M2 * T) * esz;
      const size_t blk = st->slab_yc ? st->sBc * H : st->sblkS;  /* peer block */
      for (int part = 0; part < 2; part++) { /* full tiles in one launch, the ragged tile in another */
        const int ntile = part == 0 ? nfull : (tail > 0 ? 1 : 0);
        if (!ntile) continue;
        const int first = part == 0 ? 0 : nfull;
        offt_pass_desc d;
        desc_init(&d, st, po->Ny, -1, 1);
        d.ncols = part == 0 ? T : tail; d.nb1 = nz; d.nb2 = ntile;
        d.in_b2_stride = st->slab_yc ? (long long)c->M2 * Tz * T : (long long)p2 * (long long)blk;  /* next tile */
        if (st->slab_yc) { /* whole y-lines (runs of F2 per peer block), columns = x_t */
          d.in_axis_stride = 1; d.in_contig = 1;
          d.in_col_stride = (long long)Tz * c->M2; d.in_b1_stride = c->M2;
        } else {
          d.in_axis_stride = T; d.in_col_stride = 1; d.in_b1_stride = (long long)c->M2 * T;
        }
        if (p2 > 1) { d.in_split = c->F2; d.in_split_nfloor = c->b2 ? p2 - c->b2 : 0; d.in_block_stride = (long long)blk; }
        d.out_axis_stride = c->M1; d.out_col_stride = 1; d.out_b1_stride = (long long)c->M4 * c->M1; d.out_b2_stride = T;
        d.out_keep = 1; /* K3(h) reads this chunk of R2 right away: keep it in the Infinity Cache (see execute_single) */
        if (run_pass(st, &d, src + (st->slab_yc ? (size_t)first * c->M2 * Tz * T : (size_t)first * p2 * blk) * esz,
                     (char *)(st->slab_inplace ? data : st->R2) + ((size_t)z0 * c->M4 * c->M1 + (size_t)first * T) * esz, s, 0)) return -1;
      }
    }
    /* ---- K3(h): FFTx (offt-compute.c:2729-2730) on whole lines, into the caller's z-y-x layout ---- */
    if (nz > 0 && c->m4 > 0) {
      offt_pass_desc d;
      desc_init(&d, st, po->Nx, -1, 0);
      d.ncols = c->m4; d.nb1 = nz;
      d.in_axis_stride = 1; d.in_col_stride = c->M1; d.in_b1_stride = (long long)c->M4 * c->M1; d.in_contig = 1;
      d.out_axis_stride = c->ostride[0]; d.out_col_stride = c->ostride[1]; d.out_b1_stride = c->ostride[2]; d.out_contig = 1;
      d.scale = st->out_scale;
      /* (K3(h) on a second stream, so that K2(h + 1) shares the chip with it, was tried: 1.439 against 1.443 ms for the
       *  chunked phase at 1024^3 on 8 ranks -- nothing; profiles/r03_rehearse_k3_stream.txt) */
      if (run_pass(st, &d, (char *)(st->slab_inplace ? data : st->R2) + (size_t)z0 * c->M4 * c->M1 * esz, (char *)data + (size_t)z0 * c->ostride[2] * esz, s, 0)) return -1;
    }
  }
  st->rec_tag = -1;
  if (p2p && run_signal(st, &st->g1, 1, ep, s)) return -1; /* R1 is consumed: the peers may store the next transform */
  be->event_record(st->evp[2], s);
  be->event_record(st->evp[3], s);
  return 0;
}

/* ------------------------------------------------------------------------- */
/* tile pipeline for p1 x p2 ranks (forward)                                  */
/*                                                                           */
/* reference:  phase 1, for each x-tile i (T1 planes, window W1):            */
/*               FFTz+pack1(i); wait(i-W); ia2a(i); unpack1+FFTy(i-W)        */
/*                                           (offt-compute.c:3537-3679)      */
/*             [transpose]; phase 2, for each z-tile h (T2 planes, W2):      */
/*               FFTy+pack2(h); wait; ia2a(h); unpack2+FFTx(h-W)             */
/*                                           (offt-compute.c:3682-3862)      */
/* here:       phase 1, for each x-tile i: K1(i) = FFTz storing straight     */
/*             into the per-peer send blocks; a2a1(i) over the row group;    */
/*             K2(i-W) = FFTy loading straight from the receive blocks and   */
/*             storing into the column-exchange volume, which is laid out    */
/*             per peer as [z-chunk h][x][y_l][z in chunk] (T2 planes per    */
/*             chunk); a2a2(i-W): x-tile i-W of every z-chunk over the       */
/*             column group, so exchange 2 streams behind exchange 1.        */
/*             phase 2, for each z-chunk h: K3(h) = FFTx loading from the    */
/*             received column blocks of chunk h and storing in the caller's */
/*             output layout -- launched as soon as the LAST x-tile's share  */
/*             of chunk h has arrived, i.e. under the rest of exchange 2.    */
/* Exposed at the end: the last chunk's FFTx only (W2 = 0, the reference's   */
/* blocking mode, waits for the whole exchange instead).  Three HBM round    */
/* trips per element in total.                                               */
/* ------------------------------------------------------------------------- */

static int execute_pipeline(struct _offt_plan *po, void *data, int dir) {
  hip_state *st = (hip_state *)po->hip_state;
  const offt_backend *be = st->be;
  const struct _offt_comm *c = po->comm;
  const int p1 = c->p1, p2 = c->p2, T = st->T, W = st->ring - 1;
  const int Nx = po->Nx, Ny = po->Ny, Nz = po->Nz;
  const int Tz = st->Tz2, H = st->H2;
  const size_t esz = st->esz;
  const size_t MM = (size_t)c->M1 * c->M4; /* elements per z-plane of a peer block of exchange 2 */
  void *s = st->s_compute;
  int peers1[p2 > 0 ? p2 : 1], peers2[p1 > 0 ? p1 : 1];
  for (int a = 0; a < p2; a++) peers1[a] = a; /* key = rank_y inside comm1 */
  for (int a = 0; a < p1; a++) peers2[a] = a; /* key = rank_x inside comm2 */
  const int Hf = c->m3 / Tz;                   /* chunks whose Tz planes are all owned by this rank */
  const int w2 = po->params->v[_W2_];
  /* direct-store exchange: K1 stores into the row peers' ring slots (tab_pr[r]), K2 into the column peers' recv2 (tab_p2):
   *   per tile  wait FREE1[r] | K1 | signal READY1[r] ... wait READY1[r] | K2 | signal FREE1[r], signal READY2
   *   then      wait READY2 (all tiles) | K3(h) ... | signal FREE2                                                    */
  const int pp1 = st->p2p && st->x1, pp2 = st->p2p && st->x2;
  const unsigned long long ep = (st->p2p && !st->rec) ? ++st->epoch : 0; /* (recording for the mirrored inverse: no flags, no counting) */

  /* One in-order compute stream carries the reference's software pipeline
   * "pack(i); wait(i-W); ia2a(i); unpack(i-W)" (offt-compute.c:3537-3647): K1(i) then
   * K2(i-W), so that tile i's exchange runs under K1(i+1..i+W).  (Putting K1 and K2 on
   * two streams was measured and is slower: cross-stream event edges cost more than the
   * tail waves they fill -- profiles/r01_sweep.txt.) */
  be->event_record(st->evp[0], s);
  for (int i = 0; i < st->ntiles + W; i++) {
    if (i < st->ntiles) {
      /* ---- K1(i): FFTz + pack1 (offt-compute.c:905-1206) ---- */
      const int r = i % st->ring, x0 = i * T;
      int myT = c->m1 - x0; if (myT > T) myT = T; if (myT < 0) myT = 0;
      st->rec_tag = PTAG(1, i);
      if (i >= st->ring) be->stream_wait(s, st->ev_k2[r]); /* slot's previous tile fully consumed */
      if (pp1) { /* ... by every row peer: they read what this rank stored into THEIR slot r */
        if (run_wait(st, &st->g1, st->ring + r, st->use1[r], s, EDGE_P2P_WAIT_FREE)) return -1;
        if (!st->rec) st->use1[r]++;
      }
      if (myT > 0 && c->m2 > 0) {
        offt_pass_desc d;
        desc_init(&d, st, Nz, dir, 2);
        d.real_input = po->is_r2c;
        d.ncols = c->m2; d.nb1 = myT;
        d.in_axis_stride = 1; d.in_col_stride = c->istride[1]; d.in_b1_stride = c->istride[0];
        d.in_contig = 1;
        if (st->pencil_yc) { /* send block [x_t][z_l][y]: 128-B runs of 8 y at a pitch of one y-line */
          d.out_axis_stride = c->M2; d.out_col_stride = 1; d.out_b1_stride = (long long)c->M3 * c->M2;
          d.out_contig = 0;
        } else {             /* send block [x_t][y][z-run] */
          d.out_axis_stride = 1; d.out_col_stride = c->M3; d.out_b1_stride = (long long)c->M2 * c->M3;
          d.out_contig = 1;
        }
        if (p2 > 1) { /* peer a owns z in [a*F3, ..): offt-compute.c:1015-1027 */
          d.out_split = c->F3; d.out_split_nfloor = c->b3 ? p2 - c->b3 : 0;
          d.out_block_stride = (long long)st->blk1;
          if (st->tab_r1) d.out_block_tab = st->tab_r1[r];
          if (pp1) d.out_block_tab = st->tab_pr[r];
        }
        if (run_pass(st, &d, (char *)data + (size_t)x0 * c->istride[0] * esz, st->send1[r], s, 1)) return -1;
      }
      be->event_record(st->ev_k1[r], s);
      /* ---- a2a1(i) over comm1 (offt-compute.c:862-881) ---- */
      if (pp1) {
        if (run_signal(st, &st->g1, r, st->use1[r], s)) return -1;
      } else if (st->x1) {
        SCHED_WAIT(EDGE_PENCIL_EX1_AFTER_K1, st->s_comm1, st->ev_k1[r]);
        const void *sp[p2]; void *rp[p2]; size_t sb[p2], rb[p2]; int pr[p2], cnt = 0;
        for (int a = 0; a < p2; a++) {
          if (st->tab_r1 && a == po->rank % p2) continue; /* K1 stored this rank's own block straight into recv1[r] */
          pr[cnt] = peers1[a];
          sp[cnt] = (char *)st->send1[r] + (size_t)a * st->blk1 * esz;
          rp[cnt] = (char *)st->recv1[r] + (size_t)a * st->blk1 * esz;
          sb[cnt] = rb[cnt] = (size_t)myT * c->M2 * c->M3 * esz;
          cnt++;
        }
        if (myT > 0 && cnt && run_a2a(st, 1, cnt, pr, sp, sb, rp, rb, st->s_comm1)) return -1;
        be->event_record(st->ev_a1[r], st->s_comm1);
      }
    }
    const int k = i - W;
    if (k >= 0 && k < st->ntiles) {
      /* ---- K2(k): unpack1 + FFTy (+ pack2) (offt-compute.c:1208-1520, 1636-2345) ---- */
      const int r = k % st->ring, x0 = k * T;
      int myT = c->m1 - x0; if (myT > T) myT = T; if (myT < 0) myT = 0;
      st->rec_tag = PTAG(2, k);
      if (pp1) { if (run_wait(st, &st->g1, r, st->use1[r], s, EDGE_P2P_WAIT_READY)) return -1; }
      else if (st->x1) SCHED_WAIT(EDGE_PENCIL_K2_AFTER_EX1, s, st->ev_a1[r]);
      if (pp2 && k == 0 && run_wait(st, &st->g2, 1, ep - 1, s, EDGE_P2P_WAIT_FREE)) return -1; /* the column peers have consumed the previous transform's recv2 */
      if (st->pencil_yc && myT > 0) {
        /* whole y-lines out of the receive blocks [x_t][z_l][y_l] (runs of F2 per peer), columns = x_t, into the
         * column-exchange volume [chunk][peer][x-tile][z in chunk][y_l][x_t]: x contiguous, one (chunk, peer, tile)
         * block = one message of exchange 2 */
        const size_t B2 = (size_t)Tz * c->M4 * T;
        offt_pass_desc d;
        desc_init(&d, st, Ny, dir, 1);
        d.ncols = myT; d.nb1 = Tz; d.nb2 = H;
        d.in_axis_stride = 1; d.in_contig = 1;
        d.in_col_stride = (long long)c->M3 * c->M2; d.in_b1_stride = c->M2; d.in_b2_stride = (long long)Tz * c->M2;
        if (p2 > 1) {
          d.in_split = c->F2; d.in_split_nfloor = c->b2 ? p2 - c->b2 : 0;
          d.in_block_stride = (long long)st->blk1;
        }
        d.out_axis_stride = T; d.out_col_stride = 1; d.out_b1_stride = (long long)c->M4 * T;
        d.out_b2_stride = (long long)B2; /* volume [peer][x-tile][z-chunk]: all chunks of a (peer, tile) are one contiguous message */
        if (p1 > 1) { /* peer a owns y in [a*F4, ..): offt-compute.c:1758-1776 */
          d.out_split = c->F4; d.out_split_nfloor = c->b4 ? p1 - c->b4 : 0;
          d.out_block_stride = (long long)st->blk2;
          d.out_block_tab = pp2 ? st->tab_p2 : st->tab_x2; /* (relative to the launch pointer: tile / chunk offsets are the same in send2 and every recv2) */
        }
        if (run_pass(st, &d, st->recv1[r], (char *)st->send2 + (size_t)k * st->B2t * esz, s, 0)) return -1;
      }
      for (int part = 0; part < 2 && myT > 0 && !st->pencil_yc; part++) {
        /* the chunks this rank fills completely go in one launch (chunk = second batch dimension), the ragged last
         * chunk (m3 not a multiple of T2) in another */
        const int h0 = part == 0 ? 0 : Hf, nh = part == 0 ? Hf : (c->m3 > Hf * Tz ? 1 : 0);
        if (!nh) continue;
        const int z0 = h0 * Tz;
        int tzh = c->M3 - z0; if (tzh > Tz) tzh = Tz;                   /* planes of the chunk in the (padded) layout */
        const int nz = part == 0 ? Tz : c->m3 - z0;                    /* ... of which this rank owns nz */
        offt_pass_desc d;
        desc_init(&d, st, Ny, dir, 1);
        d.ncols = nz; d.nb1 = myT; d.nb2 = nh;
        d.in_axis_stride = c->M3; d.in_col_stride = 1; d.in_b1_stride = (long long)c->M2 * c->M3; d.in_b2_stride = Tz;
        if (p2 > 1) {
          d.in_split = c->F2; d.in_split_nfloor = c->b2 ? p2 - c->b2 : 0;
          d.in_block_stride = (long long)st->blk1;
        }
        d.out_axis_stride = tzh; d.out_col_stride = 1; d.out_b1_stride = (long long)c->M4 * tzh;
        d.out_b2_stride = (long long)MM * Tz;
        if (p1 > 1) { /* peer a owns y in [a*F4, ..): offt-compute.c:1758-1776 */
          d.out_split = c->F4; d.out_split_nfloor = c->b4 ? p1 - c->b4 : 0;
          d.out_block_stride = (long long)st->blk2;
          d.out_block_tab = pp2 ? st->tab_p2 : st->tab_x2;
        }
        if (run_pass(st, &d, (char *)st->recv1[r] + (size_t)z0 * esz,
                     (char *)st->send2 + ((size_t)z0 * MM + (size_t)x0 * c->M4 * tzh) * esz, s, 0)) return -1;
      }
      be->event_record(st->ev_k2[r], s);
      if (pp1 && run_signal(st, &st->g1, st->ring + r, st->use1[r], s)) return -1; /* slot r consumed */
      if (pp2 && run_signal(st, &st->g2, 0, st->rec ? 0 : ++st->tiles2, s)) return -1; /* tile k is in the column peers' recv2 */
      /* ---- a2a2(k) over comm2: x-tile k of every z-chunk of every column block.  All chunks of a tile go in one
       * grouped call, except for the last tile: there each chunk is its own call with an event behind it, so that
       * K3(h) can start while chunks h+1.. are still on the wire ---- */
      if (st->x2 && !pp2) {
        SCHED_WAIT(EDGE_PENCIL_EX2_AFTER_K2, st->s_comm2, st->ev_k2[r]);
        const int last = (k == st->ntiles - 1);
        const int merged = st->pencil_yc && !last; /* one contiguous message per peer holds every chunk of the tile */
        const int ngroups = last ? H : 1, per = (last || merged) ? 1 : H;
        for (int g = 0; g < ngroups; g++) {
          int cnt = 0;
          const void *sp[per * p1]; void *rp[per * p1]; size_t sb[per * p1], rb[per * p1]; int pr[per * p1];
          int any = 0;
          for (int hh = 0; hh < per; hh++) {
            const int h = last ? g : hh, z0 = h * Tz;
            int tzh = c->M3 - z0; if (tzh > Tz) tzh = Tz;
            for (int a = 0; a < p1; a++) {
              if (st->tab_x2 && a == po->rank / p2) continue; /* K2 stored this rank's own block straight into recv2 */
              const int e = cnt++;
              int ma = blk_size(a, c->F1, c->b1, p1) - x0; if (ma > T) ma = T; if (ma < 0) ma = 0;
              const size_t off = st->pencil_yc
                                     ? ((size_t)a * st->blk2 + (size_t)k * st->B2t + (merged ? 0 : (size_t)h) * (size_t)Tz * c->M4 * T) * esz
                                     : ((size_t)a * st->blk2 + (size_t)z0 * MM + (size_t)x0 * c->M4 * tzh) * esz;
              const size_t nch = merged ? (size_t)H : (size_t)1;
              pr[e] = peers2[a];
              sp[e] = (char *)st->send2 + off;
              rp[e] = (char *)st->recv2 + off;
              sb[e] = nch * myT * c->M4 * tzh * esz;
              rb[e] = nch * ma * c->M4 * tzh * esz;
              any |= (sb[e] || rb[e]);
            }
          }
          st->rec_tag = last ? PTAG(3, g) : PTAG(2, k);
          if (any && run_a2a(st, 2, cnt, pr, sp, sb, rp, rb, st->s_comm2)) return -1;
          if (last) be->event_record(st->ev_a2[g], st->s_comm2);
        }
      }
    }
  }
  be->event_record(st->evp[2], s);
  if (pp2 && run_wait(st, &st->g2, 0, st->tiles2, s, EDGE_P2P_WAIT_READY)) return -1; /* every column peer has stored all its tiles (they count like this rank) */
  /* ---- K3(h): unpack2 + FFTx into the caller's layout (offt-compute.c:2347-2993), z-chunk by z-chunk ---- */
  for (int h = 0; h < H; h++) {
    const int z0 = h * Tz;
    int tzh = c->M3 - z0; if (tzh > Tz) tzh = Tz;
    int nz = c->m3 - z0; if (nz > tzh) nz = tzh;
    if (st->x2 && !pp2) SCHED_WAIT(EDGE_PENCIL_K3_AFTER_EX2, s, st->ev_a2[w2 == 0 ? H - 1 : h]);
    st->rec_tag = PTAG(4, h);
    if (nz <= 0 || c->m4 <= 0) continue;
    offt_pass_desc d;
    desc_init(&d, st, Nx, dir, 0);
    d.out_axis_stride = c->ostride[0];
    d.out_contig = (c->ostride[0] == 1);
    d.scale = st->out_scale;
    if (st->pencil_yc) {
      /* whole x-lines: x = (peer, tile, x_t) and the blocks (peer, tile) of a chunk are consecutive, so one split of length
       * T addresses both levels; columns = y_l */
      const size_t B2 = (size_t)Tz * c->M4 * T;
      d.ncols = c->m4; d.nb1 = nz;
      d.in_axis_stride = 1; d.in_contig = 1; d.in_col_stride = T; d.in_b1_stride = (long long)c->M4 * T;
      d.in_split = T; d.in_block_stride = (long long)st->B2t; /* (peer, tile) blocks are H chunks (+ pad) apart, peers ntiles of them */
      d.out_col_stride = c->ostride[1]; d.out_b1_stride = c->ostride[2];
      if (run_pass(st, &d, (char *)st->recv2 + (size_t)h * B2 * esz, (char *)data + (size_t)z0 * c->ostride[2] * esz, s, 0)) return -1;
      continue;
    }
    d.ncols = nz; d.nb1 = c->m4;
    d.in_axis_stride = (long long)c->M4 * tzh; d.in_col_stride = 1; d.in_b1_stride = tzh;
    if (p1 > 1) { /* peer a owns x in [a*F1, ..): offt-compute.c:2432-2450 */
      d.in_split = c->F1; d.in_split_nfloor = c->b1 ? p1 - c->b1 : 0;
      d.in_block_stride = (long long)st->blk2;
    }
    d.out_col_stride = c->ostride[2]; d.out_b1_stride = c->ostride[1];
    if (run_pass(st, &d, (char *)st->recv2 + (size_t)z0 * MM * esz, (char *)data + (size_t)z0 * c->ostride[2] * esz, s, 0)) return -1;
  }
  if (pp2 && run_signal(st, &st->g2, 1, ep, s)) return -1; /* recv2 is consumed */
  be->event_record(st->evp[3], s);
  return 0;
}

/* ------------------------------------------------------------------------- */
/* execute (offt-compute.c:3864-4048)                                         */
/* ------------------------------------------------------------------------- */
/* a communicator of this plan (or the world) reported an asynchronous error, or a wait ran out of time: abort the
 * communicators so that their kernels leave the GPU, and make every later call fail fast.  The reference has no failure
 * path here at all -- communicate_wait (offt-compute.c:883-890) sits in MPI_Wait forever when a peer dies. */
static void comm_fail(hip_state *st) {
  G.comm_failed = 1;
  if (R.CommAbort) {
    if (st->have_comm1) { (void)R.CommAbort(st->comm1); st->have_comm1 = 0; st->comm1 = NULL; }
    if (st->have_comm2) { (void)R.CommAbort(st->comm2); st->have_comm2 = 0; st->comm2 = NULL; }
    if (st->comm1 == G.world) st->comm1 = NULL; /* a group that spans the world uses the world communicator itself */
    if (st->comm2 == G.world) st->comm2 = NULL;
    if (G.have_comm) { (void)R.CommAbort(G.world); G.have_comm = 0; G.world = NULL; }
  }
  /* an RCCL without ncclCommAbort: the communicators stay as they are -- their kernels may still be spinning, so they
   * are neither destroyed (ncclCommDestroy would block) nor forgotten; G.comm_failed makes every later call fail fast
   * and mesh_teardown / offt_hip_finalize_world skip them (a documented leak, the process is expected to exit) */
}

static int comm_async_error(hip_state *st) {
  if (!R.CommGetAsyncError) return 0;
  ncclComm_t cs[3] = {G.have_comm ? G.world : NULL, st->have_comm1 ? st->comm1 : NULL, st->have_comm2 ? st->comm2 : NULL};
  for (int i = 0; i < 3; i++) {
    ncclResult_t ar = 0;
    if (!cs[i]) continue;
    if (R.CommGetAsyncError(cs[i], &ar) != 0 || (ar != 0 && ar != NCCL_IN_PROGRESS)) {
      SET_ERR("RCCL communicator %d reported an asynchronous error: %s", i, R.GetErrorString(ar));
      return -1;
    }
  }
  return 0;
}

/* wait for the compute stream.  A plan that exchanges over RCCL polls instead of blocking: every millisecond the
 * communicators are asked for asynchronous errors, and the wait is bounded (OFFT_EXEC_TIMEOUT seconds, default 120) --
 * a dead or stuck peer becomes the failure marker t[ALL] = 99999999 plus offt_hip_last_error(), not a hang. */
static int wait_compute_ex(hip_state *st, int watch) {
  if (!watch) return st->be->stream_sync(st->s_compute);
  const double limit = st->opt.exec_timeout_s;
  const double t0 = wall_seconds();
  double tchk = t0;
  for (;;) {
    const hipError_t q = hipStreamQuery((hipStream_t)st->s_compute);
    if (q == hipSuccess) break;
    if (q != hipErrorNotReady) { SET_ERR("compute stream failed: %s", hipGetErrorString(q)); comm_fail(st); return -1; }
    const double now = wall_seconds();
    if (now - tchk < 1e-3) continue;
    tchk = now;
    if (comm_async_error(st)) { comm_fail(st); return -1; }
    if (now - t0 > limit) {
      SET_ERR("offt_3d_execute: no completion after %.0f s (a peer is missing or an exchange is stuck); communicators aborted", limit);
      comm_fail(st);
      return -1;
    }
  }
  if (comm_async_error(st)) { comm_fail(st); return -1; }
  return 0;
}
static int wait_compute(hip_state *st) {
  const int rc = wait_compute_ex(st, st->uses_rccl);
  if (!rc && st->p2p && st->p2p_status && *(volatile unsigned long long *)st->p2p_status) {
    /* a wait kernel of the direct-store exchange ran out of time: a peer never signalled */
    SET_ERR("offt_3d_execute: a flag of the direct-store exchange did not arrive (a peer is missing or stuck)");
    G.comm_failed = 1;
    return -1;
  }
  return rc;
}

void offt_3d_execute_dir(struct _offt_plan *po, void *in, void *out, int direction) {
  hip_state *st = (hip_state *)po->hip_state;
  const offt_backend *be = st->be;
  double *t = po->t;
  memset(t, 0, GES * sizeof(double));
  TEST_NOTE_MESH(po);
  if (in != out && !st->warned_in) {
    /* offt-compute.c:3866: "in must be equal to out" -- the reference silently ignores `in` */
    fprintf(stderr, "offt(hip): offt_3d_execute is in-place; `in` is ignored (as in the reference)\n");
    st->warned_in = 1;
  }
  if ((st->uses_rccl || st->p2p) && G.comm_failed) {
    SET_ERR("offt_3d_execute: the communicator failed earlier; make a new world (offt_hip_set_world) and plan");
    t[ALL] = 99999999.0;
    return;
  }
  double t0 = wall_seconds();
  void *data = out;
  int staged = 0;
  const size_t bytes = local_elems(po->comm) * st->esz;
  if (!g_backend && !is_device_ptr(out)) {
    /* host buffer handed over the boundary (the reference's calloc'ed array,
     * run-fft.c:304): stage through HBM; PCIe-inclusive, not the measured path */
    if (!st->stage) { st->stage = be->dmalloc(bytes); st->stage_bytes = bytes; }
    if (!st->stage) { t[ALL] = 99999999.0; return; }
    HCHECK(hipMemcpy(st->stage, out, bytes, hipMemcpyHostToDevice), { t[ALL] = 99999999.0; return; });
    data = st->stage;
    staged = 1;
  }
  const int timed = st->timed = !(st->async && !staged);
  if (timed) be->event_record(st->ev0, st->s_compute);
  int rc;
  if (!st->use_pipeline) rc = execute_single(po, data, direction);
  else if (direction > 0) rc = execute_inverse_multi(po, data);
  else rc = st->slab_zyx ? execute_slab(po, data) : execute_pipeline(po, data, direction);
  if (timed) be->event_record(st->ev1, st->s_compute);
  if (rc) { /* the reference's failure marker, offt-compute.c:3881 */
    if (st->uses_rccl) comm_fail(st); /* peers may already sit in an exchange this rank will never join */
    t[ALL] = 99999999.0;
    return;
  }
  if (st->async && !staged) { t[ALL] = wall_seconds() - t0; return; }
  if (wait_compute(st)) { t[ALL] = 99999999.0; return; }
  st->last_dev_s = 1e-3 * be->event_ms(st->ev0, st->ev1);
  double a = 1e-3 * be->event_ms(st->evp[0], st->evp[1]);
  double b = 1e-3 * be->event_ms(st->evp[1], st->evp[2]);
  double cc = 1e-3 * be->event_ms(st->evp[2], st->evp[3]);
  if (st->use_pipeline && st->slab_zyx) {
    /* K1 phase (all FFTz + pack), then the chunked exchange / FFTy / FFTx phase */
    st->pass_s[0] = a; st->pass_s[1] = 0; st->pass_s[2] = b;
    t[PACK1] = a; t[FFTx] = b;
  } else if (st->use_pipeline) {
    double ph1 = 1e-3 * be->event_ms(st->evp[0], st->evp[2]);
    st->pass_s[0] = ph1; st->pass_s[1] = 0; st->pass_s[2] = cc;
    t[PACK1] = ph1; t[FFTx] = cc;
  } else {
    double tt[3] = {a, b, cc};
    /* the y and x launches alternate (execute_single): their shared time, halved */
    if (st->yx_fused) { const int i0 = st->yx_fused - 1; const double pr = tt[i0] + tt[i0 + 1]; tt[i0] = tt[i0 + 1] = 0.5 * pr; }
    for (int i = 0; i < 3; i++) st->pass_s[st->pass_slot[i]] = tt[i];
    t[FFTz] = st->pass_s[0]; t[FFTy1] = st->pass_s[1]; t[FFTx] = st->pass_s[2];
  }
  if (staged) HCHECK(hipMemcpy(out, st->stage, bytes, hipMemcpyDeviceToHost), { t[ALL] = 99999999.0; return; });
  t[ALL] = wall_seconds() - t0;
}

/* asynchronous mode (offt_hip_set_async): wait for everything enqueued so far -- the same bounded, error-polling
 * wait a synchronous execute ends with.  0 on success; on failure the communicators are aborted and t[ALL] carries the
 * reference's failure marker. */
int offt_hip_wait(struct _offt_plan *po) {
  hip_state *st = (hip_state *)po->hip_state;
  if (wait_compute(st)) { po->t[ALL] = 99999999.0; return -1; }
  return 0;
}

void offt_3d_execute(struct _offt_plan *po, double *in, double *out, int is_tuning) {
  (void)is_tuning;
  offt_3d_execute_dir(po, in, out, -1);
}

/* ------------------------------------------------------------------------- */
/* device helpers                                                             */
/* ------------------------------------------------------------------------- */
void *offt_hip_malloc(long long bytes) { return hb_malloc((size_t)bytes); }
void offt_hip_free(void *p) { hb_free(p); }
int offt_hip_memcpy_h2d(void *dst, const void *src, long long bytes) {
  HCHECK(hipMemcpy(dst, src, (size_t)bytes, hipMemcpyHostToDevice), return -1);
  return 0;
}
int offt_hip_memcpy_d2h(void *dst, const void *src, long long bytes) {
  HCHECK(hipMemcpy(dst, src, (size_t)bytes, hipMemcpyDeviceToHost), return -1);
  return 0;
}
int offt_hip_device_synchronize(void) { HCHECK(hipDeviceSynchronize(), return -1); return 0; }

int offt_hip_fill_input(struct _offt_plan *po, void *buf, int kind) {
  hip_state *st = (hip_state *)po->hip_state;
  const struct _offt_comm *c = po->comm;
  int rc;
  if (po->is_r2c) /* real rows: element (x,y,z) at scalar index z + 2*istride1*y + 2*istride0*x (run-fft.c:54) */
    rc = offt_hipk_fill(buf, st->prec | 0x100, kind, c->isize[0], c->isize[1], c->isize[2], c->istart[0], c->istart[1],
                        c->istart[2], 2LL * c->istride[0], 2LL * c->istride[1], 1, st->s_compute);
  else
    rc = offt_hipk_fill(buf, st->prec, kind, c->isize[0], c->isize[1], c->isize[2], c->istart[0], c->istart[1],
                        c->istart[2], c->istride[0], c->istride[1], c->istride[2], st->s_compute);
  if (rc) { SET_ERR("%s", offt_hipk_last_error()); return rc; }
  return st->be->stream_sync(st->s_compute);
}

/* ------------------------------------------------------------------------- */
/* xGMI link probe: grouped ncclSend/ncclRecv on the world communicator, the   */
/* exact primitive of the exchanges above, timed with HIP events.              */
/*   mode 0  all-to-all: every rank sends `bytes` to every other rank          */
/*   mode 1  shift: rank r sends `bytes` to (r + shift) % size and receives    */
/*           from (r - shift) % size -- one link per direction                 */
/* Returns this rank's seconds per repetition (the caller takes the max over   */
/* ranks), negative on failure.  SURVEY.md 7 hard part #1: decide mesh shapes   */
/* with a measured link number, not a data-sheet one.                           */
/* ------------------------------------------------------------------------- */
double offt_hip_link_probe(int mode, int shift, long long bytes, int reps) {
  if (!G.have_comm || G.comm_failed) { SET_ERR("offt_hip_link_probe: no world communicator"); return -1.0; }
  if (bytes < 1 || reps < 1) { SET_ERR("offt_hip_link_probe: bad arguments"); return -1.0; }
  const int p = G.size, me = G.rank;
  const int npeer = mode == 0 ? p : 1;
  char *sbuf = NULL, *rbuf = NULL;
  hipStream_t s = NULL;
  hipEvent_t e0 = NULL, e1 = NULL;
  double result = -1.0;
  if (hipMalloc((void **)&sbuf, (size_t)bytes * npeer) != hipSuccess || hipMalloc((void **)&rbuf, (size_t)bytes * npeer) != hipSuccess ||
      hipMemset(sbuf, 1, (size_t)bytes * npeer) != hipSuccess || hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) {
    SET_ERR("offt_hip_link_probe: allocation failed");
    goto done;
  }
  int broken = 0; /* an exchange was enqueued and may never finish: the communicator is aborted before anything is freed */
  for (int r = -1; r < reps && !broken; r++) { /* r = -1: warm-up (connection set-up), not timed */
    if (r == 0 && hipEventRecord(e0, s) != hipSuccess) goto done;
    NCHECK(R.GroupStart(), goto done);
    int grc = 0;
    if (mode == 0) {
      for (int a = 0; a < p && !grc; a++) {
        if (a == me) continue;
        NCHECK(R.Send(sbuf + (size_t)a * bytes, (size_t)bytes, NCCL_INT8, a, G.world, s), grc = -1);
        if (!grc) NCHECK(R.Recv(rbuf + (size_t)a * bytes, (size_t)bytes, NCCL_INT8, a, G.world, s), grc = -1);
      }
    } else {
      const int to = (me + shift % p + p) % p, from = (me - shift % p + p) % p;
      if (to != me) {
        NCHECK(R.Send(sbuf, (size_t)bytes, NCCL_INT8, to, G.world, s), grc = -1);
        if (!grc) NCHECK(R.Recv(rbuf, (size_t)bytes, NCCL_INT8, from, G.world, s), grc = -1);
      }
    }
    if (grc) { (void)R.GroupEnd(); broken = 1; break; } /* never leave a group open behind a failed call (as hb_a2a) */
    NCHECK(R.GroupEnd(), { broken = 1; break; });
  }
  if (!broken && hipEventRecord(e1, s) != hipSuccess) broken = 1;
  if (!broken) {
    /* bounded, error-polling wait, like the end of an execute */
    const double t0 = wall_seconds();
    double tchk = t0;
    for (;;) {
      const hipError_t q = hipStreamQuery(s);
      if (q == hipSuccess) break;
      const double now = wall_seconds();
      if (q != hipErrorNotReady || now - t0 > 60.0) { SET_ERR("offt_hip_link_probe: exchange did not complete"); broken = 1; break; }
      if (now - tchk < 1e-3) continue;
      tchk = now;
      if (R.CommGetAsyncError) {
        ncclResult_t ar = 0;
        if (R.CommGetAsyncError(G.world, &ar) != 0 || (ar != 0 && ar != NCCL_IN_PROGRESS)) {
          SET_ERR("offt_hip_link_probe: RCCL reported an asynchronous error: %s", R.GetErrorString(ar));
          broken = 1; break;
        }
      }
    }
    float ms = 0;
    if (!broken && hipEventElapsedTime(&ms, e0, e1) == hipSuccess) result = 1e-3 * ms / reps;
  }
  if (broken) {
    /* RCCL kernels may still be using the buffers: abort the communicator first (hipFree synchronises the device and
     * would turn the time-out into a hang), and make every later execute on this world fail fast */
    G.comm_failed = 1;
    if (R.CommAbort) { (void)R.CommAbort(G.world); G.have_comm = 0; G.world = NULL; }
    else { sbuf = rbuf = NULL; s = NULL; } /* no abort available: leak the buffers and the stream rather than block */
  }
done:
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  if (s) (void)hipStreamDestroy(s);
  if (sbuf) (void)hipFree(sbuf);
  if (rbuf) (void)hipFree(rbuf);
  return result;
}
