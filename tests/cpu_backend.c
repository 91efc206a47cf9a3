/*
 * cpu_backend.c -- TEST INFRASTRUCTURE ONLY.
 *
 * A CPU interpreter of offt_pass_desc plus host-memory "streams", installed
 * through offt_hip_test_set_backend() so that the CPU-only test-suite can run
 * the product's real host logic (decomposition, pass descriptors, tile ring,
 * exchange schedule of offt_host.c) without a GPU, including world_size-2
 * `gloo` runs where the all-to-all is a Python callback.  It is built into
 * tests/libcpubackend.so, never into offt_amd/liboffthip.so.
 */
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "offt_backend.h"
#include "oracle.h"

typedef int (*a2a_cb_t)(int which, int npeers, const int *peer, const void *const *sendp, const size_t *sendbytes,
                        void *const *recvp, const size_t *recvbytes);
static a2a_cb_t g_a2a_cb = NULL;
static long g_pass_count = 0;

/* failure injection: the n-th allocation from now fails once (an out-of-memory on ONE rank, tests/test_host_logic.py) */
static long g_alloc_count = 0, g_fail_alloc_at = -1;
static void *cb_malloc(size_t b) {
  if (++g_alloc_count == g_fail_alloc_at) { g_fail_alloc_at = -1; return NULL; }
  return calloc(1, b ? b : 16);
}
static void cb_free(void *p) { free(p); }
static int cb_prepare(int n, int prec) { (void)n; (void)prec; return 0; }

static long long split_off(int k, int split, int nfloor, long long blk, long long axis, const long long *tab) {
  if (split == 0 && nfloor == 0) return (long long)k * axis;
  int a, r;
  if (nfloor > 0 && k >= split * nfloor) { int kk = k - split * nfloor; a = nfloor + kk / (split + 1); r = kk % (split + 1); }
  else { a = k / split; r = k % split; }
  return (tab ? tab[a] : (long long)a * blk) + (long long)r * axis; /* tab: offt_pass_desc::in_block_tab / out_block_tab */
}

static int cb_pass(const offt_pass_desc *d, const void *in, void *out, void *stream) {
  (void)stream;
  g_pass_count++;
  if (d->n < 1 || d->ncols < 1 || d->nb1 < 1 || d->nb2 < 1) return 0;
  const int n = d->n, f32 = d->precision == OFFT_PREC_F32;
  orc_fft_plan *pl = orc_fft_plan_create(n);
  double *line = (double *)malloc(sizeof(double) * 2 * (size_t)n), *scr = (double *)malloc(sizeof(double) * 6 * (size_t)n + 64);
  for (int b2 = 0; b2 < d->nb2; b2++)
    for (int b1 = 0; b1 < d->nb1; b1++)
      for (int c = 0; c < d->ncols; c++) {
        long long ib = (long long)b1 * d->in_b1_stride + (long long)b2 * d->in_b2_stride + (long long)c * d->in_col_stride;
        long long ob = (long long)b1 * d->out_b1_stride + (long long)b2 * d->out_b2_stride + (long long)c * d->out_col_stride;
        for (int k = 0; k < n; k++) {
          long long o = ib + split_off(k, d->in_split, d->in_split_nfloor, d->in_block_stride, d->in_axis_stride, d->in_block_tab);
          double re, im;
          if (d->real_input) { re = f32 ? ((const float *)in)[2 * ib + k] : ((const double *)in)[2 * ib + k]; im = 0.0; }
          else if (f32) { re = ((const float *)in)[2 * o]; im = ((const float *)in)[2 * o + 1]; }
          else { re = ((const double *)in)[2 * o]; im = ((const double *)in)[2 * o + 1]; }
          line[2 * k] = re; line[2 * k + 1] = d->direction > 0 ? -im : im;
        }
        orc_fft_execute(pl, line, 1, 0, 1, scr);
        for (int k = 0; k < (d->real_input ? n / 2 + 1 : n); k++) {
          long long o = ob + split_off(k, d->out_split, d->out_split_nfloor, d->out_block_stride, d->out_axis_stride, d->out_block_tab);
          double re = line[2 * k] * d->scale, im = (d->direction > 0 ? -line[2 * k + 1] : line[2 * k + 1]) * d->scale;
          if (f32) { ((float *)out)[2 * o] = (float)re; ((float *)out)[2 * o + 1] = (float)im; }
          else { ((double *)out)[2 * o] = re; ((double *)out)[2 * o + 1] = im; }
        }
      }
  free(line); free(scr); orc_fft_plan_destroy(pl);
  return 0;
}
static void *cb_stream_create(void) { return malloc(8); }
static void cb_stream_destroy(void *s) { free(s); }
static void *cb_event_create(void) { return calloc(1, sizeof(double)); }
static void cb_event_destroy(void *e) { free(e); }
static int cb_event_record(void *e, void *s) { /* everything runs synchronously: an event is a wall-clock stamp */
  (void)s;
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  if (e) *(double *)e = 1e3 * (double)ts.tv_sec + 1e-6 * (double)ts.tv_nsec;
  return 0;
}
static int cb_stream_wait(void *s, void *e) { (void)e; (void)s; return 0; }
static int cb_stream_sync(void *s) { (void)s; return 0; }
static double cb_event_ms(void *a, void *b) { return (a && b) ? *(double *)b - *(double *)a : 0.0; }
static int cb_a2a(void *ctx, int which, int npeers, const int *peer, const void *const *sendp, const size_t *sendbytes,
                  void *const *recvp, const size_t *recvbytes, void *stream) {
  (void)ctx; (void)stream;
  if (!g_a2a_cb) return -1;
  return g_a2a_cb(which, npeers, peer, sendp, sendbytes, recvp, recvbytes);
}
static int cb_memcpy_dd(void *dst, const void *src, size_t bytes, void *s) { (void)s; memmove(dst, src, bytes); return 0; }

static int cb_upload(void *dst, const void *src, size_t bytes) { memcpy(dst, src, bytes); return 0; }

/* direct-store exchange on host memory: several ranks as THREADS of one process see each other's buffers anyway; the
 * test hands the pointers round through a callback (ranks as processes have no shared memory: -1, the library then falls
 * back to the staged exchange).  Flags are plain atomics; a wait spins -- the passes run synchronously, so a rank really
 * blocks here until its peers have stored and signalled. */
typedef int (*peer_cb_t)(int which, int npeers, int self, void *local, size_t bytes, void **peers);
static peer_cb_t g_peer_cb = NULL;
static int cb_peer_open(void *ctx, int which, int npeers, int self, void *local, size_t bytes, void **peers) {
  (void)ctx;
  if (!g_peer_cb) return -1;
  return g_peer_cb(which, npeers, self, local, bytes, peers);
}
static void cb_peer_close(void *ctx, int npeers, int self, void **peers) { (void)ctx; (void)npeers; (void)self; (void)peers; }
static void *cb_flag_alloc(size_t bytes, int host_visible) { (void)host_visible; return calloc(1, bytes); }
static void cb_flag_free(void *p, int host_visible) { (void)host_visible; free(p); }
static int cb_flag_signal(int n, unsigned long long *const *addr, unsigned long long value, void *stream) {
  (void)stream;
  for (int i = 0; i < n; i++) __atomic_store_n(addr[i], value, __ATOMIC_RELEASE);
  return 0;
}
static int cb_flag_wait(int n, unsigned long long *const *addr, unsigned long long value, unsigned long long *status, double timeout_s, void *stream) {
  (void)stream;
  struct timespec t0, t;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  for (int i = 0; i < n; i++)
    while (__atomic_load_n(addr[i], __ATOMIC_ACQUIRE) < value) {
      clock_gettime(CLOCK_MONOTONIC, &t);
      if ((double)(t.tv_sec - t0.tv_sec) + 1e-9 * (double)(t.tv_nsec - t0.tv_nsec) > timeout_s) { if (status) *status = 1; return 0; }
      struct timespec nap = {0, 50000};
      nanosleep(&nap, NULL);
    }
  return 0;
}

static const offt_backend k_cpu_backend = {cb_malloc, cb_free, cb_prepare, cb_pass, cb_stream_create, cb_stream_destroy,
                                           cb_event_create, cb_event_destroy, cb_event_record, cb_stream_wait,
                                           cb_stream_sync, cb_event_ms, cb_a2a, cb_memcpy_dd, cb_upload,
                                           cb_peer_open, cb_peer_close, cb_flag_alloc, cb_flag_free, cb_flag_signal, cb_flag_wait};

const offt_backend *cpu_backend_table(void) { return &k_cpu_backend; }
/* run one descriptor on host arrays (descriptor-level parity tests against the HIP kernels) */
int cpu_backend_run_pass(const offt_pass_desc *d, const void *in, void *out) { return cb_pass(d, in, out, NULL); }
void cpu_backend_set_a2a(a2a_cb_t cb) { g_a2a_cb = cb; }
void cpu_backend_set_peer_open(peer_cb_t cb) { g_peer_cb = cb; }
long cpu_backend_pass_count(void) { return g_pass_count; }
long cpu_backend_alloc_count(void) { return g_alloc_count; }
void cpu_backend_fail_alloc_at(long n_from_now) { g_fail_alloc_at = n_from_now > 0 ? g_alloc_count + n_from_now : -1; }
