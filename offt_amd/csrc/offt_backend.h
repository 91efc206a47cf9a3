/*
 * offt_backend.h -- the handful of device operations the host pipeline needs,
 * as a table of C function pointers.  The library ships exactly ONE table:
 * HIP kernels + HIP streams/events + RCCL (k_hip_backend in offt_host.c).
 *
 * offt_hip_test_set_backend() (compiled only with -DOFFT_TEST_SEAMS, i.e. into
 * tests/liboffthip_test.so, never into the product) exists so that the CPU-only test-suite can run
 * the real host logic (decomposition, pass descriptors, tile ring, exchange
 * schedule) in world_size-2 `gloo` processes with a descriptor interpreter
 * that lives under tests/ -- the library itself contains no CPU FFT and never
 * installs another table on its own.
 */
#ifndef OFFT_BACKEND_H
#define OFFT_BACKEND_H
#include <stddef.h>
#include "offt_hipk.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct offt_backend {
  void *(*dmalloc)(size_t bytes);
  void (*dfree)(void *p);
  int (*prepare)(int n, int precision);
  int (*pass)(const offt_pass_desc *d, const void *in, void *out, void *stream);
  void *(*stream_create)(void);
  void (*stream_destroy)(void *s);
  void *(*event_create)(void);
  void (*event_destroy)(void *e);
  int (*event_record)(void *e, void *s);
  int (*stream_wait)(void *s, void *e);
  int (*stream_sync)(void *s);
  double (*event_ms)(void *a, void *b);
  /* all-to-all of one tile: which = 1 (row group, p2 peers) or 2 (column group,
   * p1 peers); peer a sends sendbytes[a] from sendp[a], receives recvbytes[a]
   * into recvp[a]; peer index == rank inside the group */
  int (*a2a)(void *ctx, int which, int npeers, const int *peer, const void *const *sendp,
             const size_t *sendbytes, void *const *recvp, const size_t *recvbytes, void *stream);
  int (*memcpy_dd)(void *dst, const void *src, size_t bytes, void *stream);
  /* host -> device copy of a small table, synchronous (plan time) */
  int (*upload)(void *dst, const void *src, size_t bytes);
  /* ---- direct-store exchange (offt_host.c, "p2p") ----
   * peer_open: collective over exchange group `which` (1 row / 2 column / 0 world) of npeers members, this rank being
   * member `self`: every member offers the allocation `local` and gets in peers[a] an address valid HERE for member a's
   * (peers[self] = local).  Non-zero: this backend cannot map peer memory (the plan falls back to the staged exchange). */
  int (*peer_open)(void *ctx, int which, int npeers, int self, void *local, size_t bytes, void **peers);
  void (*peer_close)(void *ctx, int npeers, int self, void **peers);
  /* zeroed 64-bit words that peers write and this rank polls / a status word the host can read while a kernel may still write it */
  void *(*flag_alloc)(size_t bytes, int host_visible);
  void (*flag_free)(void *p, int host_visible);
  /* offt_hipk_flag_signal / offt_hipk_flag_wait (offt_hipk.h) */
  int (*flag_signal)(int n, unsigned long long *const *addr, unsigned long long value, void *stream);
  int (*flag_wait)(int n, unsigned long long *const *addr, unsigned long long value, unsigned long long *status, double timeout_s, void *stream);
} offt_backend;

void offt_hip_test_set_backend(const offt_backend *b, int rank, int size);

/* keep the HIP backend but route the exchange through `fn` (called with the comm stream
 * drained; device pointers): several test ranks can then share one GPU */
typedef int (*offt_test_transport_fn)(int which, int npeers, const int *peer, const void *const *sendp,
                                      const size_t *sendbytes, void *const *recvp, const size_t *recvbytes);
void offt_hip_test_set_transport(offt_test_transport_fn fn, int rank, int size);
/* ... the same, but ASYNCHRONOUS: `fn` is called without draining the stream and gets it as its last argument; it has to
 * enqueue the copies itself (ordered by events between the ranks' streams), so that the schedules' own event edges between
 * compute and comm streams are all that orders kernels and exchanges -- as with RCCL.  Set after offt_hip_test_set_transport. */
typedef int (*offt_test_transport_async_fn)(int which, int npeers, const int *peer, const void *const *sendp, const size_t *sendbytes,
                                            void *const *recvp, const size_t *recvbytes, void *stream);
void offt_hip_test_set_transport_async(offt_test_transport_async_fn fn);
/* several ranks as threads of ONE process (one GPU, or the CPU backend): hipIpc cannot open a handle in the process that
 * made it, so peer_open goes through `fn` (same arguments as offt_backend::peer_open without ctx), which hands out the
 * other threads' pointers; `hook` is called before every wait the direct-store schedule enqueues -- a barrier among the
 * rank threads there guarantees that every signal a wait depends on is already enqueued (streams of one process may
 * share a hardware queue, where a wait kernel ahead of the signal it waits for would never end). */
typedef int (*offt_test_peer_open_fn)(int which, int npeers, int self, void *local, size_t bytes, void **peers);
typedef void (*offt_test_hook_fn)(void);
void offt_hip_test_set_p2p(offt_test_peer_open_fn fn, offt_test_hook_fn hook);
/* p1 of the plan whose exchange is in progress on this thread: with it a transport maps (which, group member) to a
 * world rank -- which 1 = row group (rank_x * p2 + member), 2 = column group (member * p2 + rank_y), 0 = world   */
int offt_hip_test_current_p1(void);

#ifdef __cplusplus
}
#endif
#endif
