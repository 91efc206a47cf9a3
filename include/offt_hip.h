/*
 * offt_hip.h -- MI355X-specific extensions around the offt.h boundary.
 *
 * The reference binds ranks through MPI_COMM_WORLD inside offt_3d_init
 * (offt-compute.c:3315-3316) and exchanges tiles with MPI_Ialltoall
 * (offt-compute.c:835-900).  Here one process drives one GPU and the exchange
 * is an RCCL all-to-all over xGMI, so the world (rank, size, RCCL unique id) is
 * handed in through this C ABI by whatever launcher is in use (torchrun +
 * torch.distributed store, MPI_Bcast, a file): plain pointers and sizes only.
 *
 * Everything in this header is an extension: callers that only use offt.h get
 * the reference's behaviour (single process == world of one rank).
 */
#ifndef OFFT_HIP_INCLUDE
#define OFFT_HIP_INCLUDE

#include "offt.h"

#ifdef __cplusplus
extern "C" {
#endif

#define OFFT_HIP_UNIQUE_ID_BYTES 128

/* ---- world bootstrap (replaces MPI_Comm_size/rank, offt-compute.c:3315) ---- */
/* rank 0 creates the RCCL unique id; the launcher ships the 128 bytes around. */
int offt_hip_get_unique_id(void *id128);
/* declare this process as `rank` of `size`; id128 may be NULL when size == 1.
 * Must precede offt_3d_init.  Binds the process to `device` (hipSetDevice).    */
int offt_hip_set_world(int rank, int size, const void *id128, int device);
/* tear the world down (destroys the RCCL communicator)                         */
int offt_hip_finalize_world(void);
int offt_hip_world_rank(void);
int offt_hip_world_size(void);
/* number of ranks that answer on the world communicator (an ncclAllReduce of ones); -1 on failure.  Collective.    */
int offt_hip_world_count(void);
/* xGMI link probe on the world communicator: grouped ncclSend/ncclRecv of `bytes` per peer, `reps` repetitions after
 * one warm-up.  mode 0 = all-to-all among all ranks, mode 1 = ring shift by `shift` (one link per direction).
 * Returns this rank's seconds per repetition, negative on failure.  Collective.                                     */
double offt_hip_link_probe(int mode, int shift, long long bytes, int reps);

/* ---- plan extensions -------------------------------------------------------- */
#define OFFT_HIP_F64 0
#define OFFT_HIP_F32 1
/* Same arguments as offt_3d_init plus the arithmetic type (the reference is
 * double only, Appendix F of SURVEY.md).  With OFFT_HIP_F32 `in`/`out` point at
 * interleaved float pairs.                                                      */
struct _offt_plan *offt_3d_init_ex(int Nx, int Ny, int Nz, void *in, void *out, int is_r2c,
                                   int fftw_flag, int is_oned, int is_a2a, int is_equalxy,
                                   int is_notest, int ah_strategy, int max_loop, int tuning_mode,
                                   int is_W0, int extrapolation_window,
                                   struct _offt_params *custom_params, int precision);
/* direction: -1 forward (what offt_3d_execute does), +1 inverse (unnormalised,
 * FFTW_BACKWARD convention).  The inverse consumes the forward's OUTPUT layout
 * (ostart/osize/ostride) and produces the INPUT layout (istart/isize/istride). */
void offt_3d_execute_dir(struct _offt_plan *po, void *in, void *out, int direction);
/* run on a caller-owned hipStream_t (NULL = the plan's own stream)             */
void offt_hip_set_stream(struct _offt_plan *po, void *stream);
/* 0: offt_3d_execute returns after the GPU finished (timers valid, reference
 *    behaviour); 1: returns after enqueueing (caller synchronises the stream);
 *    no timing events are recorded then, so back-to-back small transforms pay
 *    for the kernel launches only.                                              */
void offt_hip_set_async(struct _offt_plan *po, int async);
/* asynchronous mode: wait for everything enqueued on the plan so far.  Like the end of a synchronous execute the wait
 * is bounded (OFFT_EXEC_TIMEOUT) and polls the RCCL communicators for asynchronous errors; returns 0, or -1 with
 * t[ALL] = 99999999 and the text in offt_hip_last_error().                                                        */
int offt_hip_wait(struct _offt_plan *po);
/* plan-level options.  Each has an environment variable of the same meaning that is read ONCE, by offt_3d_init, as the
 * default; after that a plan's behaviour does not depend on the process environment.  Options marked (collective) rebuild
 * the plan's exchange buffers: every rank of the world calls them with the same value.  0 on success. */
#define OFFT_HIP_OPT_ZGROUP_MIB 0      /* single rank: MiB per group of the alternating y / x launches; 0 off, -1 library rule (OFFT_ZGROUP_MIB) */
#define OFFT_HIP_OPT_ZGROUP_STREAMS 1  /* ... 2 = consumer launches on a second stream (OFFT_ZGROUP_STREAMS) */
#define OFFT_HIP_OPT_SLAB_CHUNK_MIB 2  /* (collective) slab schedule: largest z-chunk in MiB (OFFT_SLAB_CHUNK_MIB) */
#define OFFT_HIP_OPT_COMM_STREAMS 3    /* pencil schedule: 2 = row and column exchanges on two streams (OFFT_COMM_STREAMS) */
#define OFFT_HIP_OPT_F32_PAIRS 4       /* single precision: 0 = never the column-pair kernels (OFFT_F32_PAIRS) */
#define OFFT_HIP_OPT_K1_STREAMS 5      /* slab schedule: 2 = FFTz launches of consecutive x-tiles on two streams (OFFT_K1_STREAMS) */
#define OFFT_HIP_OPT_SELF_BYPASS 6     /* (collective) 0 = a rank's own block goes through the exchange like any other (OFFT_SELF_BYPASS) */
#define OFFT_HIP_OPT_MIN_MSG 7         /* (collective) bytes a per-peer message is merged up to (OFFT_MIN_MSG) */
#define OFFT_HIP_OPT_EXEC_TIMEOUT_S 8  /* bound of the final wait of a multi-rank execute (OFFT_EXEC_TIMEOUT) */
#define OFFT_HIP_OPT_P2P_TIMEOUT_S 9   /* bound of one flag wait of the direct-store exchange (OFFT_P2P_TIMEOUT) */
int offt_hip_set_option(struct _offt_plan *po, int option, long long value);
/* (Launchers that want an exchange-only / compute-only split of a multi-rank execute link the DIAGNOSTICS build,
 *  tools/liboffthip_diag.so = the product compiled with -DOFFT_BENCH_DIAGNOSTICS, which adds
 *  void offt_hip_set_debug_skip(po, mask): mask 1 leaves out the FFT passes, 2 the exchanges.  The product library does
 *  not contain it.) */
long long offt_hip_get_option(const struct _offt_plan *po, int option);
/* exchange of a multi-rank plan.  STAGED (default): the packing passes fill a send volume, grouped RCCL send/recv moves
 * it (the reference's pack + MPI_Ialltoall, offt-compute.c:1084-1109, 835-881); a rank's own block bypasses the exchange.
 * DIRECT: the packing passes store every block straight into its owner's receive volume (peer memory mapped through
 * hipIpc at plan time) and 64-bit flags replace the exchange -- no send volume, no copy kernels.  Collective: all ranks
 * call it with the same mode.  Returns the mode in use afterwards (DIRECT falls back to STAGED on all ranks together where
 * peer memory cannot be mapped), -1 on failure.  OFFT_EXCHANGE=p2p in the environment is the init-time default. */
#define OFFT_HIP_EXCHANGE_STAGED 0
#define OFFT_HIP_EXCHANGE_DIRECT 1
int offt_hip_set_exchange(struct _offt_plan *po, int mode);
int offt_hip_get_exchange(const struct _offt_plan *po);
/* select a static-sweep kernel variant per axis (0 = x, 1 = y, 2 = z); -1 default */
void offt_hip_set_variant(struct _offt_plan *po, int axis, int variant);
/* multiply the result by `scale` in the store of the last pass (1.0 = the reference's
 * unnormalised transform); free, it rides on the kernel's stores                  */
void offt_hip_set_output_scale(struct _offt_plan *po, double scale);
/* bytes the caller must allocate for in/out on this rank (run-fft.c:294-304)   */
long long offt_hip_local_bytes(const struct _offt_plan *po);
/* device time of the last execute in seconds, from hipEvents on the plan stream */
double offt_hip_last_device_seconds(const struct _offt_plan *po);
/* per-pass device seconds of the last execute: z, y, x passes (0 if fused away).  When two of the passes ran as
 * alternating launches over groups of planes (single rank: the y and x passes of the z-y-x layout), only their SUM was
 * measured and each of the two slots holds half of it: offt_hip_last_passes_paired() returns 0, or the two slots as
 * bits (1 = z, 2 = y, 4 = x). */
void offt_hip_last_pass_seconds(const struct _offt_plan *po, double t[3]);
int offt_hip_last_passes_paired(const struct _offt_plan *po);
/* last error text ("" if none); errors also go to stderr, like the reference's
 * printf-only error handling (offt-compute.c:702-704)                           */
const char *offt_hip_last_error(void);

/* ---- device helpers for harnesses and tests ---------------------------------- */
void *offt_hip_malloc(long long bytes);
void offt_hip_free(void *p);
int offt_hip_memcpy_h2d(void *dst, const void *src, long long bytes);
int offt_hip_memcpy_d2h(void *dst, const void *src, long long bytes);
int offt_hip_device_synchronize(void);
/* fill this rank's input block (istart/isize/istride) on the device:
 * kind 0 = harness ramp re = z + 10 y + 100 x (run-fft.c:46-61), 1 = seeded
 * position hash in [-1,1)                                */
int offt_hip_fill_input(struct _offt_plan *po, void *buf, int kind);

#ifdef __cplusplus
}
#endif
#endif
