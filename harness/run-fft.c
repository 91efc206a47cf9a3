/*
 * run-fft.c -- benchmark / smoke harness for the MI355X OFFT library.
 *
 * Counterpart of the reference's run-fft.c for its own algorithm (-a 0): the same
 * command-line letters (run-fft.c:171-231), the same stdout lines ("@ INPUT", "@ FINAL",
 * "t_init", "t_<r>", "t_fin", "t_all", "t_min", "p <rank>: x y z: re im",
 * run-fft.c:235-251, 321, 358-367, 400-407, 438-449, 501) so that existing job scripts
 * (job-test.sh:9-13) and log parsers keep working.  Differences, all additive:
 *   -D        keep the grid in a host calloc()ed array like the reference (staged
 *             through HBM on every execute); default is device-resident data filled
 *             by a kernel
 *   -g        print GFLOP/s (5 E log2 E) and the achieved fraction of the HBM roofline
 *   comparison back-ends (-a 1/2/3: FFTW-MPI, P3DFFT, 2DECOMP) are not part of the path.
 * World: one process per GPU.  Built with -DOFFT_HARNESS_MPI it takes rank/size from
 * MPI and broadcasts the RCCL unique id with MPI_Bcast; without, it is a single rank.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>
#ifdef OFFT_HARNESS_MPI
#include <mpi.h>
#endif
#include "offt_hip.h"

static double now(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static void host_ramp(double *in, const struct _offt_comm *c, int is_r2c) { /* run-fft.c:46-61 */
  for (int x = 0; x < c->isize[0]; x++)
    for (int y = 0; y < c->isize[1]; y++)
      for (int z = 0; z < c->isize[2]; z++) {
        double v = z + 10 * (y + c->istart[1]) + 100 * (x + c->istart[0]);
        if (is_r2c) {
          in[(size_t)z + 2 * (size_t)c->istride[1] * y + 2 * (size_t)c->istride[0] * x] = v;
        } else {
          size_t o = 2 * ((size_t)z + (size_t)c->istride[1] * y + (size_t)c->istride[0] * x);
          in[o] = v;
          in[o + 1] = 0.0;
        }
      }
}

int main(int argc, char **argv) {
  int Nx = 32, Ny = 32, Nz = 32, p1 = -1, reps = 1, p = 1, rank = 0;
  int fftw_level = 0, ah_strategy = 0, max_loop = 0, tuning_mode = 0, is_W0 = 0, extrapolation_window = 0;
  int is_oned = 0, is_a2a = 0, is_equalxy = 0, is_notest = 0, fft_alg = 0, verbose = 0, is_r2c = 0;
  int host_data = 0, gflops = 0, rc = 0;
  unsigned fftw_flag = FFTW_ESTIMATE;
#ifndef OFFT_HARNESS_MPI
  /* no MPI: a launcher (tools/launch.py, torchrun, srun ...) may still start one process per GPU and say so in the
   * environment: RANK / WORLD_SIZE / LOCAL_RANK as torch.distributed.run sets them, plus OFFT_ID_FILE, a path on a file
   * system all ranks see -- rank 0 writes the 128-byte RCCL id there, the others wait for it */
  if (getenv("WORLD_SIZE") && atoi(getenv("WORLD_SIZE")) > 1) {
    p = atoi(getenv("WORLD_SIZE"));
    rank = getenv("RANK") ? atoi(getenv("RANK")) : 0;
    const int dev = getenv("LOCAL_RANK") ? atoi(getenv("LOCAL_RANK")) : rank;
    const char *idf = getenv("OFFT_ID_FILE");
    char id[OFFT_HIP_UNIQUE_ID_BYTES];
    if (!idf) { fprintf(stderr, "WORLD_SIZE > 1 needs OFFT_ID_FILE (or build the harness with MPI=1)\n"); return 2; }
    /* The file is bound to THIS run: the id is followed by a nonce every rank of the run knows (OFFT_RUN_NONCE, else the
     * launcher's MASTER_PORT / TORCHELASTIC_RUN_ID), and a reader accepts only a file that carries its own nonce -- a file
     * left behind by an earlier run at the same path is ignored (the ranks would otherwise read a stale id at once and
     * block in ncclCommInitRank, which has no time-out).  Without any of the three variables the path itself must be
     * fresh per run (tools/launch.py makes one in a new temporary directory). */
    char nonce[64];
    memset(nonce, 0, sizeof nonce);
    {
      const char *nv = getenv("OFFT_RUN_NONCE");
      if (!nv) nv = getenv("MASTER_PORT");
      if (!nv) nv = getenv("TORCHELASTIC_RUN_ID");
      if (nv) snprintf(nonce, sizeof nonce, "%s", nv);
      else if (rank == 0) fprintf(stderr, "run-fft: no OFFT_RUN_NONCE / MASTER_PORT: %s must not exist from an earlier run\n", idf);
    }
    if (rank == 0) {
      char tmp[4096];
      snprintf(tmp, sizeof tmp, "%s.tmp", idf);
      if (offt_hip_get_unique_id(id)) return 2;
      FILE *f = fopen(tmp, "wb");
      if (!f || fwrite(id, 1, sizeof id, f) != sizeof id || fwrite(nonce, 1, sizeof nonce, f) != sizeof nonce) { fprintf(stderr, "cannot write %s\n", tmp); return 2; }
      fclose(f);
      if (rename(tmp, idf)) { fprintf(stderr, "cannot publish %s\n", idf); return 2; }
    } else {
      int ok = 0;
      for (int tries = 0; tries < 6000 && !ok; tries++) { /* up to 60 s */
        char got[64];
        FILE *f = fopen(idf, "rb");
        if (f) {
          ok = fread(id, 1, sizeof id, f) == sizeof id && fread(got, 1, sizeof got, f) == sizeof got && !memcmp(got, nonce, sizeof got);
          fclose(f);
        }
        if (!ok) usleep(10000);
      }
      if (!ok) { fprintf(stderr, "rank %d: no RCCL id of this run (nonce \"%s\") in %s\n", rank, nonce, idf); return 2; }
    }
    if (offt_hip_set_world(rank, p, id, dev)) return 3;
  }
#endif
#ifdef OFFT_HARNESS_MPI
  MPI_Init(&argc, &argv);
  MPI_Comm_size(MPI_COMM_WORLD, &p);
  MPI_Comm_rank(MPI_COMM_WORLD, &rank);
  {
    char id[OFFT_HIP_UNIQUE_ID_BYTES];
    memset(id, 0, sizeof id);
    if (rank == 0 && p > 1 && offt_hip_get_unique_id(id)) MPI_Abort(MPI_COMM_WORLD, 2);
    MPI_Bcast(id, sizeof id, MPI_BYTE, 0, MPI_COMM_WORLD);
    const char *lr = getenv("OFFT_LOCAL_RANK");
    int dev = lr ? atoi(lr) : rank % (getenv("OFFT_GPUS_PER_NODE") ? atoi(getenv("OFFT_GPUS_PER_NODE")) : 8);
    if (offt_hip_set_world(rank, p, p > 1 ? id : NULL, dev)) MPI_Abort(MPI_COMM_WORLD, 3);
  }
#endif
  struct _offt_params *cp = (struct _offt_params *)malloc(sizeof *cp);
  for (int i = 0; i < PARAM_COUNT; i++) cp->v[i] = -1;
  static const struct { char opt; int idx; } pmap[] = {
      {'T', _T1_}, {'W', _W1_}, {'t', _T2_}, {'w', _W2_}, {'Y', _Ry_}, {'P', _Px1_}, {'p', _Py1_}, {'X', _Px2_},
      {'Z', _Pz2_}, {'U', _Ux1_}, {'u', _Uz1_}, {'y', _Uy2_}, {'z', _Uz2_}, {'E', _Fz_}, {'F', _Fy1_}, {'f', _Fy2_},
      {'G', _Fx_}, {'H', _FP1_}, {'h', _FP2_}, {'I', _FU1_}, {'i', _FU2_}, {'V', _V_}, {'S', _S_}};
  int c;
  while ((c = getopt(argc, argv, "N:n:L:va:Rr:m:A:O:Qobecs:l:d:T:W:t:w:P:p:X:Y:Z:U:u:y:z:E:F:f:G:H:h:I:i:V:S:Dg")) >= 0) {
    switch (c) {
      case 'N': Nx = atoi(optarg); break;
      case 'n': Ny = atoi(optarg); break;
      case 'L': Nz = atoi(optarg); break;
      case 'v': verbose = 1; break;
      case 'a': fft_alg = atoi(optarg); break;
      case 'R': is_r2c = 1; break;
      case 'r': reps = atoi(optarg); break;
      case 'm': {
        static const unsigned lv[4] = {FFTW_ESTIMATE, FFTW_MEASURE, FFTW_PATIENT, FFTW_EXHAUSTIVE};
        fftw_level = atoi(optarg); fftw_flag = lv[fftw_level % 4];
      } break;
      case 'o': is_oned = 1; break;
      case 'b': is_a2a = 1; break;
      case 'e': is_equalxy = 1; break;
      case 'c': is_notest = 1; break;
      case 's': ah_strategy = atoi(optarg); break;
      case 'l': max_loop = atoi(optarg); break;
      case 'O': tuning_mode = atoi(optarg); break;
      case 'Q': is_W0 = 1; break;
      case 'A': extrapolation_window = atoi(optarg); break;
      case 'd': p1 = cp->v[_P1_] = atoi(optarg); break;
      case 'D': host_data = 1; break;
      case 'g': gflops = 1; break;
      default:
        for (size_t k = 0; k < sizeof pmap / sizeof pmap[0]; k++)
          if (pmap[k].opt == c) cp->v[pmap[k].idx] = atoi(optarg);
    }
  }
  if (!rank) {
    for (int a = 0; a < argc; a++) printf("%s ", argv[a]);
    printf("\nNx %d Ny %d Nz %d p %d p1 %d r %d a %d m %d o %d b %d e %d c %d s %d l %d O %d Q %d A %d\n", Nx, Ny, Nz, p, p1,
           reps, fft_alg, fftw_level, is_oned, is_a2a, is_equalxy, is_notest, ah_strategy, max_loop, tuning_mode, is_W0,
           extrapolation_window);
    printf("@ INPUT "); print_params(cp->v);
  }
  if (fft_alg != 0) {
    if (!rank) printf("only -a 0 (OFFT) is part of this library; FFTW-MPI / P3DFFT / 2DECOMP back-ends are not built\nt_min 999999999.00000\n");
    goto finish;
  }
  if (p1 == -1 && max_loop == 0) {
    /* run-fft.c:290-293: without -d the reference only SIZES its buffer for p1 = p and leaves custom_params->v[_P1_] at -1, so
     * the library picks its default mesh (the largest divisor of p that is <= sqrt(p)); the buffer here is sized after init
     * from the plan itself (offt_hip_local_bytes), for whichever mesh was chosen */
    p1 = p;
    if (!rank) printf("set p1 = %d\n", p1);
  }
  double t = -now();
  struct _offt_plan *po = offt_3d_init(Nx, Ny, Nz, NULL, NULL, is_r2c, (int)fftw_flag, is_oned, is_a2a, is_equalxy, is_notest,
                                       ah_strategy, max_loop, tuning_mode, is_W0, extrapolation_window, cp);
  if (!po) { fprintf(stderr, "offt_3d_init failed: %s\n", offt_hip_last_error()); if (!rank) printf("t_min 999999999.00000\n"); rc = 3; goto finish; }
  p1 = po->params->v[_P1_];
  long long bytes = offt_hip_local_bytes(po);
  if (!rank) printf("allocate memory for total # elements %lld\n", bytes / 16);
  double *out = host_data ? (double *)calloc((size_t)bytes, 1) : (double *)offt_hip_malloc(bytes);
  if (!out) { fprintf(stderr, "allocation of %lld bytes failed\n", bytes); if (!rank) printf("t_min 999999999.00000\n"); rc = 4; offt_3d_fin(po); goto finish; }
  if (!rank) { printf("@ FINAL "); print_params(po->params->v); }
  t += now();
  if (!rank) printf("t_init %.5f %.5f %.5f %.5f\n", po->t_init[INIT_ALL], po->t_init[INIT_FFTW], po->t_init[INIT_AH], po->t_init[INIT_BUFFER]);

  double t_min = 999999999.0, t_min_arr[GES];
  memset(t_min_arr, 0, sizeof t_min_arr);
  for (int r = 0; r < reps; r++) {
    if (host_data) host_ramp(out, po->comm, is_r2c); else offt_hip_fill_input(po, out, 0);
#ifdef OFFT_HARNESS_MPI
    MPI_Barrier(MPI_COMM_WORLD);
#else
    if (p > 1 && offt_hip_world_count() != p) { fprintf(stderr, "rank %d: not all %d ranks answered\n", rank, p); rc = 6; goto finish_plan; }
#endif
    double t0 = now();
    offt_3d_execute(po, out, out, 0);
    double tc = now() - t0;
    if (po->t[ALL] >= 99999999.0) {
      /* the library's only failure channel (offt-compute.c:3881): a failed launch, an RCCL error, a length no kernel takes.
       * Say so and fail the run instead of printing timings of an untransformed buffer. */
      fprintf(stderr, "offt_3d_execute failed on rank %d: %s\n", rank, offt_hip_last_error());
      printf("t_min 999999999.00000\n");
      fflush(stdout);
#ifdef OFFT_HARNESS_MPI
      MPI_Abort(MPI_COMM_WORLD, 5);
#endif
      rc = 5;
      goto finish_plan;
    }
    t += tc;
    if (!rank) { printf("t_%d ", r); po->t[ALL] = tc; offt_print_time(po->t); }
    if (tc < t_min) { t_min = tc; memcpy(t_min_arr, po->t, sizeof t_min_arr); }
  }
  double spot[4][2];
  int nspot = 0;
  if (verbose && rank == 0) { /* run-fft.c:452-503: out[x=0, y=0, z=0..3] through ostride */
    int MM3 = (is_r2c ? Nz / 2 + 1 : Nz) / (p / p1), zEnd = 4 > MM3 ? MM3 : 4;
    for (int z = 0; z < zEnd; z++) {
      size_t o = 2 * (size_t)z * po->comm->ostride[2];
      if (host_data) { spot[z][0] = out[o]; spot[z][1] = out[o + 1]; }
      else offt_hip_memcpy_d2h(spot[z], (char *)out + o * 8, 16);
      nspot++;
    }
  }
  double tf = -now();
  int MM3p = (is_r2c ? Nz / 2 + 1 : Nz) / (p / p1), MM4p = Ny / p1;
  offt_3d_fin(po);
  tf += now();
  t += tf;
  if (!rank) {
    printf("t_fin %.5f\nt_all %.5f\nt_min ", tf, t);
    offt_print_time(t_min_arr);
    if (gflops) {
      double E = (double)Nx * Ny * Nz, fl = 5.0 * E * log2(E), dev = t_min_arr[FFTz] + t_min_arr[FFTy1] + t_min_arr[FFTx] + t_min_arr[PACK1];
      double bytes = 6.0 * 16.0 * E;
      const char *model = "6*16*E/p bytes";
      if (is_r2c) { /* real rows in, Nz/2+1 complex out: half the flops of the z pass, y and x passes on the half spectrum */
        const double Eh = (double)Nx * Ny * (Nz / 2 + 1);
        fl = 2.5 * E * log2((double)Nz) + 5.0 * Eh * log2((double)Nx * Ny);
        bytes = 8.0 * E + 16.0 * Eh + 4.0 * 16.0 * Eh;
        model = "r2c: (8*E + 5*16*Eh)/p bytes, Eh = Nx*Ny*(Nz/2+1)";
      }
      printf("gflops_wall %.1f gflops_device %.1f hbm_roofline_frac %.4f (%s, 8 TB/s)\n", fl / t_min / 1e9,
             dev > 0 ? fl / dev / 1e9 : 0.0, dev > 0 ? bytes / p / dev / 8e12 : 0.0, model);
    }
    if (verbose) {
      printf("p1 %d p2 %d MM3 %d MM4 %d\n", p1, p / p1, MM3p, MM4p);
      for (int z = 0; z < nspot; z++) printf("p %d: %d %d %d: %.5f %.5f\n", rank, 0, 0, z, spot[z][0], spot[z][1]);
    }
  }
  goto finish_out;
finish_plan:
  offt_3d_fin(po);
finish_out:
  if (host_data) free(out); else offt_hip_free(out);
finish:
  free(cp);
#ifdef OFFT_HARNESS_MPI
  offt_hip_finalize_world();
  MPI_Finalize();
#else
  if (p > 1) offt_hip_finalize_world();
#endif
  return rc;
}
