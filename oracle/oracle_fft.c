/*
 * oracle_fft.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * 1-D complex forward DFT of arbitrary length, standing in for the FFTW plans
 * the reference creates in setup_p1d (offt-compute.c:329-489):
 * fftw_plan_dft_1d (contiguous) and fftw_plan_many_dft (strided).  FFTW itself
 * is a third-party dependency that is not vendored under the reference; this
 * is the textbook algorithm FFTW implements: mixed-radix Cooley-Tukey
 * decimation in time (radix 4 / 2 butterflies, generic odd radices by direct
 * summation), sign convention exp(-2 pi i n k / N), unnormalised.
 * Twiddles come from one table computed in long double.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "oracle.h"

struct orc_fft_plan {
  int n;
  int nfac;
  int fac[64];
  double *w; /* w[2k], w[2k+1] = cos, -sin of 2 pi k / n */
};

orc_fft_plan *orc_fft_plan_create(int n) {
  orc_fft_plan *p = (orc_fft_plan *)calloc(1, sizeof *p);
  p->n = n;
  int m = n;
  while (m % 4 == 0) { p->fac[p->nfac++] = 4; m /= 4; }
  while (m % 2 == 0) { p->fac[p->nfac++] = 2; m /= 2; }
  for (int f = 3; f * f <= m; f += 2)
    while (m % f == 0) { p->fac[p->nfac++] = f; m /= f; }
  if (m > 1) p->fac[p->nfac++] = m;
  p->w = (double *)malloc(sizeof(double) * 2 * (size_t)n);
  const long double pi = 3.14159265358979323846264338327950288419716939937510L;
  for (int k = 0; k < n; k++) {
    /* reduce to the first octant so the table has the exact symmetries */
    long long oct = (8LL * k) / n, rem = 8LL * k - oct * n;
    long double a = (long double)rem / (long double)n * (pi / 4), c, s;
    long double b = pi / 4 - a;
    switch ((int)(oct & 7)) {
      case 0: c = cosl(a); s = sinl(a); break;
      case 1: c = sinl(b); s = cosl(b); break;
      case 2: c = -sinl(a); s = cosl(a); break;
      case 3: c = -cosl(b); s = sinl(b); break;
      case 4: c = -cosl(a); s = -sinl(a); break;
      case 5: c = -sinl(b); s = -cosl(b); break;
      case 6: c = sinl(a); s = -cosl(a); break;
      default: c = cosl(b); s = -sinl(b); break;
    }
    p->w[2 * k] = (double)c;
    p->w[2 * k + 1] = (double)(-s);
  }
  return p;
}

void orc_fft_plan_destroy(orc_fft_plan *p) {
  if (!p) return;
  free(p->w);
  free(p);
}

/* out[0..n) (contiguous) = DFT of in[0], in[is], in[2 is], ...; ws = N / n */
static void rec(const orc_fft_plan *p, double *out, const double *in, long is, int n, int fi, int ws) {
  if (n == 1) { out[0] = in[0]; out[1] = in[1]; return; }
  const int r = p->fac[fi], m = n / r, N = p->n;
  const double *W = p->w;
  for (int q = 0; q < r; q++) rec(p, out + 2 * (size_t)q * m, in + 2 * q * is, is * r, m, fi + 1, ws * r);
  if (r == 2) {
    for (int k = 0; k < m; k++) {
      double *a = out + 2 * k, *b = out + 2 * (k + m);
      const double wr = W[2 * (k * ws)], wi = W[2 * (k * ws) + 1];
      double br = b[0] * wr - b[1] * wi, bi = b[0] * wi + b[1] * wr;
      double ar = a[0], ai = a[1];
      a[0] = ar + br; a[1] = ai + bi;
      b[0] = ar - br; b[1] = ai - bi;
    }
  } else if (r == 4) {
    for (int k = 0; k < m; k++) {
      double *x0 = out + 2 * k, *x1 = out + 2 * (k + m), *x2 = out + 2 * (k + 2 * m), *x3 = out + 2 * (k + 3 * m);
      const double *w1 = W + 2 * (size_t)(k * ws), *w2 = W + 2 * (size_t)(2 * k * ws), *w3 = W + 2 * (size_t)(3 * k * ws);
      double ar = x0[0], ai = x0[1];
      double br = x1[0] * w1[0] - x1[1] * w1[1], bi = x1[0] * w1[1] + x1[1] * w1[0];
      double cr = x2[0] * w2[0] - x2[1] * w2[1], ci = x2[0] * w2[1] + x2[1] * w2[0];
      double dr = x3[0] * w3[0] - x3[1] * w3[1], di = x3[0] * w3[1] + x3[1] * w3[0];
      double t0r = ar + cr, t0i = ai + ci, t1r = ar - cr, t1i = ai - ci;
      double t2r = br + dr, t2i = bi + di, t3r = br - dr, t3i = bi - di;
      x0[0] = t0r + t2r; x0[1] = t0i + t2i;
      x2[0] = t0r - t2r; x2[1] = t0i - t2i;
      /* -i * t3 = (t3i, -t3r) */
      x1[0] = t1r + t3i; x1[1] = t1i - t3r;
      x3[0] = t1r - t3i; x3[1] = t1i + t3r;
    }
  } else {
    double tr[r], ti[r];
    const int wr_step = N / r;
    for (int k = 0; k < m; k++) {
      for (int q = 0; q < r; q++) {
        const double *x = out + 2 * ((size_t)q * m + k);
        const double *w = W + 2 * (size_t)(((long long)q * k * ws) % N);
        tr[q] = x[0] * w[0] - x[1] * w[1];
        ti[q] = x[0] * w[1] + x[1] * w[0];
      }
      for (int j = 0; j < r; j++) {
        double sr = 0, si = 0;
        for (int q = 0; q < r; q++) {
          const double *w = W + 2 * (size_t)(((long long)q * j % r) * wr_step);
          sr += tr[q] * w[0] - ti[q] * w[1];
          si += tr[q] * w[1] + ti[q] * w[0];
        }
        out[2 * ((size_t)j * m + k)] = sr;
        out[2 * ((size_t)j * m + k) + 1] = si;
      }
    }
  }
}

void orc_fft_execute(const orc_fft_plan *p, double *data, long stride, long dist, int howmany, double *scratch) {
  const int n = p->n;
  double *tmp = scratch, *res = scratch + 2 * (size_t)n;
  for (int h = 0; h < howmany; h++) {
    double *line = data + 2 * (size_t)h * dist;
    if (stride == 1) {
      rec(p, res, line, 1, n, 0, 1);
      memcpy(line, res, sizeof(double) * 2 * (size_t)n);
    } else {
      for (int k = 0; k < n; k++) { tmp[2 * k] = line[2 * k * stride]; tmp[2 * k + 1] = line[2 * k * stride + 1]; }
      rec(p, res, tmp, 1, n, 0, 1);
      for (int k = 0; k < n; k++) { line[2 * k * stride] = res[2 * k]; line[2 * k * stride + 1] = res[2 * k + 1]; }
    }
  }
}
