/* TEST INFRASTRUCTURE (oracle) -- not part of the product.  See fft_ref.h.
 *
 * Mixed strategy: iterative radix-2 Stockham for powers of two (twiddles from a
 * table built with cos/sin in double), direct O(n^2) DFT otherwise (test sizes
 * such as 10 or 12 only).  Real transforms go through the full complex transform
 * of the zero-imaginary signal -- simple and obviously right; this is a checker,
 * not a fast library.  Every golden fixture was written through that path.
 *
 * FFT_REF_FAST=1 (environment; bench.py's cpu_baseline leg sets it, nothing else does)
 * sends real transforms of power-of-two length through the usual half-length complex
 * transform + split / merge step with the plan's own work arrays instead -- no
 * allocation per line, half the butterflies -- so that the host baseline times the
 * reference's Fortran and not the checker's way of doing a real transform.  Same
 * transform, rounding differs in the last bits (tests/test_oracle_vs_reference.py).
 */
#include "fft_ref.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846264338327950288
#endif

struct fft_ref_plan {
  int n;
  int pow2;
  int fast;     /* FFT_REF_FAST: half-length real transforms */
  double *tw;   /* n complex twiddles exp(-2 pi i k / n) */
  double *wa;   /* 2n doubles work */
  double *wb;   /* 2n doubles work */
};

fft_ref_plan *fft_ref_plan_create(int n) {
  fft_ref_plan *p = (fft_ref_plan *)calloc(1, sizeof(*p));
  p->n = n;
  p->pow2 = (n > 0) && ((n & (n - 1)) == 0);
  {
    const char *e = getenv("FFT_REF_FAST");
    p->fast = e && atoi(e) != 0 && p->pow2 && n >= 4;
  }
  p->tw = (double *)malloc(sizeof(double) * 2 * (size_t)n);
  p->wa = (double *)malloc(sizeof(double) * 2 * (size_t)n);
  p->wb = (double *)malloc(sizeof(double) * 2 * (size_t)n);
  for (int k = 0; k < n; ++k) {
    double a = -2.0 * M_PI * (double)k / (double)n;
    p->tw[2 * k] = cos(a);
    p->tw[2 * k + 1] = sin(a);
  }
  return p;
}

void fft_ref_plan_destroy(fft_ref_plan *p) {
  if (!p) return;
  free(p->tw); free(p->wa); free(p->wb); free(p);
}

/* Stockham autosort radix-2 of n complex (n a power of two dividing p->n: its twiddles are every (p->n / n)-th of the
 * plan's): x -> result returned in either x or y */
static double *stockham2n(const fft_ref_plan *p, int n, double *x, double *y, int sign) {
  const int ts = p->n / n;
  int l = n / 2, m = 1;
  while (l >= 1) {
    /* butterflies: for j in [0,l), k in [0,m) */
    for (int j = 0; j < l; ++j) {
      /* twiddle w = exp(sign * 2 pi i j / (2l)) = tw[j * (n/(2l))] (conj for +) */
      int ti = j * (n / (2 * l)) * ts;
      double wr = p->tw[2 * ti], wi = p->tw[2 * ti + 1];
      if (sign > 0) wi = -wi;
      for (int k = 0; k < m; ++k) {
        const double *c0 = x + 2 * (k + j * m);
        const double *c1 = x + 2 * (k + j * m + l * m);
        double ar = c0[0], ai = c0[1], br = c1[0], bi = c1[1];
        double *d0 = y + 2 * (k + 2 * j * m);
        double *d1 = y + 2 * (k + 2 * j * m + m);
        d0[0] = ar + br; d0[1] = ai + bi;
        double tr = ar - br, ti2 = ai - bi;
        d1[0] = tr * wr - ti2 * wi;
        d1[1] = tr * wi + ti2 * wr;
      }
    }
    double *t = x; x = y; y = t;
    l /= 2; m *= 2;
  }
  return x;
}
static double *stockham2(const fft_ref_plan *p, double *x, double *y, int sign) { return stockham2n(p, p->n, x, y, sign); }

void fft_ref_c2c(fft_ref_plan *p, const double *in, double *out, int sign) {
  const int n = p->n;
  if (p->pow2 && n > 1) {
    memcpy(p->wa, in, sizeof(double) * 2 * (size_t)n);
    double *r = stockham2(p, p->wa, p->wb, sign);
    memcpy(out, r, sizeof(double) * 2 * (size_t)n);
    return;
  }
  for (int k = 0; k < n; ++k) {
    double sr = 0.0, si = 0.0;
    for (int j = 0; j < n; ++j) {
      int ti = (int)(((long long)j * k) % n);
      double wr = p->tw[2 * ti], wi = p->tw[2 * ti + 1];
      if (sign > 0) wi = -wi;
      sr += in[2 * j] * wr - in[2 * j + 1] * wi;
      si += in[2 * j] * wi + in[2 * j + 1] * wr;
    }
    out[2 * k] = sr; out[2 * k + 1] = si;
  }
}

void fft_ref_r2c(fft_ref_plan *p, const double *in, double *out) {
  const int n = p->n;
  if (p->fast) {
    /* z[j] = x[2j] + i x[2j+1]; Z = FFT_m(z), m = n/2; X[k] = E[k] + W_n^k O[k] with E = (Z[k] + conj Z[m-k]) / 2,
     * O = (Z[k] - conj Z[m-k]) / (2 i) */
    const int m = n / 2;
    memcpy(p->wa, in, sizeof(double) * (size_t)n);
    const double *Z = stockham2n(p, m, p->wa, p->wb, -1);
    for (int k = 0; k <= m; ++k) {
      const int k1 = k == m ? 0 : k, k2 = k == 0 ? 0 : m - k;
      const double ar = Z[2 * k1], ai = Z[2 * k1 + 1], br = Z[2 * k2], bi = -Z[2 * k2 + 1];
      const double er = 0.5 * (ar + br), ei = 0.5 * (ai + bi);
      const double dr = 0.5 * (ar - br), di = 0.5 * (ai - bi);      /* (Z - conj Z') / 2; O = that / i = (di, -dr) */
      const double orr = di, oi = -dr;
      const double wr = p->tw[2 * k], wi = p->tw[2 * k + 1];
      out[2 * k] = er + (wr * orr - wi * oi);
      out[2 * k + 1] = ei + (wr * oi + wi * orr);
    }
    return;
  }
  double *z = (double *)malloc(sizeof(double) * 4 * (size_t)n);
  double *zo = z + 2 * (size_t)n;
  for (int j = 0; j < n; ++j) { z[2 * j] = in[j]; z[2 * j + 1] = 0.0; }
  fft_ref_c2c(p, z, zo, -1);
  for (int k = 0; k <= n / 2; ++k) { out[2 * k] = zo[2 * k]; out[2 * k + 1] = zo[2 * k + 1]; }
  free(z);
}

void fft_ref_c2r(fft_ref_plan *p, const double *in, double *out) {
  const int n = p->n;
  if (p->fast) {
    /* x[2j] + i x[2j+1] = IDFT_m(A + i B), A[k] = X[k] + conj X[m-k], B[k] = (X[k] - conj X[m-k]) conj(W_n^k); unnormalised.
     * (FFTW's c2r ignores the imaginary parts of the DC and Nyquist bins) */
    const int m = n / 2;
    for (int k = 0; k < m; ++k) {
      double ar = in[2 * k], ai = (k == 0) ? 0.0 : in[2 * k + 1];
      double br = in[2 * (m - k)], bi = (k == 0) ? 0.0 : -in[2 * (m - k) + 1];
      const double sr = ar + br, si = ai + bi, dr = ar - br, di = ai - bi;
      const double wr = p->tw[2 * k], wi = -p->tw[2 * k + 1];
      const double Br = dr * wr - di * wi, Bi = dr * wi + di * wr;
      p->wa[2 * k] = sr - Bi;
      p->wa[2 * k + 1] = si + Br;
    }
    const double *z = stockham2n(p, m, p->wa, p->wb, +1);
    memcpy(out, z, sizeof(double) * (size_t)n);
    return;
  }
  double *z = (double *)malloc(sizeof(double) * 4 * (size_t)n);
  double *zo = z + 2 * (size_t)n;
  /* FFTW c2r ignores the imaginary parts of the DC and Nyquist bins */
  for (int k = 0; k <= n / 2; ++k) { z[2 * k] = in[2 * k]; z[2 * k + 1] = in[2 * k + 1]; }
  z[1] = 0.0;
  if (n % 2 == 0) z[2 * (n / 2) + 1] = 0.0;
  for (int k = n / 2 + 1; k < n; ++k) { z[2 * k] = in[2 * (n - k)]; z[2 * k + 1] = -in[2 * (n - k) + 1]; }
  fft_ref_c2c(p, z, zo, +1);
  for (int j = 0; j < n; ++j) out[j] = zo[2 * j];
  free(z);
}

void fft_ref_redft10(int n, const double *in, double *out) {
  for (int k = 0; k < n; ++k) {
    double s = 0.0;
    for (int j = 0; j < n; ++j) s += in[j] * cos(M_PI * ((double)j + 0.5) * (double)k / (double)n);
    out[k] = 2.0 * s;
  }
}

void fft_ref_redft01(int n, const double *in, double *out) {
  for (int k = 0; k < n; ++k) {
    double s = 0.0;
    for (int j = 1; j < n; ++j) s += in[j] * cos(M_PI * (double)j * ((double)k + 0.5) / (double)n);
    out[k] = in[0] + 2.0 * s;
  }
}
