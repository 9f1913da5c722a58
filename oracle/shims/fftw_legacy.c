/* TEST INFRASTRUCTURE (oracle/_ref all-reference builds only) -- not part of the product; the drop-in program never links it
 * (its modpois calls libudcore, nothing calls FFTW).
 *
 * The legacy FFTW3 Fortran entry points the reference calls (src/modpois.f90:110-111,120-121,...,481): dfftw_plan_*_1d capture
 * the buffer addresses, dfftw_execute runs the unnormalised transform on them.  The transforms are the published FFTW
 * definitions restated in oracle/fft_ref.c (pinned against numpy/scipy golden vectors).
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../fft_ref.h"

/* ------------------------------------------------------------ FFTW legacy */
enum { K_R2C = 1, K_C2R = 2, K_REDFT10 = 3, K_REDFT01 = 4 };
#define FFTW_REDFT01_CODE 4
#define FFTW_REDFT10_CODE 5

typedef struct {
  int kind, n;
  double *in, *out;
  fft_ref_plan *p;
} shim_plan;

void dfftw_plan_dft_r2c_1d_(int64_t *plan, int *n, double *in, double *out, int *flags) {
  (void)flags;
  shim_plan *s = (shim_plan *)calloc(1, sizeof(*s));
  s->kind = K_R2C; s->n = *n; s->in = in; s->out = out; s->p = fft_ref_plan_create(*n);
  *plan = (int64_t)(intptr_t)s;
}
void dfftw_plan_dft_c2r_1d_(int64_t *plan, int *n, double *in, double *out, int *flags) {
  (void)flags;
  shim_plan *s = (shim_plan *)calloc(1, sizeof(*s));
  s->kind = K_C2R; s->n = *n; s->in = in; s->out = out; s->p = fft_ref_plan_create(*n);
  *plan = (int64_t)(intptr_t)s;
}
void dfftw_plan_r2r_1d_(int64_t *plan, int *n, double *in, double *out, int *kind, int *flags) {
  (void)flags;
  shim_plan *s = (shim_plan *)calloc(1, sizeof(*s));
  s->n = *n; s->in = in; s->out = out;
  if (*kind == FFTW_REDFT10_CODE) s->kind = K_REDFT10;
  else if (*kind == FFTW_REDFT01_CODE) s->kind = K_REDFT01;
  else { fprintf(stderr, "ERROR: unsupported r2r kind %d in oracle FFT shim\n", *kind); exit(1); }
  *plan = (int64_t)(intptr_t)s;
}
void dfftw_execute_(int64_t *plan) {
  shim_plan *s = (shim_plan *)(intptr_t)(*plan);
  switch (s->kind) {
    case K_R2C: fft_ref_r2c(s->p, s->in, s->out); break;
    case K_C2R: fft_ref_c2r(s->p, s->in, s->out); break;
    case K_REDFT10: fft_ref_redft10(s->n, s->in, s->out); break;
    case K_REDFT01: fft_ref_redft01(s->n, s->in, s->out); break;
    default: fprintf(stderr, "ERROR: bad plan in oracle FFT shim\n"); exit(1);
  }
}
void dfftw_destroy_plan_(int64_t *plan) {
  shim_plan *s = (shim_plan *)(intptr_t)(*plan);
  if (s) { fft_ref_plan_destroy(s->p); free(s); }
}
