/* TEST INFRASTRUCTURE (oracle/_ref build only) -- not part of the product.
 *
 * Link-time externals for the single-rank reference build:
 *   (1) np=1 MPI entry points with the Fortran-77 calling convention.  With one
 *       rank a broadcast is a no-op and a reduction is a copy, so none of these
 *       performs arithmetic.  Datatype handles encode their byte size (see
 *       mpi_np1.f90: handle / 100 = size in bytes).
 *   (2) the legacy FFTW3 Fortran entry points the reference calls
 *       (src/modpois.f90:110-111,120-121,...,481): dfftw_plan_*_1d capture the
 *       buffer addresses, dfftw_execute runs the unnormalised transform on them.
 *       The transforms are the published FFTW definitions restated in
 *       oracle/fft_ref.c (pinned against numpy/scipy golden vectors).
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "../fft_ref.h"

/* ---------------------------------------------------------------- MPI np=1 */
#ifndef WITH_REAL_MPI   /* the MPI baseline build links MPICH's libmpifort instead */
static size_t tsize(int handle) { return (size_t)(handle / 100); }

void mpi_init_(int *ierr) { *ierr = 0; }
void mpi_finalize_(int *ierr) { *ierr = 0; }
void mpi_abort_(int *comm, int *code, int *ierr) { (void)comm; (void)ierr; exit(*code ? *code : 1); }
void mpi_comm_rank_(int *comm, int *rank, int *ierr) { (void)comm; *rank = 0; *ierr = 0; }
void mpi_comm_size_(int *comm, int *size, int *ierr) { (void)comm; *size = 1; *ierr = 0; }
void mpi_barrier_(int *comm, int *ierr) { (void)comm; *ierr = 0; }
void mpi_bcast_(void *buf, int *count, int *type, int *root, int *comm, int *ierr) {
  (void)buf; (void)count; (void)type; (void)root; (void)comm; *ierr = 0;
}
void mpi_allreduce_(const void *s, void *r, int *count, int *type, int *op, int *comm, int *ierr) {
  (void)op; (void)comm;
  if (s != r) memmove(r, s, (size_t)(*count) * tsize(*type));
  *ierr = 0;
}
void mpi_reduce_(const void *s, void *r, int *count, int *type, int *op, int *root, int *comm, int *ierr) {
  (void)op; (void)comm; (void)root;
  if (s != r) memmove(r, s, (size_t)(*count) * tsize(*type));
  *ierr = 0;
}
/* point-to-point only appears in the unused legacy excis/excjs (src/modmpi.f90:389,526) */
static void p2p_unavailable(const char *what) {
  fprintf(stderr, "ERROR: %s called in the single-rank oracle build\n", what);
  exit(1);
}
void mpi_isend_(void) { p2p_unavailable("MPI_ISEND"); }
void mpi_irecv_(void) { p2p_unavailable("MPI_IRECV"); }
void mpi_send_(void) { p2p_unavailable("MPI_SEND"); }
void mpi_recv_(void) { p2p_unavailable("MPI_RECV"); }
void mpi_sendrecv_(void) { p2p_unavailable("MPI_SENDRECV"); }
void mpi_wait_(void) { p2p_unavailable("MPI_WAIT"); }
/* the whole-tree build: src/modstartup.f90:681-688 asks for this rank's coordinates and neighbours in the (1 x 1) grid; with one
   rank in a direction 2DECOMP's grid is not periodic there (:662-672), so both neighbours are MPI_PROC_NULL (-2, mpi_np1.f90) */
void mpi_cart_coords_(int *comm, int *rank, int *maxdims, int *coords, int *ierr) {
  (void)comm; (void)rank;
  for (int i = 0; i < *maxdims; i++) coords[i] = 0;
  *ierr = 0;
}
void mpi_cart_shift_(int *comm, int *dir, int *disp, int *src, int *dst, int *ierr) {
  (void)comm; (void)dir; (void)disp;
  *src = -2; *dst = -2; *ierr = 0;
}
double mpi_wtime_(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

#endif /* WITH_REAL_MPI */

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
