/* TEST INFRASTRUCTURE -- CPU oracle, NOT the product.  See udcore_oracle.h.
 *
 * Straight restatement, loop for loop, of the reference Fortran; expression
 * order follows the cited lines so that differences against the reference's own
 * compiled code stay at the few-ulp level (compiled with -ffp-contract=off).
 */
#include "udcore_oracle.h"
#include "fft_ref.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846264338327950288
#endif

#define SX (g->nx + 2)
#define SY (g->ny + 2)
#define M(a, i, j, k) a[(size_t)(i) + (size_t)SX * ((size_t)(j) + (size_t)SY * (size_t)(k))]
#define SXC (g->nx + 4)
#define SYC (g->ny + 4)
#define C(a, i, j, k) a[(size_t)((i) + 1) + (size_t)SXC * ((size_t)((j) + 1) + (size_t)SYC * (size_t)((k) + 1))]

static size_t msize(const orc_grid *g) { return (size_t)(g->nx + 2) * (g->ny + 2) * (g->nz + 2); }
static size_t csize(const orc_grid *g) { return (size_t)(g->nx + 4) * (g->ny + 4) * (g->nz + 4); }

/* metrics exactly as src/modglobal.f90:812-838 derives them */
typedef struct {
  double dxi, dyi, dxiq, dyiq, dx2i, dy2i, dxi5, dyi5, dx2, dy2;
  double *dzfi, *dzfi5, *dzfiq, *dzf2, *dzhi, *dzhiq, *dzh2i; /* indexed by Fortran k */
} metrics;

static void metrics_init(const orc_grid *g, metrics *m) {
  int n = g->nz + 2;
  m->dxi = 1. / g->dx; m->dyi = 1. / g->dy;
  m->dx2 = g->dx * g->dx; m->dy2 = g->dy * g->dy;
  m->dxiq = 0.25 * m->dxi; m->dyiq = 0.25 * m->dyi;
  m->dx2i = m->dxi * m->dxi; m->dy2i = m->dyi * m->dyi;
  m->dxi5 = 0.5 * m->dxi; m->dyi5 = 0.5 * m->dyi;
  m->dzfi = (double *)calloc(7 * (size_t)n, sizeof(double));
  m->dzfi5 = m->dzfi + n; m->dzfiq = m->dzfi5 + n; m->dzf2 = m->dzfiq + n;
  m->dzhi = m->dzf2 + n; m->dzhiq = m->dzhi + n; m->dzh2i = m->dzhiq + n;
  for (int k = 0; k < n; ++k) {
    m->dzfi[k] = 1. / g->dzf[k];
    m->dzfi5[k] = 0.5 * m->dzfi[k];
    m->dzfiq[k] = 0.25 * m->dzfi[k];
    m->dzf2[k] = g->dzf[k] * g->dzf[k];
  }
  for (int k = 1; k < n; ++k) {
    m->dzhi[k] = 1. / g->dzh[k];
    m->dzhiq[k] = 0.25 * m->dzhi[k];
    m->dzh2i[k] = m->dzhi[k] * m->dzhi[k];
  }
}
static void metrics_free(metrics *m) { free(m->dzfi); }

/* ====================================================================== advection */

/* src/modadvection.f90:158-212 */
void orc_advecu_2nd(const orc_grid *g, const double *u0, const double *v0, const double *w0,
                    const double *pres0, double *up) {
  metrics m; metrics_init(g, &m);
  const double *dzf = g->dzf;
  for (int k = 1; k <= g->nz; ++k)
    for (int j = 1; j <= g->ny; ++j)
      for (int i = 1; i <= g->nx; ++i) {
        int im = i - 1, ip = i + 1, jm = j - 1, jp = j + 1;
        M(up, i, j, k) = M(up, i, j, k) - (
            ((M(u0, i, j, k) + M(u0, ip, j, k)) * (M(u0, i, j, k) + M(u0, ip, j, k))
           - (M(u0, i, j, k) + M(u0, im, j, k)) * (M(u0, i, j, k) + M(u0, im, j, k))) * m.dxiq
          + ((M(u0, i, j, k) + M(u0, i, jp, k)) * (M(v0, i, jp, k) + M(v0, im, jp, k))
           - (M(u0, i, j, k) + M(u0, i, jm, k)) * (M(v0, i, j, k) + M(v0, im, j, k))) * m.dyiq)
          - ((M(pres0, i, j, k) - M(pres0, i - 1, j, k)) * m.dxi);
      }
  for (int j = 1; j <= g->ny; ++j)
    for (int i = 1; i <= g->nx; ++i)
      for (int k = 1; k <= g->nz; ++k) {
        int im = i - 1, km = k - 1, kp = k + 1;
        M(up, i, j, k) = M(up, i, j, k) - (
            (M(u0, i, j, kp) * dzf[k] + M(u0, i, j, k) * dzf[kp]) * m.dzhi[kp]
              * (M(w0, i, j, kp) + M(w0, im, j, kp))
          - (M(u0, i, j, k) * dzf[km] + M(u0, i, j, km) * dzf[k]) * m.dzhi[k]
              * (M(w0, i, j, k) + M(w0, im, j, k))) * 0.5 * m.dzfi5[k];
      }
  metrics_free(&m);
}

/* src/modadvection.f90:215-270 */
void orc_advecv_2nd(const orc_grid *g, const double *u0, const double *v0, const double *w0,
                    const double *pres0, double *vp) {
  metrics m; metrics_init(g, &m);
  const double *dzf = g->dzf;
  for (int k = 1; k <= g->nz; ++k)
    for (int j = 1; j <= g->ny; ++j)
      for (int i = 1; i <= g->nx; ++i) {
        int im = i - 1, ip = i + 1, jm = j - 1, jp = j + 1;
        M(vp, i, j, k) = M(vp, i, j, k) - (
            ((M(u0, ip, j, k) + M(u0, ip, jm, k)) * (M(v0, i, j, k) + M(v0, ip, j, k))
           - (M(u0, i, j, k) + M(u0, i, jm, k)) * (M(v0, i, j, k) + M(v0, im, j, k))) * m.dxiq
          + ((M(v0, i, jp, k) + M(v0, i, j, k)) * (M(v0, i, j, k) + M(v0, i, jp, k))
           - (M(v0, i, jm, k) + M(v0, i, j, k)) * (M(v0, i, j, k) + M(v0, i, jm, k))) * m.dyiq)
          - ((M(pres0, i, j, k) - M(pres0, i, jm, k)) * m.dyi);
      }
  for (int j = 1; j <= g->ny; ++j)
    for (int i = 1; i <= g->nx; ++i)
      for (int k = 1; k <= g->nz; ++k) {
        int jm = j - 1, km = k - 1, kp = k + 1;
        M(vp, i, j, k) = M(vp, i, j, k) - (
            (M(w0, i, j, kp) + M(w0, i, jm, kp))
              * (M(v0, i, j, kp) * dzf[k] + M(v0, i, j, k) * dzf[kp]) * m.dzhi[kp]
          - (M(w0, i, j, k) + M(w0, i, jm, k))
              * (M(v0, i, j, km) * dzf[k] + M(v0, i, j, k) * dzf[km]) * m.dzhi[k]) * 0.5 * m.dzfi5[k];
      }
  metrics_free(&m);
}

/* src/modadvection.f90:273-314 */
void orc_advecw_2nd(const orc_grid *g, const double *u0, const double *v0, const double *w0,
                    const double *pres0, double *wp) {
  metrics m; metrics_init(g, &m);
  const double *dzf = g->dzf;
  for (int k = 2; k <= g->nz; ++k)
    for (int j = 1; j <= g->ny; ++j)
      for (int i = 1; i <= g->nx; ++i) {
        int im = i - 1, ip = i + 1, jm = j - 1, jp = j + 1, km = k - 1, kp = k + 1;
        M(wp, i, j, k) = M(wp, i, j, k) - (
            ((M(w0, ip, j, k) + M(w0, i, j, k)) * (dzf[km] * M(u0, ip, j, k) + dzf[k] * M(u0, ip, j, km))
           - (M(w0, i, j, k) + M(w0, im, j, k)) * (dzf[km] * M(u0, i, j, k) + dzf[k] * M(u0, i, j, km)))
              * m.dxiq * m.dzhi[k]
          + ((M(w0, i, jp, k) + M(w0, i, j, k)) * (dzf[km] * M(v0, i, jp, k) + dzf[k] * M(v0, i, jp, km))
           - (M(w0, i, j, k) + M(w0, i, jm, k)) * (dzf[km] * M(v0, i, j, k) + dzf[k] * M(v0, i, j, km)))
              * m.dyiq * m.dzhi[k]
          + ((M(w0, i, j, k) + M(w0, i, j, kp)) * (M(w0, i, j, k) + M(w0, i, j, kp))
           - (M(w0, i, j, k) + M(w0, i, j, km)) * (M(w0, i, j, k) + M(w0, i, j, km))) * m.dzhiq[k])
          - ((M(pres0, i, j, k) - M(pres0, i, j, km)) * m.dzhi[k]);
      }
  metrics_free(&m);
}

/* src/modadvection.f90:410-421, eps1 = 1e-10 (src/modglobal.f90:318) */
static double rlim(double d1, double d2) {
  const double eps1 = 1.e-10;
  double ri = (d2 + eps1) / (d1 + eps1);
  double phir = fmax(0., fmin(2. * ri, fmin(1. / 3. + 2. / 3. * ri, 2.)));
  return 0.5 * phir * d1;
}

/* src/modadvection.f90:316-407; kappa grids src/modglobal.f90:841-867.
 * The reference accumulates each direction through two temporaries (dumu, duml)
 * and adds `dumu + duml` to the tendency; the same order is kept here. */
void orc_advecc_kappa(const orc_grid *g, const double *u0, const double *v0, const double *w0,
                      const double *c, double *cp) {
  const int nx = g->nx, ny = g->ny, nz = g->nz;
  const double dyi = 1. / g->dy;
  /* x grid as src/modglobal.f90:771-791 builds it (uniform, but via differences) */
  double *xh = (double *)calloc((size_t)8 * (nx + 8), sizeof(double));
  double *xf = xh + (nx + 8), *dxf = xf + (nx + 8), *dxh = dxf + (nx + 8);
  double *dxfc = dxh + (nx + 8), *dxfci = dxfc + (nx + 8), *dxhci = dxfci + (nx + 8);
  /* arrays indexed by Fortran i + 2 (range -1..nx+2) */
#define X(a, i) a[(i) + 2]
  for (int i = 1; i <= nx + 1; ++i) { X(xh, i) = (i - 1) * g->dx; X(xf, i) = X(xh, i) + g->dx / 2; }
  for (int i = 1; i <= nx; ++i) X(dxf, i) = X(xh, i + 1) - X(xh, i);
  X(dxf, nx + 1) = X(dxf, nx); X(dxf, 0) = X(dxf, 1);
  X(dxh, 1) = 2 * X(xf, 1);
  for (int i = 2; i <= nx + 1; ++i) X(dxh, i) = X(xf, i) - X(xf, i - 1);
  for (int i = 0; i <= nx + 1; ++i) X(dxfc, i) = X(dxf, i);
  X(dxfc, -1) = X(dxfc, 0); X(dxfc, nx + 2) = X(dxfc, nx + 1);
  for (int i = 1; i <= nx + 1; ++i) X(dxhci, i) = 1. / X(dxh, i);
  X(dxhci, 0) = X(dxhci, 1); X(dxhci, nx + 2) = X(dxhci, nx + 1);
  for (int i = -1; i <= nx + 2; ++i) X(dxfci, i) = 1. / X(dxfc, i);
  /* z grids, Fortran k + 2 (range -1..nz+2) */
  double *dzfc = (double *)calloc((size_t)3 * (nz + 8), sizeof(double));
  double *dzfci = dzfc + (nz + 8), *dzhci = dzfci + (nz + 8);
  for (int k = 0; k <= nz + 1; ++k) X(dzfc, k) = g->dzf[k];
  X(dzfc, -1) = X(dzfc, 0); X(dzfc, nz + 2) = X(dzfc, nz + 1);
  for (int k = 1; k <= nz + 1; ++k) X(dzhci, k) = 1. / g->dzh[k];
  X(dzhci, 0) = X(dzhci, 1); X(dzhci, nz + 2) = X(dzhci, nz + 1);
  for (int k = -1; k <= nz + 2; ++k) X(dzfci, k) = 1. / X(dzfc, k);

  size_t n = csize(g);
  double *dumu = (double *)calloc(2 * n, sizeof(double));
  double *duml = dumu + n;
  double d1, d2, cf;
  /* -d(uc)/dx */
  for (int k = 1; k <= nz; ++k)
    for (int j = 1; j <= ny; ++j)
      for (int i = 1; i <= nx + 1; ++i) {
        if (M(u0, i, j, k) > 0) {
          d1 = (C(c, i - 1, j, k) - C(c, i - 2, j, k)) * X(dxhci, i - 1);
          d2 = (C(c, i, j, k) - C(c, i - 1, j, k)) * X(dxhci, i);
          cf = C(c, i - 1, j, k);
        } else {
          d1 = (C(c, i, j, k) - C(c, i + 1, j, k)) * X(dxhci, i + 1);
          d2 = (C(c, i - 1, j, k) - C(c, i, j, k)) * X(dxhci, i);
          cf = C(c, i, j, k);
        }
        cf = cf + X(dxfc, i) * rlim(d1, d2);
        C(dumu, i - 1, j, k) = -cf * M(u0, i, j, k) * X(dxfci, i - 1);
        C(duml, i, j, k) = cf * M(u0, i, j, k) * X(dxfci, i);
      }
  for (size_t q = 0; q < n; ++q) cp[q] = cp[q] + dumu[q] + duml[q];
  memset(dumu, 0, 2 * n * sizeof(double));
  /* -d(vc)/dy */
  for (int k = 1; k <= nz; ++k)
    for (int j = 1; j <= ny + 1; ++j)
      for (int i = 1; i <= nx; ++i) {
        if (M(v0, i, j, k) > 0) {
          d1 = C(c, i, j - 1, k) - C(c, i, j - 2, k);
          d2 = C(c, i, j, k) - C(c, i, j - 1, k);
          cf = C(c, i, j - 1, k);
        } else {
          d1 = C(c, i, j, k) - C(c, i, j + 1, k);
          d2 = C(c, i, j - 1, k) - C(c, i, j, k);
          cf = C(c, i, j, k);
        }
        cf = cf + rlim(d1, d2);
        C(duml, i, j, k) = cf * M(v0, i, j, k) * dyi;
        C(dumu, i, j - 1, k) = -cf * M(v0, i, j, k) * dyi;
      }
  for (size_t q = 0; q < n; ++q) cp[q] = cp[q] + dumu[q] + duml[q];
  memset(dumu, 0, 2 * n * sizeof(double));
  /* -d(wc)/dz, faces kb+1..ke+1 */
  for (int k = 2; k <= nz + 1; ++k)
    for (int j = 1; j <= ny; ++j)
      for (int i = 1; i <= nx; ++i) {
        if (M(w0, i, j, k) > 0) {
          d1 = (C(c, i, j, k - 1) - C(c, i, j, k - 2)) * X(dzhci, k - 1);
          d2 = (C(c, i, j, k) - C(c, i, j, k - 1)) * X(dzhci, k);
          cf = C(c, i, j, k - 1);
        } else {
          d1 = (C(c, i, j, k) - C(c, i, j, k + 1)) * X(dzhci, k + 1);
          d2 = (C(c, i, j, k - 1) - C(c, i, j, k)) * X(dzhci, k);
          cf = C(c, i, j, k);
        }
        cf = cf + X(dzfc, k) * rlim(d1, d2);
        C(duml, i, j, k) = cf * M(w0, i, j, k) * X(dzfci, k);
        C(dumu, i, j, k - 1) = -cf * M(w0, i, j, k) * X(dzfci, k - 1);
      }
  for (size_t q = 0; q < n; ++q) cp[q] = cp[q] + dumu[q] + duml[q];
#undef X
  free(dumu); free(dzfc); free(xh);
}

/* ====================================================================== subgrid */

/* closure: src/modsubgrid.f90:159-412 (Smagorinsky :208-264, Vreman :269-360,
 * DNS :401-404), followed by closurebc.  As in the reference the molecular
 * viscosity is added to the WHOLE array (halos included) before closurebc
 * rewrites the halos. */
static double orc_dthvdz(const orc_grid *g, const double *thl0, int i, int j, int k);
/* thl0 of the state the closure works on, for the Vreman buoyancy correction (orc_substep sets it; NULL: no correction) */
static const double *closure_thl0 = NULL;
void orc_set_closure_thl(const double *thl0) { closure_thl0 = thl0; }
void orc_closure(const orc_grid *g, const double *u0, const double *v0, const double *w0,
                 double *ekm, double *ekh) {
  metrics m; metrics_init(g, &m);
  const double *dzf = g->dzf;
  const size_t n = msize(g);
  const int nx = g->nx, ny = g->ny, nz = g->nz;
  if (g->sgs == 1) {
    for (int k = 1; k <= nz; ++k) {
      int kp = k + 1, km = k - 1;
      for (int i = 1; i <= nx; ++i) {
        int ip = i + 1, im = i - 1;
        /* delta(i,k) = (dxf(i)*dy*dzf(k))**(1/3)  src/modglobal.f90:793-797; dxf(i) = xh(i+1)-xh(i) */
        double dxf_i = (double)i * g->dx - (double)(i - 1) * g->dx;
        double delta = pow(dxf_i * g->dy * dzf[k], 1. / 3.);
        double mlen = g->csz * delta;
        for (int j = 1; j <= ny; ++j) {
          int jp = j + 1, jm = j - 1;
          double damp = 1.;
          double t, strain2;
          t = (M(u0, ip, j, k) - M(u0, i, j, k)) * m.dxi; strain2 = t * t;
          t = (M(v0, i, jp, k) - M(v0, i, j, k)) * m.dyi; strain2 = strain2 + t * t;
          t = (M(w0, i, j, kp) - M(w0, i, j, k)) * m.dzfi[k]; strain2 = strain2 + t * t;
          double a1 = (M(w0, i, j, kp) - M(w0, im, j, kp)) * m.dxi + (M(u0, i, j, kp) - M(u0, i, j, k)) * m.dzhi[kp];
          double a2 = (M(w0, i, j, k) - M(w0, im, j, k)) * m.dxi + (M(u0, i, j, k) - M(u0, i, j, km)) * m.dzhi[k];
          double a3 = (M(w0, ip, j, k) - M(w0, i, j, k)) * m.dxi + (M(u0, ip, j, k) - M(u0, ip, j, km)) * m.dzhi[k];
          double a4 = (M(w0, ip, j, kp) - M(w0, i, j, kp)) * m.dxi + (M(u0, ip, j, kp) - M(u0, ip, j, k)) * m.dzhi[kp];
          strain2 = strain2 + 0.125 * (a1 * a1 + a2 * a2 + a3 * a3 + a4 * a4);
          a1 = (M(u0, i, jp, k) - M(u0, i, j, k)) * m.dyi + (M(v0, i, jp, k) - M(v0, im, jp, k)) * m.dxi;
          a2 = (M(u0, i, j, k) - M(u0, i, jm, k)) * m.dyi + (M(v0, i, j, k) - M(v0, im, j, k)) * m.dxi;
          a3 = (M(u0, ip, j, k) - M(u0, ip, jm, k)) * m.dyi + (M(v0, ip, j, k) - M(v0, i, j, k)) * m.dxi;
          a4 = (M(u0, ip, jp, k) - M(u0, ip, j, k)) * m.dyi + (M(v0, ip, jp, k) - M(v0, i, jp, k)) * m.dxi;
          strain2 = strain2 + 0.125 * (a1 * a1 + a2 * a2 + a3 * a3 + a4 * a4);
          a1 = (M(v0, i, j, kp) - M(v0, i, j, k)) * m.dzhi[kp] + (M(w0, i, j, kp) - M(w0, i, jm, kp)) * m.dyi;
          a2 = (M(v0, i, j, k) - M(v0, i, j, km)) * m.dzhi[k] + (M(w0, i, j, k) - M(w0, i, jm, k)) * m.dyi;
          a3 = (M(v0, i, jp, k) - M(v0, i, jp, km)) * m.dzhi[k] + (M(w0, i, jp, k) - M(w0, i, j, k)) * m.dyi;
          a4 = (M(v0, i, jp, kp) - M(v0, i, jp, k)) * m.dzhi[kp] + (M(w0, i, jp, kp) - M(w0, i, j, kp)) * m.dyi;
          strain2 = strain2 + 0.125 * (a1 * a1 + a2 * a2 + a3 * a3 + a4 * a4);
          double md = mlen * damp;
          M(ekm, i, j, k) = (md * md) * sqrt(2. * strain2);
          M(ekh, i, j, k) = M(ekm, i, j, k) * g->prandtli;
        }
      }
    }
    for (size_t q = 0; q < n; ++q) ekm[q] = ekm[q] + g->numol;
    for (size_t q = 0; q < n; ++q) ekh[q] = ekh[q] + g->numol * g->prandtlmoli;
  } else if (g->sgs == 2) {
    for (int k = 1; k <= nz; ++k) {
      int kp = k + 1, km = k - 1;
      for (int j = 1; j <= ny; ++j) {
        int jp = j + 1, jm = j - 1;
        for (int i = 1; i <= nx; ++i) {
          int ip = i + 1, im = i - 1;
          double a11 = (M(u0, ip, j, k) - M(u0, i, j, k)) * m.dxi;
          double a12 = (M(v0, ip, jp, k) + M(v0, ip, j, k) - M(v0, im, jp, k) - M(v0, im, j, k)) * m.dxiq;
          double a13 = (M(w0, ip, j, kp) + M(w0, ip, j, k) - M(w0, im, j, kp) - M(w0, im, j, k)) * m.dxiq;
          double a21 = (M(u0, ip, jp, k) + M(u0, i, jp, k) - M(u0, ip, jm, k) - M(u0, i, jm, k)) * m.dyiq;
          double a22 = (M(v0, i, jp, k) - M(v0, i, j, k)) * m.dyi;
          double a23 = (M(w0, i, jp, kp) + M(w0, i, jp, k) - M(w0, i, jm, kp) - M(w0, i, jm, k)) * m.dyiq;
          double a31 = (((M(u0, ip, j, kp) + M(u0, i, j, kp)) * dzf[k] + (M(u0, ip, j, k) + M(u0, i, j, k)) * dzf[kp]) * m.dzhi[kp]
                      - ((M(u0, ip, j, k) + M(u0, i, j, k)) * dzf[km] + (M(u0, ip, j, km) + M(u0, i, j, km)) * dzf[k]) * m.dzhi[k])
                       * m.dzfiq[k];
          double a32 = (((M(v0, i, jp, kp) + M(v0, i, j, kp)) * dzf[k] + (M(v0, i, jp, k) + M(v0, i, j, k)) * dzf[kp]) * m.dzhi[kp]
                      - ((M(v0, i, jp, k) + M(v0, i, j, k)) * dzf[km] + (M(v0, i, jp, km) + M(v0, i, j, km)) * dzf[k]) * m.dzhi[k])
                       * m.dzfiq[k];
          double a33 = (M(w0, i, j, kp) - M(w0, i, j, k)) * m.dzfi[k];
          double aa = a11 * a11 + a21 * a21 + a31 * a31 + a12 * a12 + a22 * a22 + a32 * a32
                    + a13 * a13 + a23 * a23 + a33 * a33;
          double dz2 = m.dzf2[k];
          double b11 = m.dx2 * a11 * a11 + m.dy2 * a21 * a21 + dz2 * a31 * a31;
          double b22 = m.dx2 * a12 * a12 + m.dy2 * a22 * a22 + dz2 * a32 * a32;
          double b12 = m.dx2 * a11 * a12 + m.dy2 * a21 * a22 + dz2 * a31 * a32;
          double b33 = m.dx2 * a13 * a13 + m.dy2 * a23 * a23 + dz2 * a33 * a33;
          double b13 = m.dx2 * a11 * a13 + m.dy2 * a21 * a23 + dz2 * a31 * a33;
          double b23 = m.dx2 * a12 * a13 + m.dy2 * a22 * a23 + dz2 * a32 * a33;
          double bb = b11 * b22 - b12 * b12 + b11 * b33 - b13 * b13 + b22 * b33 - b23 * b23;
          if (bb < 1.e-8) M(ekm, i, j, k) = 0.;
          else M(ekm, i, j, k) = g->c_vreman * sqrt(bb / aa);
        }
      }
    }
    /* buoyancy correction for stable stratification, src/modsubgrid.f90:330-353 (dthvdz: calthv's, from the same thl0) */
    if (g->lbuoyancy && g->lbuoycorr && closure_thl0) {
      const double *thl0 = closure_thl0;
      for (int k = 1; k <= nz; ++k) {
        int kp = k + 1, km = k - 1;
        for (int j = 1; j <= ny; ++j) {
          int jp = j + 1;
          for (int i = 1; i <= nx; ++i) {
            int ip = i + 1;
            double du0dz = 0.5 * ((M(u0, i, j, kp) + M(u0, ip, j, kp)) - (M(u0, i, j, km) + M(u0, ip, j, km))) / (g->dzh[kp] + g->dzh[k]);
            double dv0dz = 0.5 * ((M(v0, i, j, kp) + M(v0, i, jp, kp)) - (M(v0, i, j, km) + M(v0, i, jp, km))) / (g->dzh[kp] + g->dzh[k]);
            double Rig = ((9.81 / M(thl0, i, j, k)) * orc_dthvdz(g, thl0, i, j, k)) / (du0dz * du0dz + dv0dz * dv0dz + 1.e-10);
            M(ekm, i, j, k) = M(ekm, i, j, k) * sqrt(1.0 - fmin(fmax(Rig, 0.0), g->Rigc) / g->Rigc);
          }
        }
      }
    }
    for (size_t q = 0; q < n; ++q) ekh[q] = ekm[q] * g->prandtli;
    for (size_t q = 0; q < n; ++q) ekm[q] = ekm[q] + g->numol;
    for (size_t q = 0; q < n; ++q) ekh[q] = ekh[q] + g->numol * g->prandtlmoli;
  } else {
    for (size_t q = 0; q < n; ++q) { ekm[q] = g->numol; ekh[q] = g->numol * g->prandtlmoli; }
  }
  metrics_free(&m);
  orc_closurebc(g, ekm, ekh);
}

/* src/modboundary.f90:434-505, single rank, periodic x and y.
 * (reassure_fluxtop_boundary :392-431 is applied by the caller, orc_substep.) */
/* Inflow / outflow in x (&BC BCxm = 2, BCxm_profile): orc_set_open_x switches the x ghost columns of the m-arrays from periodic
 * images to the reference's inflow / outflow values -- the arrays here carry the reference's ghosts, so only the routines that
 * WRITE them change: closurebc (src/modboundary.f90:467-475), bcpup (:1257-1280), bcp (:1376-1394), the cosine transform in x of
 * poisson (src/modpois.f90:113-121, 492-500, 689-697), tstep_integrate's u0(ie+1) (src/modtstep.f90:262-264), halos (no periodic
 * refresh in x, src/modboundary.f90:95-100) and boundary's xmi_profile / xmo_convective (:688-717, 908-926).
 * uprof, vprof: [nz+2] indexed by k (entry nz+1 = what the reference's uprof(ke+1) holds: zero).  uouttot is a variable of the
 * run (the last `boundary`'s value is what bcpup reads): orc_set_open_x_uouttot / orc_open_x_uouttot; wlev[nz] (or NULL: keep the
 * value) makes orc_boundary refresh it as the reference does, sum_k wlev(k-1) u0av(k) with the u0av handed to orc_boundary_open_x. */
static int xo_on = 0;
static const double *xo_uprof = NULL, *xo_vprof = NULL;
static double xo_uouttot = 0.;
void orc_set_open_x(int on, const double *uprof, const double *vprof) { xo_on = on; xo_uprof = uprof; xo_vprof = vprof; }
void orc_set_open_x_uouttot(double u) { xo_uouttot = u; }
double orc_open_x_uouttot(void) { return xo_uouttot; }

void orc_closurebc(const orc_grid *g, double *ekm, double *ekh) {
  const int nx = g->nx, ny = g->ny, nz = g->nz;
  const double nm = g->numol, nh = g->numol * g->prandtlmoli;
  for (int j = 0; j <= ny + 1; ++j)
    for (int i = 0; i <= nx + 1; ++i) {
      if (g->bctopm == 2) {
        M(ekm, i, j, nz + 1) = 2. * nm - M(ekm, i, j, nz);
        M(ekh, i, j, nz + 1) = (2. * nh) - M(ekh, i, j, nz);
      } else {
        M(ekm, i, j, nz + 1) = M(ekm, i, j, nz);
        M(ekh, i, j, nz + 1) = M(ekh, i, j, nz);
      }
      M(ekm, i, j, 0) = 2. * nm - M(ekm, i, j, 1);
      M(ekh, i, j, 0) = (2. * nh) - M(ekh, i, j, 1);
    }
  for (int k = 0; k <= nz + 1; ++k)
    for (int j = 0; j <= ny + 1; ++j) {
      if (xo_on) {      /* src/modboundary.f90:467-475 */
        M(ekm, 0, j, k) = M(ekm, 1, j, k); M(ekm, nx + 1, j, k) = M(ekm, nx, j, k);
        M(ekh, 0, j, k) = M(ekh, 1, j, k); M(ekh, nx + 1, j, k) = M(ekh, nx, j, k);
        continue;
      }
      M(ekm, 0, j, k) = M(ekm, nx, j, k); M(ekm, nx + 1, j, k) = M(ekm, 1, j, k);
      M(ekh, 0, j, k) = M(ekh, nx, j, k); M(ekh, nx + 1, j, k) = M(ekh, 1, j, k);
    }
  for (int k = 0; k <= nz + 1; ++k)
    for (int i = 0; i <= nx + 1; ++i) {
      M(ekm, i, 0, k) = M(ekm, i, ny, k); M(ekm, i, ny + 1, k) = M(ekm, i, 1, k);
      M(ekh, i, 0, k) = M(ekh, i, ny, k); M(ekh, i, ny + 1, k) = M(ekh, i, 1, k);
    }
}

/* src/modsubgrid.f90:672-775 (LES :685-732, DNS :734-771) */
void orc_diffu(const orc_grid *g, const double *u0, const double *v0, const double *w0,
               const double *ekm, double *up) {
  metrics m; metrics_init(g, &m);
  const double *dzf = g->dzf;
  const double nu = g->numol;
  for (int k = 1; k <= g->nz; ++k)
    for (int j = 1; j <= g->ny; ++j)
      for (int i = 1; i <= g->nx; ++i) {
        int kp = k + 1, km = k - 1, jp = j + 1, jm = j - 1;
        if (g->sgs != 0) {
          double emom = (dzf[km] * (M(ekm, i, j, k) + M(ekm, i - 1, j, k)) + dzf[k] * (M(ekm, i, j, km) + M(ekm, i - 1, j, km))) * m.dzhiq[k];
          double emop = (dzf[kp] * (M(ekm, i, j, k) + M(ekm, i - 1, j, k)) + dzf[k] * (M(ekm, i, j, kp) + M(ekm, i - 1, j, kp))) * m.dzhiq[kp];
          double empo = 0.25 * ((M(ekm, i, j, k) + M(ekm, i, jp, k)) + (M(ekm, i - 1, j, k) + M(ekm, i - 1, jp, k)));
          double emmo = 0.25 * ((M(ekm, i, j, k) + M(ekm, i, jm, k)) + (M(ekm, i - 1, jm, k) + M(ekm, i - 1, j, k)));
          M(up, i, j, k) = M(up, i, j, k)
            + (M(ekm, i, j, k) * (M(u0, i + 1, j, k) - M(u0, i, j, k))
             - M(ekm, i - 1, j, k) * (M(u0, i, j, k) - M(u0, i - 1, j, k))) * 2. * m.dx2i
            + (empo * ((M(u0, i, jp, k) - M(u0, i, j, k)) * m.dyi + (M(v0, i, jp, k) - M(v0, i - 1, jp, k)) * m.dxi)
             - emmo * ((M(u0, i, j, k) - M(u0, i, jm, k)) * m.dyi + (M(v0, i, j, k) - M(v0, i - 1, j, k)) * m.dxi)) * m.dyi
            + (emop * ((M(u0, i, j, kp) - M(u0, i, j, k)) * m.dzhi[kp] + (M(w0, i, j, kp) - M(w0, i - 1, j, kp)) * m.dxi)
             - emom * ((M(u0, i, j, k) - M(u0, i, j, km)) * m.dzhi[k] + (M(w0, i, j, k) - M(w0, i - 1, j, k)) * m.dxi)) * m.dzfi[k];
        } else {
          M(up, i, j, k) = M(up, i, j, k)
            + (nu * (M(u0, i + 1, j, k) - M(u0, i, j, k)) * m.dxi
             - nu * (M(u0, i, j, k) - M(u0, i - 1, j, k)) * m.dxi) * 2. * m.dxi
            + (nu * ((M(u0, i, jp, k) - M(u0, i, j, k)) * m.dyi + (M(v0, i, jp, k) - M(v0, i - 1, jp, k)) * m.dxi)
             - nu * ((M(u0, i, j, k) - M(u0, i, jm, k)) * m.dyi + (M(v0, i, j, k) - M(v0, i - 1, j, k)) * m.dxi)) * m.dyi
            + (nu * ((M(u0, i, j, kp) - M(u0, i, j, k)) * m.dzhi[kp] + (M(w0, i, j, kp) - M(w0, i - 1, j, kp)) * m.dxi)
             - nu * ((M(u0, i, j, k) - M(u0, i, j, km)) * m.dzhi[k] + (M(w0, i, j, k) - M(w0, i - 1, j, k)) * m.dxi)) * m.dzfi[k];
        }
      }
  metrics_free(&m);
}

/* src/modsubgrid.f90:778-886 */
void orc_diffv(const orc_grid *g, const double *u0, const double *v0, const double *w0,
               const double *ekm, double *vp) {
  metrics m; metrics_init(g, &m);
  const double *dzf = g->dzf;
  const double nu = g->numol;
  for (int k = 1; k <= g->nz; ++k)
    for (int j = 1; j <= g->ny; ++j)
      for (int i = 1; i <= g->nx; ++i) {
        int kp = k + 1, km = k - 1, jp = j + 1, jm = j - 1;
        if (g->sgs != 0) {
          double eomm = (dzf[km] * (M(ekm, i, j, k) + M(ekm, i, jm, k)) + dzf[k] * (M(ekm, i, j, km) + M(ekm, i, jm, km))) * m.dzhiq[k];
          double eomp = (dzf[kp] * (M(ekm, i, j, k) + M(ekm, i, jm, k)) + dzf[k] * (M(ekm, i, j, kp) + M(ekm, i, jm, kp))) * m.dzhiq[kp];
          double emmo = 0.25 * (M(ekm, i, j, k) + M(ekm, i, jm, k) + M(ekm, i - 1, jm, k) + M(ekm, i - 1, j, k));
          double epmo = 0.25 * (M(ekm, i, j, k) + M(ekm, i, jm, k) + M(ekm, i + 1, jm, k) + M(ekm, i + 1, j, k));
          M(vp, i, j, k) = M(vp, i, j, k)
            + (epmo * ((M(v0, i + 1, j, k) - M(v0, i, j, k)) * m.dxi + (M(u0, i + 1, j, k) - M(u0, i + 1, jm, k)) * m.dyi)
             - emmo * ((M(v0, i, j, k) - M(v0, i - 1, j, k)) * m.dxi + (M(u0, i, j, k) - M(u0, i, jm, k)) * m.dyi)) * m.dxi
            + (M(ekm, i, j, k) * (M(v0, i, jp, k) - M(v0, i, j, k))
             - M(ekm, i, jm, k) * (M(v0, i, j, k) - M(v0, i, jm, k))) * 2. * m.dy2i
            + (eomp * ((M(v0, i, j, kp) - M(v0, i, j, k)) * m.dzhi[kp] + (M(w0, i, j, kp) - M(w0, i, jm, kp)) * m.dyi)
             - eomm * ((M(v0, i, j, k) - M(v0, i, j, km)) * m.dzhi[k] + (M(w0, i, j, k) - M(w0, i, jm, k)) * m.dyi)) * m.dzfi[k];
        } else {
          M(vp, i, j, k) = M(vp, i, j, k)
            + (nu * ((M(v0, i + 1, j, k) - M(v0, i, j, k)) * m.dxi + (M(u0, i + 1, j, k) - M(u0, i + 1, jm, k)) * m.dyi)
             - nu * ((M(v0, i, j, k) - M(v0, i - 1, j, k)) * m.dxi + (M(u0, i, j, k) - M(u0, i, jm, k)) * m.dyi)) * m.dxi
            + (nu * (M(v0, i, jp, k) - M(v0, i, j, k)) - nu * (M(v0, i, j, k) - M(v0, i, jm, k))) * 2. * m.dy2i
            + (nu * ((M(v0, i, j, kp) - M(v0, i, j, k)) * m.dzhi[kp] + (M(w0, i, j, kp) - M(w0, i, jm, kp)) * m.dyi)
             - nu * ((M(v0, i, j, k) - M(v0, i, j, km)) * m.dzhi[k] + (M(w0, i, j, k) - M(w0, i, jm, k)) * m.dyi)) * m.dzfi[k];
        }
      }
  metrics_free(&m);
}

/* src/modsubgrid.f90:890-997 */
void orc_diffw(const orc_grid *g, const double *u0, const double *v0, const double *w0,
               const double *ekm, double *wp) {
  metrics m; metrics_init(g, &m);
  const double *dzf = g->dzf;
  const double nu = g->numol;
  for (int k = 2; k <= g->nz; ++k)
    for (int j = 1; j <= g->ny; ++j)
      for (int i = 1; i <= g->nx; ++i) {
        int kp = k + 1, km = k - 1, jp = j + 1, jm = j - 1;
        if (g->sgs != 0) {
          double emom = (dzf[km] * (M(ekm, i, j, k) + M(ekm, i - 1, j, k)) + dzf[k] * (M(ekm, i, j, km) + M(ekm, i - 1, j, km))) * m.dzhiq[k];
          double eomm = (dzf[km] * (M(ekm, i, j, k) + M(ekm, i, jm, k)) + dzf[k] * (M(ekm, i, j, km) + M(ekm, i, jm, km))) * m.dzhiq[k];
          double eopm = (dzf[km] * (M(ekm, i, j, k) + M(ekm, i, jp, k)) + dzf[k] * (M(ekm, i, j, km) + M(ekm, i, jp, km))) * m.dzhiq[k];
          double epom = (dzf[km] * (M(ekm, i, j, k) + M(ekm, i + 1, j, k)) + dzf[k] * (M(ekm, i, j, km) + M(ekm, i + 1, j, km))) * m.dzhiq[k];
          M(wp, i, j, k) = M(wp, i, j, k)
            + (epom * ((M(w0, i + 1, j, k) - M(w0, i, j, k)) * m.dxi + (M(u0, i + 1, j, k) - M(u0, i + 1, j, km)) * m.dzhi[k])
             - emom * ((M(w0, i, j, k) - M(w0, i - 1, j, k)) * m.dxi + (M(u0, i, j, k) - M(u0, i, j, km)) * m.dzhi[k])) * m.dxi
            + (eopm * ((M(w0, i, jp, k) - M(w0, i, j, k)) * m.dyi + (M(v0, i, jp, k) - M(v0, i, jp, km)) * m.dzhi[k])
             - eomm * ((M(w0, i, j, k) - M(w0, i, jm, k)) * m.dyi + (M(v0, i, j, k) - M(v0, i, j, km)) * m.dzhi[k])) * m.dyi
            + (M(ekm, i, j, k) * (M(w0, i, j, kp) - M(w0, i, j, k)) * m.dzfi[k]
             - M(ekm, i, j, km) * (M(w0, i, j, k) - M(w0, i, j, km)) * m.dzfi[km]) * 2. * m.dzhi[k];
        } else {
          M(wp, i, j, k) = M(wp, i, j, k)
            + (nu * ((M(w0, i + 1, j, k) - M(w0, i, j, k)) * m.dxi + (M(u0, i + 1, j, k) - M(u0, i + 1, j, km)) * m.dzhi[k])
             - nu * ((M(w0, i, j, k) - M(w0, i - 1, j, k)) * m.dxi + (M(u0, i, j, k) - M(u0, i, j, km)) * m.dzhi[k])) * m.dxi
            + (nu * ((M(w0, i, jp, k) - M(w0, i, j, k)) * m.dyi + (M(v0, i, jp, k) - M(v0, i, jp, km)) * m.dzhi[k])
             - nu * ((M(w0, i, j, k) - M(w0, i, jm, k)) * m.dyi + (M(v0, i, j, k) - M(v0, i, j, km)) * m.dzhi[k])) * m.dyi
            + (nu * (M(w0, i, j, kp) - M(w0, i, j, k)) * m.dzfi[k]
             - nu * (M(w0, i, j, k) - M(w0, i, j, km)) * m.dzfi[km]) * 2. * m.dzhi[k];
        }
      }
  metrics_free(&m);
}

/* src/modsubgrid.f90:540-623 for a halo-2 scalar (ekh is a halo-1 array) */
void orc_diffc(const orc_grid *g, const double *c, const double *ekh, double *cp) {
  metrics m; metrics_init(g, &m);
  const double *dzf = g->dzf;
  const double cekh = g->numol * g->prandtlmoli;
  for (int k = 1; k <= g->nz; ++k)
    for (int j = 1; j <= g->ny; ++j)
      for (int i = 1; i <= g->nx; ++i) {
        int kp = k + 1, km = k - 1, jp = j + 1, jm = j - 1, ip = i + 1, im = i - 1;
        if (g->sgs != 0) {
          C(cp, i, j, k) = C(cp, i, j, k) + 0.5 * (
              ((M(ekh, ip, j, k) + M(ekh, i, j, k)) * (C(c, ip, j, k) - C(c, i, j, k))
             - (M(ekh, i, j, k) + M(ekh, im, j, k)) * (C(c, i, j, k) - C(c, im, j, k))) * m.dx2i
            + ((M(ekh, i, jp, k) + M(ekh, i, j, k)) * (C(c, i, jp, k) - C(c, i, j, k))
             - (M(ekh, i, j, k) + M(ekh, i, jm, k)) * (C(c, i, j, k) - C(c, i, jm, k))) * m.dy2i
            + ((dzf[kp] * M(ekh, i, j, k) + dzf[k] * M(ekh, i, j, kp)) * (C(c, i, j, kp) - C(c, i, j, k)) * m.dzh2i[kp]
             - (dzf[km] * M(ekh, i, j, k) + dzf[k] * M(ekh, i, j, km)) * (C(c, i, j, k) - C(c, i, j, km)) * m.dzh2i[k]) * m.dzfi[k]);
        } else {
          C(cp, i, j, k) = C(cp, i, j, k) + (
              (cekh * (C(c, ip, j, k) - C(c, i, j, k)) - cekh * (C(c, i, j, k) - C(c, im, j, k))) * m.dx2i
            + (cekh * (C(c, i, jp, k) - C(c, i, j, k)) - cekh * (C(c, i, j, k) - C(c, i, jm, k))) * m.dy2i
            + (cekh * (C(c, i, j, kp) - C(c, i, j, k)) * m.dzhi[kp]
             - cekh * (C(c, i, j, k) - C(c, i, j, km)) * m.dzhi[k]) * m.dzfi[k]);
        }
      }
  metrics_free(&m);
}

/* ====================================================================== forces */
/* src/modforces.f90:84-127, lbuoyancy = .false. */
void orc_forces(const orc_grid *g, const double *dpdxl, const double *dpdyl,
                double *up, double *vp, double *wp) {
  for (int k = 1; k <= g->nz; ++k)
    for (int j = 1; j <= g->ny; ++j)
      for (int i = 1; i <= g->nx; ++i) {
        M(up, i, j, k) = M(up, i, j, k) - dpdxl[k];
        M(vp, i, j, k) = M(vp, i, j, k) - dpdyl[k];
      }
  for (int j = 1; j <= g->ny; ++j)
    for (int i = 1; i <= g->nx; ++i) M(wp, i, j, 1) = 0.0;
}

/* ====================================================================== pressure */

/* The open lid, BCtopm = 3 (BCtopm_pressure, src/modglobal.f90:142): w(ke+1) "can vary according to pressure gradient".
 * bcpup's branch src/modboundary.f90:1234-1243: pres0ij = avexy_ibm(pres0) over the fluid c cells (src/modmpi.f90:623-664),
 * pwp(ke+1) = wm(ke+1) rk3coefi + 2 pres0ij(ke) dzhi(ke+1), and wp(ke+1) is OVERWRITTEN with pwp(ke+1) - wm(ke+1) rk3coefi.
 * orc_fillps needs pres0 and a writable wp for it: orc_set_lid hands them over (orc_substep does; NULL, NULL = off). */
static const double *lid_pres0 = NULL;
static double *lid_wp = NULL;
void orc_set_lid(const double *pres0, double *wp) { lid_pres0 = pres0; lid_wp = wp; }
static const double *ibm_ctx_mask(int grid);
/* avexy_ibm's mean of level k over the fluid c cells (all cells without an immersed boundary) */
static double lid_level_mean(const orc_grid *g, const double *f, int k) {
  const double *mask = ibm_ctx_mask(3);
  double s = 0., c = 0.;
  for (int j = 1; j <= g->ny; ++j)
    for (int i = 1; i <= g->nx; ++i) {
      const double w = mask ? (M(mask, i, j, k) > 0.5 ? 1. : 0.) : 1.;
      s += M(f, i, j, k) * w; c += w;
    }
  return c > 0. ? s / c : -999.;
}
void orc_bcpup_lid(const orc_grid *g, double rk3coef, const double *pres0, const double *wm, double *wp, double *pwp) {
  const int nx = g->nx, ny = g->ny, nz = g->nz;
  const double rk3coefi = 1. / rk3coef, dzhi = 1. / g->dzh[nz + 1];
  const double pres0ij = lid_level_mean(g, pres0, nz);
  for (int j = 1; j <= ny; ++j)
    for (int i = 1; i <= nx; ++i) {
      M(pwp, i, j, 1) = 0.;
      M(pwp, i, j, nz + 1) = M(wm, i, j, nz + 1) * rk3coefi + 2 * pres0ij * dzhi;
      M(wp, i, j, nz + 1) = M(pwp, i, j, nz + 1) - M(wm, i, j, nz + 1) * rk3coefi;
    }
}
/* tderive's branch src/modpois.f90:1058-1069: wp(ke+1) += 2 pij(ke) dzhi(ke+1), pij = avexy_ibm(p) */
void orc_tderive_lid(const orc_grid *g, const double *p, double *wp) {
  const int nz = g->nz;
  const double pij = lid_level_mean(g, p, nz), dzhi = 1. / g->dzh[nz + 1];
  for (int i = 1; i <= g->nx; ++i)
    for (int j = 1; j <= g->ny; ++j) M(wp, i, j, nz + 1) = M(wp, i, j, nz + 1) + 2 * pij * dzhi;
}

/* fillps src/modpois.f90:911-998 + bcpup src/modboundary.f90:1191-1341
 * (free-slip / no-slip top, or the open lid after orc_set_lid; periodic x and y on one rank) */
/* bcpup's BCxm_profile branch reads u0 and overwrites up(ib), up(ie+1): orc_set_open_x_fields hands them to orc_fillps */
static const double *xo_u0 = NULL;
static double *xo_up = NULL;
void orc_set_open_x_fields(const double *u0, double *up) { xo_u0 = u0; xo_up = up; }
void orc_fillps(const orc_grid *g, double rk3coef, const double *up, const double *vp,
                const double *wp, const double *um, const double *vm, const double *wm,
                double *pup, double *pvp, double *pwp, double *p) {
  const int nx = g->nx, ny = g->ny, nz = g->nz;
  const double rk3coefi = 1. / rk3coef;
  const double dxi = 1. / g->dx, dyi = 1. / g->dy;
  for (int k = 1; k <= nz; ++k)
    for (int j = 1; j <= ny; ++j)
      for (int i = 1; i <= nx; ++i) {
        M(pup, i, j, k) = M(up, i, j, k) + M(um, i, j, k) * rk3coefi;
        M(pvp, i, j, k) = M(vp, i, j, k) + M(vm, i, j, k) * rk3coefi;
        M(pwp, i, j, k) = M(wp, i, j, k) + M(wm, i, j, k) * rk3coefi;
      }
  for (int j = 1; j <= ny; ++j)
    for (int i = 1; i <= nx; ++i) { M(pwp, i, j, 1) = 0.; M(pwp, i, j, nz + 1) = 0.; }
  if (g->bctopm == 3 && lid_pres0 && lid_wp) orc_bcpup_lid(g, rk3coef, lid_pres0, wm, lid_wp, pwp);
  if (xo_on && xo_u0 && xo_up) {      /* src/modboundary.f90:1257-1280 */
    for (int k = 1; k <= nz; ++k)
      for (int j = 0; j <= ny + 1; ++j) { M(pup, 1, j, k) = xo_uprof[k] * rk3coefi; M(xo_up, 1, j, k) = 0.; }
    for (int k = 2; k <= nz; ++k)
      for (int j = 0; j <= ny + 1; ++j) {
        M(pup, nx + 1, j, k) = M(um, nx + 1, j, k) * rk3coefi - (M(xo_u0, nx + 1, j, k) - M(xo_u0, nx, j, k)) * dxi * xo_uouttot;
        M(xo_up, nx + 1, j, k) = M(pup, nx + 1, j, k) - M(um, nx + 1, j, k) * rk3coefi;
      }
    for (int j = 0; j <= ny + 1; ++j) {
      M(pup, nx + 1, j, 1) = M(pup, nx, j, 1);
      M(xo_up, nx + 1, j, 1) = M(pup, nx + 1, j, 1) - M(um, nx + 1, j, 1) * rk3coefi;
    }
  } else
  for (int k = 1; k <= nz; ++k)
    for (int j = 1; j <= ny; ++j) M(pup, nx + 1, j, k) = M(pup, 1, j, k);
  for (int k = 1; k <= nz; ++k)
    for (int i = 1; i <= nx; ++i) M(pvp, i, ny + 1, k) = M(pvp, i, 1, k);
  for (int k = 1; k <= nz; ++k) {
    double dzfi = 1. / g->dzf[k];
    for (int j = 1; j <= ny; ++j)
      for (int i = 1; i <= nx; ++i)
        M(p, i, j, k) = (M(pup, i + 1, j, k) - M(pup, i, j, k)) * dxi
                      + (M(pvp, i, j + 1, k) - M(pvp, i, j, k)) * dyi
                      + (M(pwp, i, j, k + 1) - M(pwp, i, j, k)) * dzfi;
  }
}

/* initpois (eigenvalues, tridiagonal coefficients) src/modpois.f90:99-220 and the
 * POISS_FFT2D branch of poisson :440-712 with solmpj :1107-1166.
 * Spectral coefficients stay in real arrays with half-complex ordering
 * [Re0, Re1, Im1, ..., Re(N/2)], exactly like the reference. */
/* &BC BCzp (src/modglobal.f90:159): 1 = tridiagonal solve in z (solmpj), 2 = cosine transform in z.  Set by orc_set_poisson_bczp. */
static int poisson_bczp = 1;
void orc_set_poisson_bczp(int bczp) { poisson_bczp = bczp; }
void orc_poisson_solve(const orc_grid *g, double *p) {
  const int nx = g->nx, ny = g->ny, nz = g->nz;
  const double dxi = 1. / g->dx, dyi = 1. / g->dy;
  /* all 1-based like the Fortran */
  double *xrt = (double *)calloc((size_t)nx + 2, sizeof(double));
  double *yrt = (double *)calloc((size_t)ny + 2, sizeof(double));
  double *a = (double *)calloc(3 * ((size_t)nz + 2), sizeof(double));
  double *b = a + nz + 2, *c = b + nz + 2;
  double fac = 1. / (2. * nx);
  for (int i = 3; i <= nx; i += 2) {
    double s = sin((double)(i - 1) * M_PI * fac);
    xrt[i - 1] = -4. * dxi * dxi * (s * s);
    xrt[i] = xrt[i - 1];
  }
  xrt[1] = 0.; xrt[nx] = -4. * dxi * dxi;
  if (xo_on)      /* Neumann - Neumann, src/modpois.f90:113-117 */
    for (int i = 1; i <= nx; ++i) { double s = sin((double)(i - 1) * M_PI * (1. / (2. * nx))); xrt[i] = -4. * dxi * dxi * (s * s); }
  fac = 1. / (2. * ny);
  for (int j = 3; j <= ny; j += 2) {
    double s = sin((double)(j - 1) * M_PI * fac);
    yrt[j - 1] = -4. * dyi * dyi * (s * s);
    yrt[j] = yrt[j - 1];
  }
  yrt[1] = 0.; yrt[ny] = -4. * dyi * dyi;
  /* rhobf = rhobh = 1 (src/modfields.f90:571-572) */
  for (int k = 1; k <= nz; ++k) {
    a[k] = 1. / (g->dzf[k] * g->dzh[k]);
    c[k] = 1. / (g->dzf[k] * g->dzh[k + 1]);
    b[k] = -(a[k] + c[k]);
  }
  b[1] = b[1] + a[1];
  double b_top_N = b[nz] + c[nz];
  double b_top_D = b[nz] - c[nz];
  b[nz] = b_top_N;
  a[1] = 0.; c[nz] = 0.;

  size_t nn = (size_t)nx * ny * nz;
  double *w = (double *)malloc(sizeof(double) * 3 * nn);      /* w[i + nx*(j + ny*k)], 0-based */
  double *bxyzrt = w + nn, *d = bxyzrt + nn;
#define W(a_, i, j, k) a_[(size_t)((i) - 1) + (size_t)nx * ((size_t)((j) - 1) + (size_t)ny * (size_t)((k) - 1))]
  for (int k = 1; k <= nz; ++k)
    for (int j = 1; j <= ny; ++j)
      for (int i = 1; i <= nx; ++i) {
        W(w, i, j, k) = M(p, i, j, k);
        double xyzrt = 1. * (xrt[i] + yrt[j] + 0.);
        if (xyzrt == 0. && k == nz) W(bxyzrt, i, j, k) = b_top_D;
        else W(bxyzrt, i, j, k) = b[k] + xyzrt;
      }
  /* forward x : src/modpois.f90:478-490 */
  fft_ref_plan *px = fft_ref_plan_create(nx), *py = fft_ref_plan_create(ny);
  int nmax = nx > ny ? nx : ny;
  double *line = (double *)malloc(sizeof(double) * (3 * (size_t)nmax + 8));
  double *spec = line + nmax;
  fac = 1. / sqrt(nx * 1.);
  if (xo_on) {      /* REDFT10, src/modpois.f90:492-500 */
    fac = 1. / sqrt(2. * nx);
    for (int k = 1; k <= nz; ++k)
      for (int j = 1; j <= ny; ++j) {
        for (int i = 1; i <= nx; ++i) line[i - 1] = W(w, i, j, k);
        fft_ref_redft10(nx, line, spec);
        for (int i = 1; i <= nx; ++i) W(w, i, j, k) = spec[i - 1] * fac;
      }
  } else
  for (int k = 1; k <= nz; ++k)
    for (int j = 1; j <= ny; ++j) {
      for (int i = 1; i <= nx; ++i) line[i - 1] = W(w, i, j, k);
      fft_ref_r2c(px, line, spec);
      W(w, 1, j, k) = spec[0];
      for (int i = 1; i <= nx / 2 - 1; ++i) { W(w, 2 * i, j, k) = spec[2 * i]; W(w, 2 * i + 1, j, k) = spec[2 * i + 1]; }
      W(w, nx, j, k) = spec[2 * (nx / 2)];
      for (int i = 1; i <= nx; ++i) W(w, i, j, k) = W(w, i, j, k) * fac;
    }
  /* forward y : :522-534 */
  fac = 1. / sqrt(ny * 1.);
  for (int i = 1; i <= nx; ++i)
    for (int k = 1; k <= nz; ++k) {
      for (int j = 1; j <= ny; ++j) line[j - 1] = W(w, i, j, k);
      fft_ref_r2c(py, line, spec);
      W(w, i, 1, k) = spec[0];
      for (int j = 1; j <= ny / 2 - 1; ++j) { W(w, i, 2 * j, k) = spec[2 * j]; W(w, i, 2 * j + 1, k) = spec[2 * j + 1]; }
      W(w, i, ny, k) = spec[2 * (ny / 2)];
      for (int j = 1; j <= ny; ++j) W(w, i, j, k) = W(w, i, j, k) * fac;
    }
  if (poisson_bczp == 2) {
    /* BCzp = 2, src/modpois.f90:179-191, 559-590: cosine transform in z (equidistant levels), division by the eigenvalues
     * xyzrt = xrt + yrt + zrt with zrt(k) = -4 dzi^2 sin^2((k-1) pi / (2 ktot)); the modes with xyzrt = 0 are set to zero */
    const double dzi = 1. / g->dzf[1];
    double *col = (double *)malloc(sizeof(double) * 2 * (size_t)nz), *tr = col + nz;
    const double fz = 1. / sqrt(2. * nz);
    for (int i = 1; i <= nx; ++i)
      for (int j = 1; j <= ny; ++j) {
        for (int k = 1; k <= nz; ++k) col[k - 1] = W(w, i, j, k);
        fft_ref_redft10(nz, col, tr);
        for (int k = 1; k <= nz; ++k) {
          double sz = sin((double)(k - 1) * M_PI * (1. / (2. * nz)));
          double zrt = k == 1 ? 0. : -4. * dzi * dzi * (sz * sz);
          double xyzrt = 1. * (xrt[i] + yrt[j] + zrt);
          double v = tr[k - 1] * fz;
          col[k - 1] = xyzrt != 0. ? v / xyzrt : 0.;
        }
        fft_ref_redft01(nz, col, tr);
        for (int k = 1; k <= nz; ++k) W(w, i, j, k) = tr[k - 1] * fz;
      }
    free(col);
  } else {
  /* solmpj : :1107-1166 */
  for (int j = 1; j <= ny; ++j)
    for (int i = 1; i <= nx; ++i) {
      double z = 1. / W(bxyzrt, i, j, 1);
      W(d, i, j, 1) = c[1] * z;
      W(w, i, j, 1) = W(w, i, j, 1) * z;
    }
  for (int k = 2; k <= nz - 1; ++k)
    for (int j = 1; j <= ny; ++j)
      for (int i = 1; i <= nx; ++i) {
        double bbk = W(bxyzrt, i, j, k);
        double z = 1. / (bbk - a[k] * W(d, i, j, k - 1));
        W(d, i, j, k) = c[k] * z;
        W(w, i, j, k) = (W(w, i, j, k) - a[k] * W(w, i, j, k - 1)) * z;
      }
  {
    double ak = a[nz];
    for (int j = 1; j <= ny; ++j)
      for (int i = 1; i <= nx; ++i) {
        double bbk = W(bxyzrt, i, j, nz);
        double z = bbk - ak * W(d, i, j, nz - 1);
        W(w, i, j, nz) = (W(w, i, j, nz) - ak * W(w, i, j, nz - 1)) / z;
      }
  }
  for (int k = nz - 1; k >= 1; --k)
    for (int j = 1; j <= ny; ++j)
      for (int i = 1; i <= nx; ++i) W(w, i, j, k) = W(w, i, j, k) - W(d, i, j, k) * W(w, i, j, k + 1);
  }
  /* backward y : :615-625 */
  fac = 1. / sqrt(ny * 1.);
  for (int i = 1; i <= nx; ++i)
    for (int k = 1; k <= nz; ++k) {
      spec[0] = W(w, i, 1, k); spec[1] = 0.;
      for (int j = 1; j <= ny / 2 - 1; ++j) { spec[2 * j] = W(w, i, 2 * j, k); spec[2 * j + 1] = W(w, i, 2 * j + 1, k); }
      spec[2 * (ny / 2)] = W(w, i, ny, k); spec[2 * (ny / 2) + 1] = 0.;
      fft_ref_c2r(py, spec, line);
      for (int j = 1; j <= ny; ++j) W(w, i, j, k) = line[j - 1] * fac;
    }
  /* backward x : :669-679 */
  fac = 1. / sqrt(nx * 1.);
  if (xo_on) {      /* REDFT01, src/modpois.f90:689-697 */
    fac = 1. / sqrt(2. * nx);
    for (int k = 1; k <= nz; ++k)
      for (int j = 1; j <= ny; ++j) {
        for (int i = 1; i <= nx; ++i) spec[i - 1] = W(w, i, j, k);
        fft_ref_redft01(nx, spec, line);
        for (int i = 1; i <= nx; ++i) W(w, i, j, k) = line[i - 1] * fac;
      }
  } else
  for (int k = 1; k <= nz; ++k)
    for (int j = 1; j <= ny; ++j) {
      spec[0] = W(w, 1, j, k); spec[1] = 0.;
      for (int i = 1; i <= nx / 2 - 1; ++i) { spec[2 * i] = W(w, 2 * i, j, k); spec[2 * i + 1] = W(w, 2 * i + 1, j, k); }
      spec[2 * (nx / 2)] = W(w, nx, j, k); spec[2 * (nx / 2) + 1] = 0.;
      fft_ref_c2r(px, spec, line);
      for (int i = 1; i <= nx; ++i) W(w, i, j, k) = line[i - 1] * fac;
    }
  for (int k = 1; k <= nz; ++k)
    for (int j = 1; j <= ny; ++j)
      for (int i = 1; i <= nx; ++i) M(p, i, j, k) = W(w, i, j, k);
#undef W
  free(line); free(w);
  fft_ref_plan_destroy(px); fft_ref_plan_destroy(py);
  free(xrt); free(yrt); free(a);
}

/* tderive src/modpois.f90:1001-1105 + bcp src/modboundary.f90:1344-1430 */
void orc_tderive(const orc_grid *g, double *p, double *up, double *vp, double *wp, double *pres0) {
  const int nx = g->nx, ny = g->ny, nz = g->nz;
  const double dxi = 1. / g->dx, dyi = 1. / g->dy;
  if (xo_on) {      /* bcp, src/modboundary.f90:1376-1394 (p and pres0; j = jb-1 .. je+1 of the rows as they are at this point) */
    for (int k = 1; k <= nz; ++k)
      for (int j = 0; j <= ny + 1; ++j) {
        M(p, 0, j, k) = M(p, 1, j, k); M(pres0, 0, j, k) = M(pres0, 1, j, k);
        M(p, nx + 1, j, k) = M(p, nx, j, k); M(pres0, nx + 1, j, k) = M(pres0, nx, j, k);
      }
  } else
  for (int j = 1; j <= ny; ++j)
    for (int k = 1; k <= nz; ++k) { M(p, 0, j, k) = M(p, nx, j, k); M(p, nx + 1, j, k) = M(p, 1, j, k); }
  for (int i = 1; i <= nx; ++i)
    for (int k = 1; k <= nz; ++k) { M(p, i, 0, k) = M(p, i, ny, k); M(p, i, ny + 1, k) = M(p, i, 1, k); }
  for (int i = 1; i <= nx; ++i)
    for (int j = 1; j <= ny; ++j) {
      M(up, i, j, 1) = M(up, i, j, 1) - (M(p, i, j, 1) - M(p, i - 1, j, 1)) * dxi;
      M(vp, i, j, 1) = M(vp, i, j, 1) - (M(p, i, j, 1) - M(p, i, j - 1, 1)) * dyi;
      for (int k = 2; k <= nz; ++k) {
        M(up, i, j, k) = M(up, i, j, k) - (M(p, i, j, k) - M(p, i - 1, j, k)) * dxi;
        M(vp, i, j, k) = M(vp, i, j, k) - (M(p, i, j, k) - M(p, i, j - 1, k)) * dyi;
        M(wp, i, j, k) = M(wp, i, j, k) - (M(p, i, j, k) - M(p, i, j, k - 1)) * (1. / g->dzh[k]);
      }
    }
  if (g->bctopm == 3) orc_tderive_lid(g, p, wp);
  for (int k = 0; k <= nz + 1; ++k)
    for (int j = 0; j <= ny + 1; ++j)
      for (int i = 0; i <= nx + 1; ++i) M(pres0, i, j, k) = M(pres0, i, j, k) + M(p, i, j, k);
}

/* ====================================================================== time step */
/* src/modtstep.f90:191-338 (loneeqn, ltempeq, lmoist false) */
void orc_tstep_integrate(const orc_grid *g, int rk3step, double dt, double *u0, double *v0,
                         double *w0, double *um, double *vm, double *wm, double *up, double *vp,
                         double *wp, double *sv0, double *svm, double *svp) {
  const double rk3coef = dt / (4. - (double)rk3step);
  const size_t n = msize(g), nc = csize(g);
  for (int k = 1; k <= g->nz; ++k)
    for (int j = 1; j <= g->ny; ++j)
      for (int i = 1; i <= g->nx; ++i) {
        M(u0, i, j, k) = M(um, i, j, k) + rk3coef * M(up, i, j, k);
        M(v0, i, j, k) = M(vm, i, j, k) + rk3coef * M(vp, i, j, k);
        M(w0, i, j, k) = M(wm, i, j, k) + rk3coef * M(wp, i, j, k);
        for (int s = 0; s < g->nsv; ++s) {
          const double *pm = svm + s * nc, *pp = svp + s * nc;
          double *p0 = sv0 + s * nc;
          C(p0, i, j, k) = C(pm, i, j, k) + rk3coef * C(pp, i, j, k);
        }
      }
  if (xo_on)                                                      /* src/modtstep.f90:262-264 */
    for (int k = 1; k <= g->nz; ++k)
      for (int j = 1; j <= g->ny; ++j) M(u0, g->nx + 1, j, k) = M(um, g->nx + 1, j, k) + rk3coef * M(up, g->nx + 1, j, k);
  if (g->bctopm == 3)                                             /* src/modtstep.f90:270-286 */
    for (int j = 1; j <= g->ny; ++j)
      for (int i = 1; i <= g->nx; ++i) M(w0, i, j, g->nz + 1) = M(wm, i, j, g->nz + 1) + rk3coef * M(wp, i, j, g->nz + 1);
  memset(up, 0, n * sizeof(double));
  memset(vp, 0, n * sizeof(double));
  memset(wp, 0, n * sizeof(double));
  if (g->nsv > 0) memset(svp, 0, (size_t)g->nsv * nc * sizeof(double));
  if (g->lchem && g->nsv >= 3 && rk3step == 3) orc_chem(g, dt, sv0);      /* src/modtstep.f90:236-238 */
  if (rk3step == 3) {
    memcpy(um, u0, n * sizeof(double));
    memcpy(vm, v0, n * sizeof(double));
    memcpy(wm, w0, n * sizeof(double));
    if (g->nsv > 0) memcpy(svm, sv0, (size_t)g->nsv * nc * sizeof(double));
  }
}

/* ====================================================================== halos, boundary */
/* xm_periodic then ym_periodic: src/modboundary.f90:508-539, 596-627 (over ALL j / ALL i) */
void orc_halos_m(const orc_grid *g, double *a) {
  const int nx = g->nx, ny = g->ny, nz = g->nz;
  if (!xo_on)      /* (BCxm = 2: no periodic refresh in x, src/modboundary.f90:95-100) */
  for (int k = 0; k <= nz + 1; ++k)
    for (int j = 0; j <= ny + 1; ++j) { M(a, 0, j, k) = M(a, nx, j, k); M(a, nx + 1, j, k) = M(a, 1, j, k); }
  for (int k = 0; k <= nz + 1; ++k)
    for (int i = 0; i <= nx + 1; ++i) { M(a, i, 0, k) = M(a, i, ny, k); M(a, i, ny + 1, k) = M(a, i, 1, k); }
}
/* xT_periodic / xq_periodic (:543-577) ahead of the y refresh: the temperature and the total water where they stay periodic in x beside an
 * inflow / outflow (BCxT = BCxq = 1 with BCxm = 2 / 3, the reference's defaults: `halos` calls them under `ibrank .and. ierank`, :95-100) */
static void halos_m_periodic_x(const orc_grid *g, double *a) {
  const int nx = g->nx, ny = g->ny, nz = g->nz;
  if (xo_on)
    for (int k = 0; k <= nz + 1; ++k)
      for (int j = 0; j <= ny + 1; ++j) { M(a, 0, j, k) = M(a, nx, j, k); M(a, nx + 1, j, k) = M(a, 1, j, k); }
  orc_halos_m(g, a);
}
/* xs_periodic, ys_periodic: :580-593, 670-685 (halo 2) */
void orc_halos_c(const orc_grid *g, double *a) {
  const int nx = g->nx, ny = g->ny, nz = g->nz;
  if (!xo_on)      /* (BCxs = 2 with BCxm = 2: no periodic refresh in x) */
  for (int m = 1; m <= 2; ++m)
    for (int k = -1; k <= nz + 2; ++k)
      for (int j = -1; j <= ny + 2; ++j) {
        C(a, 1 - m, j, k) = C(a, nx + 1 - m, j, k);
        C(a, nx + m, j, k) = C(a, m, j, k);
      }
  for (int m = 1; m <= 2; ++m)
    for (int k = -1; k <= nz + 2; ++k)
      for (int i = -1; i <= nx + 2; ++i) {
        C(a, i, 1 - m, k) = C(a, i, ny + 1 - m, k);
        C(a, i, ny + m, k) = C(a, i, m, k);
      }
}

static void top_row_m(const orc_grid *g, double *a, double val) {
  /* fluxtop with zero flux (:1494-1507) or valuetop (:1509-1519), whole padded xy plane */
  for (int j = 0; j <= g->ny + 1; ++j)
    for (int i = 0; i <= g->nx + 1; ++i) {
      if (g->bctopm == 2) M(a, i, j, g->nz + 1) = 2 * val - M(a, i, j, g->nz);
      else M(a, i, j, g->nz + 1) = M(a, i, j, g->nz);
    }
}

/* boundary: src/modboundary.f90:163-247 (w(kb)=0; top ghost rows; scalars zero-flux top
 * :1521-1537 with flux = 0, which adds exactly 0.0) */
/* anything other than the zero-flux copy orc_boundary already makes? */
static int scalar_top_active(const orc_grid *g) {
  if (g->bctops == 2) return 1;
  for (int s = 0; s < g->nsv && s < 4; ++s) if (g->wsvtop[s] != 0.) return 1;
  return 0;
}
void orc_scalar_tops(const orc_grid *g, const double *ekh, double *sv0, double *svm) {
  const size_t nc = csize(g);
  const int nz = g->nz;
  for (int s = 0; s < g->nsv; ++s) {
    double *p0 = sv0 + s * nc, *pm = svm + s * nc;
    for (int mm = 1; mm <= 2; ++mm)
      for (int j = 0; j <= g->ny + 1; ++j)
        for (int i = 0; i <= g->nx + 1; ++i) {
          if (g->bctops == 2) {
            C(p0, i, j, nz + mm) = 2 * g->sv_top[s] - C(p0, i, j, nz);
            C(pm, i, j, nz + mm) = 2 * g->sv_top[s] - C(pm, i, j, nz);
          } else {
            const double d = g->dzh[nz + 1] * g->wsvtop[s] /
                             ((1. / g->dzh[nz + 1]) * (0.5 * (g->dzf[nz] * M(ekh, i, j, nz + 1) + g->dzf[nz + 1] * M(ekh, i, j, nz))));
            C(p0, i, j, nz + mm) = C(p0, i, j, nz) + d;
            C(pm, i, j, nz + mm) = C(pm, i, j, nz) + d;
          }
        }
  }
}
void orc_boundary(const orc_grid *g, double *u0, double *v0, double *w0, double *um, double *vm,
                  double *wm, double *sv0, double *svm) {
  const int nx = g->nx, ny = g->ny, nz = g->nz;
  const size_t nc = csize(g);
  for (int j = 0; j <= ny + 1; ++j)
    for (int i = 0; i <= nx + 1; ++i) { M(wm, i, j, 1) = 0.; M(w0, i, j, 1) = 0.; }
  top_row_m(g, um, g->uinf); top_row_m(g, u0, g->uinf);
  top_row_m(g, vm, g->vinf); top_row_m(g, v0, g->vinf);
  if (g->bctopm != 3)                                             /* open lid: "w considered in modpois", src/modboundary.f90:191-200 */
    for (int j = 0; j <= ny + 1; ++j)
      for (int i = 0; i <= nx + 1; ++i) { M(w0, i, j, nz + 1) = 0.; M(wm, i, j, nz + 1) = 0.; }
  for (int s = 0; s < g->nsv; ++s) {
    double *p0 = sv0 + s * nc, *pm = svm + s * nc;
    for (int mm = 1; mm <= 2; ++mm)
      for (int j = 0; j <= ny + 1; ++j)      /* reference slices ib-ih:ie+ih, jb-jh:je+jh */
        for (int i = 0; i <= nx + 1; ++i) {
          C(p0, i, j, nz + mm) = C(p0, i, j, nz) + 0.0;
          C(pm, i, j, nz + mm) = C(pm, i, j, nz) + 0.0;
        }
  }
}

/* BCxm = 2: `boundary`'s lateral part, src/modboundary.f90:250-262, 376 -- xmi_profile (:688-706) then xmo_convective (:908-919) with
 * the substep's rk3coef (:914) and the outlet speed orc_set_open_x_uouttot / orc_set_open_x_outflow left */
void orc_boundary_open_x(const orc_grid *g, double rk3coef, double *u0, double *v0, double *w0, double *um, double *vm, double *wm) {
  const int nx = g->nx, ny = g->ny, nz = g->nz;
  const double dxi = 1. / g->dx;
  if (!xo_on) return;
  for (int j = 0; j <= ny + 1; ++j)
    for (int k = 1; k <= nz + 1; ++k) {
      M(u0, 1, j, k) = xo_uprof[k];
      M(um, 1, j, k) = xo_uprof[k];
      M(u0, 0, j, k) = 2 * M(u0, 1, j, k) - M(u0, 2, j, k);
      M(um, 0, j, k) = 2 * M(um, 1, j, k) - M(um, 2, j, k);
      M(v0, 0, j, k) = 2 * xo_vprof[k] - M(v0, 1, j, k);
      M(vm, 0, j, k) = 2 * xo_vprof[k] - M(vm, 1, j, k);
      M(w0, 0, j, k) = -M(w0, 1, j, k);
      M(wm, 0, j, k) = -M(wm, 1, j, k);
    }
  for (int k = 0; k <= nz + 1; ++k)
    for (int j = 0; j <= ny + 1; ++j) {
      M(v0, nx + 1, j, k) = M(v0, nx + 1, j, k) - (M(v0, nx + 1, j, k) - M(v0, nx, j, k)) * dxi * rk3coef * xo_uouttot;
      M(w0, nx + 1, j, k) = M(w0, nx + 1, j, k) - (M(w0, nx + 1, j, k) - M(w0, nx, j, k)) * dxi * rk3coef * xo_uouttot;
      M(vm, nx + 1, j, k) = M(vm, nx + 1, j, k) - (M(vm, nx + 1, j, k) - M(vm, nx, j, k)) * dxi * rk3coef * xo_uouttot;
      M(wm, nx + 1, j, k) = M(wm, nx + 1, j, k) - (M(wm, nx + 1, j, k) - M(wm, nx, j, k)) * dxi * rk3coef * xo_uouttot;
    }
}
/* ... and the temperature's, BCxT = 2: xTi_profile (src/modboundary.f90:766-793), xTo_convective (:947-957); thlprof [nz+2] by k */
static const double *xo_thlprof = NULL, *xo_qtprof = NULL;
void orc_set_open_x_thl(const double *thlprof) { xo_thlprof = thlprof; }
void orc_boundary_open_x_thl(const orc_grid *g, double rk3coef, double *thl0, double *thlm) {
  const int nx = g->nx, ny = g->ny, nz = g->nz;
  const double dxi = 1. / g->dx;
  if (!xo_on || !xo_thlprof) return;
  for (int j = 0; j <= ny + 1; ++j)
    for (int k = 1; k <= nz + 1; ++k) { M(thl0, 0, j, k) = xo_thlprof[k]; M(thlm, 0, j, k) = xo_thlprof[k]; }
  for (int j = 0; j <= ny + 1; ++j)
    for (int k = 1; k <= nz; ++k) { M(thl0, 1, j, k) = xo_thlprof[k]; M(thlm, 1, j, k) = xo_thlprof[k]; }
  for (int k = 0; k <= nz + 1; ++k)
    for (int j = 0; j <= ny + 1; ++j) {
      M(thl0, nx + 1, j, k) = M(thl0, nx + 1, j, k) - (M(thl0, nx + 1, j, k) - M(thl0, nx, j, k)) * dxi * rk3coef * xo_uouttot;
      M(thlm, nx + 1, j, k) = M(thlm, nx + 1, j, k) - (M(thlm, nx + 1, j, k) - M(thlm, nx, j, k)) * dxi * rk3coef * xo_uouttot;
    }
}
/* ... and the passive scalars', BCxs = 2: xsi_profile (src/modboundary.f90:844-861), xso_convective (:983-996); svprof [nsv][nz+2] by k */
/* BCxq = 2: xqi_profile (src/modboundary.f90:811-823: the ghost mirrored about qtprof) and xqo_convective (:961-971 -- which starts from
 * qt(ie), not from qt(ie+1) like its siblings: as it is) */
void orc_set_open_x_qt(const double *qtprof) { xo_qtprof = qtprof; }
void orc_boundary_open_x_qt(const orc_grid *g, double rk3coef, double *qt0, double *qtm) {
  const int nx = g->nx, ny = g->ny, nz = g->nz;
  const double dxi = 1. / g->dx;
  if (!xo_on || !xo_qtprof) return;
  for (int j = 0; j <= ny + 1; ++j)
    for (int k = 1; k <= nz + 1; ++k) {
      M(qt0, 0, j, k) = 2 * xo_qtprof[k] - M(qt0, 1, j, k);
      M(qtm, 0, j, k) = 2 * xo_qtprof[k] - M(qtm, 1, j, k);
    }
  for (int k = 0; k <= nz + 1; ++k)
    for (int j = 0; j <= ny + 1; ++j) {
      M(qt0, nx + 1, j, k) = M(qt0, nx, j, k) - (M(qt0, nx + 1, j, k) - M(qt0, nx, j, k)) * dxi * rk3coef * xo_uouttot;
      M(qtm, nx + 1, j, k) = M(qtm, nx, j, k) - (M(qtm, nx + 1, j, k) - M(qtm, nx, j, k)) * dxi * rk3coef * xo_uouttot;
    }
}
static const double *xo_svprof = NULL;
void orc_set_open_x_scalars(const double *svprof) { xo_svprof = svprof; }
void orc_boundary_open_x_sv(const orc_grid *g, double rk3coef, double *sv0, double *svm) {
  const int nx = g->nx, ny = g->ny, nz = g->nz;
  const double dxi = 1. / g->dx;
  const size_t nc = csize(g);
  if (!xo_on || !xo_svprof) return;
  for (int n = 0; n < g->nsv; ++n) {
    double *p0 = sv0 + n * nc, *pm = svm + n * nc;
    const double *prof = xo_svprof + (size_t)n * (nz + 2);
    for (int j = 1; j <= ny; ++j)
      for (int k = 1; k <= nz + 1; ++k)
        for (int m = 1; m <= 2; ++m) {
          C(p0, 1 - m, j, k) = 2 * prof[k] - C(p0, 1 - m + 1, j, k);
          C(pm, 1 - m, j, k) = 2 * prof[k] - C(pm, 1 - m + 1, j, k);
        }
    for (int k = -1; k <= nz + 2; ++k)
      for (int j = -1; j <= ny + 2; ++j) {
        C(p0, nx + 1, j, k) = C(p0, nx + 1, j, k) - (C(p0, nx + 1, j, k) - C(p0, nx, j, k)) * dxi * rk3coef * xo_uouttot;
        C(pm, nx + 1, j, k) = C(pm, nx + 1, j, k) - (C(pm, nx + 1, j, k) - C(pm, nx, j, k)) * dxi * rk3coef * xo_uouttot;
      }
  }
}
/* uouttot without a prescribed volume flow (src/modboundary.f90:143-156): sum_k u0av(k) dzf(k) / (zh(ke+1) - zh(kb+1)), u0av = diagfld's
 * slab average over the fluid u points (src/modthermodynamics.f90:271) of the state the substep starts from.  wlev[nz]: those weights
 * (NULL: uouttot stays what orc_set_open_x_uouttot said); hold_first: the next refresh is skipped -- the first substep's `boundary` still
 * reads the u0av the start-up formed before ITS `boundary` (src/modstartup.f90:1604, src/program.f90:118) */
static const double *xo_wlev = NULL;
static int xo_hold = 0;
void orc_set_open_x_outflow(const double *wlev, double uouttot, int hold_first) { xo_wlev = wlev; xo_uouttot = uouttot; xo_hold = hold_first; }
static double xo_outlet_speed(const orc_grid *g, const double *u0) {
  const double *mask = ibm_ctx_mask(0);
  double u = 0.;
  for (int k = 1; k <= g->nz; ++k) {
    double s = 0., c = 0.;
    for (int j = 1; j <= g->ny; ++j)
      for (int i = 1; i <= g->nx; ++i) {
        const double w = mask ? (M(mask, i, j, k) > 0.5 ? 1. : 0.) : 1.;
        s += M(u0, i, j, k) * w; c += w;
      }
    u = u + (c > 0. ? s / c : -999.) * xo_wlev[k - 1];
  }
  return u;
}

/* ====================================================================== floor wall function */
/* `bottom` (src/modibm.f90:1998-2100), lbottom branch with BCbotm = 3 -> wfmneutral case 91
 * (src/modwallfunctions.f90:263-350); scalars BCbots = 1 (src/modibm.f90:2073-2090).
 * dxf/dxhi follow src/modglobal.f90:770-790 (equal to dx up to the rounding of (i-1)*dx). */
void orc_bottom(const orc_grid *g, const double *u0, const double *v0, const double *ekm, const double *ekh,
                const double *sv0, double *up, double *vp, double *svp, double *momfluxb) {
  if (!g->lbottom) return;
  metrics m;
  metrics_init(g, &m);
  const int nx = g->nx, ny = g->ny;
  const double *dzf = g->dzf;
  const double fkar = 0.41;                       /* src/modglobal.f90:317 */
  const double fkar2 = fkar * fkar;
  const double umin = 0.0001;                     /* src/modwallfunctions.f90:286 */
  double *xh = (double *)calloc(4 * (size_t)(nx + 3), sizeof(double));   /* Fortran i = 0..nx+1 */
  double *xf = xh + (nx + 3), *dxf = xf + (nx + 3), *dxhi = dxf + (nx + 3);
  for (int i = 1; i <= nx + 1; ++i) { xh[i] = (double)(i - 1) * g->dx; xf[i] = xh[i] + g->dx / 2; }
  for (int i = 1; i <= nx; ++i) dxf[i] = xh[i + 1] - xh[i];
  dxf[nx + 1] = dxf[nx]; dxf[0] = dxf[1];
  dxhi[1] = 1. / (2 * xf[1]);
  for (int i = 2; i <= nx + 1; ++i) dxhi[i] = 1. / (xf[i] - xf[i - 1]);

  const int k = 1, km = 0;
  const double delta = 0.5 * dzf[k];
  const double l_ = log(delta / g->z0);
  const double logdz2 = l_ * l_;
  for (int j = 1; j <= ny; ++j)        /* u component, :318-331 */
    for (int i = 1; i <= nx; ++i) {
      const double utang1Int = M(u0, i, j, k);
      const double utang2Int = (M(v0, i, j, k) + M(v0, i - 1, j, k) + M(v0, i, j + 1, k) + M(v0, i - 1, j + 1, k)) * 0.25;
      const double utangInt = fmax(umin, (utang1Int * utang1Int + utang2Int * utang2Int));
      const double ctm = fkar2 / (logdz2);
      const double dummy = fabs(utang1Int) * sqrt(utangInt) * ctm;
      const double bcmomflux = copysign(dummy, utang1Int);
      if (momfluxb) M(momfluxb, i, j, k) = M(momfluxb, i, j, k) + bcmomflux * m.dzfi[k];
      const double emom = (dzf[km] * (M(ekm, i, j, k) * dxf[i - 1] + M(ekm, i - 1, j, k) * dxf[i]) +
                           dzf[k] * (M(ekm, i, j, km) * dxf[i - 1] + M(ekm, i - 1, j, km) * dxf[i])) * dxhi[i] * m.dzhiq[k];
      M(up, i, j, k) = M(up, i, j, k) + (M(u0, i, j, k) - M(u0, i, j, km)) * emom * m.dzhi[k] * m.dzfi[k] - bcmomflux * m.dzfi[k];
    }
  for (int j = 1; j <= ny; ++j)        /* v component, :333-346 */
    for (int i = 1; i <= nx; ++i) {
      const double utang1Int = (M(u0, i, j, k) + M(u0, i, j - 1, k) + M(u0, i + 1, j - 1, k) + M(u0, i + 1, j, k)) * 0.25;
      const double utang2Int = M(v0, i, j, k);
      const double utangInt = fmax(umin, (utang1Int * utang1Int + utang2Int * utang2Int));
      const double ctm = fkar2 / (logdz2);
      const double dummy = fabs(utang2Int) * sqrt(utangInt) * ctm;
      const double bcmomflux = copysign(dummy, utang2Int);
      if (momfluxb) M(momfluxb, i, j, k) = M(momfluxb, i, j, k) + bcmomflux * m.dzfi[k];
      const double eomm = (dzf[km] * (M(ekm, i, j, k) + M(ekm, i, j - 1, k)) + dzf[k] * (M(ekm, i, j, km) + M(ekm, i, j - 1, km))) * m.dzhiq[k];
      M(vp, i, j, k) = M(vp, i, j, k) + (M(v0, i, j, k) - M(v0, i, j, km)) * eomm * m.dzhi[k] * m.dzfi[k] - bcmomflux * m.dzfi[k];
    }
  const size_t nc = csize(g);          /* zero-flux floor for the scalars */
  for (int n = 0; n < g->nsv; ++n) {
    const double *c0 = sv0 + n * nc;
    double *cp = svp + n * nc;
    for (int j = 1; j <= ny; ++j)
      for (int i = 1; i <= nx; ++i)
        C(cp, i, j, 1) = C(cp, i, j, 1) + (0.5 * (dzf[0] * M(ekh, i, j, 1) + dzf[1] * M(ekh, i, j, 0))
                                           * (C(c0, i, j, 1) - C(c0, i, j, 0)) * m.dzh2i[1] + 0.) * m.dzfi[1];
  }
  free(xh);
  metrics_free(&m);
}

/* ====================================================================== chemistry */
void orc_chem(const orc_grid *g, double dt, double *sv0) {
  const size_t nc = csize(g);
  double *a = sv0, *b = sv0 + nc, *c = sv0 + 2 * nc;
  const double k1 = g->k1, J = g->JNO2;
  for (int k = 1; k <= g->nz + 2; ++k)
    for (int j = -1; j <= g->ny + 2; ++j)
      for (int i = -1; i <= g->nx + 2; ++i) {
        const double dNO = 1 * C(a, i, j, k) / 30.006, dNO2 = 1 * C(b, i, j, k) / 46.005, dO3 = 1 * C(c, i, j, k) / 47.997;
        const double r = (dt * (-k1 * dNO * dO3 + J * dNO2)) / (1. + ((dNO + dO3) * k1 + J) * dt);
        C(a, i, j, k) = 30.006 * ((C(a, i, j, k) / 30.006) + r);
        C(b, i, j, k) = 46.005 * ((C(b, i, j, k) / 46.005) - r);
        C(c, i, j, k) = 47.997 * ((C(c, i, j, k) / 47.997) + r);
      }
}

/* ====================================================================== scalar sources */
void orc_scalsource(const orc_grid *g, int npoint, const double *points, int nline, const double *lines, double *cp) {
  const double pi = 3.141592653589793116;                 /* src/modglobal.f90:270 */
  const double dx = g->dx, dy = g->dy, dxi = 1. / dx, dyi = 1. / dy;
  metrics m; metrics_init(g, &m);
  for (int ns = 0; ns < npoint; ++ns) {                   /* :392-415 */
    const double xS = points[5 * ns], yS = points[5 * ns + 1], zS = points[5 * ns + 2], SS = points[5 * ns + 3], sigS = points[5 * ns + 4];
    for (int k = 1; k <= g->nz; ++k)
      for (int j = 1; j <= g->ny; ++j)
        for (int i = 1; i <= g->nx; ++i) {
          const double ax = (i - 0.5) * dx - xS, ay = (j - 0.5) * dy - yS, az = g->zf[k] - zS;
          const double ra2 = ax * ax + ay * ay + az * az;
          if (ra2 <= 9 * (sigS * sigS))
            C(cp, i, j, k) = C(cp, i, j, k) + dxi * dyi * m.dzfi[k] * SS * exp(-ra2 / (2 * (sigS * sigS)));
        }
  }
  for (int ns = 0; ns < nline; ++ns) {                    /* :420-478 */
    const double *r = lines + 8 * ns;
    const double xSb = r[0], ySb = r[1], zSb = r[2], xSe = r[3], ySe = r[4], zSe = r[5], SS = r[6], sigS = r[7];
    const double lsx = xSe - xSb, lsy = ySe - ySb, lsz = zSe - zSb;
    for (int k = 1; k <= g->nz; ++k)
      for (int j = 1; j <= g->ny; ++j)
        for (int i = 1; i <= g->nx; ++i) {
          const double px = (i - 0.5) * dx, py = (j - 0.5) * dy, pz = g->zf[k];
          const double vx = px - xSb, vy = py - ySb, vz = pz - zSb;
          const double dot = (vx * lsx + vy * lsy + vz * lsz) / (lsx * lsx + lsy * lsy + lsz * lsz);
          double ra2;
          if (dot < 0.0) ra2 = (px - xSb) * (px - xSb) + (py - ySb) * (py - ySb) + (pz - zSb) * (pz - zSb);
          else if (dot > 1.0) ra2 = (px - xSe) * (px - xSe) + (py - ySe) * (py - ySe) + (pz - zSe) * (pz - zSe);
          else {
            const double qx = px - (xSb + dot * lsx), qy = py - (ySb + dot * lsy), qz = pz - (zSb + dot * lsz);
            ra2 = qx * qx + qy * qy + qz * qz;
          }
          if (ra2 <= 9 * (sigS * sigS))
            C(cp, i, j, k) = C(cp, i, j, k) + dxi * dyi * m.dzfi[k] *
                             sqrt(2.0 * pi) * SS * sigS * exp(-ra2 / (2 * (sigS * sigS))) * erf(sqrt((9 * (sigS * sigS) - ra2) / (2 * (sigS * sigS))));
        }
  }
  metrics_free(&m);
}

/* ---- wfuno, src/modwallfunctions.f90:24-170: Louis (1979) / Uno et al. (1995) transfer coefficients over a rough wall */
static void uno_F(double logdz, double sqdz, double Ri, double fkar2, double *Fm, double *Fh) {
  const double b1 = 9.4, b2 = 4.7, dm = 7.4, dh = 5.3;                                  /* :184-187 */
  if (Ri > 0.) { *Fm = 1. / ((1. + b2 * Ri) * (1. + b2 * Ri)); *Fh = *Fm; }             /* Eq. 4, stable */
  else {
    const double cm = (dm * fkar2) / (logdz * logdz) * b1 * sqdz, ch = (dh * fkar2) / (logdz * logdz) * b1 * sqdz;   /* Eq. 5 */
    *Fm = 1. - (b1 * Ri) / (1. + cm * sqrt(fabs(Ri)));                                   /* Eq. 3 */
    *Fh = 1. - (b1 * Ri) / (1. + ch * sqrt(fabs(Ri)));
  }
}
/* unom, :224-261 */
static double uno_m(double prt, double logdz, double logzh, double sqdz, double Ribl0, double fkar2) {
  double Fm, Fh;
  uno_F(logdz, sqdz, Ribl0, fkar2, &Fm, &Fh);
  const double Mm = prt * logdz * sqrt(Fm) / Fh;                                        /* Eq. 14 */
  const double Ribl1 = Ribl0 - Ribl0 * prt * logzh / (prt * logzh + Mm);                 /* Eq. 17 */
  uno_F(logdz, sqdz, Ribl1, fkar2, &Fm, &Fh);
  return fkar2 / (logdz * logdz) * Fm;                                                   /* Eq. 7 */
}
/* unoh, :176-220: returns the heat flux otf */
static double uno_h(double prt, double logdz, double logzh, double sqdz, double utangInt, double dT, double Ribl0, double fkar2) {
  double Fm, Fh;
  uno_F(logdz, sqdz, Ribl0, fkar2, &Fm, &Fh);
  double Mm = prt * logdz * sqrt(Fm) / Fh;
  const double Ribl1 = Ribl0 - Ribl0 * prt * logzh / (prt * logzh + Mm);
  uno_F(logdz, sqdz, Ribl1, fkar2, &Fm, &Fh);
  Mm = prt * logdz * sqrt(Fm) / Fh;
  const double dTrough = dT * 1. / (prt * logzh / Mm + 1.);                              /* Eq. 13a */
  const double octh = sqrt(utangInt) * fkar2 / (logdz * logdz) * Fh / prt;               /* Eq. 8 */
  return octh * dTrough;
}
void orc_bottom_uno(const orc_grid *g, const double *u0, const double *v0, const double *thl0, const double *ekm, const double *ekh,
                    double *up, double *vp, double *thlp) {
  if (!g->lbottom) return;
  metrics m;
  metrics_init(g, &m);
  const int nx = g->nx, ny = g->ny, k = 1, km = 0;
  const double *dzf = g->dzf;
  const double fkar2 = 0.41 * 0.41, umin = 0.0001, grav = 9.81, Twall = g->thls, prt = g->prandtlturb;
  double *xh = (double *)calloc(4 * (size_t)(nx + 3), sizeof(double));   /* dxf, dxhi as src/modglobal.f90 builds them */
  double *xf = xh + (nx + 3), *dxf = xf + (nx + 3), *dxhi = dxf + (nx + 3);
  for (int i = 1; i <= nx + 1; ++i) { xh[i] = (double)(i - 1) * g->dx; xf[i] = xh[i] + g->dx / 2; }
  for (int i = 1; i <= nx; ++i) dxf[i] = xh[i + 1] - xh[i];
  dxf[nx + 1] = dxf[nx]; dxf[0] = dxf[1];
  dxhi[1] = 1. / (2 * xf[1]);
  for (int i = 2; i <= nx + 1; ++i) dxhi[i] = 1. / (xf[i] - xf[i - 1]);
  const double delta = 0.5 * dzf[k];
  const double logdz = log(delta / g->z0), logzh = log(g->z0 / g->z0h), sqdz = sqrt(delta / g->z0);
  if (g->bcbotm == 2) {
    for (int j = 1; j <= ny; ++j)        /* case 91, u component, :92-109 */
      for (int i = 1; i <= nx; ++i) {
        const double utang1Int = M(u0, i, j, k);
        const double utang2Int = (M(v0, i, j, k) + M(v0, i - 1, j, k) + M(v0, i, j + 1, k) + M(v0, i - 1, j + 1, k)) * 0.25;
        const double utangInt = fmax(umin, (utang1Int * utang1Int + utang2Int * utang2Int));
        const double dT = ((M(thl0, i, j, k) + M(thl0, i - 1, j, k)) - (Twall + Twall)) * 0.5;
        const double Ribl0 = grav * delta * dT * 2 / ((Twall + Twall) * utangInt);
        const double ctm = uno_m(prt, logdz, logzh, sqdz, Ribl0, fkar2);
        const double bcmomflux = copysign(fabs(utang1Int) * sqrt(utangInt) * ctm, utang1Int);
        const double emom = (dzf[km] * (M(ekm, i, j, k) * dxf[i - 1] + M(ekm, i - 1, j, k) * dxf[i]) +
                             dzf[k] * (M(ekm, i, j, km) * dxf[i - 1] + M(ekm, i - 1, j, km) * dxf[i])) * dxhi[i] * m.dzhiq[k];
        M(up, i, j, k) = M(up, i, j, k) + (M(u0, i, j, k) - M(u0, i, j, km)) * emom * m.dzhi[k] * m.dzfi[k] - bcmomflux * m.dzfi[k];
      }
    for (int j = 1; j <= ny; ++j)        /* v component, :111-127 */
      for (int i = 1; i <= nx; ++i) {
        const double utang1Int = (M(u0, i, j, k) + M(u0, i, j - 1, k) + M(u0, i + 1, j - 1, k) + M(u0, i + 1, j, k)) * 0.25;
        const double utang2Int = M(v0, i, j, k);
        const double utangInt = fmax(umin, (utang1Int * utang1Int + utang2Int * utang2Int));
        const double dT = ((M(thl0, i, j, k) + M(thl0, i, j - 1, k)) - (Twall + Twall)) * 0.5;
        const double Ribl0 = grav * delta * dT * 2 / ((Twall + Twall) * utangInt);
        const double ctm = uno_m(prt, logdz, logzh, sqdz, Ribl0, fkar2);
        const double bcmomflux = copysign(fabs(utang2Int) * sqrt(utangInt) * ctm, utang2Int);
        const double eomm = (dzf[km] * (M(ekm, i, j, k) + M(ekm, i, j - 1, k)) + dzf[k] * (M(ekm, i, j, km) + M(ekm, i, j - 1, km))) * m.dzhiq[k];
        M(vp, i, j, k) = M(vp, i, j, k) + (M(v0, i, j, k) - M(v0, i, j, km)) * eomm * m.dzhi[k] * m.dzfi[k] - bcmomflux * m.dzfi[k];
      }
  }
  if (g->bcbott == 2 && thlp)            /* case 92, :131-165 */
    for (int j = 1; j <= ny; ++j)
      for (int i = 1; i <= nx; ++i) {
        const double utang1Int = (M(u0, i, j, k) + M(u0, i + 1, j, k)) * 0.5;
        const double utang2Int = (M(v0, i, j, k) + M(v0, i, j + 1, k)) * 0.5;
        const double utangInt = fmax(umin, (utang1Int * utang1Int + utang2Int * utang2Int));
        const double dT = (M(thl0, i, j, k) - Twall);
        const double Ribl0 = grav * delta * dT / (Twall * utangInt);
        const double bcTflux = uno_h(prt, logdz, logzh, sqdz, utangInt, dT, Ribl0, fkar2);
        M(thlp, i, j, k) = M(thlp, i, j, k) + 0.5 * (dzf[k - 1] * M(ekh, i, j, k) + dzf[k] * M(ekh, i, j, k - 1)) *
                           (M(thl0, i, j, k) - M(thl0, i, j, k - 1)) * m.dzh2i[k] * m.dzfi[k] - bcTflux * m.dzfi[k];
      }
  free(xh);
  metrics_free(&m);
}

/* ====================================================================== temperature equation (passive) */
/* advecc_2nd, src/modadvection.f90:103-155, on an m-array (halo 1: advecc_2nd(ih,jh,kh,thl0,thlp), :68) */
void orc_advecc_2nd(const orc_grid *g, const double *u0, const double *v0, const double *w0, const double *c, double *cp) {
  metrics m; metrics_init(g, &m);
  const double *dzf = g->dzf;
  for (int k = 1; k <= g->nz; ++k)
    for (int j = 1; j <= g->ny; ++j)
      for (int i = 1; i <= g->nx; ++i) {
        int ip = i + 1, im = i - 1, jp = j + 1, jm = j - 1;
        M(cp, i, j, k) = M(cp, i, j, k) - (
            (M(u0, ip, j, k) * (M(c, ip, j, k) + M(c, i, j, k)) - M(u0, i, j, k) * (M(c, im, j, k) + M(c, i, j, k))) * m.dxi5
          + (M(v0, i, jp, k) * (M(c, i, jp, k) + M(c, i, j, k)) - M(v0, i, j, k) * (M(c, i, jm, k) + M(c, i, j, k))) * m.dyi5);
      }
  for (int j = 1; j <= g->ny; ++j)
    for (int i = 1; i <= g->nx; ++i)
      for (int k = 1; k <= g->nz; ++k) {
        int kp = k + 1, km = k - 1;
        M(cp, i, j, k) = M(cp, i, j, k) - (
            M(w0, i, j, kp) * (M(c, i, j, kp) * dzf[k] + M(c, i, j, k) * dzf[kp]) * m.dzhi[kp]
          - M(w0, i, j, k) * (M(c, i, j, km) * dzf[k] + M(c, i, j, k) * dzf[km]) * m.dzhi[k]) * m.dzfi5[k];
      }
  metrics_free(&m);
}
/* diffc, src/modsubgrid.f90:540-623, on an m-array (diffc(ih,jh,kh,thl0,thlp), :146) */
void orc_diffc_m(const orc_grid *g, const double *c, const double *ekh, double *cp) {
  metrics m; metrics_init(g, &m);
  const double *dzf = g->dzf;
  const double cekh = g->numol * g->prandtlmoli;
  for (int k = 1; k <= g->nz; ++k)
    for (int j = 1; j <= g->ny; ++j)
      for (int i = 1; i <= g->nx; ++i) {
        int kp = k + 1, km = k - 1, jp = j + 1, jm = j - 1, ip = i + 1, im = i - 1;
        if (g->sgs != 0) {
          M(cp, i, j, k) = M(cp, i, j, k) + 0.5 * (
              ((M(ekh, ip, j, k) + M(ekh, i, j, k)) * (M(c, ip, j, k) - M(c, i, j, k))
             - (M(ekh, i, j, k) + M(ekh, im, j, k)) * (M(c, i, j, k) - M(c, im, j, k))) * m.dx2i
            + ((M(ekh, i, jp, k) + M(ekh, i, j, k)) * (M(c, i, jp, k) - M(c, i, j, k))
             - (M(ekh, i, j, k) + M(ekh, i, jm, k)) * (M(c, i, j, k) - M(c, i, jm, k))) * m.dy2i
            + ((dzf[kp] * M(ekh, i, j, k) + dzf[k] * M(ekh, i, j, kp)) * (M(c, i, j, kp) - M(c, i, j, k)) * m.dzh2i[kp]
             - (dzf[km] * M(ekh, i, j, k) + dzf[k] * M(ekh, i, j, km)) * (M(c, i, j, k) - M(c, i, j, km)) * m.dzh2i[k]) * m.dzfi[k]);
        } else {
          M(cp, i, j, k) = M(cp, i, j, k) + (
              (cekh * (M(c, ip, j, k) - M(c, i, j, k)) - cekh * (M(c, i, j, k) - M(c, im, j, k))) * m.dx2i
            + (cekh * (M(c, i, jp, k) - M(c, i, j, k)) - cekh * (M(c, i, j, k) - M(c, i, jm, k))) * m.dy2i
            + (cekh * (M(c, i, j, kp) - M(c, i, j, k)) * m.dzhi[kp]
             - cekh * (M(c, i, j, k) - M(c, i, j, km)) * m.dzhi[k]) * m.dzfi[k]);
        }
      }
  metrics_free(&m);
}
void orc_thl0c_from(const orc_grid *g, const double *thl0, double *thl0c) {
  memset(thl0c, 0, csize(g) * sizeof(double));
  for (int k = 1; k <= g->nz; ++k)
    for (int j = 1; j <= g->ny; ++j)
      for (int i = 1; i <= g->nx; ++i) C(thl0c, i, j, k) = M(thl0, i, j, k);
  orc_halos_c(g, thl0c);
  if (g->bctopt != 2)
    for (int n = 1; n <= 2; ++n)
      for (int j = -1; j <= g->ny + 2; ++j)
        for (int i = -1; i <= g->nx + 2; ++i) C(thl0c, i, j, g->nz + n) = C(thl0c, i, j, g->nz + n - 1);
}
void orc_advec_thl_kappa(const orc_grid *g, const double *u0, const double *v0, const double *w0, const double *thl0c, double *thlp) {
  double *pc = (double *)calloc(csize(g), sizeof(double));
  for (int k = 1; k <= g->nz; ++k)
    for (int j = 1; j <= g->ny; ++j)
      for (int i = 1; i <= g->nx; ++i) C(pc, i, j, k) = M(thlp, i, j, k);
  orc_advecc_kappa(g, u0, v0, w0, thl0c, pc);
  for (int k = 1; k <= g->nz; ++k)
    for (int j = 1; j <= g->ny; ++j)
      for (int i = 1; i <= g->nx; ++i) M(thlp, i, j, k) = C(pc, i, j, k);
  free(pc);
}
/* top condition of thl / qt: fluxtop (src/modboundary.f90:1494-1507) / valuetop (:1509-1519), BCtopT, BCtopq 1 / 2 */
static void scalar_top(const orc_grid *g, const double *ekh, double *a, int bctop, double wtop, double top_value) {
  const int ke = g->nz;
  const double eps1 = 1e-10;                      /* src/modglobal.f90:318 */
  for (int j = 0; j <= g->ny + 1; ++j)
    for (int i = 0; i <= g->nx + 1; ++i) {
      if (bctop == 2) M(a, i, j, ke + 1) = 2 * top_value - M(a, i, j, ke);
      else if (fabs(wtop) <= eps1) M(a, i, j, ke + 1) = M(a, i, j, ke);
      else M(a, i, j, ke + 1) = M(a, i, j, ke) + g->dzh[ke + 1] * wtop /
               ((1. / g->dzh[ke + 1]) * (0.5 * (g->dzf[ke] * M(ekh, i, j, ke + 1) + g->dzf[ke + 1] * M(ekh, i, j, ke))));
    }
}
void orc_thl_top(const orc_grid *g, const double *ekh, double *a) { scalar_top(g, ekh, a, g->bctopt, g->wttop, g->thl_top); }
/* src/modboundary.f90:222-231 */
void orc_qt_top(const orc_grid *g, const double *ekh, double *a) { scalar_top(g, ekh, a, g->bctopq, g->wqtop, g->qt_top); }
/* floor of qt in `bottom`, BCbotq = 1: src/modibm.f90:2050-2061 (the flux enters with a plus sign there) */
void orc_qt_floor(const orc_grid *g, const double *ekh, const double *qt0, double *qtp) {
  if (!g->lbottom) return;
  metrics m; metrics_init(g, &m);
  const double *dzf = g->dzf;
  for (int j = 1; j <= g->ny; ++j)
    for (int i = 1; i <= g->nx; ++i)
      M(qtp, i, j, 1) = M(qtp, i, j, 1) + (0.5 * (dzf[0] * M(ekh, i, j, 1) + dzf[1] * M(ekh, i, j, 0))
                                           * (M(qt0, i, j, 1) - M(qt0, i, j, 0)) * m.dzh2i[1] + g->wqsurf) * m.dzfi[1];
  metrics_free(&m);
}
/* floor of thl in `bottom`, BCbotT = 1: src/modibm.f90:2035-2047 */
void orc_thl_floor(const orc_grid *g, const double *ekh, const double *thl0, double *thlp) {
  if (!g->lbottom) return;
  metrics m; metrics_init(g, &m);
  const double *dzf = g->dzf;
  for (int j = 1; j <= g->ny; ++j)
    for (int i = 1; i <= g->nx; ++i)
      M(thlp, i, j, 1) = M(thlp, i, j, 1) + (0.5 * (dzf[0] * M(ekh, i, j, 1) + dzf[1] * M(ekh, i, j, 0))
                                             * (M(thl0, i, j, 1) - M(thl0, i, j, 0)) * m.dzh2i[1] - g->wtsurf) * m.dzfi[1];
  metrics_free(&m);
}

static const double *ibm_ctx_mask(int grid);
/* buoyancy term of forces with lbuoyancy for dry air (src/modforces.f90:73-84): thv0h = thl0h (calthv,
 * src/modthermodynamics.f90:208; calc_halflev :518-524), thvh = its slab average (:76, avexy_ibm) */
void orc_buoyancy(const orc_grid *g, const double *thl0, double *wp) {
  if (!g->lbuoyancy) return;
  const double grav = 9.81;                        /* src/modglobal.f90:271 */
  const double *dzf = g->dzf, *dzh = g->dzh;
  const double *mask_w = ibm_ctx_mask(2);          /* avexy_ibm with IIw, IIws: the fluid w points of the level only */
  for (int k = 2; k <= g->nz; ++k) {
    double s = 0., cnt = 0.;
    for (int j = 1; j <= g->ny; ++j)
      for (int i = 1; i <= g->nx; ++i)
        if (!mask_w || M(mask_w, i, j, k) > 0.5) {
          s += (M(thl0, i, j, k) * dzf[k - 1] + M(thl0, i, j, k - 1) * dzf[k]) / (2 * dzh[k]);
          cnt += 1.;
        }
    const double thvh = s / cnt;
    for (int j = 1; j <= g->ny; ++j)
      for (int i = 1; i <= g->nx; ++i) {
        const double th = (M(thl0, i, j, k) * dzf[k - 1] + M(thl0, i, j, k - 1) * dzf[k]) / (2 * dzh[k]);
        M(wp, i, j, k) = M(wp, i, j, k) + grav * (th - thvh) / thvh;
      }
  }
}

/* ====================================================================== coriolis */
/* src/modforces.f90:600-717: lcoriol (:627-676) or lprofforc (:682-710); ug[k] indexed by Fortran k */
void orc_coriolis(const orc_grid *g, const double *u0, const double *v0, const double *w0, const double *ug,
                  double *up, double *vp, double *wp) {
  const double *dzf = g->dzf, *dzh = g->dzh;
  const double om22 = g->om22, om23 = g->om23;
  if (g->coriolis_mode == 1) {
    for (int k = 2; k <= g->nz; ++k) {
      int kp = k + 1, km = k - 1;
      for (int j = 1; j <= g->ny; ++j) {
        int jp = j + 1, jm = j - 1;
        for (int i = 1; i <= g->nx; ++i) {
          M(up, i, j, k) = M(up, i, j, k)
              + ((M(v0, i, j, k) + M(v0, i, jp, k) + M(v0, i - 1, j, k) + M(v0, i - 1, jp, k)) * om23 * 0.25)
              - ((M(w0, i, j, k) + M(w0, i, j, kp) + M(w0, i - 1, j, kp) + M(w0, i - 1, j, k)) * om22 * 0.25);
          M(vp, i, j, k) = M(vp, i, j, k)
              - ((M(u0, i, j, k) + M(u0, i, jm, k) + M(u0, i + 1, jm, k) + M(u0, i + 1, j, k)) * om23 * 0.25);
          M(wp, i, j, k) = M(wp, i, j, k) + (((dzf[km] * (M(u0, i, j, k) + M(u0, i + 1, j, k))
                         + dzf[k] * (M(u0, i, j, km) + M(u0, i + 1, j, km))) / dzh[k]) * om22 * 0.25);
        }
      }
    }
    for (int j = 1; j <= g->ny; ++j) {
      int jp = j + 1, jm = j - 1;
      for (int i = 1; i <= g->nx; ++i) {
        M(up, i, j, 1) = M(up, i, j, 1)
            + (M(v0, i, j, 1) + M(v0, i, jp, 1) + M(v0, i - 1, j, 1) + M(v0, i - 1, jp, 1)) * om23 * 0.25
            - (M(w0, i, j, 1) + M(w0, i, j, 2) + M(w0, i - 1, j, 2) + M(w0, i - 1, j, 1)) * om22 * 0.25;
        M(vp, i, j, 1) = M(vp, i, j, 1)
            - (M(u0, i, j, 1) + M(u0, i, jm, 1) + M(u0, i + 1, jm, 1) + M(u0, i + 1, j, 1)) * om23 * 0.25;
        M(wp, i, j, 1) = 0.0;
      }
    }
  } else if (g->coriolis_mode == 2) {
    for (int k = 1; k <= g->nz; ++k)
      for (int j = 1; j <= g->ny; ++j)
        for (int i = 1; i <= g->nx; ++i) M(up, i, j, k) = M(up, i, j, k) + om23 * (ug[k] - M(u0, i, j, k));
  }
}

/* ====================================================================== one-equation (TKE) closure */
/* ---- moist thermodynamics, src/modthermodynamics.f90 ------------------------------------------------------------ */
static const double TH_RD = 287.04, TH_RV = 461.5, TH_CP = 1004., TH_RLV = 2.26e6, TH_GRAV = 9.81, TH_PREF0 = 1.e5,
                    TH_TMELT = 273.16, TH_ES0 = 610.78, TH_AT = 17.27, TH_BT = 35.86;   /* src/modglobal.f90:271-313 */
/* thermo with lqlnr (:448-473): Newton-Raphson on the temperature, first guess tl */
static double th_ql_nr(double thl, double qt, double pressure, double exner) {
  const double tl = thl * exner;
  double Tnr = tl, Tnr_old = 0., qsatur = 0.;
  while (fabs(Tnr - Tnr_old) / Tnr > 1e-5) {
    Tnr_old = Tnr;
    const double es = TH_ES0 * exp(TH_AT * (Tnr - TH_TMELT) / (Tnr - TH_BT));
    qsatur = TH_RD / TH_RV * es / (pressure - (1 - TH_RD / TH_RV) * es);
    Tnr = Tnr - (Tnr + (TH_RLV / TH_CP) * qsatur - tl - (TH_RLV / TH_CP) * qt) / (1 + (TH_RLV * TH_RLV * qsatur) / (TH_RV * TH_CP * (Tnr * Tnr)));
  }
  return qt - qsatur > 0. ? qt - qsatur : 0.;
}
/* thermo (:430-503, lqlnr false): "all-or-nothing" condensate of one point */
static double th_ql(double thl, double qt, double pressure, double exner) {
  double tl = thl * exner;
  if (tl < 100.0) tl = 100.0;                                                         /* :483-485 */
  const double es = TH_ES0 * exp(TH_AT * (tl - TH_TMELT) / (tl - TH_BT));
  const double qsl = TH_RD / TH_RV * es / (pressure - (1 - TH_RD / TH_RV) * es);
  const double b1 = TH_RLV * TH_RLV / (tl * tl * TH_CP * TH_RV);
  const double qs = qsl * (1. + b1 * qt) / (1. + b1 * qsl);
  return qt - qs > 0. ? qt - qs : 0.;                                                 /* dim(qt - qs, 0) */
}
#define TH(t) (s->thermo + (size_t)(t) * (g->nz + 2))
/* fromztop, :366-422 */
static void th_fromztop(const orc_grid *g, orc_state *s) {
  const int ke1 = g->nz + 1;
  const double rdocp = TH_RD / TH_CP;
  const double *dzf = g->dzf, *dzh = g->dzh, *th0av = TH(ORC_TH_TH0AV), *qt0av = TH(ORC_TH_QT0AV), *ql0av = TH(ORC_TH_QL0AV);
  double *presf = TH(ORC_TH_PRESF), *presh = TH(ORC_TH_PRESH);
  const double thvs = g->thls * (1. + (TH_RV / TH_RD - 1.) * g->qts);                 /* src/modstartup.f90:522 */
  double thvh = thvs;
  presf[1] = pow(g->ps, rdocp) - TH_GRAV * pow(TH_PREF0, rdocp) * g->zf[1] / (TH_CP * thvh);
  presf[1] = pow(presf[1], 1. / rdocp);
  for (int k = 2; k <= ke1; ++k) {
    const double thetah = (th0av[k] * dzf[k - 1] + th0av[k - 1] * dzf[k]) / (2 * dzh[k]);
    const double qth = (qt0av[k] * dzf[k - 1] + qt0av[k - 1] * dzf[k]) / (2 * dzh[k]);
    const double qlh = (ql0av[k] * dzf[k - 1] + ql0av[k - 1] * dzf[k]) / (2 * dzh[k]);
    thvh = thetah * (1 + (TH_RV / TH_RD - 1) * qth - TH_RV / TH_RD * qlh);
    presf[k] = pow(presf[k - 1], rdocp) - TH_GRAV * pow(TH_PREF0, rdocp) * dzh[k] / (TH_CP * thvh);
    presf[k] = pow(presf[k], 1. / rdocp);
  }
  presh[1] = g->ps;
  for (int k = 2; k <= ke1; ++k) {
    const double thvf = th0av[k - 1] * (1 + (TH_RV / TH_RD - 1) * qt0av[k - 1] - TH_RV / TH_RD * ql0av[k - 1]);
    presh[k] = pow(presh[k - 1], rdocp) - TH_GRAV * pow(TH_PREF0, rdocp) * dzf[k - 1] / (TH_CP * thvf);
    presh[k] = pow(presh[k], 1. / rdocp);
  }
}
/* diagfld, :241-350 (the parts the moist path needs: thl0av, qt0av, ql0av, pressures, exner functions) */
static void th_diagfld(const orc_grid *g, orc_state *s) {
  const int ke1 = g->nz + 1;
  const double cnt = (double)g->nx * (double)g->ny;
  double *thl0av = TH(ORC_TH_THL0AV), *qt0av = TH(ORC_TH_QT0AV), *ql0av = TH(ORC_TH_QL0AV), *th0av = TH(ORC_TH_TH0AV);
  double *exnf = TH(ORC_TH_EXNF), *exnh = TH(ORC_TH_EXNH), *presf = TH(ORC_TH_PRESF), *presh = TH(ORC_TH_PRESH);
  for (int k = 1; k <= ke1; ++k) {
    double a = 0., b = 0., c = 0.;
    for (int j = 1; j <= g->ny; ++j)
      for (int i = 1; i <= g->nx; ++i) { a += M(s->thl0, i, j, k); b += M(s->qt0, i, j, k); c += M(s->ql0, i, j, k); }
    thl0av[k] = a / cnt; qt0av[k] = b / cnt; ql0av[k] = c / cnt;
  }
  for (int k = 1; k <= ke1; ++k) {                                                    /* :290-292 */
    exnf[k] = 1 - TH_GRAV * g->zf[k] / (TH_CP * g->thls);
    exnh[k] = 1 - TH_GRAV * g->zh[k] / (TH_CP * g->thls);
    th0av[k] = thl0av[k] + (TH_RLV / TH_CP) * ql0av[k] / exnf[k];
  }
  th_fromztop(g, s);                                                                  /* :311 */
  for (int k = 1; k <= ke1; ++k) {
    exnf[k] = pow(presf[k] / TH_PREF0, TH_RD / TH_CP);
    th0av[k] = thl0av[k] + (TH_RLV / TH_CP) * ql0av[k] / exnf[k];
  }
  th_fromztop(g, s);                                                                  /* :318 */
  exnh[1] = pow(g->ps / TH_PREF0, TH_RD / TH_CP);                                     /* :329-334 */
  exnf[1] = pow(presf[1] / TH_PREF0, TH_RD / TH_CP);
  for (int k = 2; k <= ke1; ++k) {
    exnf[k] = pow(presf[k] / TH_PREF0, TH_RD / TH_CP);
    exnh[k] = pow(presh[k] / TH_PREF0, TH_RD / TH_CP);
  }
}
/* thl0h, qt0h of calc_halflev (:508-539) and thv0h of calthv (:142-152) at one half level point */
static double th_thv0h(const orc_grid *g, const orc_state *s, int i, int j, int k) {
  const double *dzf = g->dzf, *dzh = g->dzh;
  double thl0h = (M(s->thl0, i, j, k) * dzf[k - 1] + M(s->thl0, i, j, k - 1) * dzf[k]) / (2 * dzh[k]);
  double qt0h = (M(s->qt0, i, j, k) * dzf[k - 1] + M(s->qt0, i, j, k - 1) * dzf[k]) / (2 * dzh[k]);
  if (k == 1) { thl0h = g->thls; qt0h = g->qts; }
  const double exnh = s->thermo[(size_t)ORC_TH_EXNH * (g->nz + 2) + k];
  const double ql0h = (g->lqlnr ? th_ql_nr : th_ql)(thl0h, qt0h, s->thermo[(size_t)ORC_TH_PRESH * (g->nz + 2) + k], exnh);
  return (thl0h + TH_RLV * ql0h / (TH_CP * exnh)) * (1 + (TH_RV / TH_RD - 1) * qt0h - TH_RV / TH_RD * ql0h);
}
void orc_thermodynamics(const orc_grid *g, orc_state *s) {
  const int ke1 = g->nz + 1;
  double *valid = s->thermo + (size_t)ORC_TH_N * (g->nz + 2);
  if (*valid == 0.) { th_diagfld(g, s); *valid = 1.; }                                /* if (timee==0) call diagfld, :64 */
  const double *presf = TH(ORC_TH_PRESF), *exnf = TH(ORC_TH_EXNF);
  /* thermo(thl0,qt0,ql0,presf,exnf), :66.  thermo declares its result ql(..., kb:ke+kh) (:441) while ql0 is allocated
   * from kb-kh (alloc_z, src/modfields.f90:505), so by sequence association the value of level k lands in ql0(k-1):
   * the reference's ql0 -- and with it ql0av in diagfld -- is one level low, and ql0(ke+kh) is never written.  Kept. */
  for (int k = 1; k <= ke1; ++k)
    for (int j = 1; j <= g->ny; ++j)
      for (int i = 1; i <= g->nx; ++i) M(s->ql0, i, j, k - 1) = (g->lqlnr ? th_ql_nr : th_ql)(M(s->thl0, i, j, k), M(s->qt0, i, j, k), presf[k], exnf[k]);
  th_diagfld(g, s);                                                                   /* :69 */
  double *thvh = TH(ORC_TH_THVH);
  const double cnt = (double)g->nx * (double)g->ny;
  for (int k = 1; k <= ke1; ++k) {                                                    /* :70-76 */
    double a = 0.;
    for (int j = 1; j <= g->ny; ++j)
      for (int i = 1; i <= g->nx; ++i) a += th_thv0h(g, s, i, j, k);
    thvh[k] = a / cnt;
  }
  thvh[1] = TH(ORC_TH_TH0AV)[1] * (1 + (TH_RV / TH_RD - 1) * TH(ORC_TH_QT0AV)[1] - TH_RV / TH_RD * TH(ORC_TH_QL0AV)[1]);   /* :90 */
}
void orc_buoyancy_moist(const orc_grid *g, const orc_state *s, double *wp) {
  if (!g->lbuoyancy) return;
  const double *thvh = s->thermo + (size_t)ORC_TH_THVH * (g->nz + 2);
  for (int k = 2; k <= g->nz; ++k)
    for (int j = 1; j <= g->ny; ++j)
      for (int i = 1; i <= g->nx; ++i)
        M(wp, i, j, k) = M(wp, i, j, k) + TH_GRAV * (th_thv0h(g, s, i, j, k) - thvh[k]) / thvh[k];
}
#undef TH

/* dthvdz of calthv, dry air (src/modthermodynamics.f90:208-232); thl0 may be NULL (no temperature equation) */
/* moist air (:154-205): set by orc_set_moist_context / orc_substep -- the state whose qt0, ql0 (level-shifted, as the
 * reference has it) and thermodynamics tables calthv read.  NULL: dry */
static const orc_state *moist_ctx = NULL;
void orc_set_moist_context(const orc_state *s) { moist_ctx = s; }
static double orc_dthvdz(const orc_grid *g, const double *thl0, int i, int j, int k) {
  const double eps1 = 1e-10;
  double d = 0.;
  if (thl0 && k >= 2 && g->lmoist && moist_ctx && moist_ctx->ql0) {
    const orc_state *s = moist_ctx;
    const double chi_half = 0.5;                                    /* src/modthermodynamics.f90:39 */
    const double epsilon = TH_RD / TH_RV, eps_I = 1 / epsilon - 1.;
    const double qt = M(s->qt0, i, j, k), th = M(thl0, i, j, k), ql = M(s->ql0, i, j, k);
    const double a_dry = 1. + eps_I * qt, b_dry = eps_I * th;
    const double dth = M(thl0, i, j, k + 1) - M(thl0, i, j, k - 1), dq = M(s->qt0, i, j, k + 1) - M(s->qt0, i, j, k - 1);
    const double del_thv_dry = a_dry * dth + b_dry * dq;
    double dthv = del_thv_dry;
    if (ql > 0) {
      const double exnf = s->thermo[(size_t)ORC_TH_EXNF * (g->nz + 2) + k];
      const double temp = th * exnf + (TH_RLV / TH_CP) * ql;
      const double qs = qt - ql;
      const double a_moist = (1. - qt + qs / epsilon * (1. + TH_RLV / (TH_RV * temp))) / (1. + TH_RLV * TH_RLV * qs / (TH_CP * TH_RV * (temp * temp)));
      const double b_moist = a_moist * TH_RLV / TH_CP - temp;
      const double c_liquid = a_dry * TH_RLV / TH_CP - th / epsilon;
      const double del_thv_sat = a_moist * dth + b_moist * dq;
      const double chi = 2 * chi_half * (g->zf[k] - g->zf[k - 1]) / (g->dzh[k] + g->dzh[k + 1]);
      const double chi_sat = c_liquid * ql / (del_thv_dry - del_thv_sat);
      if (chi < chi_sat) dthv = del_thv_sat;
    }
    d = dthv / (g->dzh[k + 1] + g->dzh[k]);
  } else if (thl0 && k >= 2) d = (M(thl0, i, j, k + 1) - M(thl0, i, j, k - 1)) / (g->dzh[k + 1] + g->dzh[k]);
  if (fabs(d) < eps1) d = copysign(eps1, d);
  return d;
}
static double orc_delta(const orc_grid *g, int i, int k) {
  double dxf_i = (double)i * g->dx - (double)(i - 1) * g->dx;    /* src/modglobal.f90:783,793-797 */
  return pow(dxf_i * g->dy * g->dzf[k], 1. / 3.);
}
/* closure, loneeqn branch: src/modsubgrid.f90:363-400 (damp = 1); closurebc applied by the caller as for the others */
void orc_closure_tke(const orc_grid *g, const double *e120, const double *thl0, double *ekm, double *ekh) {
  for (int k = 1; k <= g->nz; ++k)
    for (int j = 1; j <= g->ny; ++j)
      for (int i = 1; i <= g->nx; ++i) {
        const double delta = orc_delta(g, i, k), dth = orc_dthvdz(g, thl0, i, j, k), e = M(e120, i, j, k);
        if (g->ldelta || dth <= 0) {
          const double zlt = delta;
          M(ekm, i, j, k) = g->cm * zlt * 1. * e;
          M(ekh, i, j, k) = (g->ch1 + g->ch2) * M(ekm, i, j, k);
        } else {
          const double zlt = fmin(delta, g->cn * e / sqrt(9.81 / g->thvs * fabs(dth)));
          M(ekm, i, j, k) = g->cm * zlt * 1. * e;
          M(ekh, i, j, k) = (g->ch1 + g->ch2 * zlt / delta) * M(ekm, i, j, k);
        }
        M(ekm, i, j, k) = M(ekm, i, j, k) + g->numol;
        M(ekh, i, j, k) = M(ekh, i, j, k) + g->numol * g->prandtlmoli;
      }
  orc_closurebc(g, ekm, ekh);
}
/* diffe, src/modsubgrid.f90:627-669 */
void orc_diffe(const orc_grid *g, const double *e120, const double *ekm, double *e12p) {
  metrics m; metrics_init(g, &m);
  const double *dzf = g->dzf;
  for (int k = 1; k <= g->nz; ++k)
    for (int j = 1; j <= g->ny; ++j)
      for (int i = 1; i <= g->nx; ++i) {
        int kp = k + 1, km = k - 1, jp = j + 1, jm = j - 1, ip = i + 1, im = i - 1;
        M(e12p, i, j, k) = M(e12p, i, j, k) + 1.0 * (
            ((M(ekm, ip, j, k) + M(ekm, i, j, k)) * (M(e120, ip, j, k) - M(e120, i, j, k))
           - (M(ekm, i, j, k) + M(ekm, im, j, k)) * (M(e120, i, j, k) - M(e120, im, j, k))) * m.dx2i
          + ((M(ekm, i, jp, k) + M(ekm, i, j, k)) * (M(e120, i, jp, k) - M(e120, i, j, k))
           - (M(ekm, i, j, k) + M(ekm, i, jm, k)) * (M(e120, i, j, k) - M(e120, i, jm, k))) * m.dy2i
          + ((dzf[kp] * M(ekm, i, j, k) + dzf[k] * M(ekm, i, j, kp)) * (M(e120, i, j, kp) - M(e120, i, j, k)) * m.dzh2i[kp]
           - (dzf[km] * M(ekm, i, j, k) + dzf[k] * M(ekm, i, j, km)) * (M(e120, i, j, k) - M(e120, i, j, km)) * m.dzh2i[k]) * m.dzfi[k]);
      }
  metrics_free(&m);
}
/* sources, src/modsubgrid.f90:415-538: k = kb+1..ke (sbshr, sbbuo, sbdiss stay unwritten at kb) */
void orc_sources(const orc_grid *g, const double *u0, const double *v0, const double *w0, const double *e120,
                 const double *thl0, const double *ekm, const double *ekh, double *e12p) {
  metrics m; metrics_init(g, &m);
  const double dxi = m.dxi, dyi = m.dyi;
#define SQ(x) ((x) * (x))
  for (int k = 2; k <= g->nz; ++k)
    for (int j = 1; j <= g->ny; ++j)
      for (int i = 1; i <= g->nx; ++i) {
        int kp = k + 1, km = k - 1, jp = j + 1, jm = j - 1, ip = i + 1, im = i - 1;
        double tdef2 = 2. * (SQ((M(u0, ip, j, k) - M(u0, i, j, k)) * dxi) + SQ((M(v0, i, jp, k) - M(v0, i, j, k)) * dyi)
                           + SQ((M(w0, i, j, kp) - M(w0, i, j, k)) * m.dzfi[k]));
        tdef2 = tdef2 + 0.25 * (
            SQ((M(w0, i, j, kp) - M(w0, im, j, kp)) * dxi + (M(u0, i, j, kp) - M(u0, i, j, k)) * m.dzhi[kp])
          + SQ((M(w0, i, j, k) - M(w0, im, j, k)) * dxi + (M(u0, i, j, k) - M(u0, i, j, km)) * m.dzhi[k])
          + SQ((M(w0, ip, j, k) - M(w0, i, j, k)) * dxi + (M(u0, ip, j, k) - M(u0, ip, j, km)) * m.dzhi[k])
          + SQ((M(w0, ip, j, kp) - M(w0, i, j, kp)) * dxi + (M(u0, ip, j, kp) - M(u0, ip, j, k)) * m.dzhi[kp]));
        tdef2 = tdef2 + 0.25 * (
            SQ((M(u0, i, jp, k) - M(u0, i, j, k)) * dyi + (M(v0, i, jp, k) - M(v0, im, jp, k)) * dxi)
          + SQ((M(u0, i, j, k) - M(u0, i, jm, k)) * dyi + (M(v0, i, j, k) - M(v0, im, j, k)) * dxi)
          + SQ((M(u0, ip, j, k) - M(u0, ip, jm, k)) * dyi + (M(v0, ip, j, k) - M(v0, i, j, k)) * dxi)
          + SQ((M(u0, ip, jp, k) - M(u0, ip, j, k)) * dyi + (M(v0, ip, jp, k) - M(v0, i, jp, k)) * dxi));
        tdef2 = tdef2 + 0.25 * (
            SQ((M(v0, i, j, kp) - M(v0, i, j, k)) * m.dzhi[kp] + (M(w0, i, j, kp) - M(w0, i, jm, kp)) * dyi)
          + SQ((M(v0, i, j, k) - M(v0, i, j, km)) * m.dzhi[k] + (M(w0, i, j, k) - M(w0, i, jm, k)) * dyi)
          + SQ((M(v0, i, jp, k) - M(v0, i, jp, km)) * m.dzhi[k] + (M(w0, i, jp, k) - M(w0, i, j, k)) * dyi)
          + SQ((M(v0, i, jp, kp) - M(v0, i, jp, k)) * m.dzhi[kp] + (M(w0, i, jp, kp) - M(w0, i, j, kp)) * dyi));
        const double e = M(e120, i, j, k), delta = orc_delta(g, i, k), dth = orc_dthvdz(g, thl0, i, j, k);
        const double zlt = (g->ldelta || dth <= 0) ? delta : fmin(delta, g->cn * e / sqrt(9.81 / g->thvs * fabs(dth)));
        const double sbshr = (M(ekm, i, j, k) - g->numol) * tdef2 / (2 * e);
        const double sbbuo = -(M(ekh, i, j, k) - g->numol * g->prandtlmoli) * 9.81 / g->thvs * dth / (2 * e);
        const double sbdiss = -2. * (g->ce1 + g->ce2 * zlt / delta) * (e * e) / (2. * 1. * zlt);
        M(e12p, i, j, k) = M(e12p, i, j, k) + sbshr + sbbuo + sbdiss;
      }
#undef SQ
  metrics_free(&m);
}

/* ====================================================================== masscorr */
/* src/modforces.f90:328-497, volume-flow branches: luvolflowr (:389-417) and lvvolflowr (:467-494);
 * avexy_ibm without IBM (src/modmpi.f90:623-664): slab sums divided by IIus(k) = itot*jtot. */
static const double *ibm_ctx_mask(int grid);
/* mask (m-array of 1 / 0, or NULL) of the velocity component masscorr averages: avexy_ibm with IIu / IIv
 * (src/modforces.f90:404-405, src/modmpi.f90:623-664); set by orc_substep from the immersed-boundary context */
static const double *flow_mask = NULL;
static double volflow_def(const orc_grid *g, double rk3coef, const double *tp, const double *tm, double target) {
  double zsize = 0.;
  for (int k = 1; k <= g->nz; ++k) zsize += g->dzf[k];               /* zh(ke+1) */
  double svol = 0., svolold = 0.;
  for (int k = 1; k <= g->nz; ++k) {
    double a = 0., b = 0., cnt = 0.;
    for (int j = 1; j <= g->ny; ++j)
      for (int i = 1; i <= g->nx; ++i) {
        const double mk = flow_mask ? M(flow_mask, i, j, k) : 1.;
        a += M(tp, i, j, k) * mk; b += M(tm, i, j, k) * mk; cnt += mk;
      }
    if (cnt == 0.) continue;                                         /* (-999 in the reference: no such level in the tests) */
    svol += (a / cnt) * g->dzf[k];
    svolold += (b / cnt) * g->dzf[k];
  }
  const double outflow = rk3coef * svol / zsize;
  const double flowrateold = svolold / zsize;
  return target - (outflow + flowrateold);
}
void orc_masscorr(const orc_grid *g, int rk3step, double dt, double *up, const double *um, double *vp, const double *vm) {
  const double rk3coef = dt / (4. - (double)rk3step);
  const double rk3coefi = 1 / rk3coef;
  if (xo_on) return;      /* every branch sits under `.not. linoutflow` (src/modforces.f90:352, 393, 424, 467) */
  if (g->luvolflowr) {
    flow_mask = ibm_ctx_mask(0);
    const double udef = volflow_def(g, rk3coef, up, um, g->uflowrate);
    flow_mask = NULL;
    for (int k = 1; k <= g->nz; ++k)
      for (int j = 1; j <= g->ny; ++j)
        for (int i = 1; i <= g->nx; ++i) M(up, i, j, k) = M(up, i, j, k) + udef * rk3coefi;
  }
  if (g->lvvolflowr) {
    flow_mask = ibm_ctx_mask(1);
    const double vdef = volflow_def(g, rk3coef, vp, vm, g->vflowrate);
    flow_mask = NULL;
    for (int k = 1; k <= g->nz; ++k)
      for (int j = 1; j <= g->ny; ++j)
        for (int i = 1; i <= g->nx; ++i) M(vp, i, j, k) = M(vp, i, j, k) + vdef * rk3coefi;
  }
}

/* ====================================================================== substep */
/* ------------------------------------------------------------------------------------------------------------------
 * Immersed boundary, sparse corrections: src/modibm.f90.  Point lists as the reference reads them (solid_*.txt,
 * fluid_boundary_*.txt): pts[3*n + 0..2] = (i, j, k), 1-based; single rank.  Masks are m-arrays holding 1 (fluid) / 0.
 * orc_ibm_mask restates initibm's mask set-up (:150-167): ones, zero at k = kb-1, mask_w also zero at kb, zero at the
 * solid points.  The lateral ghost cells come from exchange_halo_z alone (:163-165), which wraps a direction only when the
 * run splits it over more than one rank (periodic_bc, src/modstartup.f90:662-672): wrapx / wrapy say which; without it
 * the ghost cells keep the initial 1 ("fluid"), which is what the single-rank reference build has. */
void orc_ibm_mask(const orc_grid *g, int is_w, const int *solid, int nsolid, double *mask, int wrapx, int wrapy) {
  const size_t n = msize(g);
  const int nx = g->nx, ny = g->ny, nz = g->nz;
  for (size_t q = 0; q < n; ++q) mask[q] = 1.;
  for (int j = 0; j <= ny + 1; ++j)
    for (int i = 0; i <= nx + 1; ++i) { M(mask, i, j, 0) = 0.; if (is_w) M(mask, i, j, 1) = 0.; }
  for (int q = 0; q < nsolid; ++q) M(mask, solid[3 * q], solid[3 * q + 1], solid[3 * q + 2]) = 0.;
  if (wrapx)
    for (int k = 0; k <= nz + 1; ++k)
      for (int j = 0; j <= ny + 1; ++j) { M(mask, 0, j, k) = M(mask, nx, j, k); M(mask, nx + 1, j, k) = M(mask, 1, j, k); }
  if (wrapy)
    for (int k = 0; k <= nz + 1; ++k)
      for (int i = 0; i <= nx + 1; ++i) { M(mask, i, 0, k) = M(mask, i, ny, k); M(mask, i, ny + 1, k) = M(mask, i, 1, k); }
}
/* diffu_corr :990-1030 */
void orc_ibm_diffu_corr(const orc_grid *g, const int *bnd, int nbnd, const double *mask_u, const double *u0, const double *ekm, double *up) {
  metrics m; metrics_init(g, &m);
  const double *dzf = g->dzf;
  for (int n = 0; n < nbnd; ++n) {
    const int i = bnd[3 * n], j = bnd[3 * n + 1], k = bnd[3 * n + 2];
    if (fabs(M(mask_u, i, j + 1, k)) < 1e-10) {
      double empo = 0.25 * ((M(ekm, i, j, k) + M(ekm, i, j + 1, k)) + (M(ekm, i - 1, j, k) + M(ekm, i - 1, j + 1, k)));
      M(up, i, j, k) = M(up, i, j, k) - empo * (M(u0, i, j + 1, k) - M(u0, i, j, k)) * m.dy2i;
    }
    if (fabs(M(mask_u, i, j - 1, k)) < 1e-10) {
      double emmo = 0.25 * ((M(ekm, i, j, k) + M(ekm, i, j - 1, k)) + (M(ekm, i - 1, j - 1, k) + M(ekm, i - 1, j, k)));
      M(up, i, j, k) = M(up, i, j, k) + emmo * (M(u0, i, j, k) - M(u0, i, j - 1, k)) * m.dy2i;
    }
    if (fabs(M(mask_u, i, j, k + 1)) < 1e-10) {
      double emop = (dzf[k + 1] * (M(ekm, i, j, k) + M(ekm, i - 1, j, k)) + dzf[k] * (M(ekm, i, j, k + 1) + M(ekm, i - 1, j, k + 1))) * m.dzhiq[k + 1];
      M(up, i, j, k) = M(up, i, j, k) - emop * (M(u0, i, j, k + 1) - M(u0, i, j, k)) * m.dzhi[k + 1] * m.dzfi[k];
    }
    if (fabs(M(mask_u, i, j, k - 1)) < 1e-10) {
      double emom = (dzf[k - 1] * (M(ekm, i, j, k) + M(ekm, i - 1, j, k)) + dzf[k] * (M(ekm, i, j, k - 1) + M(ekm, i - 1, j, k - 1))) * m.dzhiq[k];
      M(up, i, j, k) = M(up, i, j, k) + emom * (M(u0, i, j, k) - M(u0, i, j, k - 1)) * m.dzhi[k] * m.dzfi[k];
    }
  }
  metrics_free(&m);
}
/* diffv_corr :1033-1072 */
void orc_ibm_diffv_corr(const orc_grid *g, const int *bnd, int nbnd, const double *mask_v, const double *v0, const double *ekm, double *vp) {
  metrics m; metrics_init(g, &m);
  const double *dzf = g->dzf;
  for (int n = 0; n < nbnd; ++n) {
    const int i = bnd[3 * n], j = bnd[3 * n + 1], k = bnd[3 * n + 2];
    if (fabs(M(mask_v, i + 1, j, k)) < 1e-10) {
      double epmo = 0.25 * (M(ekm, i, j, k) + M(ekm, i, j - 1, k) + M(ekm, i + 1, j - 1, k) + M(ekm, i + 1, j, k));
      M(vp, i, j, k) = M(vp, i, j, k) - epmo * (M(v0, i + 1, j, k) - M(v0, i, j, k)) * m.dx2i;
    }
    if (fabs(M(mask_v, i - 1, j, k)) < 1e-10) {
      double emmo = 0.25 * (M(ekm, i, j, k) + M(ekm, i, j - 1, k) + M(ekm, i - 1, j - 1, k) + M(ekm, i - 1, j, k));
      M(vp, i, j, k) = M(vp, i, j, k) + emmo * (M(v0, i, j, k) - M(v0, i - 1, j, k)) * m.dx2i;
    }
    if (fabs(M(mask_v, i, j, k + 1)) < 1e-10) {
      double eomp = (dzf[k + 1] * (M(ekm, i, j, k) + M(ekm, i, j - 1, k)) + dzf[k] * (M(ekm, i, j, k + 1) + M(ekm, i, j - 1, k + 1))) * m.dzhiq[k + 1];
      M(vp, i, j, k) = M(vp, i, j, k) - eomp * (M(v0, i, j, k + 1) - M(v0, i, j, k)) * m.dzhi[k + 1] * m.dzfi[k];
    }
    if (fabs(M(mask_v, i, j, k - 1)) < 1e-10) {
      double eomm = (dzf[k - 1] * (M(ekm, i, j, k) + M(ekm, i, j - 1, k)) + dzf[k] * (M(ekm, i, j, k - 1) + M(ekm, i, j - 1, k - 1))) * m.dzhiq[k];
      M(vp, i, j, k) = M(vp, i, j, k) + eomm * (M(v0, i, j, k) - M(v0, i, j, k - 1)) * m.dzhi[k] * m.dzfi[k];
    }
  }
  metrics_free(&m);
}
/* diffw_corr :1075-1117 */
void orc_ibm_diffw_corr(const orc_grid *g, const int *bnd, int nbnd, const double *mask_w, const double *w0, const double *ekm, double *wp) {
  metrics m; metrics_init(g, &m);
  const double *dzf = g->dzf;
  for (int n = 0; n < nbnd; ++n) {
    const int i = bnd[3 * n], j = bnd[3 * n + 1], k = bnd[3 * n + 2];
    if (fabs(M(mask_w, i + 1, j, k)) < 1e-10) {
      double epom = (dzf[k - 1] * (M(ekm, i, j, k) + M(ekm, i + 1, j, k)) + dzf[k] * (M(ekm, i, j, k - 1) + M(ekm, i + 1, j, k - 1))) * m.dzhiq[k];
      M(wp, i, j, k) = M(wp, i, j, k) - epom * (M(w0, i + 1, j, k) - M(w0, i, j, k)) * m.dx2i;
    }
    if (fabs(M(mask_w, i - 1, j, k)) < 1e-10) {
      double emom = (dzf[k - 1] * (M(ekm, i, j, k) + M(ekm, i - 1, j, k)) + dzf[k] * (M(ekm, i, j, k - 1) + M(ekm, i - 1, j, k - 1))) * m.dzhiq[k];
      M(wp, i, j, k) = M(wp, i, j, k) + emom * (M(w0, i, j, k) - M(w0, i - 1, j, k)) * m.dx2i;
    }
    if (fabs(M(mask_w, i, j + 1, k)) < 1e-10) {
      double eopm = (dzf[k - 1] * (M(ekm, i, j, k) + M(ekm, i, j + 1, k)) + dzf[k] * (M(ekm, i, j, k - 1) + M(ekm, i, j + 1, k - 1))) * m.dzhiq[k];
      M(wp, i, j, k) = M(wp, i, j, k) - eopm * (M(w0, i, j + 1, k) - M(w0, i, j, k)) * m.dy2i;
    }
    if (fabs(M(mask_w, i, j - 1, k)) < 1e-10) {
      double eomm = (dzf[k - 1] * (M(ekm, i, j, k) + M(ekm, i, j - 1, k)) + dzf[k] * (M(ekm, i, j, k - 1) + M(ekm, i, j - 1, k - 1))) * m.dzhiq[k];
      M(wp, i, j, k) = M(wp, i, j, k) + eomm * (M(w0, i, j, k) - M(w0, i, j - 1, k)) * m.dy2i;
    }
  }
  metrics_free(&m);
}
/* diffc_corr :1120-1164 for a kappa-advected scalar (c-arrays, as ibmwallfun passes sv0 / svp with the wide halo :1262-1264) */
void orc_ibm_diffc_corr(const orc_grid *g, const int *bnd, int nbnd, const double *mask_c, const double *var, const double *ekh, double *rhs) {
  metrics m; metrics_init(g, &m);
  const double *dzf = g->dzf;
  for (int n = 0; n < nbnd; ++n) {
    const int i = bnd[3 * n], j = bnd[3 * n + 1], k = bnd[3 * n + 2];
    if (fabs(M(mask_c, i + 1, j, k)) < 1e-10)
      C(rhs, i, j, k) = C(rhs, i, j, k) - 0.5 * (M(ekh, i + 1, j, k) + M(ekh, i, j, k)) * (C(var, i + 1, j, k) - C(var, i, j, k)) * m.dx2i;
    if (fabs(M(mask_c, i - 1, j, k)) < 1e-10)
      C(rhs, i, j, k) = C(rhs, i, j, k) + 0.5 * (M(ekh, i, j, k) + M(ekh, i - 1, j, k)) * (C(var, i, j, k) - C(var, i - 1, j, k)) * m.dx2i;
    if (fabs(M(mask_c, i, j + 1, k)) < 1e-10)
      C(rhs, i, j, k) = C(rhs, i, j, k) - 0.5 * (M(ekh, i, j + 1, k) + M(ekh, i, j, k)) * (C(var, i, j + 1, k) - C(var, i, j, k)) * m.dy2i;
    if (fabs(M(mask_c, i, j - 1, k)) < 1e-10)
      C(rhs, i, j, k) = C(rhs, i, j, k) + 0.5 * (M(ekh, i, j, k) + M(ekh, i, j - 1, k)) * (C(var, i, j, k) - C(var, i, j - 1, k)) * m.dy2i;
    if (fabs(M(mask_c, i, j, k + 1)) < 1e-10)
      C(rhs, i, j, k) = C(rhs, i, j, k) - 0.5 * (dzf[k + 1] * M(ekh, i, j, k) + dzf[k] * M(ekh, i, j, k + 1))
                                              * (C(var, i, j, k + 1) - C(var, i, j, k)) * m.dzh2i[k + 1] * m.dzfi[k];
    if (fabs(M(mask_c, i, j, k - 1)) < 1e-10)
      C(rhs, i, j, k) = C(rhs, i, j, k) + 0.5 * (dzf[k - 1] * M(ekh, i, j, k) + dzf[k] * M(ekh, i, j, k - 1))
                                              * (C(var, i, j, k) - C(var, i, j, k - 1)) * m.dzh2i[k] * m.dzfi[k];
  }
  metrics_free(&m);
}
/* solid :748-826 without a mask, m-arrays (ibmnorm :706-708: um / up ... with val = 0) */
void orc_ibm_solid_m(const orc_grid *g, const int *pts, int n, double *var, double *rhs, double val) {
  for (int q = 0; q < n; ++q) {
    const int i = pts[3 * q], j = pts[3 * q + 1], k = pts[3 * q + 2];
    M(var, i, j, k) = val;
    M(rhs, i, j, k) = 0.;
  }
}
/* solid with the c mask on a c-array pair (ibmnorm :733-734: svm / svp): mean over the fluid neighbours */
void orc_ibm_solid_c(const orc_grid *g, const int *pts, int n, const double *mask, double *var, double *rhs, double val) {
  const int di[6] = {0, 0, 0, 0, 1, -1}, dj[6] = {1, -1, 0, 0, 0, 0}, dk[6] = {0, 0, 1, -1, 0, 0};      /* the reference's order */
  for (int q = 0; q < n; ++q) {
    const int i = pts[3 * q], j = pts[3 * q + 1], k = pts[3 * q + 2];
    double count = 0.;
    C(var, i, j, k) = val;
    C(rhs, i, j, k) = 0.;
    for (int b = 0; b < 6; ++b)
      if (fabs(M(mask, i + di[b], j + dj[b], k + dk[b]) - 1.) < 1e-10) {
        count = count + 1;
        C(var, i, j, k) = C(var, i, j, k) + C(var, i + di[b], j + dj[b], k + dk[b]);
        C(rhs, i, j, k) = C(rhs, i, j, k) + C(rhs, i + di[b], j + dj[b], k + dk[b]);
      }
    if (count > 0) {
      C(var, i, j, k) = (C(var, i, j, k) - val) / count;
      C(rhs, i, j, k) = C(rhs, i, j, k) / count;
    }
  }
}
/* ---- thl / qt with an immersed boundary: the same routines on m-arrays (halo 1) */
/* diffc_corr :1120-1164 as ibmwallfun calls it for thl0 / thlp and qt0 / qtp (:1232-1233) */
void orc_ibm_diffc_corr_m(const orc_grid *g, const int *bnd, int nbnd, const double *mask_c, const double *var, const double *ekh, double *rhs) {
  metrics m; metrics_init(g, &m);
  const double *dzf = g->dzf;
  for (int n = 0; n < nbnd; ++n) {
    const int i = bnd[3 * n], j = bnd[3 * n + 1], k = bnd[3 * n + 2];
    if (fabs(M(mask_c, i + 1, j, k)) < 1e-10)
      M(rhs, i, j, k) = M(rhs, i, j, k) - 0.5 * (M(ekh, i + 1, j, k) + M(ekh, i, j, k)) * (M(var, i + 1, j, k) - M(var, i, j, k)) * m.dx2i;
    if (fabs(M(mask_c, i - 1, j, k)) < 1e-10)
      M(rhs, i, j, k) = M(rhs, i, j, k) + 0.5 * (M(ekh, i, j, k) + M(ekh, i - 1, j, k)) * (M(var, i, j, k) - M(var, i - 1, j, k)) * m.dx2i;
    if (fabs(M(mask_c, i, j + 1, k)) < 1e-10)
      M(rhs, i, j, k) = M(rhs, i, j, k) - 0.5 * (M(ekh, i, j + 1, k) + M(ekh, i, j, k)) * (M(var, i, j + 1, k) - M(var, i, j, k)) * m.dy2i;
    if (fabs(M(mask_c, i, j - 1, k)) < 1e-10)
      M(rhs, i, j, k) = M(rhs, i, j, k) + 0.5 * (M(ekh, i, j, k) + M(ekh, i, j - 1, k)) * (M(var, i, j, k) - M(var, i, j - 1, k)) * m.dy2i;
    if (fabs(M(mask_c, i, j, k + 1)) < 1e-10)
      M(rhs, i, j, k) = M(rhs, i, j, k) - 0.5 * (dzf[k + 1] * M(ekh, i, j, k) + dzf[k] * M(ekh, i, j, k + 1))
                                              * (M(var, i, j, k + 1) - M(var, i, j, k)) * m.dzh2i[k + 1] * m.dzfi[k];
    if (fabs(M(mask_c, i, j, k - 1)) < 1e-10)
      M(rhs, i, j, k) = M(rhs, i, j, k) + 0.5 * (dzf[k - 1] * M(ekh, i, j, k) + dzf[k] * M(ekh, i, j, k - 1))
                                              * (M(var, i, j, k) - M(var, i, j, k - 1)) * m.dzh2i[k] * m.dzfi[k];
  }
  metrics_free(&m);
}
/* solid :748-826 with the c mask on m-arrays (ibmnorm :715 thlm / thlp with the volume mean of thl0av, :726 qtm / qtp with 0) */
void orc_ibm_solid_cm(const orc_grid *g, const int *pts, int n, const double *mask, double *var, double *rhs, double val) {
  const int di[6] = {0, 0, 0, 0, 1, -1}, dj[6] = {1, -1, 0, 0, 0, 0}, dk[6] = {0, 0, 1, -1, 0, 0};      /* the reference's order */
  for (int q = 0; q < n; ++q) {
    const int i = pts[3 * q], j = pts[3 * q + 1], k = pts[3 * q + 2];
    double count = 0.;
    M(var, i, j, k) = val;
    M(rhs, i, j, k) = 0.;
    for (int b = 0; b < 6; ++b)
      if (fabs(M(mask, i + di[b], j + dj[b], k + dk[b]) - 1.) < 1e-10) {
        count = count + 1;
        M(var, i, j, k) = M(var, i, j, k) + M(var, i + di[b], j + dj[b], k + dk[b]);
        M(rhs, i, j, k) = M(rhs, i, j, k) + M(rhs, i + di[b], j + dj[b], k + dk[b]);
      }
    if (count > 0) {
      M(var, i, j, k) = (M(var, i, j, k) - val) / count;
      M(rhs, i, j, k) = M(rhs, i, j, k) / count;
    }
  }
}
/* advecc2nd_corr_conservative :889-933 / advecc2nd_corr_liberal :936-987 at the fluid-boundary points of the c grid */
void orc_ibm_advecc2nd_corr_m(const orc_grid *g, int conservative, const int *bnd, int nbnd, const double *mask_u, const double *mask_v,
                              const double *mask_w, const double *mask_c, const double *u0, const double *v0, const double *w0,
                              const double *var, double *rhs) {
  metrics m; metrics_init(g, &m);
  const double *dzf = g->dzf, eps1 = 1e-10;
  for (int n = 0; n < nbnd; ++n) {
    const int i = bnd[3 * n], j = bnd[3 * n + 1], k = bnd[3 * n + 2];
    const double v0c = M(var, i, j, k);
    if (conservative) {
      if (fabs(M(mask_u, i + 1, j, k)) < eps1 || fabs(M(mask_c, i + 1, j, k)) < eps1)
        M(rhs, i, j, k) = M(rhs, i, j, k) + M(u0, i + 1, j, k) * (M(var, i + 1, j, k) + v0c) * m.dxi5;
      if (fabs(M(mask_u, i, j, k)) < eps1 || fabs(M(mask_c, i - 1, j, k)) < eps1)
        M(rhs, i, j, k) = M(rhs, i, j, k) - M(u0, i, j, k) * (M(var, i - 1, j, k) + v0c) * m.dxi5;
      if (fabs(M(mask_v, i, j + 1, k)) < eps1 || fabs(M(mask_c, i, j + 1, k)) < eps1)
        M(rhs, i, j, k) = M(rhs, i, j, k) + M(v0, i, j + 1, k) * (M(var, i, j + 1, k) + v0c) * m.dyi5;
      if (fabs(M(mask_v, i, j, k)) < eps1 || fabs(M(mask_c, i, j - 1, k)) < eps1)
        M(rhs, i, j, k) = M(rhs, i, j, k) - M(v0, i, j, k) * (M(var, i, j - 1, k) + v0c) * m.dyi5;
      if (fabs(M(mask_w, i, j, k + 1)) < eps1 || fabs(M(mask_c, i, j, k + 1)) < eps1)
        M(rhs, i, j, k) = M(rhs, i, j, k) + M(w0, i, j, k + 1) * (M(var, i, j, k + 1) * dzf[k] + v0c * dzf[k + 1]) * m.dzhi[k + 1] * m.dzfi5[k];
      if (fabs(M(mask_w, i, j, k)) < eps1 || fabs(M(mask_c, i, j, k - 1)) < eps1)
        M(rhs, i, j, k) = M(rhs, i, j, k) - M(w0, i, j, k) * (M(var, i, j, k - 1) * dzf[k] + v0c * dzf[k - 1]) * m.dzhi[k] * m.dzfi5[k];
    } else {
      if (fabs(M(mask_c, i + 1, j, k)) < eps1)
        M(rhs, i, j, k) = M(rhs, i, j, k) + M(u0, i + 1, j, k) * (M(var, i + 1, j, k) + v0c) * m.dxi5
                                          - M(u0, i + 1, j, k) * (v0c + v0c) * m.dxi5;
      if (fabs(M(mask_c, i - 1, j, k)) < eps1)
        M(rhs, i, j, k) = M(rhs, i, j, k) - M(u0, i, j, k) * (M(var, i - 1, j, k) + v0c) * m.dxi5
                                          + M(u0, i, j, k) * (v0c + v0c) * m.dxi5;
      if (fabs(M(mask_c, i, j + 1, k)) < eps1)
        M(rhs, i, j, k) = M(rhs, i, j, k) + M(v0, i, j + 1, k) * (M(var, i, j + 1, k) + v0c) * m.dyi5
                                          - M(v0, i, j + 1, k) * (v0c + v0c) * m.dyi5;
      if (fabs(M(mask_c, i, j - 1, k)) < eps1)
        M(rhs, i, j, k) = M(rhs, i, j, k) - M(v0, i, j, k) * (M(var, i, j - 1, k) + v0c) * m.dyi5
                                          + M(v0, i, j, k) * (v0c + v0c) * m.dyi5;
      if (fabs(M(mask_c, i, j, k + 1)) < eps1)
        M(rhs, i, j, k) = M(rhs, i, j, k) + M(w0, i, j, k + 1) * (M(var, i, j, k + 1) * dzf[k] + v0c * dzf[k + 1]) * m.dzhi[k + 1] * m.dzfi5[k]
                                          - M(w0, i, j, k + 1) * (v0c * dzf[k] + v0c * dzf[k + 1]) * m.dzhi[k + 1] * m.dzfi5[k];
      if (fabs(M(mask_c, i, j, k - 1)) < eps1)
        M(rhs, i, j, k) = M(rhs, i, j, k) - M(w0, i, j, k) * (M(var, i, j, k - 1) * dzf[k] + v0c * dzf[k - 1]) * m.dzhi[k] * m.dzfi5[k]
                                          + M(w0, i, j, k) * (v0c * dzf[k] + v0c * dzf[k - 1]) * m.dzhi[k] * m.dzfi5[k];
    }
  }
  metrics_free(&m);
}
/* avexy_ibm (src/modmpi.f90:623-664): the mean of f over the fluid cells of level k */
double orc_ibm_level_mean_m(const orc_grid *g, const double *f, const double *mask, int k) {
  double s = 0., c = 0.;
  for (int j = 1; j <= g->ny; ++j)
    for (int i = 1; i <= g->nx; ++i)
      if (M(mask, i, j, k) > 0.5) { s += M(f, i, j, k); c += 1.; }
  return c > 0. ? s / c : -999.;
}
/* what the substep does with an immersed boundary: ibmwallfun without facet wall functions (src/program.f90:166) and
 * ibmnorm (:171).  Set with orc_set_ibm (NULL: none). */
static const orc_ibm *ibm_ctx = NULL;
void orc_set_ibm(const orc_ibm *b) { ibm_ctx = b; }
static const double *ibm_ctx_mask(int grid) { return ibm_ctx ? ibm_ctx->mask[grid] : NULL; }
void orc_ibmwallfun(const orc_grid *g, const orc_ibm *b, orc_state *s) {
  const size_t nc = csize(g);
  orc_ibm_diffu_corr(g, b->bnd[0], b->nbnd[0], b->mask[0], s->u0, s->ekm, s->up);
  orc_ibm_diffv_corr(g, b->bnd[1], b->nbnd[1], b->mask[1], s->v0, s->ekm, s->vp);
  orc_ibm_diffw_corr(g, b->bnd[2], b->nbnd[2], b->mask[2], s->w0, s->ekm, s->wp);
  /* (:1227-1231 wallfunheat: not restated -- with prescribed zero wall fluxes it adds nothing) */
  if (g->ltempeq) orc_ibm_diffc_corr_m(g, b->bnd[3], b->nbnd[3], b->mask[3], s->thl0, s->ekh, s->thlp);      /* :1232 */
  if (g->lmoist) orc_ibm_diffc_corr_m(g, b->bnd[3], b->nbnd[3], b->mask[3], s->qt0, s->ekh, s->qtp);         /* :1233 */
  for (int n = 0; n < g->nsv; ++n) orc_ibm_diffc_corr(g, b->bnd[3], b->nbnd[3], b->mask[3], s->sv0 + n * nc, s->ekh, s->svp + n * nc);
}
void orc_ibmnorm(const orc_grid *g, const orc_ibm *b, orc_state *s) {
  const size_t nc = csize(g);
  orc_ibm_solid_m(g, b->sol[0], b->nsol[0], s->um, s->up, 0.);
  orc_ibm_solid_m(g, b->sol[1], b->nsol[1], s->vm, s->vp, 0.);
  orc_ibm_solid_m(g, b->sol[2], b->nsol[2], s->wm, s->wp, 0.);
  if (g->ltempeq) {                                                                     /* :714-722 */
    /* thl0av of the last thermodynamics call (diagfld :277, fluid cells only): thl0 has not changed since */
    double val = 0., zh_top = 0.;
    for (int k = 1; k <= g->nz; ++k) { val += orc_ibm_level_mean_m(g, s->thl0, b->mask[3], k) * g->dzf[k]; zh_top += g->dzf[k]; }
    val = val / zh_top;
    orc_ibm_solid_cm(g, b->sol[3], b->nsol[3], b->mask[3], s->thlm, s->thlp, val);
    if (g->iadv_thl != 7)
      orc_ibm_advecc2nd_corr_m(g, b->conservative, b->bnd[3], b->nbnd[3], b->mask[0], b->mask[1], b->mask[2], b->mask[3],
                               s->u0, s->v0, s->w0, s->thl0, s->thlp);
  }
  if (g->lmoist) {                                                                      /* :725-731 */
    orc_ibm_solid_cm(g, b->sol[3], b->nsol[3], b->mask[3], s->qtm, s->qtp, 0.);
    orc_ibm_advecc2nd_corr_m(g, b->conservative, b->bnd[3], b->nbnd[3], b->mask[0], b->mask[1], b->mask[2], b->mask[3],
                             s->u0, s->v0, s->w0, s->qt0, s->qtp);
  }
  for (int n = 0; n < g->nsv; ++n) orc_ibm_solid_c(g, b->sol[3], b->nsol[3], b->mask[3], s->svm + n * nc, s->svp + n * nc, 0.);
}

/* src/program.f90:132-222: advection, subgrid, forces, poisson, tstep_integrate, halos, boundary */
void orc_substep(const orc_grid *g, orc_state *s, int rk3step, double dt) {
  const size_t nc = csize(g);
  moist_ctx = (g->lmoist && s->thermo && s->ql0) ? s : NULL;
  const double rk3coef = dt / (4. - (double)rk3step);
  double uouttot_next = xo_uouttot;      /* BCxm = 2: what this substep's `boundary` will convect the outlet with */
  if (xo_on && xo_wlev) { if (xo_hold) xo_hold = 0; else uouttot_next = xo_outlet_speed(g, s->u0); }
  orc_advecu_2nd(g, s->u0, s->v0, s->w0, s->pres0, s->up);
  orc_advecv_2nd(g, s->u0, s->v0, s->w0, s->pres0, s->vp);
  orc_advecw_2nd(g, s->u0, s->v0, s->w0, s->pres0, s->wp);
  if (g->sgs == 3) orc_advecc_2nd(g, s->u0, s->v0, s->w0, s->e120, s->e12p);              /* src/modadvection.f90:56-58 */
  if (g->ltempeq && g->iadv_thl == 7) orc_advec_thl_kappa(g, s->u0, s->v0, s->w0, s->thl0c, s->thlp);   /* src/modadvection.f90:69-72 */
  else if (g->ltempeq) orc_advecc_2nd(g, s->u0, s->v0, s->w0, s->thl0, s->thlp);         /* src/modadvection.f90:66-68 */
  if (g->lmoist) orc_advecc_2nd(g, s->u0, s->v0, s->w0, s->qt0, s->qtp);                 /* src/modadvection.f90:78-86 */
  for (int n = 0; n < g->nsv; ++n) orc_advecc_kappa(g, s->u0, s->v0, s->w0, s->sv0 + n * nc, s->svp + n * nc);
  if (g->sgs == 3) orc_closure_tke(g, s->e120, g->ltempeq ? s->thl0 : NULL, s->ekm, s->ekh);
  else { closure_thl0 = g->ltempeq ? s->thl0 : NULL; orc_closure(g, s->u0, s->v0, s->w0, s->ekm, s->ekh); }
  /* reassure_fluxtop_boundary src/modboundary.f90:392-431 (free-slip: re-impose top rows) */
  if (g->bctopm != 2) {
    top_row_m(g, s->um, 0.); top_row_m(g, s->u0, 0.); top_row_m(g, s->vm, 0.); top_row_m(g, s->v0, 0.);
    for (int n = 0; n < g->nsv; ++n) {
      double *p0 = s->sv0 + n * nc, *pm = s->svm + n * nc;
      for (int mm = 1; mm <= 2; ++mm)
        for (int j = 0; j <= g->ny + 1; ++j)
          for (int i = 0; i <= g->nx + 1; ++i) {
            C(p0, i, j, g->nz + mm) = C(p0, i, j, g->nz) + 0.0;
            C(pm, i, j, g->nz + mm) = C(pm, i, j, g->nz) + 0.0;
          }
    }
  }
  if (g->nsv > 0 && g->bctops != 2 && scalar_top_active(g)) orc_scalar_tops(g, s->ekh, s->sv0, s->svm);      /* :427-430 */
  if (g->ltempeq && g->bctopt != 2) { orc_thl_top(g, s->ekh, s->thlm); orc_thl_top(g, s->ekh, s->thl0); }   /* :417-420 */
  if (g->lmoist && g->bctopq != 2) { orc_qt_top(g, s->ekh, s->qtm); orc_qt_top(g, s->ekh, s->qt0); }         /* :422-425 */
  orc_diffu(g, s->u0, s->v0, s->w0, s->ekm, s->up);
  orc_diffv(g, s->u0, s->v0, s->w0, s->ekm, s->vp);
  orc_diffw(g, s->u0, s->v0, s->w0, s->ekm, s->wp);
  if (g->sgs == 3) orc_diffe(g, s->e120, s->ekm, s->e12p);                               /* src/modsubgrid.f90:144 */
  if (g->ltempeq) orc_diffc_m(g, s->thl0, s->ekh, s->thlp);                              /* src/modsubgrid.f90:146 */
  if (g->lmoist) orc_diffc_m(g, s->qt0, s->ekh, s->qtp);                                 /* src/modsubgrid.f90:147 */
  for (int n = 0; n < g->nsv; ++n) orc_diffc(g, s->sv0 + n * nc, s->ekh, s->svp + n * nc);
  if (g->sgs == 3) {
    orc_sources(g, s->u0, s->v0, s->w0, s->e120, g->ltempeq ? s->thl0 : NULL, s->ekm, s->ekh, s->e12p);   /* :151 */
    for (int j = 0; j <= g->ny + 1; ++j)                      /* `bottom`, src/modibm.f90:2012-2013 */
      for (int i = 0; i <= g->nx + 1; ++i) { M(s->e120, i, j, 0) = M(s->e120, i, j, 1); M(s->e12m, i, j, 0) = M(s->e12m, i, j, 1); }
  }
  if (g->bcbotm == 2) {                                                                  /* wfuno for momentum; scalars as before */
    double *zu = (double *)calloc(2 * msize(g), sizeof(double));
    orc_bottom(g, s->u0, s->v0, s->ekm, s->ekh, s->sv0, zu, zu + msize(g), s->svp, NULL);
    free(zu);
    orc_bottom_uno(g, s->u0, s->v0, s->thl0, s->ekm, s->ekh, s->up, s->vp, g->bcbott == 2 ? s->thlp : NULL);
  } else {
    orc_bottom(g, s->u0, s->v0, s->ekm, s->ekh, s->sv0, s->up, s->vp, s->svp, NULL);   /* src/program.f90:152 */
    if (g->bcbott == 2) orc_bottom_uno(g, s->u0, s->v0, s->thl0, s->ekm, s->ekh, s->up, s->vp, s->thlp);
  }
  if (g->ltempeq && g->bcbott != 2) orc_thl_floor(g, s->ekh, s->thl0, s->thlp);
  if (g->lmoist) orc_qt_floor(g, s->ekh, s->qt0, s->qtp);
  if (s->dpdxl && g->coriolis_mode) orc_coriolis(g, s->u0, s->v0, s->w0, s->ug, s->up, s->vp, s->wp);   /* src/program.f90:158 */
  if (s->dpdxl) orc_forces(g, s->dpdxl, s->dpdyl, s->up, s->vp, s->wp);
  if (s->dpdxl && g->ltempeq && !g->lmoist) orc_buoyancy(g, s->thl0, s->wp);
  if (s->dpdxl && g->lmoist && s->thermo) orc_buoyancy_moist(g, s, s->wp);
  if (g->ltempeq && s->dpdxl && s->thlpcar)                                             /* src/modforces.f90:104-110 */
    for (int k = 1; k <= g->nz; ++k)
      for (int j = 1; j <= g->ny; ++j)
        for (int i = 1; i <= g->nx; ++i) M(s->thlp, i, j, k) = M(s->thlp, i, j, k) + s->thlpcar[k];
  if (ibm_ctx) orc_ibmwallfun(g, ibm_ctx, s);                                          /* src/program.f90:166 */
  orc_masscorr(g, rk3step, dt, s->up, s->um, s->vp, s->vm);                            /* src/program.f90:169 */
  if (ibm_ctx) orc_ibmnorm(g, ibm_ctx, s);                                             /* src/program.f90:171 */
  if (s->svsrc)                                                                          /* scalsource, src/program.f90:181 */
    for (size_t q = 0; q < (size_t)g->nsv * nc; ++q) s->svp[q] = s->svp[q] + s->svsrc[q];
  orc_set_lid(g->bctopm == 3 ? s->pres0 : NULL, g->bctopm == 3 ? s->wp : NULL);
  orc_set_open_x_fields(xo_on ? s->u0 : NULL, xo_on ? s->up : NULL);
  orc_fillps(g, rk3coef, s->up, s->vp, s->wp, s->um, s->vm, s->wm, s->pup, s->pvp, s->pwp, s->p);
  orc_set_lid(NULL, NULL);
  orc_set_open_x_fields(NULL, NULL);
  orc_poisson_solve(g, s->p);
  orc_tderive(g, s->p, s->up, s->vp, s->wp, s->pres0);
  orc_tstep_integrate(g, rk3step, dt, s->u0, s->v0, s->w0, s->um, s->vm, s->wm, s->up, s->vp, s->wp,
                      s->sv0, s->svm, s->svp);
  if (g->sgs == 3) {                                                                     /* src/modtstep.f90:209-211,327,335 */
    const size_t nm = msize(g);
    const double rk3c = dt / (4. - (double)rk3step), e12min = 5.e-5;
    for (int k = 1; k <= g->nz; ++k)
      for (int j = 1; j <= g->ny; ++j)
        for (int i = 1; i <= g->nx; ++i) {
          M(s->e120, i, j, k) = M(s->e12m, i, j, k) + rk3c * M(s->e12p, i, j, k);
          M(s->e120, i, j, k) = fmax(e12min, M(s->e120, i, j, k));
          M(s->e12m, i, j, k) = fmax(e12min, M(s->e12m, i, j, k));
        }
    memset(s->e12p, 0, nm * sizeof(double));
    if (rk3step == 3) memcpy(s->e12m, s->e120, nm * sizeof(double));
    orc_halos_m(g, s->e120); orc_halos_m(g, s->e12m);
    for (int j = 0; j <= g->ny + 1; ++j)                      /* boundary, src/modboundary.f90:180-181 */
      for (int i = 0; i <= g->nx + 1; ++i) { M(s->e120, i, j, g->nz + 1) = e12min; M(s->e12m, i, j, g->nz + 1) = e12min; }
  }
  if (g->ltempeq) {                                                                      /* src/modtstep.f90:240-249,325,334 */
    const size_t nm = msize(g);
    const double rk3c = dt / (4. - (double)rk3step);
    for (int k = 1; k <= g->nz; ++k)
      for (int j = 1; j <= g->ny; ++j)
        for (int i = 1; i <= g->nx; ++i) M(s->thl0, i, j, k) = M(s->thlm, i, j, k) + rk3c * M(s->thlp, i, j, k);
    memset(s->thlp, 0, nm * sizeof(double));
    if (rk3step == 3) memcpy(s->thlm, s->thl0, nm * sizeof(double));
    if (xo_thlprof) { orc_halos_m(g, s->thl0); orc_halos_m(g, s->thlm); }
    else { halos_m_periodic_x(g, s->thl0); halos_m_periodic_x(g, s->thlm); }      /* (BCxT = 1 beside an open flow) */
  }
  if (g->lmoist) {                                                                       /* src/modtstep.f90:256,328,337 */
    const size_t nm = msize(g);
    const double rk3c = dt / (4. - (double)rk3step);
    for (int k = 1; k <= g->nz; ++k)
      for (int j = 1; j <= g->ny; ++j)
        for (int i = 1; i <= g->nx; ++i) M(s->qt0, i, j, k) = M(s->qtm, i, j, k) + rk3c * M(s->qtp, i, j, k);
    memset(s->qtp, 0, nm * sizeof(double));
    if (rk3step == 3) memcpy(s->qtm, s->qt0, nm * sizeof(double));
    if (xo_qtprof) { orc_halos_m(g, s->qt0); orc_halos_m(g, s->qtm); }
    else { halos_m_periodic_x(g, s->qt0); halos_m_periodic_x(g, s->qtm); }      /* (BCxq = 1 beside an open flow) */
  }
  orc_halos_m(g, s->u0); orc_halos_m(g, s->v0); orc_halos_m(g, s->w0);
  orc_halos_m(g, s->um); orc_halos_m(g, s->vm); orc_halos_m(g, s->wm);
  for (int n = 0; n < g->nsv; ++n) { orc_halos_c(g, s->sv0 + n * nc); orc_halos_c(g, s->svm + n * nc); }
  orc_boundary(g, s->u0, s->v0, s->w0, s->um, s->vm, s->wm, s->sv0, s->svm);
  if (xo_on) { xo_uouttot = uouttot_next; orc_boundary_open_x(g, rk3coef, s->u0, s->v0, s->w0, s->um, s->vm, s->wm); }
  if (g->nsv > 0 && scalar_top_active(g)) orc_scalar_tops(g, s->ekh, s->sv0, s->svm);      /* src/modboundary.f90:236-247 */
  if (g->ltempeq) { orc_thl_top(g, s->ekh, s->thlm); orc_thl_top(g, s->ekh, s->thl0); }     /* src/modboundary.f90:207-217 */
  if (g->ltempeq && xo_on) orc_boundary_open_x_thl(g, rk3coef, s->thl0, s->thlm);            /* :270-283, 377 */
  if (g->nsv > 0 && xo_on) orc_boundary_open_x_sv(g, rk3coef, s->sv0, s->svm);                /* :300-314, 379 */
  if (g->ltempeq && g->iadv_thl == 7) orc_thl0c_from(g, s->thl0, s->thl0c);                 /* src/modtstep.f90:249 + halos + boundary */
  if (g->lmoist) { orc_qt_top(g, s->ekh, s->qtm); orc_qt_top(g, s->ekh, s->qt0); }          /* src/modboundary.f90:222-231 */
  if (g->lmoist && xo_on) orc_boundary_open_x_qt(g, rk3coef, s->qt0, s->qtm);               /* :285-297, 378 */
  if (g->lmoist && s->thermo) orc_thermodynamics(g, s);                                     /* src/program.f90:214 */
}
