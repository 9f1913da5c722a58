// Inflow / outflow in x (&BC BCxm = 2: inflow from prof.inp's profile, convective outflow; the reference then opens the lid too,
// src/modstartup.f90:830-849).
//
// The device grid of such a handle (udc_create_open_x) is the deck's grid plus ONE ghost column at either end of every row:
// device column c is the reference's i = c (ib - 1 = 0, ib = 1, ..., ie = itot, ie + 1 = itot + 1), g.nx = itot + 2, g.xg = 1.  With passive
// scalars two ghost columns (g.xg = 2, g.nx = itot + 4, column c is i = c - 1): the kappa scheme reads i-2 .. i+1.
// Every sweep runs on it unchanged, as if those itot + 2 columns were periodic: an interior cell then finds the reference's ghost
// values where its stencil looks for them, and what the sweeps leave IN the two ghost columns (a tendency formed across the seam, a
// velocity integrated from it) is never read before the kernels below have put the reference's values there:
//
//   closurebc     src/modboundary.f90:467-475    ekm, ekh: ghost = the column next to it                  k_xo_ek_ghosts
//   bcpup         :1257-1280                     pup(ib) = uprof / rk3coef, up(ib) = 0; pup(ie+1) convective,
//                                                at kb a copy of pup(ie)                                   k_xo_bcpup
//   poisson       src/modpois.f90:113-121        cosine transform in x (Neumann at both ends)             k_xo_poisson
//   bcp           src/modboundary.f90:1376-1394  p, pres0: ghost = the column next to it                   k_xo_poisson, k_xo_boundary
//   xmi_profile   :688-717                       u(ib) = uprof; the inlet ghosts mirrored about the profile
//   xmo_convective :908-926                      v, w (0 and m) at ie+1 carried by uouttot                 k_xo_boundary
//   tstep_integrate src/modtstep.f90:191-264, 322 v, w at ie+1 untouched; vm = v0 at stage 3 there too       k_xo_after_integrate
//
// u(ie+1) is prognostic in the reference (src/modtstep.f90:262-264) and here simply a cell of the extended row: bcpup gives it its
// tendency, the projection finds p(ie+1) = p(ie) and leaves it alone, the integration advances it.  v and w at ie+1 are the outlet's
// own state: they live in planes of their own (xo_east), which every integration writes back over what it left in the ghost column
// and every `boundary` advances.  The reductions that span a level or the domain (slab sums, Courant and diffusion numbers, the divergence
// check) leave the ghost columns out (Geo::xg).
//
// The pressure solve: a cosine transform in x is the Fourier transform of the row followed by its mirror image, and the Neumann
// problem's solution is the first half of the periodic solution on that doubled row (same second-order Laplacian, same eigenvalues
// -4 dxi^2 sin^2(pi m / (2 itot))).  So the solve runs, unchanged, on a second handle that is 2 itot wide and holds nothing but p and
// the solver's arrays: twice the transform work of a dedicated DCT (the solve is ~40 % of a substep), no second set of line kernels.
#include "udc_internal.h"
#include <algorithm>
#include <vector>

namespace {

// one thread per (row, plane) of the padded array: jj = j + HY, kk = k + HZ
__device__ __forceinline__ bool plane_decode(const Geo &g, int &jj, int &kk) {
  jj = blockIdx.x * blockDim.x + threadIdx.x;
  kk = blockIdx.y;
  return jj < g.py;
}
inline dim3 plane_grid(const Geo &g) { return dim3((unsigned)((g.py + 63) / 64), (unsigned)g.pz); }

__global__ void xo_ek_kernel(Geo g, double *__restrict__ ekm, double *__restrict__ ekh) {
  int jj, kk;
  if (!plane_decode(g, jj, kk)) return;
  const long r = (long)g.sy * jj + g.sz * kk + (g.xg - 1);      // the ghost column ib-1 (xg = 2: one more beyond it, for the kappa scalars)
  const int e = g.nx - 2 * g.xg + 1;                             // ... from there to ie+1
  ekm[r] = ekm[r + 1]; ekm[r + e] = ekm[r + e - 1];
  ekh[r] = ekh[r + 1]; ekh[r + e] = ekh[r + e - 1];
}

// bcpup, BCxm_profile.  PUP: the tendency array holds the predicted velocity pup = up + um / rk3coef
// inlet: the profile's uprof(k), or (BCxm_driver, :1282-1303) the precursor plane u0driver(j, k) -- there the outlet is convective
// at kb too
template <bool PUP>
__global__ void xo_bcpup_kernel(Geo g, double rk3coefi, double dxi, const double *__restrict__ prof, const double *__restrict__ inlet,
                                const double *__restrict__ uout,
                                const double *__restrict__ u0, const double *__restrict__ um, double *__restrict__ up,
                                const double *__restrict__ pres0) {
  int jj, kk;
  if (!plane_decode(g, jj, kk)) return;
  const int j = jj - HY, k = kk - HZ;
  if (j < -1 || j > g.ny || k < 0 || k >= g.nz) return;
  const long r = (long)g.sy * jj + g.sz * kk + (g.xg - 1);      // the ghost column ib-1 (xg = 2: one more beyond it, for the kappa scalars)
  const int e = g.nx - 2 * g.xg + 1;                             // ... from there to ie+1
  const double uin = inlet ? inlet[(long)kk * g.py + jj] : prof[k + 1];
  up[r + 1] = PUP ? uin * rk3coefi : 0.;
  const double ume = um[r + e];
  double pe;
  if (k > 0 || inlet) pe = ume * rk3coefi - (u0[r + e] - u0[r + e - 1]) * dxi * uout[0];
  else {
    pe = PUP ? up[r + e - 1] : up[r + e - 1] + um[r + e - 1] * rk3coefi;      // "Neumann at bottom": pup(ie+1, kb) = pup(ie, kb)
    // (pressure-total form: the sweep left grad pres0 out of pup(ie); the reference's copy carries it -- and nothing else at ie+1 does)
    if (pres0) pe -= (pres0[r + e - 1] - pres0[r + e - 2]) * dxi;
  }
  up[r + e] = PUP ? pe : pe - ume * rk3coefi;
}

// xmi_profile, then xmo_convective on the outlet's planes; bcp's columns of pres0
__global__ void xo_boundary_kernel(Geo g, const double *__restrict__ prof, const double *__restrict__ inlet, double dxi, double rk3coef,
                                   const double *__restrict__ uout, double *__restrict__ u0, double *__restrict__ v0, double *__restrict__ w0,
                                   double *__restrict__ um, double *__restrict__ vm, double *__restrict__ wm,
                                   double *__restrict__ pres0, double *__restrict__ east, double *__restrict__ west, int stage3) {
  int jj, kk;
  if (!plane_decode(g, jj, kk)) return;
  const int j = jj - HY, k = kk - HZ;
  const long r = (long)g.sy * jj + g.sz * kk + (g.xg - 1);      // the ghost column ib-1 (xg = 2: one more beyond it, for the kappa scalars)
  const int e = g.nx - 2 * g.xg + 1;                             // ... from there to ie+1
  const long P = (long)g.py * g.pz, q = (long)kk * g.py + jj;
  if (inlet) {      // xmi_driver, src/modboundary.f90:720-749: u at ib and ib-1, v at ib-1 on kb .. ke; w at ib-1 on kb .. ke+1
    if (j >= -1 && j <= g.ny && k >= 0 && k <= g.nz) {
      if (k < g.nz) {
        u0[r + 1] = inlet[q]; um[r + 1] = inlet[P + q];
        u0[r] = inlet[q]; um[r] = inlet[P + q];
        v0[r] = inlet[2 * P + q]; vm[r] = inlet[3 * P + q];
      }
      w0[r] = inlet[4 * P + q]; wm[r] = inlet[5 * P + q];
    }
  } else
  if (j >= -1 && j <= g.ny && k >= 0 && k <= g.nz) {      // j = jb-1 .. je+1, k = kb .. ke+1
    const double up = prof[k + 1], vp = prof[g.nz + 2 + k + 1];
    u0[r + 1] = up; um[r + 1] = up;
    u0[r] = 2 * up - u0[r + 2]; um[r] = 2 * up - um[r + 2];
    v0[r] = 2 * vp - v0[r + 1]; vm[r] = 2 * vp - vm[r + 1];
    w0[r] = -w0[r + 1]; wm[r] = -wm[r + 1];
  }
  if (j >= -1 && j <= g.ny && k >= 0 && k < g.nz) { pres0[r] = pres0[r + 1]; pres0[r + e] = pres0[r + e - 1]; }
  // the outlet: every row and plane the arrays hold
  double ev0 = east[q], ew0 = east[P + q], evm = east[2 * P + q], ewm = east[3 * P + q];
  // (stage3 > 0: the fused substep -- this launch stands for xo_restore_kernel too, whose every array element it overwrites: the m planes
  //  take the 0 planes first, as tstep_integrate's vm = v0 on RK stage 3)
  if (stage3 > 0) { evm = ev0; ewm = ew0; }
  const double uo = uout[0];
  ev0 = ev0 - (ev0 - v0[r + e - 1]) * dxi * rk3coef * uo;
  ew0 = ew0 - (ew0 - w0[r + e - 1]) * dxi * rk3coef * uo;
  evm = evm - (evm - vm[r + e - 1]) * dxi * rk3coef * uo;
  ewm = ewm - (ewm - wm[r + e - 1]) * dxi * rk3coef * uo;
  east[q] = ev0; east[P + q] = ew0; east[2 * P + q] = evm; east[3 * P + q] = ewm;
  v0[r + e] = ev0; w0[r + e] = ew0; vm[r + e] = evm; wm[r + e] = ewm;
  // the inlet's ghost column as it stands now: what the next integration must leave there
  west[q] = u0[r]; west[P + q] = v0[r]; west[2 * P + q] = w0[r]; west[3 * P + q] = um[r]; west[4 * P + q] = vm[r]; west[5 * P + q] = wm[r];
}

// after the integration: v, w at ie+1 are not the integration's to touch (src/modtstep.f90:191-264 runs over ib:ie) -- the ghost
// column takes the outlet's plane back; on RK stage 3 the m planes first take the 0 planes (vm = v0, wm = w0 are whole-array copies,
// :322-324)
__global__ void xo_restore_kernel(Geo g, int stage3, double *__restrict__ u0, double *__restrict__ v0, double *__restrict__ w0,
                                  double *__restrict__ um, double *__restrict__ vm, double *__restrict__ wm,
                                  double *__restrict__ east, double *__restrict__ west) {
  int jj, kk;
  if (!plane_decode(g, jj, kk)) return;
  const long r0 = (long)g.sy * jj + g.sz * kk + (g.xg - 1), r = r0 + g.nx - 2 * g.xg + 1;      // ib-1, ie+1
  const long P = (long)g.py * g.pz, q = (long)kk * g.py + jj;
  const double ev0 = east[q], ew0 = east[P + q];
  double evm = east[2 * P + q], ewm = east[3 * P + q];
  if (stage3) { evm = ev0; ewm = ew0; east[2 * P + q] = evm; east[3 * P + q] = ewm; }
  v0[r] = ev0; w0[r] = ew0; vm[r] = evm; wm[r] = ewm;
  // ... and the inlet's ghost column ib-1 (what host code sees between the integration and `boundary`: statsdump, fielddump, checksim)
  const double a0 = west[q], a1 = west[P + q], a2 = west[2 * P + q];
  double b0 = west[3 * P + q], b1 = west[4 * P + q], b2 = west[5 * P + q];
  if (stage3) { b0 = a0; b1 = a1; b2 = a2; west[3 * P + q] = b0; west[4 * P + q] = b1; west[5 * P + q] = b2; }
  u0[r0] = a0; v0[r0] = a1; w0[r0] = a2; um[r0] = b0; vm[r0] = b1; wm[r0] = b2;
}

// the temperature (m-array in scalar slot 15, central scheme): xTi_profile (src/modboundary.f90:766-793: thl(ib-1) = thlprof(k) on
// kb .. ke+1, thl(ib) = thlprof(k) on kb .. ke), xTo_convective (:947-957) on the outlet's planes; the columns as they stand are kept
// for the next integration to put back
// MODE 2: the profile (thl: both columns take it; QT, xqi_profile :811-823: the ghost mirrored about it); MODE 3: the ghost column from the
// planes f0driver, fmdriver of a precursor run (xTi_driver :795-808, xqi_driver :826-839).  QT's outlet as the reference has it
// (xqo_convective :961-971 starts from qt(ie), not from qt(ie+1))
template <bool QT>
__global__ void xo_thl_boundary_kernel(Geo g, const double *__restrict__ prof, const double *__restrict__ inlet, double dxi, double rk3coef,
                                       const double *__restrict__ uout,
                                       double *__restrict__ t0, double *__restrict__ tm, double *__restrict__ east, double *__restrict__ west) {
  int jj, kk;
  if (!plane_decode(g, jj, kk)) return;
  const int j = jj - HY, k = kk - HZ;
  const long r = (long)g.sy * jj + g.sz * kk + (g.xg - 1);      // the ghost column ib-1 (xg = 2: one more beyond it, for the kappa scalars)
  const int e = g.nx - 2 * g.xg + 1;                             // ... from there to ie+1
  const long P = (long)g.py * g.pz, q = (long)kk * g.py + jj;
  if (j >= -1 && j <= g.ny && k >= 0 && k <= g.nz) {
    if (inlet) { t0[r] = inlet[q]; tm[r] = inlet[P + q]; }
    else {
      const double tp = prof[k + 1];
      if (QT) { t0[r] = 2 * tp - t0[r + 1]; tm[r] = 2 * tp - tm[r + 1]; }
      else {
        t0[r] = tp; tm[r] = tp;
        if (k < g.nz) { t0[r + 1] = tp; tm[r + 1] = tp; }
      }
    }
  }
  double e0 = east[q], em = east[P + q];
  const double uo = uout[0];
  e0 = (QT ? t0[r + e - 1] : e0) - (e0 - t0[r + e - 1]) * dxi * rk3coef * uo;
  em = (QT ? tm[r + e - 1] : em) - (em - tm[r + e - 1]) * dxi * rk3coef * uo;
  east[q] = e0; east[P + q] = em;
  t0[r + e] = e0; tm[r + e] = em;
  west[q] = t0[r]; west[P + q] = tm[r];
}
__global__ void xo_thl_restore_kernel(Geo g, int stage3, double *__restrict__ t0, double *__restrict__ tm, double *__restrict__ east,
                                      double *__restrict__ west) {
  int jj, kk;
  if (!plane_decode(g, jj, kk)) return;
  const long r = (long)g.sy * jj + g.sz * kk + (g.xg - 1);
  const int e = g.nx - 2 * g.xg + 1;
  const long P = (long)g.py * g.pz, q = (long)kk * g.py + jj;
  const double e0 = east[q], w0 = west[q];
  double em = east[P + q], wm = west[P + q];
  if (stage3) { em = e0; wm = w0; east[P + q] = em; west[P + q] = wm; }      // thlm = thl0: a whole-array copy (src/modtstep.f90:325)
  t0[r + e] = e0; tm[r + e] = em;
  t0[r] = w0; tm[r] = wm;
}

// BCxT = 2 and a zero-flux top (BCtopT = 1, wttop = 0): reassure_fluxtop_boundary (src/modboundary.f90:392-431, from closurebc) re-derives
// thl(ke+1) = thl(ke) between advection and diffusion.  Everywhere that is the identity -- except in the first interior column, which
// xTi_profile has set to the profile AFTER `boundary` formed the top row from what the integration had left there.  Nothing of this reaches
// the state (the column is overwritten again), but statsdump samples thlm(ib, j, ke) as integrated (src/program.f90:199-214).
// FIX = false: the row as closurebc leaves it, the value advection saw kept aside ([py]);
// FIX = true (the fused sweep, which advects and diffuses in one pass over the reassured row): the top face's advective flux, the one
// term that should have seen the row of `boundary` (advecc_2nd, src/modadvection.f90:146-150), corrected in the tendency
template <bool FIX>
__global__ void xo_thl_top_kernel(Geo g, Metrics m, double *__restrict__ t0, double *__restrict__ tm, double *__restrict__ keep,
                                  const double *__restrict__ w0, double *__restrict__ tp) {
  const int jj = blockIdx.x * blockDim.x + threadIdx.x;
  if (jj >= g.py) return;
  const long r = (long)g.sy * jj + g.sz * (long)(g.nz - 1 + HZ) + g.xg, gh = r + g.sz;      // (ib, j, ke) and the row above
  if (!FIX) {
    keep[jj] = t0[gh];
    t0[gh] = t0[r]; tm[gh] = tm[r];
  } else {
    const int ke = g.nz;
    tp[r] = tp[r] - (w0[gh] * ((keep[jj] - t0[gh]) * m.dzf[ke]) * m.dzhi[ke + 1]) * m.dzfi5[ke];
  }
}

// the temperature and the total water where they stay periodic in x while the flow enters and leaves (&BC BCxT = 1 / BCxq = 1 next to
// BCxm = 2 / 3: the reference's defaults, its tests/cases/525): `halos` refreshes their x ghosts right after every integration
// (xT_periodic, xq_periodic, src/modboundary.f90:543-577, called under `ibrank .and. ierank`, :95-100) -- every row and level the arrays hold
__global__ void xo_wrap_kernel(Geo g, double *__restrict__ f0, double *__restrict__ fm) {
  int jj, kk;
  if (!plane_decode(g, jj, kk)) return;
  const long r = (long)g.sy * jj + g.sz * kk + (g.xg - 1);
  const int e = g.nx - 2 * g.xg + 1;
  f0[r] = f0[r + e - 1]; f0[r + e] = f0[r + 1];
  fm[r] = fm[r + e - 1]; fm[r + e] = fm[r + 1];
}

// passive scalars (c-arrays, kappa scheme: two ghost columns either side).  xsi_profile (src/modboundary.f90:844-861: on jb .. je, kb .. ke+1,
// sv(ib-1) = 2 svprof - sv(ib), sv(ib-2) = 2 svprof - sv(ib-1), sv0 and svm), xso_convective (:983-996: sv(ie+1) on every row and level; ie+2
// stays what it is); cols [8][P]: sv0 at ib-2, ib-1, ie+1, ie+2, then svm
// inlet: xsi_driver (:883-901): both ghost columns take the precursor's plane, rows jb-1 .. je+1
__global__ void xo_sv_boundary_kernel(Geo g, const double *__restrict__ prof, const double *__restrict__ inlet, double dxi, double rk3coef,
                                      const double *__restrict__ uout,
                                      double *__restrict__ s0, double *__restrict__ sm, double *__restrict__ cols) {
  int jj, kk;
  if (!plane_decode(g, jj, kk)) return;
  const int j = jj - HY, k = kk - HZ;
  const long r = (long)g.sy * jj + g.sz * kk + (g.xg - 1);
  const int e = g.nx - 2 * g.xg + 1;
  const long P = (long)g.py * g.pz, q = (long)kk * g.py + jj;
  if (inlet) {
    if (j >= -1 && j <= g.ny && k >= 0 && k <= g.nz) { s0[r] = inlet[q]; s0[r - 1] = inlet[q]; sm[r] = inlet[P + q]; sm[r - 1] = inlet[P + q]; }
  } else
  if (j >= 0 && j < g.ny && k >= 0 && k <= g.nz) {
    const double sp = prof[k + 1];
    s0[r] = 2 * sp - s0[r + 1]; sm[r] = 2 * sp - sm[r + 1];
    s0[r - 1] = 2 * sp - s0[r]; sm[r - 1] = 2 * sp - sm[r];
  }
  double e0 = cols[2 * P + q], em = cols[6 * P + q];
  const double uo = uout[0];
  e0 = e0 - (e0 - s0[r + e - 1]) * dxi * rk3coef * uo;
  em = em - (em - sm[r + e - 1]) * dxi * rk3coef * uo;
  s0[r + e] = e0; sm[r + e] = em;
  s0[r + e + 1] = cols[3 * P + q]; sm[r + e + 1] = cols[7 * P + q];
  cols[q] = s0[r - 1]; cols[P + q] = s0[r]; cols[2 * P + q] = e0;
  cols[4 * P + q] = sm[r - 1]; cols[5 * P + q] = sm[r]; cols[6 * P + q] = em;
}
__global__ void xo_sv_restore_kernel(Geo g, int stage3, double *__restrict__ s0, double *__restrict__ sm, double *__restrict__ cols) {
  int jj, kk;
  if (!plane_decode(g, jj, kk)) return;
  const long r = (long)g.sy * jj + g.sz * kk + (g.xg - 1);
  const int e = g.nx - 2 * g.xg + 1;
  const long P = (long)g.py * g.pz, q = (long)kk * g.py + jj;
  double c[8];
  for (int t = 0; t < 8; ++t) c[t] = cols[t * P + q];
  if (stage3) for (int t = 0; t < 4; ++t) { c[4 + t] = c[t]; cols[(4 + t) * P + q] = c[t]; }      // svm = sv0: a whole-array copy (src/modtstep.f90:326)
  s0[r - 1] = c[0]; s0[r] = c[1]; s0[r + e] = c[2]; s0[r + e + 1] = c[3];
  sm[r - 1] = c[4]; sm[r] = c[5]; sm[r + e] = c[6]; sm[r + e + 1] = c[7];
}
__global__ void xo_sv_capture_kernel(Geo g, const double *__restrict__ f, double *__restrict__ cols4) {
  int jj, kk;
  if (!plane_decode(g, jj, kk)) return;
  const long r = (long)g.sy * jj + g.sz * kk + (g.xg - 1);
  const int e = g.nx - 2 * g.xg + 1;
  const long P = (long)g.py * g.pz, q = (long)kk * g.py + jj;
  cols4[q] = f[r - 1]; cols4[P + q] = f[r]; cols4[2 * P + q] = f[r + e]; cols4[3 * P + q] = f[r + e + 1];
}

// the ghost column of an uploaded v0 / w0 / vm / wm -> the outlet's plane
__global__ void xo_capture_kernel(Geo g, const double *__restrict__ f, double *__restrict__ plane, int col) {
  int jj, kk;
  if (!plane_decode(g, jj, kk)) return;
  plane[(long)kk * g.py + jj] = f[(long)g.sy * jj + g.sz * kk + col];
}

// the right-hand side's interior columns and their mirror image -> the doubled row; back: the first half, and bcp's ghost columns
__global__ __launch_bounds__(256) void xo_gather_kernel(Geo g, Geo g2, const double *__restrict__ p, double *__restrict__ p2) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y, k = blockIdx.z;
  const int n = g.nx - 2 * g.xg;
  if (i >= n) return;
  const double v = p[g.idx(i + g.xg, j, k)];
  p2[g2.idx(i, j, k)] = v;
  p2[g2.idx(2 * n - 1 - i, j, k)] = v;
}
__global__ __launch_bounds__(256) void xo_scatter_kernel(Geo g, Geo g2, const double *__restrict__ p2, double *__restrict__ p) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y, k = blockIdx.z;
  const int n = g.nx - 2 * g.xg;
  if (i >= n) return;
  const double v = p2[g2.idx(i, j, k)];
  p[g.idx(i + g.xg, j, k)] = v;
  if (i == 0) p[g.idx(g.xg - 1, j, k)] = v;
  if (i == n - 1) p[g.idx(n + g.xg, j, k)] = v;
}

}  // namespace

int xo_init(udc_handle *h, const double *uprof, const double *vprof) {
  const Geo &g = h->g;
  const size_t nk = (size_t)g.nz + 2, np = (size_t)g.py * g.pz;
  HIP_OK(hipMalloc(&h->xo_prof, sizeof(double) * 2 * nk));
  HIP_OK(hipMemcpy(h->xo_prof, uprof, sizeof(double) * nk, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(h->xo_prof + nk, vprof, sizeof(double) * nk, hipMemcpyHostToDevice));
  HIP_OK(hipMalloc(&h->xo_east, sizeof(double) * 4 * np));
  HIP_OK(hipMemset(h->xo_east, 0, sizeof(double) * 4 * np));
  HIP_OK(hipMalloc(&h->xo_west, sizeof(double) * 6 * np));
  HIP_OK(hipMemset(h->xo_west, 0, sizeof(double) * 6 * np));
  if (!h->bcx_uout_dev) {
    HIP_OK(hipMalloc(&h->bcx_uout_dev, sizeof(double)));
    HIP_OK(hipMemset(h->bcx_uout_dev, 0, sizeof(double)));
  }
  return 0;
}

void xo_destroy(udc_handle *h) {
  if (h->xo_inlet_now) { hipFree(h->xo_inlet_now); h->xo_inlet_now = nullptr; }
  if (h->xo_inlet_next) { hipFree(h->xo_inlet_next); h->xo_inlet_next = nullptr; }
  if (h->xo_prof) { hipFree(h->xo_prof); h->xo_prof = nullptr; }
  for (double **q : {&h->xo_thl_top, &h->xo_thl_prof, &h->xo_thl_east, &h->xo_thl_west, &h->xo_sv_prof, &h->xo_qt_prof, &h->xo_qt_east, &h->xo_qt_west})
    if (*q) { hipFree(*q); *q = nullptr; }
  for (int t = 0; t < 15; ++t)
    for (double **q : {&h->xo_sc_in_now[t], &h->xo_sc_in_next[t]}) if (*q) { hipFree(*q); *q = nullptr; }
  for (double *&q : h->xo_sv_cols) if (q) { hipFree(q); q = nullptr; }
  if (h->xo_east) { hipFree(h->xo_east); h->xo_east = nullptr; }
  if (h->xo_west) { hipFree(h->xo_west); h->xo_west = nullptr; }
  if (h->xpois) { udc_destroy(h->xpois); h->xpois = nullptr; }
}

// an m-scalar's outlet and inlet columns as they stand ([2][pz][py] each: the 0 and the m array), allocated on first use and filled from the
// fields' own ghost columns (an upload that came before the call has put the host's there)
static int xo_alloc_east_west(udc_handle *h, int f0, int fm, double **east, double **west) {
  if (*east) return 0;
  const Geo &g = h->g;
  const size_t np = (size_t)g.py * g.pz;
  HIP_OK(hipMalloc(east, sizeof(double) * 2 * np));
  HIP_OK(hipMalloc(west, sizeof(double) * 2 * np));
  for (int t = 0; t < 2; ++t) {
    const double *f = h->fields[t ? fm : f0];
    hipLaunchKernelGGL(xo_capture_kernel, plane_grid(g), dim3(64), 0, h->stream, g, f, *east + t * np, g.nx - g.xg);
    hipLaunchKernelGGL(xo_capture_kernel, plane_grid(g), dim3(64), 0, h->stream, g, f, *west + t * np, g.xg - 1);
  }
  HIP_OK(hipGetLastError());
  return 0;
}

// planes (lb[0]:ub[0], lb[1]:ub[1]) in the reference's j, k (j fastest) -> [n][pz][py] on the device: into `next`, and into `now` too the
// first time (what bcpup reads before the first `boundary` has run)
static int stage_planes(udc_handle *h, int n, const double *const *src, const int lb[2], const int ub[2], double **now, double **next) {
  const Geo &g = h->g;
  const size_t np = (size_t)g.py * g.pz;
  const bool first = !*next;
  if (first) {
    HIP_OK(hipMalloc(next, sizeof(double) * n * np));
    HIP_OK(hipMalloc(now, sizeof(double) * n * np));
  }
  std::vector<double> st(n * np, 0.);
  const int nj = ub[0] - lb[0] + 1;
  for (int f = 0; f < n; ++f)
    for (int k = std::max(lb[1], 1 - HZ); k <= std::min(ub[1], g.nz + HZ); ++k)
      for (int j = std::max(lb[0], 1 - HY); j <= std::min(ub[0], g.ny + HY); ++j)
        st[f * np + (size_t)(k - 1 + HZ) * g.py + (j - 1 + HY)] = src[f][(size_t)(k - lb[1]) * nj + (j - lb[0])];
  HIP_OK(hipStreamSynchronize(h->stream));
  HIP_OK(hipMemcpy(*next, st.data(), sizeof(double) * n * np, hipMemcpyHostToDevice));
  if (first) HIP_OK(hipMemcpy(*now, st.data(), sizeof(double) * n * np, hipMemcpyHostToDevice));
  return 0;
}

// BCxm = 3: the six inlet planes of the reference's drivergen (modinletdata u0driver, umdriver, v0driver, vmdriver, w0driver, wmdriver),
// each (lb[0]:ub[0], lb[1]:ub[1]) in the reference's j, k, j fastest
extern "C" int udc_set_open_x_inlet(udc_handle *h, const double *u0d, const double *umd, const double *v0d, const double *vmd,
                                    const double *w0d, const double *wmd, const int lb[2], const int ub[2]) {
  if (!h) { udc_set_error("null handle"); return 1; }
  HIP_OK(hipSetDevice(h->device));
  if (udc_flush_pending(h)) return 1;
  if (!h->xg) { udc_set_error("udc_set_open_x_inlet: not a handle of udc_create_open_x"); return 1; }
  const double *src[6] = {u0d, umd, v0d, vmd, w0d, wmd};
  for (const double *s : src) if (!s) { udc_set_error("udc_set_open_x_inlet: six planes"); return 1; }
  const Geo &g = h->g;
  if (lb[0] > 0 || ub[0] < g.ny + 1 || lb[1] > 1 || ub[1] < g.nz + 1) {
    udc_set_error("udc_set_open_x_inlet: the planes must cover j = jb-1 .. je+1, k = kb .. ke+1");
    return 1;
  }
  if (stage_planes(h, 6, src, lb, ub, &h->xo_inlet_now, &h->xo_inlet_next)) return 1;
  h->xo_driver = 1;
  h->xo_inlet_fresh = true;      // (takes effect with the next `boundary` that runs; it does not ask for one)
  return 0;
}

// BCxT / BCxq / BCxs = 3 (with BCxm = 3): the temperature's, the total water's or scalar n's inlet from the planes drivergen leaves in
// modinletdata (thl0driver / thlmdriver, qt0driver / qtmdriver: (jb-jh : je+jh, kb-kh : ke+kh); sv0driver / svmdriver(:, :, n):
// (jb-jhc : je+jhc, kb-khc : ke+khc)), handed over after every drivergen like the flow's
extern "C" int udc_set_open_x_inlet_scalar(udc_handle *h, int field, const double *f0d, const double *fmd, const int lb[2], const int ub[2]) {
  if (!h) { udc_set_error("null handle"); return 1; }
  HIP_OK(hipSetDevice(h->device));
  if (udc_flush_pending(h)) return 1;
  if (!h->xg || !f0d || !fmd) { udc_set_error("udc_set_open_x_inlet_scalar: a handle of udc_create_open_x and both planes"); return 1; }
  const Geo &g = h->g;
  if (lb[0] > 0 || ub[0] < g.ny + 1 || lb[1] > 1 || ub[1] < g.nz + 1) {
    udc_set_error("udc_set_open_x_inlet_scalar: the planes must cover j = jb-1 .. je+1, k = kb .. ke+1");
    return 1;
  }
  int slot = -1;
  if (field == UDC_THL0 || field == UDC_QT0) {
    if ((int)h->fields.size() <= field || !h->fields[field]) { udc_set_error("udc_set_open_x_inlet_scalar: udc_set_tempeq / udc_set_moisture first"); return 1; }
    slot = field == UDC_THL0 ? 0 : 1;
    if (xo_alloc_east_west(h, field, field + 1, slot ? &h->xo_qt_east : &h->xo_thl_east, slot ? &h->xo_qt_west : &h->xo_thl_west)) return 1;
  } else if (field >= UDC_SV0 && (field - UDC_SV0) % 3 == 0 && (field - UDC_SV0) / 3 < 13 && (field - UDC_SV0) / 3 < h->cfg.nsv) {
    slot = 2 + (field - UDC_SV0) / 3;
    if (!h->xo_sv_cols[slot - 2]) { udc_set_error("udc_set_open_x_inlet_scalar: udc_set_open_x_scalars first"); return 1; }
  } else { udc_set_error("udc_set_open_x_inlet_scalar: UDC_THL0, UDC_QT0 or the sv0 of a passive scalar"); return 1; }
  const double *src[2] = {f0d, fmd};
  if (stage_planes(h, 2, src, lb, ub, &h->xo_sc_in_now[slot], &h->xo_sc_in_next[slot])) return 1;
  h->xo_sc_fresh[slot] = true;
  return 0;
}

// the total water's inflow profile (&BC BCxq = 2: xqi_profile, xqo_convective), qtprof [ktot+2] by the reference's k; after udc_set_moisture
extern "C" int udc_set_open_x_qt(udc_handle *h, const double *qtprof) {
  if (!h || !qtprof) { udc_set_error("udc_set_open_x_qt: null argument"); return 1; }
  HIP_OK(hipSetDevice(h->device));
  if (udc_flush_pending(h)) return 1;
  if (!h->xg) { udc_set_error("udc_set_open_x_qt: not a handle of udc_create_open_x"); return 1; }
  if (!h->lmoist || (int)h->fields.size() <= UDC_QT0 || !h->fields[UDC_QT0]) { udc_set_error("udc_set_open_x_qt: call udc_set_moisture first"); return 1; }
  const Geo &g = h->g;
  const size_t nk = (size_t)g.nz + 2;
  HIP_OK(hipStreamSynchronize(h->stream));
  if (!h->xo_qt_prof) HIP_OK(hipMalloc(&h->xo_qt_prof, sizeof(double) * nk));
  if (xo_alloc_east_west(h, UDC_QT0, UDC_QTM, &h->xo_qt_east, &h->xo_qt_west)) return 1;
  HIP_OK(hipMemcpy(h->xo_qt_prof, qtprof, sizeof(double) * nk, hipMemcpyHostToDevice));
  return 0;
}

int xo_capture_east(udc_handle *h, int field, const double *, const int lb[3], const int ub[3]) {
  if (!h->xg) return 0;
  const Geo &g = h->g;
  const int itot = g.nx - 2 * g.xg;
  int slot = -1;
  if (field == UDC_V0) slot = 0; else if (field == UDC_W0) slot = 1; else if (field == UDC_VM) slot = 2; else if (field == UDC_WM) slot = 3;
  if (slot >= 0 && lb[0] <= itot + 1 && ub[0] >= itot + 1)      // (else the host array does not carry the column: the plane stays as it is)
    hipLaunchKernelGGL(xo_capture_kernel, plane_grid(g), dim3(64), 0, h->stream, g, (const double *)h->fields[field],
                       h->xo_east + (size_t)slot * g.py * g.pz, g.nx - g.xg);
  const int wslot = (field >= UDC_U0 && field <= UDC_WM) ? field - UDC_U0 : -1;      // u0 v0 w0 um vm wm
  if (wslot >= 0 && lb[0] <= 0 && ub[0] >= 0)
    hipLaunchKernelGGL(xo_capture_kernel, plane_grid(g), dim3(64), 0, h->stream, g, (const double *)h->fields[field],
                       h->xo_west + (size_t)wslot * g.py * g.pz, g.xg - 1);
  if (field >= UDC_SV0 && (field - UDC_SV0) % 3 < 2 && (field - UDC_SV0) / 3 < 13 && h->xo_sv_cols[(field - UDC_SV0) / 3] &&
      lb[0] <= -1 && ub[0] >= itot + 2)      // sv0 / svm with both ghost columns either side (the reference's c-arrays carry them)
    hipLaunchKernelGGL(xo_sv_capture_kernel, plane_grid(g), dim3(64), 0, h->stream, g, (const double *)h->fields[field],
                       h->xo_sv_cols[(field - UDC_SV0) / 3] + (size_t)4 * ((field - UDC_SV0) % 3) * g.py * g.pz);
  const int tslot = field == UDC_THL0 ? 0 : (field == UDC_THLM ? 1 : -1);
  if (tslot >= 0 && h->xo_thl_east) {
    if (lb[0] <= itot + 1 && ub[0] >= itot + 1)
      hipLaunchKernelGGL(xo_capture_kernel, plane_grid(g), dim3(64), 0, h->stream, g, (const double *)h->fields[field],
                         h->xo_thl_east + (size_t)tslot * g.py * g.pz, g.nx - g.xg);
    if (lb[0] <= 0 && ub[0] >= 0)
      hipLaunchKernelGGL(xo_capture_kernel, plane_grid(g), dim3(64), 0, h->stream, g, (const double *)h->fields[field],
                         h->xo_thl_west + (size_t)tslot * g.py * g.pz, g.xg - 1);
  }
  const int qslot = field == UDC_QT0 ? 0 : (field == UDC_QTM ? 1 : -1);
  if (qslot >= 0 && h->xo_qt_east) {
    if (lb[0] <= itot + 1 && ub[0] >= itot + 1)
      hipLaunchKernelGGL(xo_capture_kernel, plane_grid(g), dim3(64), 0, h->stream, g, (const double *)h->fields[field],
                         h->xo_qt_east + (size_t)qslot * g.py * g.pz, g.nx - g.xg);
    if (lb[0] <= 0 && ub[0] >= 0)
      hipLaunchKernelGGL(xo_capture_kernel, plane_grid(g), dim3(64), 0, h->stream, g, (const double *)h->fields[field],
                         h->xo_qt_west + (size_t)qslot * g.py * g.pz, g.xg - 1);
  }
  HIP_OK(hipGetLastError());
  return 0;
}

int k_xo_ek_ghosts(udc_handle *h) {
  if (!h->xg) return 0;
  const Geo &g = h->g;
  PROF(h, "xo_ghosts");
  hipLaunchKernelGGL(xo_ek_kernel, plane_grid(g), dim3(64), 0, h->stream, g, h->fields[UDC_EKM], h->fields[UDC_EKH]);
  HIP_OK(hipGetLastError());
  return 0;
}

int k_xo_bcpup(udc_handle *h, double rk3coef, bool pup, bool ptotal) {
  if (!h->xg) return 0;
  const Geo &g = h->g;
  PROF(h, "xo_ghosts");
  const double *inlet = h->xo_driver ? h->xo_inlet_now : nullptr;      // (u0driver: the first of its six planes)
  if (pup)
    hipLaunchKernelGGL(xo_bcpup_kernel<true>, plane_grid(g), dim3(64), 0, h->stream, g, 1. / rk3coef, h->m.dxi, (const double *)h->xo_prof, inlet,
                       (const double *)h->bcx_uout_dev, (const double *)h->fields[UDC_U0], (const double *)h->fields[UDC_UM], h->fields[UDC_UP],
                       (const double *)(ptotal ? h->fields[UDC_PRES0] : nullptr));
  else
    hipLaunchKernelGGL(xo_bcpup_kernel<false>, plane_grid(g), dim3(64), 0, h->stream, g, 1. / rk3coef, h->m.dxi, (const double *)h->xo_prof, inlet,
                       (const double *)h->bcx_uout_dev, (const double *)h->fields[UDC_U0], (const double *)h->fields[UDC_UM], h->fields[UDC_UP],
                       (const double *)nullptr);
  HIP_OK(hipGetLastError());
  return 0;
}

int k_xo_boundary(udc_handle *h, int merged_stage3) {
  if (!h->xg) return 0;
  const Geo &g = h->g;
  PROF(h, "xo_ghosts");
  if (h->xo_driver && h->xo_inlet_fresh) {      // the planes handed over since the last `boundary` are this one's (drivergen, src/modboundary.f90:262-265)
    HIP_OK(hipMemcpyAsync(h->xo_inlet_now, h->xo_inlet_next, sizeof(double) * 6 * (size_t)g.py * g.pz, hipMemcpyDeviceToDevice, h->stream));
    h->xo_inlet_fresh = false;
  }
  hipLaunchKernelGGL(xo_boundary_kernel, plane_grid(g), dim3(64), 0, h->stream, g, (const double *)h->xo_prof,
                     (const double *)(h->xo_driver ? h->xo_inlet_now : nullptr), h->m.dxi, h->bcx_rk3coef,
                     (const double *)h->bcx_uout_dev, h->fields[UDC_U0], h->fields[UDC_V0], h->fields[UDC_W0],
                     h->fields[UDC_UM], h->fields[UDC_VM], h->fields[UDC_WM], h->fields[UDC_PRES0], h->xo_east, h->xo_west, merged_stage3);
  for (int t = 0; t < 15; ++t)      // the scalars' planes handed over since the last `boundary`
    if (h->xo_sc_in_next[t] && h->xo_sc_fresh[t]) {
      HIP_OK(hipMemcpyAsync(h->xo_sc_in_now[t], h->xo_sc_in_next[t], sizeof(double) * 2 * (size_t)g.py * g.pz, hipMemcpyDeviceToDevice, h->stream));
      h->xo_sc_fresh[t] = false;
    }
  if (h->xo_thl_prof || h->xo_sc_in_now[0])
    hipLaunchKernelGGL(xo_thl_boundary_kernel<false>, plane_grid(g), dim3(64), 0, h->stream, g, (const double *)h->xo_thl_prof,
                       (const double *)h->xo_sc_in_now[0], h->m.dxi, h->bcx_rk3coef,
                       (const double *)h->bcx_uout_dev, h->fields[UDC_THL0], h->fields[UDC_THLM], h->xo_thl_east, h->xo_thl_west);
  if (h->xo_qt_prof || h->xo_sc_in_now[1])
    hipLaunchKernelGGL(xo_thl_boundary_kernel<true>, plane_grid(g), dim3(64), 0, h->stream, g, (const double *)h->xo_qt_prof,
                       (const double *)h->xo_sc_in_now[1], h->m.dxi, h->bcx_rk3coef,
                       (const double *)h->bcx_uout_dev, h->fields[UDC_QT0], h->fields[UDC_QTM], h->xo_qt_east, h->xo_qt_west);
  for (int n = 0; n < h->cfg.nsv && n < 13; ++n)
    if (h->xo_sv_cols[n])
      hipLaunchKernelGGL(xo_sv_boundary_kernel, plane_grid(g), dim3(64), 0, h->stream, g, (const double *)(h->xo_sv_prof + (size_t)n * (g.nz + 2)),
                         (const double *)h->xo_sc_in_now[2 + n],
                         h->m.dxi, h->bcx_rk3coef, (const double *)h->bcx_uout_dev, h->fields[UDC_SV0 + 3 * n], h->fields[UDC_SVM + 3 * n], h->xo_sv_cols[n]);
  HIP_OK(hipGetLastError());
  return 0;
}

// passive scalars on such a handle (&BC BCxs = 2): the inflow profiles svprof [nsv][ktot+2] by the reference's k
extern "C" int udc_set_open_x_scalars(udc_handle *h, const double *svprof) {
  if (!h || !svprof) { udc_set_error("udc_set_open_x_scalars: null argument"); return 1; }
  HIP_OK(hipSetDevice(h->device));
  if (udc_flush_pending(h)) return 1;
  if (!h->xg || h->cfg.nsv < 1) { udc_set_error("udc_set_open_x_scalars: a handle of udc_create_open_x with nsv > 0"); return 1; }
  const Geo &g = h->g;
  const size_t nk = (size_t)g.nz + 2, np = (size_t)g.py * g.pz;
  HIP_OK(hipStreamSynchronize(h->stream));
  if (!h->xo_sv_prof) HIP_OK(hipMalloc(&h->xo_sv_prof, sizeof(double) * nk * h->cfg.nsv));
  HIP_OK(hipMemcpy(h->xo_sv_prof, svprof, sizeof(double) * nk * h->cfg.nsv, hipMemcpyHostToDevice));
  for (int n = 0; n < h->cfg.nsv && n < 13; ++n)
    if (!h->xo_sv_cols[n]) {
      HIP_OK(hipMalloc(&h->xo_sv_cols[n], sizeof(double) * 8 * np));
      HIP_OK(hipMemset(h->xo_sv_cols[n], 0, sizeof(double) * 8 * np));
    }
  return 0;
}

extern "C" int udc_set_open_x_sample_gap(udc_handle *h, int on) {
  if (!h) { udc_set_error("udc_set_open_x_sample_gap: null handle"); return 1; }
  if (!h->xg) { udc_set_error("udc_set_open_x_sample_gap: not a handle of udc_create_open_x"); return 1; }
  h->xo_sample_gap = on != 0;      // (no flush: the stage-3 substep it is meant for may be among the recorded ones)
  return 0;
}

// the temperature's inflow profile thlprof [ktot+2] by the reference's k (entry ktot+1 as the reference's thlprof(ke+1): zero);
// after udc_set_tempeq
extern "C" int udc_set_open_x_thl(udc_handle *h, const double *thlprof) {
  if (!h || !thlprof) { udc_set_error("udc_set_open_x_thl: null argument"); return 1; }
  HIP_OK(hipSetDevice(h->device));
  if (udc_flush_pending(h)) return 1;
  if (!h->xg) { udc_set_error("udc_set_open_x_thl: not a handle of udc_create_open_x"); return 1; }
  if ((int)h->fields.size() <= UDC_THL0 || !h->fields[UDC_THL0]) { udc_set_error("udc_set_open_x_thl: call udc_set_tempeq first"); return 1; }
  const Geo &g = h->g;
  const size_t nk = (size_t)g.nz + 2;
  HIP_OK(hipStreamSynchronize(h->stream));
  if (!h->xo_thl_prof) HIP_OK(hipMalloc(&h->xo_thl_prof, sizeof(double) * nk));
  if (xo_alloc_east_west(h, UDC_THL0, UDC_THLM, &h->xo_thl_east, &h->xo_thl_west)) return 1;
  HIP_OK(hipMemcpy(h->xo_thl_prof, thlprof, sizeof(double) * nk, hipMemcpyHostToDevice));
  return 0;
}

// `halos` on such a handle: xT_periodic / xq_periodic for a temperature / total water that stays periodic in x (src/modboundary.f90:95-100),
// ahead of the y refresh like there
int k_xo_halos(udc_handle *h) {
  if (!h->xg) return 0;
  const Geo &g = h->g;
  if (!(h->xo_thl_prof || h->xo_sc_in_now[0]) && (int)h->fields.size() > UDC_THL0 && h->fields[UDC_THL0])
    hipLaunchKernelGGL(xo_wrap_kernel, plane_grid(g), dim3(64), 0, h->stream, g, h->fields[UDC_THL0], h->fields[UDC_THLM]);
  if (!(h->xo_qt_prof || h->xo_sc_in_now[1]) && h->lmoist && (int)h->fields.size() > UDC_QT0 && h->fields[UDC_QT0])
    hipLaunchKernelGGL(xo_wrap_kernel, plane_grid(g), dim3(64), 0, h->stream, g, h->fields[UDC_QT0], h->fields[UDC_QTM]);
  HIP_OK(hipGetLastError());
  return 0;
}

// boundary_follows: the fused substep -- k_xo_boundary(h, stage 3 or not) comes next on the stream and does the velocities' part itself
int k_xo_after_integrate(udc_handle *h, int rk3step, bool boundary_follows) {
  if (!h->xg) return 0;
  const Geo &g = h->g;
  PROF(h, "xo_ghosts");
  if (!boundary_follows)
  hipLaunchKernelGGL(xo_restore_kernel, plane_grid(g), dim3(64), 0, h->stream, g, rk3step == 3 ? 1 : 0, h->fields[UDC_U0], h->fields[UDC_V0],
                     h->fields[UDC_W0], h->fields[UDC_UM], h->fields[UDC_VM], h->fields[UDC_WM], h->xo_east, h->xo_west);
  if (h->xo_thl_prof || h->xo_sc_in_now[0])
    hipLaunchKernelGGL(xo_thl_restore_kernel, plane_grid(g), dim3(64), 0, h->stream, g, rk3step == 3 ? 1 : 0, h->fields[UDC_THL0],
                       h->fields[UDC_THLM], h->xo_thl_east, h->xo_thl_west);
  else if ((int)h->fields.size() > UDC_THL0 && h->fields[UDC_THL0])      // BCxT = 1: periodic (halos' xT_periodic)
    hipLaunchKernelGGL(xo_wrap_kernel, plane_grid(g), dim3(64), 0, h->stream, g, h->fields[UDC_THL0], h->fields[UDC_THLM]);
  if (h->xo_qt_prof || h->xo_sc_in_now[1])
    hipLaunchKernelGGL(xo_thl_restore_kernel, plane_grid(g), dim3(64), 0, h->stream, g, rk3step == 3 ? 1 : 0, h->fields[UDC_QT0],
                       h->fields[UDC_QTM], h->xo_qt_east, h->xo_qt_west);
  else if (h->lmoist && (int)h->fields.size() > UDC_QT0 && h->fields[UDC_QT0])      // BCxq = 1 (xq_periodic)
    hipLaunchKernelGGL(xo_wrap_kernel, plane_grid(g), dim3(64), 0, h->stream, g, h->fields[UDC_QT0], h->fields[UDC_QTM]);
  for (int n = 0; n < h->cfg.nsv && n < 13; ++n)
    if (h->xo_sv_cols[n])
      hipLaunchKernelGGL(xo_sv_restore_kernel, plane_grid(g), dim3(64), 0, h->stream, g, rk3step == 3 ? 1 : 0, h->fields[UDC_SV0 + 3 * n],
                         h->fields[UDC_SVM + 3 * n], h->xo_sv_cols[n]);
  HIP_OK(hipGetLastError());
  return 0;
}

// fix = false: before the temperature's diffusion (after closurebc); fix = true: after a sweep that advected over the reassured row too
int k_xo_thl_top(udc_handle *h, bool fix) {
  if (!h->xg || !h->xo_thl_prof || h->xo_sc_in_now[0] || h->slot[15].top != 0) return 0;      // (top = 1, 2: k_scalar_top_flux / a value)
  if ((int)h->fields.size() <= UDC_THL0 || !h->fields[UDC_THL0]) return 0;
  const Geo &g = h->g;
  if (!h->xo_thl_top) HIP_OK(hipMalloc(&h->xo_thl_top, sizeof(double) * g.py));
  const dim3 gr((unsigned)((g.py + 63) / 64)), b(64);
  if (!fix)
    hipLaunchKernelGGL(xo_thl_top_kernel<false>, gr, b, 0, h->stream, g, h->m, h->fields[UDC_THL0], h->fields[UDC_THLM], h->xo_thl_top,
                       (const double *)nullptr, (double *)nullptr);
  else
    hipLaunchKernelGGL(xo_thl_top_kernel<true>, gr, b, 0, h->stream, g, h->m, h->fields[UDC_THL0], h->fields[UDC_THLM], h->xo_thl_top,
                       (const double *)h->fields[UDC_W0], h->fields[UDC_THLP]);
  HIP_OK(hipGetLastError());
  return 0;
}

int k_xo_poisson(udc_handle *h) {
  udc_handle *hp = h->xpois;
  if (!hp) { udc_set_error("open x boundaries: the solver's handle is missing"); return 1; }
  const Geo &g = h->g, &g2 = hp->g;
  const int n = g.nx - 2 * g.xg;
  const dim3 b(256), gr((unsigned)((n + 255) / 256), (unsigned)g.ny, (unsigned)g.nz);
  hp->bczp = h->bczp;
  const bool div_in = h->div_in_fft;      // (the fused substep skipped k_divergence_rhs: the doubled row's x transform evaluates it, udc_fft.hip)
  if (div_in) { hp->div_in_fft = true; hp->xo_src = h; }
  else
  if (!h->xo_rhs_mirrored) {      // (a right-hand side that did not come from k_divergence_rhs)
    PROF(h, "xo_mirror");
    hipLaunchKernelGGL(xo_gather_kernel, gr, b, 0, h->stream, g, g2, (const double *)h->fields[UDC_P], hp->fields[UDC_P]);
    HIP_OK(hipGetLastError());
  }
  h->xo_rhs_mirrored = false;
  {
    PROF(h, "poisson_2x");      // (the doubled row's transforms and tridiagonal solves, on this handle's stream)
    const int rc = k_poisson_solve(hp);
    hp->div_in_fft = false; hp->xo_src = nullptr;
    if (rc) return 1;
  }
  PROF(h, "xo_mirror");
  hipLaunchKernelGGL(xo_scatter_kernel, gr, b, 0, h->stream, g, g2, (const double *)hp->fields[UDC_P], h->fields[UDC_P]);
  HIP_OK(hipGetLastError());
  return 0;
}
