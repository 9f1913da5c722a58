// Arithmetic of the transported scalars, shared by the direct-load kernel (udc_scalar.hip) and the LDS-staged marching
// kernel (udc_scalar_lds.hip): kappa-scheme (advecc_kappa + rlim, src/modadvection.f90:316-421) or 2nd-order central
// advection (advecc_2nd, :103-155) and eddy diffusion (diffc, src/modsubgrid.f90:540-623; diffe :627-669 with dfac = 1).
// `A` supplies c(di, dj, dk) for |d| <= 2 along one axis at a time and e(di, dj, dk) (the diffusivity) for |d| <= 1.
#pragma once
#include "udc_internal.h"

// src/modadvection.f90:410-421, eps1 = 1e-10 (src/modglobal.f90:318)
__device__ __forceinline__ double rlim(double d1, double d2) {
  const double eps1 = 1.e-10;
  const double ri = (d2 + eps1) / (d1 + eps1);
  const double phir = fmax(0., fmin(2. * ri, fmin(1. / 3. + 2. / 3. * ri, 2.)));
  return 0.5 * phir * d1;
}

// face value on the low side of cell "0" given velocity vel there:
// cm2,cm1,c0,cp1 = c at -2,-1,0,+1 ; h* = inverse half-level spacings at -1, 0, +1 ; df = cell size factor
__device__ __forceinline__ double face(double vel, double cm2, double cm1, double c0, double cp1,
                                       double hm1, double h0, double hp1, double df) {
  double d1, d2, cf;
  if (vel > 0) { d1 = (cm1 - cm2) * hm1; d2 = (c0 - cm1) * h0; cf = cm1; }
  else { d1 = (c0 - cp1) * hp1; d2 = (cm1 - c0) * h0; cf = c0; }
  return cf + df * rlim(d1, d2);
}

// Level entries of the metric tables (kf = k + 1), read from the tables (ScalMetGlobal) or from a block the marching
// kernels stage in LDS a level ahead (ScalMetLds; see udc_mom_arith.h for why).  Entry t:
//   0..2 dzf(kf-1), dzf(kf), dzf(kf+1);  3..6 dzhi(max(kf-1,1)), dzhi(kf), dzhi(kf+1), dzhi(min(kf+2,nz+1));
//   7 dzfi(kf);  8 dzfi5(kf);  9, 10 dzh2i(kf), dzh2i(kf+1)
constexpr int NSCALMET = 11;
__device__ __forceinline__ double scalmet_load(const Metrics &m, int t, int kf, int nz) {
  switch (t) {
    case 0: return m.dzf[kf - 1];
    case 1: return m.dzf[kf];
    case 2: return m.dzf[kf + 1];
    case 3: return m.dzhi[kf - 1 < 1 ? 1 : kf - 1];
    case 4: return m.dzhi[kf];
    case 5: return m.dzhi[kf + 1];
    case 6: return m.dzhi[kf + 2 > nz + 1 ? nz + 1 : kf + 2];
    case 7: return m.dzfi[kf];
    case 8: return m.dzfi5[kf];
    case 9: return m.dzh2i[kf];
    default: return m.dzh2i[kf + 1];
  }
}
struct ScalMetGlobal {
  const Metrics &m;
  int kf, nz;
  __device__ __forceinline__ double get(int t) const { return scalmet_load(m, t, kf, nz); }
};
struct ScalMetLds {
  const double *p;
  __device__ __forceinline__ double get(int t) const {      // uniform: scalar registers
    union { double d; int i[2]; } u;
    u.d = p[t];
    u.i[0] = __builtin_amdgcn_readfirstlane(u.i[0]);
    u.i[1] = __builtin_amdgcn_readfirstlane(u.i[1]);
    return u.d;
  }
};

// ADV: 0 = none, 1 = kappa, 2 = cd2.  t = the tendency so far (0 when it is known to be zero).  ul/uh, vl/vh, wl/wh: the
// velocities on the low / high faces of the cell; k = device level, nz = levels; gh = kappa_ghosts rule of thl0c.
template <int ADV, bool DIFF, bool LES, class Acc, class LM>
__device__ __forceinline__ double scalar_tend(const Acc &A, const Metrics &m, const LM &lm, int k, int nz, double t, double ul, double uh,
                                              double vl, double vh, double wl, double wh, double cekh, double dfac, int gh) {
  const double c0 = A.c(0, 0, 0);
  const double cxm1 = A.c(-1, 0, 0), cxp1 = A.c(1, 0, 0), cym1 = A.c(0, -1, 0), cyp1 = A.c(0, 1, 0), czm1 = A.c(0, 0, -1), czp1 = A.c(0, 0, 1);
  if (ADV == 2) {
    // advecc_2nd, src/modadvection.f90:127-133 and :148-151 (two statements, same order)
    const double kdzf = lm.get(1), kdzfm = lm.get(0), kdzfp = lm.get(2);
    t = t - ((uh * (cxp1 + c0) - ul * (cxm1 + c0)) * m.dxi5
           + (vh * (cyp1 + c0) - vl * (cym1 + c0)) * m.dyi5);
    t = t - (wh * (czp1 * kdzf + c0 * kdzfp) * lm.get(5)
           - wl * (czm1 * kdzf + c0 * kdzfm) * lm.get(4)) * lm.get(8);
  }
  if (ADV == 1) {
    const double cxm2 = A.c(-2, 0, 0), cxp2 = A.c(2, 0, 0);
    const double cym2 = A.c(0, -2, 0), cyp2 = A.c(0, 2, 0);
    double kzm2 = A.c(0, 0, -2), kzp2 = A.c(0, 0, 2);
    double kzm1 = czm1, kzp1 = czp1;      // (the diffusion operands below stay the field's own ghosts)
    if (gh) {
      // kappa on thl (iadv_thl = 7) runs on the reference's separate copy thl0c, whose vertical ghost planes are not
      // thl0's: nothing ever writes the two below the floor (zero), and at the top `boundary` copies level ke upwards
      // for a flux condition (src/modboundary.f90:211-213) and leaves them untouched (zero) for a value condition
      const double top = gh == 1 ? 1. : 0.;
      if (k == 0) kzm1 = 0.;
      if (k <= 1) kzm2 = 0.;
      if (k == nz - 1) { kzp1 = top * c0; kzp2 = top * c0; }
      if (k == nz - 2) kzp2 = top * kzp1;
    }
    const double dxi = m.dxi, dx = m.dx, dyi = m.dyi;
    {  // x: faces i (low) and i+1 (high); dxhci = dxi, dxfc = dx, dxfci = dxi on the uniform grid
      const double fl = face(ul, cxm2, cxm1, c0, cxp1, dxi, dxi, dxi, dx);
      const double fh = face(uh, cxm1, c0, cxp1, cxp2, dxi, dxi, dxi, dx);
      t = (t + (-fh * uh * dxi)) + fl * ul * dxi;
    }
    {  // y (no stretching: d's are plain differences, df = 1)
      const double fl = face(vl, cym2, cym1, c0, cyp1, 1., 1., 1., 1.);
      const double fh = face(vh, cym1, c0, cyp1, cyp2, 1., 1., 1., 1.);
      t = (t + (-fh * vh * dyi)) + fl * vl * dyi;
    }
    {  // z: faces kb+1..ke+1 only (no flux through the floor, src/modadvection.f90:385)
      const double hkm1 = lm.get(3), hk = lm.get(4), hkp1 = lm.get(5);
      const double hkp2 = lm.get(6);
      const double dzfci = lm.get(7);
      const double fh = face(wh, kzm1, c0, kzp1, kzp2, hk, hkp1, hkp2, lm.get(2));
      const double upper = -fh * wh * dzfci;
      double lower = 0.;
      if (k >= 1) {
        const double fl = face(wl, kzm2, kzm1, c0, kzp1, hkm1, hk, hkp1, lm.get(1));
        lower = fl * wl * dzfci;
      }
      t = (t + upper) + lower;
    }
  }
  if (DIFF) {
    const double dzf_k = lm.get(1), dzf_km = lm.get(0), dzf_kp = lm.get(2);
    if (LES) {
      const double e0 = A.e(0, 0, 0), exm = A.e(-1, 0, 0), exp_ = A.e(1, 0, 0), eym = A.e(0, -1, 0), eyp = A.e(0, 1, 0),
                   ezm = A.e(0, 0, -1), ezp = A.e(0, 0, 1);
      // dfac = 0.5 with ekh (diffc, src/modsubgrid.f90:569-584) or 1.0 with ekm (diffe, :649-663)
      t = t + dfac * (((exp_ + e0) * (cxp1 - c0) - (e0 + exm) * (c0 - cxm1)) * m.dx2i
                   + ((eyp + e0) * (cyp1 - c0) - (e0 + eym) * (c0 - cym1)) * m.dy2i
                   + ((dzf_kp * e0 + dzf_k * ezp) * (czp1 - c0) * lm.get(10)
                    - (dzf_km * e0 + dzf_k * ezm) * (c0 - czm1) * lm.get(9)) * lm.get(7));
    } else {
      t = t + ((cekh * (cxp1 - c0) - cekh * (c0 - cxm1)) * m.dx2i
             + (cekh * (cyp1 - c0) - cekh * (c0 - cym1)) * m.dy2i
             + (cekh * (czp1 - c0) * lm.get(5) - cekh * (c0 - czm1) * lm.get(4)) * lm.get(7));
    }
  }
  return t;
}
