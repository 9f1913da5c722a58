// Arithmetic of the eddy-viscosity closure, shared by the direct-load kernel (udc_mom.hip) and the
// LDS-staged marching kernel (udc_mom_lds.hip).  SGS 1 = Smagorinsky (src/modsubgrid.f90:208-264),
// 2 = Vreman (:269-360).  `A` supplies u0/v0/w0 at (i+di, j+dj, k+dk).  Molecular part added in the
// reference's statement order (ekh from ekm first, then + nu).
#pragma once
#include "udc_internal.h"

// The level entries of the metric tables the closure reads (kf = k + 1), as for the momentum sweep (udc_mom_arith.h):
// from the tables themselves (ClosMetGlobal) or from a block the marching kernel stages in LDS a level ahead (ClosMetLds).
constexpr int NCLOSMET = 9;      // dzhi(kf), dzhi(kf+1), dzfi(kf), mlen(kf), dzf(kf-1), dzf(kf), dzf(kf+1), dzfiq(kf), dzf2(kf)
__device__ __forceinline__ const double *closmet_src(const Metrics &m, int t) {
  switch (t) {
    case 0: return m.dzhi;
    case 1: return m.dzhi + 1;
    case 2: return m.dzfi;
    case 3: return m.mlen;
    case 4: return m.dzf - 1;
    case 5: return m.dzf;
    case 6: return m.dzf + 1;
    case 7: return m.dzfiq;
    default: return m.dzf2;
  }
}
struct ClosMetGlobal {
  const Metrics &m;
  int kf;
  __device__ __forceinline__ double get(int t) const { return closmet_src(m, t)[kf]; }
};
struct ClosMetLane {       // entry t sits in lane t of a vector register of every wave (loaded a level ahead): no LDS at all
  double v;
  __device__ __forceinline__ double get(int t) const {
    union { double d; int i[2]; } u;
    u.d = v;
    u.i[0] = __builtin_amdgcn_readlane(u.i[0], t);
    u.i[1] = __builtin_amdgcn_readlane(u.i[1], t);
    return u.d;
  }
};
struct ClosMetLds {
  const double *p;
  __device__ __forceinline__ double get(int t) const {      // uniform: scalar registers
    union { double d; int i[2]; } u;
    u.d = p[t];
    u.i[0] = __builtin_amdgcn_readfirstlane(u.i[0]);
    u.i[1] = __builtin_amdgcn_readfirstlane(u.i[1]);
    return u.d;
  }
};

template <int SGS, class Acc, class LM>
__device__ __forceinline__ void closure_arith(const Acc &A, const Metrics &m, const LM &lm, const Params &pr, int k,
                                              double &em, double &eh) {
  const double dxi = m.dxi, dyi = m.dyi;
  const double dzhi_k = lm.get(0), dzhi_kp = lm.get(1), dzfi_k = lm.get(2);
  if (SGS == 1) {
    double t, strain2;
    t = (A.u(1, 0, 0) - A.u(0, 0, 0)) * dxi; strain2 = t * t;
    t = (A.v(0, 1, 0) - A.v(0, 0, 0)) * dyi; strain2 = strain2 + t * t;
    t = (A.w(0, 0, 1) - A.w(0, 0, 0)) * dzfi_k; strain2 = strain2 + t * t;
    double a1 = (A.w(0, 0, 1) - A.w(-1, 0, 1)) * dxi + (A.u(0, 0, 1) - A.u(0, 0, 0)) * dzhi_kp;
    double a2 = (A.w(0, 0, 0) - A.w(-1, 0, 0)) * dxi + (A.u(0, 0, 0) - A.u(0, 0, -1)) * dzhi_k;
    double a3 = (A.w(1, 0, 0) - A.w(0, 0, 0)) * dxi + (A.u(1, 0, 0) - A.u(1, 0, -1)) * dzhi_k;
    double a4 = (A.w(1, 0, 1) - A.w(0, 0, 1)) * dxi + (A.u(1, 0, 1) - A.u(1, 0, 0)) * dzhi_kp;
    strain2 = strain2 + 0.125 * (a1 * a1 + a2 * a2 + a3 * a3 + a4 * a4);
    a1 = (A.u(0, 1, 0) - A.u(0, 0, 0)) * dyi + (A.v(0, 1, 0) - A.v(-1, 1, 0)) * dxi;
    a2 = (A.u(0, 0, 0) - A.u(0, -1, 0)) * dyi + (A.v(0, 0, 0) - A.v(-1, 0, 0)) * dxi;
    a3 = (A.u(1, 0, 0) - A.u(1, -1, 0)) * dyi + (A.v(1, 0, 0) - A.v(0, 0, 0)) * dxi;
    a4 = (A.u(1, 1, 0) - A.u(1, 0, 0)) * dyi + (A.v(1, 1, 0) - A.v(0, 1, 0)) * dxi;
    strain2 = strain2 + 0.125 * (a1 * a1 + a2 * a2 + a3 * a3 + a4 * a4);
    a1 = (A.v(0, 0, 1) - A.v(0, 0, 0)) * dzhi_kp + (A.w(0, 0, 1) - A.w(0, -1, 1)) * dyi;
    a2 = (A.v(0, 0, 0) - A.v(0, 0, -1)) * dzhi_k + (A.w(0, 0, 0) - A.w(0, -1, 0)) * dyi;
    a3 = (A.v(0, 1, 0) - A.v(0, 1, -1)) * dzhi_k + (A.w(0, 1, 0) - A.w(0, 0, 0)) * dyi;
    a4 = (A.v(0, 1, 1) - A.v(0, 1, 0)) * dzhi_kp + (A.w(0, 1, 1) - A.w(0, 0, 1)) * dyi;
    strain2 = strain2 + 0.125 * (a1 * a1 + a2 * a2 + a3 * a3 + a4 * a4);
    const double ml = lm.get(3);
    em = (ml * ml) * sqrt(2. * strain2);
    if (pr.bare) { eh = 0.; return; }
    eh = em * pr.prandtli;
    em = em + pr.numol;
    eh = eh + pr.numol * pr.prandtlmoli;
  } else {
    const double dzf_k = lm.get(5), dzf_km = lm.get(4), dzf_kp = lm.get(6);
    const double a11 = (A.u(1, 0, 0) - A.u(0, 0, 0)) * dxi;
    const double a12 = (A.v(1, 1, 0) + A.v(1, 0, 0) - A.v(-1, 1, 0) - A.v(-1, 0, 0)) * m.dxiq;
    const double a13 = (A.w(1, 0, 1) + A.w(1, 0, 0) - A.w(-1, 0, 1) - A.w(-1, 0, 0)) * m.dxiq;
    const double a21 = (A.u(1, 1, 0) + A.u(0, 1, 0) - A.u(1, -1, 0) - A.u(0, -1, 0)) * m.dyiq;
    const double a22 = (A.v(0, 1, 0) - A.v(0, 0, 0)) * dyi;
    const double a23 = (A.w(0, 1, 1) + A.w(0, 1, 0) - A.w(0, -1, 1) - A.w(0, -1, 0)) * m.dyiq;
    const double a31 = (((A.u(1, 0, 1) + A.u(0, 0, 1)) * dzf_k + (A.u(1, 0, 0) + A.u(0, 0, 0)) * dzf_kp) * dzhi_kp
                      - ((A.u(1, 0, 0) + A.u(0, 0, 0)) * dzf_km + (A.u(1, 0, -1) + A.u(0, 0, -1)) * dzf_k) * dzhi_k) * lm.get(7);
    const double a32 = (((A.v(0, 1, 1) + A.v(0, 0, 1)) * dzf_k + (A.v(0, 1, 0) + A.v(0, 0, 0)) * dzf_kp) * dzhi_kp
                      - ((A.v(0, 1, 0) + A.v(0, 0, 0)) * dzf_km + (A.v(0, 1, -1) + A.v(0, 0, -1)) * dzf_k) * dzhi_k) * lm.get(7);
    const double a33 = (A.w(0, 0, 1) - A.w(0, 0, 0)) * dzfi_k;
    const double aa = a11 * a11 + a21 * a21 + a31 * a31 + a12 * a12 + a22 * a22 + a32 * a32
                    + a13 * a13 + a23 * a23 + a33 * a33;
    const double dx2 = m.dx2, dy2 = m.dy2, dz2 = lm.get(8);
    const double b11 = dx2 * a11 * a11 + dy2 * a21 * a21 + dz2 * a31 * a31;
    const double b22 = dx2 * a12 * a12 + dy2 * a22 * a22 + dz2 * a32 * a32;
    const double b12 = dx2 * a11 * a12 + dy2 * a21 * a22 + dz2 * a31 * a32;
    const double b33 = dx2 * a13 * a13 + dy2 * a23 * a23 + dz2 * a33 * a33;
    const double b13 = dx2 * a11 * a13 + dy2 * a21 * a23 + dz2 * a31 * a33;
    const double b23 = dx2 * a12 * a13 + dy2 * a22 * a23 + dz2 * a32 * a33;
    const double bb = b11 * b22 - b12 * b12 + b11 * b33 - b13 * b13 + b22 * b33 - b23 * b23;
    em = (bb < 1.e-8) ? 0. : pr.c_vreman * sqrt(bb / aa);
    if (pr.bare) { eh = 0.; return; }      // the buoyancy correction comes first (src/modsubgrid.f90:330-357)
    eh = em * pr.prandtli;
    em = em + pr.numol;
    eh = eh + pr.numol * pr.prandtlmoli;
  }
}
