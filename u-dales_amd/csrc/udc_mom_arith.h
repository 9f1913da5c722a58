// Arithmetic of the fused momentum sweep, shared by the direct-load kernel (udc_mom.hip) and the
// LDS-staged k-marching kernel (udc_mom_lds.hip).  Expression order follows the reference:
// advecu/v/w_2nd (src/modadvection.f90:158-314), diffu/v/w (src/modsubgrid.f90:672-997),
// forces (src/modforces.f90:84-127); accumulation order: xy advection, z advection, diffusion, forcing.
#pragma once
#include "udc_internal.h"

struct MomVals {
  double u_c, u_xm, u_xp, u_ym, u_yp, u_zm, u_zp, u_xp_ym, u_xp_zm;
  double v_c, v_xm, v_xp, v_ym, v_yp, v_zm, v_zp, v_xm_yp, v_yp_zm;
  double w_c, w_xm, w_xp, w_ym, w_yp, w_zm, w_zp, w_xm_zp, w_ym_zp;
  double p_c, p_xm, p_ym, p_zm;
  double e_c, e_xm, e_xp, e_ym, e_yp, e_zm, e_zp, e_xm_yp, e_xm_ym, e_xm_zm, e_xm_zp, e_ym_zm, e_ym_zp,
         e_xp_ym, e_yp_zm, e_xp_zm;
};

// The vertical metrics and level forcings one level of the sweep reads (kf = k + 1 in the tables of Metrics).
// The direct-load kernel fills it from global memory; the marching kernel from a small LDS block staged a level ahead,
// so that no vector-memory load sits between the next plane's prefetch and the stencil (vmcnt retires in order: a wait
// for a younger L2-resident table entry is a wait for every HBM load issued before it).
constexpr int NLEVMET = 12;
struct LevelMet {           // read from the tables
  double dzf_km, dzf_k, dzf_kp, dzhi_k, dzhi_kp, dzfi_km, dzfi_k, dzfi5_k, dzhiq_k, dzhiq_kp, dpdx, dpdy;
  __device__ __forceinline__ double get(int t) const {
    const double a[NLEVMET] = {dzf_km, dzf_k, dzf_kp, dzhi_k, dzhi_kp, dzfi_km, dzfi_k, dzfi5_k, dzhiq_k, dzhiq_kp, dpdx, dpdy};
    return a[t];
  }
};
struct LevelMetLds {        // read where it is used from the staged block (entries in the order of LevelMet)
  const double *p;
  // the value is the same in every lane: keep it in scalar registers (the stencil is short of vector registers)
  __device__ __forceinline__ double get(int t) const {
    union { double d; int i[2]; } u;
    u.d = p[t];
    u.i[0] = __builtin_amdgcn_readfirstlane(u.i[0]);
    u.i[1] = __builtin_amdgcn_readfirstlane(u.i[1]);
    return u.d;
  }
};
// entry t of the block: table and offset from kf
__device__ __forceinline__ const double *levmet_src(const Metrics &m, int t) {
  switch (t) {
    case 0: return m.dzf - 1;
    case 1: return m.dzf;
    case 2: return m.dzf + 1;
    case 3: return m.dzhi;
    case 4: return m.dzhi + 1;
    case 5: return m.dzfi - 1;
    case 6: return m.dzfi;
    case 7: return m.dzfi5;
    case 8: return m.dzhiq;
    case 9: return m.dzhiq + 1;
    case 10: return m.dpdxl;
    default: return m.dpdyl;
  }
}
__device__ __forceinline__ LevelMet levmet_global(const Metrics &m, int kf) {
  return LevelMet{m.dzf[kf - 1], m.dzf[kf], m.dzf[kf + 1], m.dzhi[kf], m.dzhi[kf + 1], m.dzfi[kf - 1], m.dzfi[kf],
                  m.dzfi5[kf], m.dzhiq[kf], m.dzhiq[kf + 1], m.dpdxl[kf], m.dpdyl[kf]};
}

// PGRAD = false: the gradient of pres0 is left out (pressure-total form of the fused substep: the solve then returns pres0 + p and
// the projection applies all of it, udc_plan.h ptotal)
template <bool ADV, bool DIFF, bool LES, bool FORCES, bool PGRAD = true, class LM>
__device__ __forceinline__ void mom_arith(const MomVals &q, const Metrics &m, const LM &lm, int k, double numol,
                                          double &tu, double &tv, double &tw) {
  const double u_c = q.u_c, u_xm = q.u_xm, u_xp = q.u_xp, u_ym = q.u_ym, u_yp = q.u_yp, u_zm = q.u_zm,
               u_zp = q.u_zp, u_xp_ym = q.u_xp_ym, u_xp_zm = q.u_xp_zm;
  const double v_c = q.v_c, v_xm = q.v_xm, v_xp = q.v_xp, v_ym = q.v_ym, v_yp = q.v_yp, v_zm = q.v_zm,
               v_zp = q.v_zp, v_xm_yp = q.v_xm_yp, v_yp_zm = q.v_yp_zm;
  const double w_c = q.w_c, w_xm = q.w_xm, w_xp = q.w_xp, w_ym = q.w_ym, w_yp = q.w_yp, w_zm = q.w_zm,
               w_zp = q.w_zp, w_xm_zp = q.w_xm_zp, w_ym_zp = q.w_ym_zp;
  const double dzf_k = lm.get(1), dzf_km = lm.get(0), dzf_kp = lm.get(2);
  const double dzhi_k = lm.get(3), dzhi_kp = lm.get(4);
  const double dzfi_k = lm.get(6);

  if (ADV) {
    const double p_c = PGRAD ? q.p_c : 0., p_xm = PGRAD ? q.p_xm : 0., p_ym = PGRAD ? q.p_ym : 0., p_zm = PGRAD ? q.p_zm : 0.;
    const double dzfi5_k = lm.get(7);
    // advecu_2nd, src/modadvection.f90:178-187 and :202-207
    tu = tu - (((u_c + u_xp) * (u_c + u_xp) - (u_c + u_xm) * (u_c + u_xm)) * m.dxiq
             + ((u_c + u_yp) * (v_yp + v_xm_yp) - (u_c + u_ym) * (v_c + v_xm)) * m.dyiq);
    if (PGRAD) tu = tu - ((p_c - p_xm) * m.dxi);
    tu = tu - ((u_zp * dzf_k + u_c * dzf_kp) * dzhi_kp * (w_zp + w_xm_zp)
             - (u_c * dzf_km + u_zm * dzf_k) * dzhi_k * (w_c + w_xm)) * 0.5 * dzfi5_k;
    // advecv_2nd, :235-245 and :260-265
    tv = tv - (((u_xp + u_xp_ym) * (v_c + v_xp) - (u_c + u_ym) * (v_c + v_xm)) * m.dxiq
             + ((v_yp + v_c) * (v_c + v_yp) - (v_ym + v_c) * (v_c + v_ym)) * m.dyiq);
    if (PGRAD) tv = tv - ((p_c - p_ym) * m.dyi);
    tv = tv - ((w_zp + w_ym_zp) * (v_zp * dzf_k + v_c * dzf_kp) * dzhi_kp
             - (w_c + w_ym) * (v_zm * dzf_k + v_c * dzf_km) * dzhi_k) * 0.5 * dzfi5_k;
    // advecw_2nd, :295-309 (k = kb+1..ke)
    if (k >= 1) {
      const double dzhiq_k = lm.get(8);
      tw = tw - (((w_xp + w_c) * (dzf_km * u_xp + dzf_k * u_xp_zm)
                - (w_c + w_xm) * (dzf_km * u_c + dzf_k * u_zm)) * m.dxiq * dzhi_k
               + ((w_yp + w_c) * (dzf_km * v_yp + dzf_k * v_yp_zm)
                - (w_c + w_ym) * (dzf_km * v_c + dzf_k * v_zm)) * m.dyiq * dzhi_k
               + ((w_c + w_zp) * (w_c + w_zp) - (w_c + w_zm) * (w_c + w_zm)) * dzhiq_k);
      if (PGRAD) tw = tw - ((p_c - p_zm) * dzhi_k);
    }
  }

  if (DIFF) {
    if (LES) {
      const double e_c = q.e_c, e_xm = q.e_xm, e_xp = q.e_xp, e_ym = q.e_ym, e_yp = q.e_yp,
                   e_zm = q.e_zm, e_zp = q.e_zp;
      const double e_xm_yp = q.e_xm_yp, e_xm_ym = q.e_xm_ym, e_xm_zm = q.e_xm_zm, e_xm_zp = q.e_xm_zp;
      const double e_ym_zm = q.e_ym_zm, e_ym_zp = q.e_ym_zp, e_xp_ym = q.e_xp_ym;
      const double e_yp_zm = q.e_yp_zm, e_xp_zm = q.e_xp_zm;
      const double dzhiq_k = lm.get(8), dzhiq_kp = lm.get(9);
      {  // diffu, src/modsubgrid.f90:695-729
        const double emom = (dzf_km * (e_c + e_xm) + dzf_k * (e_zm + e_xm_zm)) * dzhiq_k;
        const double emop = (dzf_kp * (e_c + e_xm) + dzf_k * (e_zp + e_xm_zp)) * dzhiq_kp;
        const double empo = 0.25 * ((e_c + e_yp) + (e_xm + e_xm_yp));
        const double emmo = 0.25 * ((e_c + e_ym) + (e_xm_ym + e_xm));
        tu = tu + (e_c * (u_xp - u_c) - e_xm * (u_c - u_xm)) * 2. * m.dx2i
                + (empo * ((u_yp - u_c) * m.dyi + (v_yp - v_xm_yp) * m.dxi)
                 - emmo * ((u_c - u_ym) * m.dyi + (v_c - v_xm) * m.dxi)) * m.dyi
                + (emop * ((u_zp - u_c) * dzhi_kp + (w_zp - w_xm_zp) * m.dxi)
                 - emom * ((u_c - u_zm) * dzhi_k + (w_c - w_xm) * m.dxi)) * dzfi_k;
      }
      {  // diffv, :802-838
        const double eomm = (dzf_km * (e_c + e_ym) + dzf_k * (e_zm + e_ym_zm)) * dzhiq_k;
        const double eomp = (dzf_kp * (e_c + e_ym) + dzf_k * (e_zp + e_ym_zp)) * dzhiq_kp;
        const double emmo = 0.25 * (e_c + e_ym + e_xm_ym + e_xm);
        const double epmo = 0.25 * (e_c + e_ym + e_xp_ym + e_xp);
        tv = tv + (epmo * ((v_xp - v_c) * m.dxi + (u_xp - u_xp_ym) * m.dyi)
                 - emmo * ((v_c - v_xm) * m.dxi + (u_c - u_ym) * m.dyi)) * m.dxi
                + (e_c * (v_yp - v_c) - e_ym * (v_c - v_ym)) * 2. * m.dy2i
                + (eomp * ((v_zp - v_c) * dzhi_kp + (w_zp - w_ym_zp) * m.dyi)
                 - eomm * ((v_c - v_zm) * dzhi_k + (w_c - w_ym) * m.dyi)) * dzfi_k;
      }
      if (k >= 1) {  // diffw, :913-951
        const double dzfi_km = lm.get(5);
        const double emom = (dzf_km * (e_c + e_xm) + dzf_k * (e_zm + e_xm_zm)) * dzhiq_k;
        const double eomm = (dzf_km * (e_c + e_ym) + dzf_k * (e_zm + e_ym_zm)) * dzhiq_k;
        const double eopm = (dzf_km * (e_c + e_yp) + dzf_k * (e_zm + e_yp_zm)) * dzhiq_k;
        const double epom = (dzf_km * (e_c + e_xp) + dzf_k * (e_zm + e_xp_zm)) * dzhiq_k;
        tw = tw + (epom * ((w_xp - w_c) * m.dxi + (u_xp - u_xp_zm) * dzhi_k)
                 - emom * ((w_c - w_xm) * m.dxi + (u_c - u_zm) * dzhi_k)) * m.dxi
                + (eopm * ((w_yp - w_c) * m.dyi + (v_yp - v_yp_zm) * dzhi_k)
                 - eomm * ((w_c - w_ym) * m.dyi + (v_c - v_zm) * dzhi_k)) * m.dyi
                + (e_c * (w_zp - w_c) * dzfi_k - e_zm * (w_c - w_zm) * dzfi_km) * 2. * dzhi_k;
      }
    } else {
      const double nu = numol;
      // DNS forms, src/modsubgrid.f90:745-768, 855-878, 967-990
      tu = tu + (nu * (u_xp - u_c) * m.dxi - nu * (u_c - u_xm) * m.dxi) * 2. * m.dxi
              + (nu * ((u_yp - u_c) * m.dyi + (v_yp - v_xm_yp) * m.dxi)
               - nu * ((u_c - u_ym) * m.dyi + (v_c - v_xm) * m.dxi)) * m.dyi
              + (nu * ((u_zp - u_c) * dzhi_kp + (w_zp - w_xm_zp) * m.dxi)
               - nu * ((u_c - u_zm) * dzhi_k + (w_c - w_xm) * m.dxi)) * dzfi_k;
      tv = tv + (nu * ((v_xp - v_c) * m.dxi + (u_xp - u_xp_ym) * m.dyi)
               - nu * ((v_c - v_xm) * m.dxi + (u_c - u_ym) * m.dyi)) * m.dxi
              + (nu * (v_yp - v_c) - nu * (v_c - v_ym)) * 2. * m.dy2i
              + (nu * ((v_zp - v_c) * dzhi_kp + (w_zp - w_ym_zp) * m.dyi)
               - nu * ((v_c - v_zm) * dzhi_k + (w_c - w_ym) * m.dyi)) * dzfi_k;
      if (k >= 1) {
        const double dzfi_km = lm.get(5);
        tw = tw + (nu * ((w_xp - w_c) * m.dxi + (u_xp - u_xp_zm) * dzhi_k)
                 - nu * ((w_c - w_xm) * m.dxi + (u_c - u_zm) * dzhi_k)) * m.dxi
                + (nu * ((w_yp - w_c) * m.dyi + (v_yp - v_yp_zm) * dzhi_k)
                 - nu * ((w_c - w_ym) * m.dyi + (v_c - v_zm) * dzhi_k)) * m.dyi
                + (nu * (w_zp - w_c) * dzfi_k - nu * (w_c - w_zm) * dzfi_km) * 2. * dzhi_k;
      }
    }
  }

  if (FORCES) {
    tu = tu - lm.get(10);
    tv = tv - lm.get(11);
    if (k == 0) tw = 0.0;
  }

}
