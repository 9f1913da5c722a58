// Closure (Smagorinsky / Vreman / DNS) and the fused momentum advection + diffusion sweep.
//
// Arithmetic follows the reference expression by expression (uDALES src/modadvection.f90:158-314,
// src/modsubgrid.f90:208-360, 672-997) so that results agree with the CPU path to round-off;
// the data movement is redesigned: one sweep reads u0,v0,w0,pres0,ekm once and updates
// up,vp,wp once (88 B/cell algorithmic) instead of the reference's ~12 sweeps.
#include "udc_internal.h"
#include "udc_mom_arith.h"
#include "udc_closure_arith.h"

namespace {

struct MomArgs {
  const double *u, *v, *w, *p, *ek;
  double *up, *vp, *wp;
};

// x is periodic on a rank: wrap the index (no ghost columns are stored)
__device__ __forceinline__ int wrapm(int i, int nx) { return i == 0 ? nx - 1 : i - 1; }
__device__ __forceinline__ int wrapp(int i, int nx) { return i == nx - 1 ? 0 : i + 1; }

// (The direct-load momentum and closure kernels of round 1 -- one thread per cell, neighbours from L1 / L2 -- were A/B switches
// until round 4 (UDC_MOM_SIMPLE); the LDS-staged sweeps of udc_mom_lds.hip run on every grid size, and the switch is gone.)
__global__ void fill_const_kernel(Geo g, double *__restrict__ a, double val, double *__restrict__ b, double valb) {
  const long q = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (q < g.n) { a[q] = val; b[q] = valb; }
}

// closurebc top/bottom rows, src/modboundary.f90:447-465, applied after the y ghosts are in
// place (equivalent to the reference's "top/bottom, x wrap, y wrap" order: see DESIGN.md).
__global__ void ek_topbot_kernel(Geo g, Params pr, double *__restrict__ ekm, double *__restrict__ ekh) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int j = (int)blockIdx.y - 1;            // rows -1 .. ny
  if (i >= g.nx) return;
  const double nm = pr.numol, nh = pr.numol * pr.prandtlmoli;
  const long top = g.idx(i, j, g.nz - 1), bot = g.idx(i, j, 0);
  if (pr.bctopm == UDC_TOP_NOSLIP) {
    ekm[top + g.sz] = 2. * nm - ekm[top];
    ekh[top + g.sz] = (2. * nh) - ekh[top];
  } else {
    ekm[top + g.sz] = ekm[top];
    ekh[top + g.sz] = ekh[top];
  }
  ekm[bot - g.sz] = 2. * nm - ekm[bot];
  ekh[bot - g.sz] = (2. * nh) - ekh[bot];
}

__global__ __launch_bounds__(256) void forces_kernel(Geo g, TileGrid tg, Metrics m, double *__restrict__ up,
                                                      double *__restrict__ vp, double *__restrict__ wp) {
  int i, j, k;
  const bool inside_ = tile_decode(g, tg, i, j, k);
  if (!inside_) return;
  const long c = g.idx(i, j, k);
  up[c] = up[c] - m.dpdxl[k + 1];
  vp[c] = vp[c] - m.dpdyl[k + 1];
  if (k == 0) wp[c] = 0.0;
}

// `bottom` with lbottom (src/modibm.f90:2021-2026, :2073-2090): wfmneutral case 91
// (src/modwallfunctions.f90:309-346) on the k = kb plane; one thread per (i, j).  The uniform grid has
// dxf = dx and dxhi = 1/dx.  tau_x, tau_y, thl_flux (src/modibm.f90:2015-2018, 2094-2097: what `bottom` added to up, vp,
// thlp, nonzero on the k = kb plane only) are kept as planes when asked for (udc_bottom_diagnostics).
struct BottomArgs {
  const double *u0, *v0, *ekm, *ekh;
  double *up, *vp;
  const double *sv0[16];
  double *svp[16];
  double mult[16];      // 1, or rk3coef where svp holds the scalar's NEW value (its RK3 update rode in its sweep, udc_scalar_lds.hip)
  double *svm_new[16];  // ... on RK stage 3: svm, which took that value too
  double flux[16];       // prescribed floor flux (0 for passive scalars, wtsurf for thl)
  int nsv, wrap_vp;
  double z0, fkar;      // fkar: von Karman constant (&WALLS fkar, src/modglobal.f90:317)
  // wfuno (UNO kernels): wall temperature, roughness length for heat, turbulent Prandtl number, the temperature the
  // stability is judged on, and which entry of sv0/svp is thl when its floor is the wall function too (BCbotT = 2), else -1
  double thls, z0h, prt;
  const double *thl0;
  double thl_air;        // thl0 == nullptr (temperature equation off): the uniform, frozen temperature of the first level
  int thl_wf;
  double *tau_x, *tau_y, *thl_flux;      // [ny_l][nx] planes or nullptr
  int thl_slot;                          // entry of sv0/svp that is thl (for thl_flux), else -1
  int jbeg, jend;                        // rows [jbeg, jend) of this launch
};
// wfuno's transfer coefficients (src/modwallfunctions.f90:176-261): Louis 1979 / Uno et al. 1995 over a rough wall
__device__ __forceinline__ void uno_F(double logdz, double sqdz, double Ri, double fkar2, double &Fm, double &Fh) {
  const double b1 = 9.4, b2 = 4.7, dm = 7.4, dh = 5.3;
  if (Ri > 0.) { Fm = 1. / ((1. + b2 * Ri) * (1. + b2 * Ri)); Fh = Fm; }
  else {
    const double cm = (dm * fkar2) / (logdz * logdz) * b1 * sqdz, ch = (dh * fkar2) / (logdz * logdz) * b1 * sqdz;
    Fm = 1. - (b1 * Ri) / (1. + cm * sqrt(fabs(Ri)));
    Fh = 1. - (b1 * Ri) / (1. + ch * sqrt(fabs(Ri)));
  }
}
__device__ __forceinline__ double uno_m(double prt, double logdz, double logzh, double sqdz, double Ribl0, double fkar2) {
  double Fm, Fh;
  uno_F(logdz, sqdz, Ribl0, fkar2, Fm, Fh);
  const double Mm = prt * logdz * sqrt(Fm) / Fh;
  const double Ribl1 = Ribl0 - Ribl0 * prt * logzh / (prt * logzh + Mm);
  uno_F(logdz, sqdz, Ribl1, fkar2, Fm, Fh);
  return fkar2 / (logdz * logdz) * Fm;
}
__device__ __forceinline__ double uno_h(double prt, double logdz, double logzh, double sqdz, double utangInt, double dT, double Ribl0,
                                        double fkar2) {
  double Fm, Fh;
  uno_F(logdz, sqdz, Ribl0, fkar2, Fm, Fh);
  double Mm = prt * logdz * sqrt(Fm) / Fh;
  const double Ribl1 = Ribl0 - Ribl0 * prt * logzh / (prt * logzh + Mm);
  uno_F(logdz, sqdz, Ribl1, fkar2, Fm, Fh);
  Mm = prt * logdz * sqrt(Fm) / Fh;
  const double dTrough = dT * 1. / (prt * logzh / Mm + 1.);
  const double octh = sqrt(utangInt) * fkar2 / (logdz * logdz) * Fh / prt;
  return octh * dTrough;
}
// UNO = false: wfmneutral (BCbotm = 3, src/modwallfunctions.f90:263-350); true: wfuno case 91 (BCbotm = 2, :72-127)
template <bool UNO>
__global__ __launch_bounds__(256) void bottom_kernel(Geo g, Metrics m, BottomArgs a) {
  const int i = blockIdx.x * 64 + threadIdx.x, j = a.jbeg + blockIdx.y * 4 + threadIdx.y;
  if (i >= g.nx || j >= a.jend) return;
  const long c = g.idx(i, j, 0);
  const long cxm = c - i + (i == 0 ? g.nx - 1 : i - 1), cxp = c - i + (i == g.nx - 1 ? 0 : i + 1);
  const long sy = g.sy, sz = g.sz;
  const int k = 1, km = 0;                       // reference level indices of the metric tables
  const double fkar2 = a.fkar * a.fkar, umin = 0.0001;
  const double delta = 0.5 * m.dzf[k];
  const double l_ = log(delta / a.z0);
  const double logdz2 = l_ * l_;
  double ctm = fkar2 / (logdz2);
  const double dzfi = m.dzfi[k], dzhi = m.dzhi[k], dzhiq = m.dzhiq[k];
  const double grav = 9.81, Twall = a.thls;
  const double logzh = (UNO || a.thl_wf >= 0) ? log(a.z0 / a.z0h) : 0., sqdz = (UNO || a.thl_wf >= 0) ? sqrt(delta / a.z0) : 0.;
  {  // u component, :318-331 (neutral) / :92-109 (uno)
    const double utang1Int = a.u0[c];
    const double utang2Int = (a.v0[c] + a.v0[cxm] + a.v0[c + sy] + a.v0[cxm + sy]) * 0.25;
    const double utangInt = fmax(umin, (utang1Int * utang1Int + utang2Int * utang2Int));
    if (UNO) {
      const double dT = (a.thl0 ? ((a.thl0[c] + a.thl0[cxm]) - (Twall + Twall)) : ((a.thl_air + a.thl_air) - (Twall + Twall))) * 0.5;
      ctm = uno_m(a.prt, l_, logzh, sqdz, grav * delta * dT * 2 / ((Twall + Twall) * utangInt), fkar2);
    }
    const double dummy = fabs(utang1Int) * sqrt(utangInt) * ctm;
    const double bcmomflux = copysign(dummy, utang1Int);
    const double emom = (m.dzf[km] * (a.ekm[c] * m.dx + a.ekm[cxm] * m.dx) +
                         m.dzf[k] * (a.ekm[c - sz] * m.dx + a.ekm[cxm - sz] * m.dx)) * m.dxi * dzhiq;
    const double old = a.up[c];
    const double t = old + (a.u0[c] - a.u0[c - sz]) * emom * dzhi * dzfi - bcmomflux * dzfi;
    a.up[c] = t;
    if (a.tau_x) a.tau_x[(size_t)j * g.nx + i] = t - old;      // tau_x = up - (up before), :2094
  }
  {  // v component, :333-346 / :111-127
    const double utang1Int = (a.u0[c] + a.u0[c - sy] + a.u0[cxp - sy] + a.u0[cxp]) * 0.25;
    const double utang2Int = a.v0[c];
    const double utangInt = fmax(umin, (utang1Int * utang1Int + utang2Int * utang2Int));
    if (UNO) {
      const double dT = (a.thl0 ? ((a.thl0[c] + a.thl0[c - sy]) - (Twall + Twall)) : ((a.thl_air + a.thl_air) - (Twall + Twall))) * 0.5;
      ctm = uno_m(a.prt, l_, logzh, sqdz, grav * delta * dT * 2 / ((Twall + Twall) * utangInt), fkar2);
    }
    const double dummy = fabs(utang2Int) * sqrt(utangInt) * ctm;
    const double bcmomflux = copysign(dummy, utang2Int);
    const double eomm = (m.dzf[km] * (a.ekm[c] + a.ekm[c - sy]) + m.dzf[k] * (a.ekm[c - sz] + a.ekm[c - sy - sz])) * dzhiq;
    const double old = a.vp[c];
    const double t = old + (a.v0[c] - a.v0[c - sz]) * eomm * dzhi * dzfi - bcmomflux * dzfi;
    a.vp[c] = t;
    if (a.tau_y) a.tau_y[(size_t)j * g.nx + i] = t - old;
    if (a.wrap_vp && j == 0) a.vp[c + sy * g.ny] = t;      // bcpup's cyclic pvp(je+1) = pvp(jb)
  }
  for (int n = 0; n < a.nsv; ++n) {   // Neumann floor: scalars src/modibm.f90:2073-2090 (flux 0), thl :2035-2047 (wtsurf)
    const double *c0 = a.sv0[n];
    if (n == a.thl_wf) {              // wfuno case 92 (BCbotT = 2, src/modwallfunctions.f90:131-165): wall at thls
      const double utang1Int = (a.u0[c] + a.u0[cxp]) * 0.5;
      const double utang2Int = (a.v0[c] + a.v0[c + sy]) * 0.5;
      const double utangInt = fmax(umin, (utang1Int * utang1Int + utang2Int * utang2Int));
      const double dT = (c0[c] - Twall);
      const double bcTflux = uno_h(a.prt, l_, logzh, sqdz, utangInt, dT, grav * delta * dT / (Twall * utangInt), fkar2);
      const double old = a.svp[n][c];
      const double t = old + 0.5 * (m.dzf[k - 1] * a.ekh[c] + m.dzf[k] * a.ekh[c - sz]) * (c0[c] - c0[c - sz]) * m.dzh2i[k] * dzfi
                       - bcTflux * dzfi;
      a.svp[n][c] = t;
      if (a.thl_flux && n == a.thl_slot) a.thl_flux[(size_t)j * g.nx + i] = t - old;
      continue;
    }
    const double old = a.svp[n][c];
    const double t = old + a.mult[n] * ((0.5 * (m.dzf[km] * a.ekh[c] + m.dzf[k] * a.ekh[c - sz]) * (c0[c] - c0[c - sz]) * m.dzh2i[k] - a.flux[n]) * dzfi);
    a.svp[n][c] = t;
    if (a.svm_new[n]) a.svm_new[n][c] = t;
    if (a.thl_flux && n == a.thl_slot) a.thl_flux[(size_t)j * g.nx + i] = t - old;
  }
}

// coriolis, src/modforces.f90:600-717.  MODE 1: lcoriol (:627-676); MODE 2: lprofforc (:682-710), relaxation of u
// towards the geostrophic profile ug(k).
template <int MODE>
__global__ __launch_bounds__(256) void coriolis_kernel(Geo g, TileGrid tg, Metrics m, double om22, double om23,
    const double *__restrict__ ug, const double *__restrict__ u0, const double *__restrict__ v0, const double *__restrict__ w0,
    double *__restrict__ up, double *__restrict__ vp, double *__restrict__ wp, int wrap_vp) {
  int i, j, k;
  if (!tile_decode(g, tg, i, j, k)) return;
  const long r0 = g.idx(0, j, k);
  const long c = r0 + i;
  if (MODE == 2) { up[c] = up[c] + om23 * (ug[k + 1] - u0[c]); return; }
  const long xm = r0 + (i == 0 ? g.nx - 1 : i - 1), xp = r0 + (i == g.nx - 1 ? 0 : i + 1);
  const long sy = g.sy, sz = g.sz;
  if (k >= 1) {
    const int kf = k + 1;
    up[c] = up[c] + ((v0[c] + v0[c + sy] + v0[xm] + v0[xm + sy]) * om23 * 0.25)
                  - ((w0[c] + w0[c + sz] + w0[xm + sz] + w0[xm]) * om22 * 0.25);
    const double tv = vp[c] - ((u0[c] + u0[c - sy] + u0[xp - sy] + u0[xp]) * om23 * 0.25);
    vp[c] = tv;
    if (wrap_vp && j == 0) vp[c + sy * g.ny] = tv;       // bcpup's cyclic pvp(je+1) = pvp(jb)
    wp[c] = wp[c] + (((m.dzf[kf - 1] * (u0[c] + u0[xp]) + m.dzf[kf] * (u0[c - sz] + u0[xp - sz])) / m.dzh[kf]) * om22 * 0.25);
  } else {
    up[c] = up[c] + (v0[c] + v0[c + sy] + v0[xm] + v0[xm + sy]) * om23 * 0.25
                  - (w0[c] + w0[c + sz] + w0[xm + sz] + w0[xm]) * om22 * 0.25;
    const double tv = vp[c] - (u0[c] + u0[c - sy] + u0[xp - sy] + u0[xp]) * om23 * 0.25;
    vp[c] = tv;
    if (wrap_vp && j == 0) vp[c + sy * g.ny] = tv;
    wp[c] = 0.0;
  }
}

// shiftedPBCs, src/modforces.f90:953-980
__global__ __launch_bounds__(256) void shifted_pbc_kernel(Geo g, TileGrid tg, double a, const double *__restrict__ sinx,
    const double *__restrict__ u0av, double dyi, const double *__restrict__ u0, const double *__restrict__ v0, const double *__restrict__ w0,
    double *__restrict__ up, double *__restrict__ vp, double *__restrict__ wp, int wrap_vp) {
  int i, j, k;
  if (!tile_decode(g, tg, i, j, k)) return;
  if (i + 1 <= g.nx / 2) return;                       // ig > int(itot/2)
  const long c = g.idx(i, j, k), sy = g.sy;
  const double vs = a * u0av[k] * sinx[i];
  up[c] = up[c] - vs * (u0[c] - u0[c - sy]) * dyi;
  const double tv = vp[c] - vs * (v0[c] - v0[c - sy]) * dyi;
  vp[c] = tv;
  if (wrap_vp && j == 0) vp[c + sy * g.ny] = tv;       // bcpup's cyclic pvp(je+1) = pvp(jb)
  wp[c] = wp[c] - vs * (w0[c] - w0[c - sy]) * dyi;
}

inline dim3 cell_grid(const Geo &g, dim3 b) {
  (void)b;
  return dim3((unsigned)tile_grid(g).tiles * (unsigned)g.nz, 1, 1);
}

}  // namespace

int k_coriolis(udc_handle *h, bool wrap_vp) {
  if (!h->coriolis_mode) return 0;
  const Geo &g = h->g;
  dim3 b(64, 4, 1), gr = cell_grid(g, b);
  PROF(h, "coriolis");
  if (h->coriolis_mode == 1)
    hipLaunchKernelGGL((coriolis_kernel<1>), gr, b, 0, h->stream, g, tile_grid(g), h->m, h->om22, h->om23, h->ug,
                       h->fields[UDC_U0], h->fields[UDC_V0], h->fields[UDC_W0], h->fields[UDC_UP], h->fields[UDC_VP], h->fields[UDC_WP],
                       wrap_vp ? 1 : 0);
  else
    hipLaunchKernelGGL((coriolis_kernel<2>), gr, b, 0, h->stream, g, tile_grid(g), h->m, h->om22, h->om23, h->ug,
                       h->fields[UDC_U0], h->fields[UDC_V0], h->fields[UDC_W0], h->fields[UDC_UP], h->fields[UDC_VP], h->fields[UDC_WP],
                       wrap_vp ? 1 : 0);
  HIP_OK(hipGetLastError());
  return 0;
}

int k_bottom(udc_handle *h, bool wrap_vp, int jbeg, int jend) {
  const Geo &g = h->g;
  BottomArgs a{};
  a.jbeg = jbeg; a.jend = jend < 0 ? g.ny : (jend < g.ny ? jend : g.ny);
  if (a.jend <= a.jbeg) return 0;
  a.u0 = h->fields[UDC_U0]; a.v0 = h->fields[UDC_V0]; a.ekm = h->fields[UDC_EKM]; a.ekh = h->fields[UDC_EKH];
  a.up = h->fields[UDC_UP]; a.vp = h->fields[UDC_VP];
  a.nsv = 0; a.wrap_vp = wrap_vp ? 1 : 0; a.z0 = h->p.z0;
  a.thls = h->floor_thls; a.z0h = h->floor_z0h; a.prt = h->floor_prt; a.fkar = h->fkar; a.thl_wf = -1; a.thl_slot = -1;
  a.tau_x = h->bottom_diag[0]; a.tau_y = h->bottom_diag[1]; a.thl_flux = h->bottom_diag[2];
  const bool have_thl = (int)h->fields.size() > UDC_THL0 && h->fields[UDC_THL0];
  a.thl0 = have_thl ? h->fields[UDC_THL0] : nullptr;
  a.thl_air = h->floor_thl_air;
  const bool uno = h->floor_bcbotm == 2;
  if (((uno && !h->floor_thl_air_on) || h->floor_bcbott == 2) && !have_thl) { udc_set_error("bottom: the wfuno floor reads the temperature of the first level: udc_set_tempeq, or udc_set_floor_air_temperature when the temperature equation is off"); return 1; }
  for (int n : h->slots) {
    if (h->slot[n].tke) continue;      // e12 has no floor-flux correction in `bottom`
    if (n == 15 && h->floor_bcbott == 2) a.thl_wf = a.nsv;
    if (n == 15) a.thl_slot = a.nsv;
    a.sv0[a.nsv] = h->fields[UDC_SV0 + 3 * n]; a.svp[a.nsv] = h->fields[UDC_SVP + 3 * n];
    a.mult[a.nsv] = h->sv_inline[n] ? h->sv_inline_rk3coef : 1.;
    a.svm_new[a.nsv] = (h->sv_inline[n] && h->sv_inline_last) ? h->fields[UDC_SVM + 3 * n] : nullptr;
    a.flux[a.nsv] = h->slot[n].floorflux; ++a.nsv;
  }
  PROF(h, "bottom");
  const dim3 gr((unsigned)((g.nx + 63) / 64), (unsigned)((a.jend - a.jbeg + 3) / 4)), bl(64, 4);
  if (uno) hipLaunchKernelGGL(bottom_kernel<true>, gr, bl, 0, h->stream, g, h->m, a);
  else hipLaunchKernelGGL(bottom_kernel<false>, gr, bl, 0, h->stream, g, h->m, a);
  HIP_OK(hipGetLastError());
  return 0;
}

int k_shifted_pbcs(udc_handle *h, bool wrap_vp) {
  const Geo &g = h->g;
  if (h->shift_a == 0. || !h->shift_tab) return 0;
  PROF(h, "shifted_pbcs");
  dim3 b(64, 4, 1), gr = cell_grid(g, b);
  hipLaunchKernelGGL(shifted_pbc_kernel, gr, b, 0, h->stream, g, tile_grid(g), h->shift_a, (const double *)h->shift_tab,
                     (const double *)(h->shift_tab + g.nx), h->m.dyi, (const double *)h->fields[UDC_U0], (const double *)h->fields[UDC_V0],
                     (const double *)h->fields[UDC_W0], h->fields[UDC_UP], h->fields[UDC_VP], h->fields[UDC_WP], wrap_vp ? 1 : 0);
  HIP_OK(hipGetLastError());
  return 0;
}

int k_forces(udc_handle *h) {
  const Geo &g = h->g;
  dim3 b(64, 4, 1), gr = cell_grid(g, b);
  PROF(h, "forces");
  hipLaunchKernelGGL(forces_kernel, gr, b, 0, h->stream, g, tile_grid(g), h->m, h->fields[UDC_UP], h->fields[UDC_VP],
                     h->fields[UDC_WP]);
  HIP_OK(hipGetLastError());
  return 0;
}

int k_closure(udc_handle *h) {
  const Geo &g = h->g;
  double *ekm = h->fields[UDC_EKM], *ekh = h->fields[UDC_EKH];
  if (h->p.sgs == UDC_SGS_ONEEQN) return k_tke_closure(h);
  if (h->p.sgs != UDC_SGS_DNS) return k_closure_lds(h, false);
  PROF(h, "closure");
  hipLaunchKernelGGL(fill_const_kernel, dim3((unsigned)((g.n + 255) / 256)), dim3(256), 0, h->stream, g, ekm,
                     h->p.numol, ekh, h->p.numol * h->p.prandtlmoli);
  HIP_OK(hipGetLastError());
  return 0;
}

int k_ek_ghosts(udc_handle *h, bool exchange) {
  const Geo &g = h->g;
  const int f[2] = {UDC_EKM, UDC_EKH};
  if (exchange && k_halo_y(h, f, 2, 1)) return 1;      // (false: the caller has exchanged the rows beside the closure sweep)
  PROF(h, "ek_topbot");
  hipLaunchKernelGGL(ek_topbot_kernel, dim3((g.nx + 63) / 64, g.ny + 2), dim3(64), 0, h->stream, g, h->p,
                     h->fields[UDC_EKM], h->fields[UDC_EKH]);
  HIP_OK(hipGetLastError());
  return 0;
}
