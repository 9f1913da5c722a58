// Slab (horizontal-plane) sums and what is built on them: the dry buoyancy term of `forces`
// (src/modforces.f90:73-84 with calc_halflev / the slab average of src/modthermodynamics.f90:76,208,518-524), diagfld's
// slab averages (:262-279) and the per-level forcings tend += A(k) + B(k) field of lstend / nudge / grwdamp
// (src/modforces.f90:719-860, src/modboundary.f90:1447-1488; tables built on the host, udcore/forcings.py).
#include "udc_internal.h"
#include <algorithm>

namespace {

inline dim3 cell_grid(const Geo &g, dim3 b) {
  (void)b;
  return dim3((unsigned)tile_grid(g).tiles * (unsigned)g.nz, 1, 1);
}

// Buoyancy (forces with lbuoyancy, src/modforces.f90:73-84; dry air: thv0h = thl0h, src/modthermodynamics.f90:208):
//   wp(k) += grav (thl0h(i,j,k) - thvh(k)) / thvh(k),  k = kb+1..ke,
//   thl0h(k) = (thl0(k) dzf(k-1) + thl0(k-1) dzf(k)) / (2 dzh(k))          (calc_halflev, :518-524)
//   thvh(k)  = slab average of thl0h(k)                                     (thermodynamics :76, avexy_ibm)
__device__ __forceinline__ double thl_half(const Geo &g, const Metrics &m, const double *__restrict__ t, long c, int k) {
  const int kf = k + 1;
  return (t[c] * m.dzf[kf - 1] + t[c - g.sz] * m.dzf[kf]) / (2 * m.dzh[kf]);
}
// stage 1: one workgroup per (xy tile of 64 x 4 LS_ROWS cells, level) -> part[level * tiles + tile]; stage 2: one workgroup per level
// (LS_ROWS rows per thread: with one cell per thread the sums of a 256^3 field ran at 2 TB/s, profiles/r06/open_x_rate_256_final.txt)
constexpr int LS_ROWS = 8;
__global__ __launch_bounds__(256) void levelsum_kernel(Geo g, Metrics m, int gx, const double *__restrict__ thl, double *__restrict__ part) {
  __shared__ double sw[4];
  const int tile = blockIdx.x, k = blockIdx.y;
  const int by = tile / gx, bx = tile - by * gx;
  const int i = bx * 64 + threadIdx.x;
  double v = 0.;
  if (i >= g.xg && i < g.nx - g.xg && k >= 1) {      // (open x boundaries: ib .. ie)
#pragma unroll
    for (int r = 0; r < LS_ROWS; ++r) {
      const int j = (by * LS_ROWS + r) * 4 + threadIdx.y;
      if (j < g.ny) v += thl_half(g, m, thl, g.idx(i, j, k), k);
    }
  }
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  if (threadIdx.x == 0) sw[threadIdx.y] = v;
  __syncthreads();
  if (threadIdx.x == 0 && threadIdx.y == 0) part[(size_t)k * gridDim.x + tile] = (sw[0] + sw[1]) + (sw[2] + sw[3]);
}
__global__ __launch_bounds__(256) void levelsum_final_kernel(int tiles, const double *__restrict__ part, double *__restrict__ S) {
  __shared__ double sw[4];
  const int k = blockIdx.x;
  double v = 0.;
  for (int q = threadIdx.x; q < tiles; q += 256) v += part[(size_t)k * tiles + q];
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  if ((threadIdx.x & 63) == 0) sw[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) S[k] = (sw[0] + sw[1]) + (sw[2] + sw[3]);
}
// immersed boundary: thvh is the average over the fluid w points of the level (avexy_ibm with IIw, IIws,
// src/modthermodynamics.f90:76): take the listed solid w points back out of the level sums.  One workgroup per level.
__global__ __launch_bounds__(256) void ibm_thvh_correct_kernel(Geo g, Metrics m, const int *__restrict__ pts, const int *__restrict__ off,
                                                                const double *__restrict__ thl, double *__restrict__ S) {
  __shared__ double sw[4];
  const int k = blockIdx.x;
  double v = 0.;
  if (k >= 1)
    for (int q = off[k] + threadIdx.x; q < off[k + 1]; q += 256) v += thl_half(g, m, thl, g.idx(pts[3 * q], pts[3 * q + 1], k), k);
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  if ((threadIdx.x & 63) == 0) sw[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) S[k] = S[k] - ((sw[0] + sw[1]) + (sw[2] + sw[3]));
}
// cntk: fluid w points per level (indexed by the reference's k) with an immersed boundary, else nullptr and cnt applies
__global__ __launch_bounds__(256) void buoyancy_kernel(Geo g, TileGrid tg, Metrics m, const double *__restrict__ thl,
                                                        const double *__restrict__ S, double cnt, const double *__restrict__ cntk,
                                                        double grav, double *__restrict__ wp) {
  int i, j, k;
  if (!tile_decode(g, tg, i, j, k) || k < 1) return;
  const long c = g.idx(i, j, k);
  const double thvh = S[k] / (cntk ? cntk[k + 1] : cnt);
  wp[c] = wp[c] + grav * (thl_half(g, m, thl, c, k) - thvh) / thvh;
}

// ---- moist thermodynamics, src/modthermodynamics.f90 (lmoist) ---------------------------------------------------------
// constants of src/modglobal.f90:271-313
constexpr double TH_RD = 287.04, TH_RV = 461.5, TH_CP = 1004., TH_RLV = 2.26e6, TH_PREF0 = 1.e5, TH_TMELT = 273.16,
                 TH_ES0 = 610.78, TH_AT = 17.27, TH_BT = 35.86;
// thermo (:430-503, lqlnr false): all-or-nothing condensate of one point
__device__ __forceinline__ double th_ql(double thl, double qt, double pressure, double exner) {
  double tl = thl * exner;
  if (tl < 100.0) tl = 100.0;
  const double es = TH_ES0 * exp(TH_AT * (tl - TH_TMELT) / (tl - TH_BT));
  const double qsl = TH_RD / TH_RV * es / (pressure - (1 - TH_RD / TH_RV) * es);
  const double b1 = TH_RLV * TH_RLV / (tl * tl * TH_CP * TH_RV);
  const double qs = qsl * (1. + b1 * qt) / (1. + b1 * qsl);
  return qt - qs > 0. ? qt - qs : 0.;
}
// thermo with lqlnr (:448-473): Newton-Raphson on the temperature from the first guess tl
__device__ __forceinline__ double th_ql_nr(double thl, double qt, double pressure, double exner) {
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
__device__ __forceinline__ double th_cond(int nr, double thl, double qt, double pressure, double exner) {
  return nr ? th_ql_nr(thl, qt, pressure, exner) : th_ql(thl, qt, pressure, exner);
}
// thv0h of calthv (:142-152) from calc_halflev's thl0h, qt0h (:508-539; kf >= kb+1 here) and thermo on the half level
__device__ __forceinline__ double thv_half(const Geo &g, const Metrics &m, const double *__restrict__ thl, const double *__restrict__ qt,
                                           double presh, double exnh, long c, int k, int nr) {
  const double thl0h = thl_half(g, m, thl, c, k), qt0h = thl_half(g, m, qt, c, k);
  const double ql0h = th_cond(nr, thl0h, qt0h, presh, exnh);
  return (thl0h + TH_RLV * ql0h / (TH_CP * exnh)) * (1 + (TH_RV / TH_RD - 1) * qt0h - TH_RV / TH_RD * ql0h);
}
// rows of 64 cells one thread of the slab-sum kernels below walks: fewer, fatter workgroups keep more loads in flight
constexpr int MS_ROWS = 8;
// slab sums of thl0, qt0 and the condensate thermo(thl0, qt0, presf, exnf) gives, levels kf = 1..nz+1 (device k = kf-1)
template <bool QL>
__global__ __launch_bounds__(256) void moist_sums_kernel(Geo g, int gx, const double *__restrict__ thl, const double *__restrict__ qt,
                                                          const double *__restrict__ presf, const double *__restrict__ exnf,
                                                          double *__restrict__ part, int nr, double *__restrict__ ql0) {
  __shared__ double sw[3][4];
  const int tile = blockIdx.x, k = blockIdx.y, kf = k + 1;
  const int by = tile / gx, bx = tile - by * gx;
  const int i = bx * 64 + threadIdx.x;
  double v[3] = {0., 0., 0.};
  const double pf = QL ? presf[kf] : 0., ef = QL ? exnf[kf] : 0.;
#pragma unroll
  for (int r = 0; r < MS_ROWS; ++r) {
    const int j = (by * MS_ROWS + r) * 4 + threadIdx.y;
    if (i < g.nx && j < g.ny) {
      const long c = g.idx(i, j, k);
      const double a = thl[c], b = qt[c];
      const bool in = i >= g.xg && i < g.nx - g.xg;      // (open x boundaries: the averages run over ib .. ie)
      if (in) { v[0] += a; v[1] += b; }
      if (QL) {
        const double ql = th_cond(nr, a, b, pf, ef);
        if (in) v[2] += ql;
        if (ql0) ql0[c - g.sz] = ql;      // the reference's ql0 holds level k at k-1 (sequence association in `thermo`)
      }
    }
  }
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    for (int o = 32; o > 0; o >>= 1) v[q] += __shfl_xor(v[q], o, 64);
    if (threadIdx.x == 0) sw[q][threadIdx.y] = v[q];
  }
  __syncthreads();
  if (threadIdx.x < 3 && threadIdx.y == 0) {
    const int q = threadIdx.x;
    part[((size_t)q * gridDim.y + k) * gridDim.x + tile] = (sw[q][0] + sw[q][1]) + (sw[q][2] + sw[q][3]);
  }
}
__global__ __launch_bounds__(256) void thv_sums_kernel(Geo g, Metrics m, int gx, const double *__restrict__ thl, const double *__restrict__ qt,
                                                        const double *__restrict__ presh, const double *__restrict__ exnh,
                                                        double *__restrict__ part, int nr) {
  __shared__ double sw[4];
  const int tile = blockIdx.x, k = blockIdx.y;
  const int by = tile / gx, bx = tile - by * gx;
  const int i = bx * 64 + threadIdx.x;
  double v = 0.;
  if (k >= 1) {
    const double ph = presh[k + 1], eh = exnh[k + 1];
#pragma unroll
    for (int r = 0; r < MS_ROWS; ++r) {
      const int j = (by * MS_ROWS + r) * 4 + threadIdx.y;
      if (i >= g.xg && i < g.nx - g.xg && j < g.ny) v += thv_half(g, m, thl, qt, ph, eh, g.idx(i, j, k), k, nr);
    }
  }
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  if (threadIdx.x == 0) sw[threadIdx.y] = v;
  __syncthreads();
  if (threadIdx.x == 0 && threadIdx.y == 0) part[(size_t)k * gridDim.x + tile] = (sw[0] + sw[1]) + (sw[2] + sw[3]);
}
// cntk: fluid points per level with an immersed boundary (entry q of a <-> cntk[q]), else null and d applies
__global__ void divide_kernel(double *a, int n, double d, const double *__restrict__ cntk) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q < n) {
    const double c = cntk ? cntk[q] : d;
    a[q] = c > 0. ? a[q] / c : -999.;      // (a level without fluid points: avexy_ibm's -999, src/modmpi.f90:655-659)
  }
}
// Immersed boundary: diagfld's slab averages run over the fluid c cells only (avexy_ibm with IIc / IIcs, :262-279), so the listed
// solid c points come back out of the level sums of moist_sums_kernel.  One workgroup per level k = 0..nz.  The reference's ql0
// holds level k+1 at k (see moist_sums_kernel), so in ql0av the mask of level k meets the condensate of level k+1.
template <bool QL>
__global__ __launch_bounds__(256) void ibm_moist_sums_correct_kernel(Geo g, int ke1, const int *__restrict__ pts, const int *__restrict__ off,
                                                                      const double *__restrict__ thl, const double *__restrict__ qt,
                                                                      const double *__restrict__ presf, const double *__restrict__ exnf,
                                                                      double *__restrict__ sums, int nr) {
  __shared__ double sw[3][4];
  const int k = blockIdx.x;
  const bool qlok = QL && k + 1 <= g.nz;
  const double pf = qlok ? presf[k + 2] : 0., ef = qlok ? exnf[k + 2] : 0.;
  double v[3] = {0., 0., 0.};
  for (int q = off[k] + threadIdx.x; q < off[k + 1]; q += 256) {
    const long c = g.idx(pts[3 * q], pts[3 * q + 1], k);
    v[0] += thl[c]; v[1] += qt[c];
    if (qlok) v[2] += th_cond(nr, thl[c + g.sz], qt[c + g.sz], pf, ef);
  }
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    for (int o = 32; o > 0; o >>= 1) v[q] += __shfl_xor(v[q], o, 64);
    if ((threadIdx.x & 63) == 0) sw[q][threadIdx.x >> 6] = v[q];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    sums[k] = sums[k] - ((sw[0][0] + sw[0][1]) + (sw[0][2] + sw[0][3]));
    sums[ke1 + k] = sums[ke1 + k] - ((sw[1][0] + sw[1][1]) + (sw[1][2] + sw[1][3]));
    if (qlok) sums[2 * ke1 + k + 1] = sums[2 * ke1 + k + 1] - ((sw[2][0] + sw[2][1]) + (sw[2][2] + sw[2][3]));
  }
}
// ... and thvh over the fluid w points (avexy_ibm with IIw / IIws, :76): S[k] = level sum of thv0h at device level k
__global__ __launch_bounds__(256) void ibm_thv_correct_kernel(Geo g, Metrics m, const int *__restrict__ pts, const int *__restrict__ off,
                                                               const double *__restrict__ thl, const double *__restrict__ qt,
                                                               const double *__restrict__ presh, const double *__restrict__ exnh,
                                                               double *__restrict__ S, int nr) {
  __shared__ double sw[4];
  const int k = blockIdx.x;
  double v = 0.;
  if (k >= 1) {
    const double ph = presh[k + 1], eh = exnh[k + 1];
    for (int q = off[k] + threadIdx.x; q < off[k + 1]; q += 256) v += thv_half(g, m, thl, qt, ph, eh, g.idx(pts[3 * q], pts[3 * q + 1], k), k, nr);
  }
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  if ((threadIdx.x & 63) == 0) sw[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) S[k] = S[k] - ((sw[0] + sw[1]) + (sw[2] + sw[3]));
}
__global__ __launch_bounds__(256) void buoyancy_moist_kernel(Geo g, TileGrid tg, Metrics m, const double *__restrict__ thl,
                                                              const double *__restrict__ qt, const double *__restrict__ presh,
                                                              const double *__restrict__ exnh, const double *__restrict__ thvh_tab,
                                                              double grav, double *__restrict__ wp, int nr) {
  int i, j, k;
  if (!tile_decode(g, tg, i, j, k) || k < 1) return;
  const long c = g.idx(i, j, k);
  const double thvh = thvh_tab[k + 1];
  wp[c] = wp[c] + grav * (thv_half(g, m, thl, qt, presh[k + 1], exnh[k + 1], c, k, nr) - thvh) / thvh;
}

// diagfld (:241-350) after the slab sums, one workgroup.  mt: the handle's tables ([n2 = nz+2] each, index = reference k);
// sums: thl0, qt0, ql0 slab sums of levels kf = 1..nz+1 at [q*(nz+1) + kf-1].  fromztop (:366-422) integrates
// p^(rd/cp) downwards-up: the recurrence runs on p^(rd/cp) itself (two serial chains of nz additions, one per wave)
// and the powers are taken in parallel; the reference raises to 1/rdocp and back at every level, a round trip that
// moves the last bit only.
struct DiagArgs { int nz; double cnt, thls, qts, ps, grav; };
__device__ void fromztop_dev(const DiagArgs &a, double *mt, const double *__restrict__ dzf, const double *__restrict__ dzh, double *lds) {
  const int n2 = a.nz + 2, ke1 = a.nz + 1;
  const double rdocp = TH_RD / TH_CP, c0 = a.grav * pow(TH_PREF0, rdocp);
  const double *th0av = mt + udc_handle::MT_TH0AV * n2, *qt0av = mt + udc_handle::MT_QT0AV * n2, *ql0av = mt + udc_handle::MT_QL0AV * n2;
  const double *zf = mt + udc_handle::MT_ZF * n2;
  double *presf = mt + udc_handle::MT_PRESF * n2, *presh = mt + udc_handle::MT_PRESH * n2;
  // the two serial chains read their increments from and write their results to LDS (a chain through global memory
  // pays an L2 round trip per level: the stores keep the compiler from hoisting the loads)
  double *__restrict__ incf = lds, *__restrict__ inch = lds + n2, *__restrict__ pf = lds + 2 * n2, *__restrict__ ph = lds + 3 * n2;
  for (int k = 2 + threadIdx.x; k <= ke1; k += blockDim.x) {
    const double thetah = (th0av[k] * dzf[k - 1] + th0av[k - 1] * dzf[k]) / (2 * dzh[k]);
    const double qth = (qt0av[k] * dzf[k - 1] + qt0av[k - 1] * dzf[k]) / (2 * dzh[k]);
    const double qlh = (ql0av[k] * dzf[k - 1] + ql0av[k - 1] * dzf[k]) / (2 * dzh[k]);
    const double thvh = thetah * (1 + (TH_RV / TH_RD - 1) * qth - TH_RV / TH_RD * qlh);
    incf[k] = c0 * dzh[k] / (TH_CP * thvh);
    const double thvf = th0av[k - 1] * (1 + (TH_RV / TH_RD - 1) * qt0av[k - 1] - TH_RV / TH_RD * ql0av[k - 1]);
    inch[k] = c0 * dzf[k - 1] / (TH_CP * thvf);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const double thvs = a.thls * (1. + (TH_RV / TH_RD - 1.) * a.qts);
    double p = pow(a.ps, rdocp) - c0 * zf[1] / (TH_CP * thvs);
    pf[1] = p;
    for (int k = 2; k <= ke1; ++k) { p = p - incf[k]; pf[k] = p; }
  } else if (threadIdx.x == 64) {
    double p = pow(a.ps, rdocp);
    ph[1] = p;
    for (int k = 2; k <= ke1; ++k) { p = p - inch[k]; ph[k] = p; }
  }
  __syncthreads();
  for (int k = 1 + threadIdx.x; k <= ke1; k += blockDim.x) {
    presf[k] = pow(pf[k], 1. / rdocp);
    presh[k] = k == 1 ? a.ps : pow(ph[k], 1. / rdocp);
  }
  __syncthreads();
}
// cntc: fluid c cells per level (reference index) with an immersed boundary, else null and a.cnt applies
__global__ __launch_bounds__(256) void diagfld_kernel(DiagArgs a, double *mt, const double *__restrict__ sums,
                                                       const double *__restrict__ dzf, const double *__restrict__ dzh, int with_ql,
                                                       const double *__restrict__ cntc) {
  extern __shared__ double lds[];      // 4 (nz+2) doubles
  const int n2 = a.nz + 2, ke1 = a.nz + 1;
  double *thl0av = mt + udc_handle::MT_THL0AV * n2, *qt0av = mt + udc_handle::MT_QT0AV * n2, *ql0av = mt + udc_handle::MT_QL0AV * n2;
  double *th0av = mt + udc_handle::MT_TH0AV * n2, *exnf = mt + udc_handle::MT_EXNF * n2, *exnh = mt + udc_handle::MT_EXNH * n2;
  const double *zf = mt + udc_handle::MT_ZF * n2, *zh = mt + udc_handle::MT_ZH * n2;
  const double *presf = mt + udc_handle::MT_PRESF * n2, *presh = mt + udc_handle::MT_PRESH * n2;
  for (int k = 1 + threadIdx.x; k <= ke1; k += blockDim.x) {
    const double cnt = cntc ? cntc[k] : a.cnt;
    // avexy_ibm's rule for a level without fluid cells: -999 (src/modmpi.f90, as divide_kernel and ibm_thl_val_kernel have it)
    const bool none = cntc && !(cnt > 0.);
    thl0av[k] = none ? -999. : sums[k - 1] / cnt;
    qt0av[k] = none ? -999. : sums[ke1 + k - 1] / cnt;
    // the reference's ql0 holds level k+1 at k and nothing at ke+kh (sequence association in `thermo`, see
    // oracle/udcore_oracle.c orc_thermodynamics): its slab average is one level low
    ql0av[k] = (with_ql && k <= a.nz) ? (none ? -999. : sums[2 * ke1 + k] / cnt) : 0.;
    exnf[k] = 1 - a.grav * zf[k] / (TH_CP * a.thls);
    exnh[k] = 1 - a.grav * zh[k] / (TH_CP * a.thls);
    th0av[k] = thl0av[k] + (TH_RLV / TH_CP) * ql0av[k] / exnf[k];
  }
  __syncthreads();
  fromztop_dev(a, mt, dzf, dzh, lds);
  for (int k = 1 + threadIdx.x; k <= ke1; k += blockDim.x) {
    exnf[k] = pow(presf[k] / TH_PREF0, TH_RD / TH_CP);
    th0av[k] = thl0av[k] + (TH_RLV / TH_CP) * ql0av[k] / exnf[k];
  }
  __syncthreads();
  fromztop_dev(a, mt, dzf, dzh, lds);
  for (int k = 1 + threadIdx.x; k <= ke1; k += blockDim.x) {
    exnf[k] = pow(presf[k] / TH_PREF0, TH_RD / TH_CP);
    exnh[k] = pow((k == 1 ? a.ps : presh[k]) / TH_PREF0, TH_RD / TH_CP);
  }
}

// plain per-level slab sums (stage 1; levelsum_final_kernel is stage 2): levels k = 0..nlev-1 (device), i.e. 1..nlev
// (a thread walks MS_ROWS rows of 64 cells, like the moist sums: one cell per thread left the level sums of a 256^3 field at 2 TB/s --
//  69 us per substep of an inflow / outflow run, whose outlet speed is the mean of u's slab averages; rocprofv3 table of profiles/r06)
__global__ __launch_bounds__(256) void levelsum_plain_kernel(Geo g, int gx, const double *__restrict__ f, double *__restrict__ part, int k0 = 0) {
  __shared__ double sw[4];
  const int tile = blockIdx.x, k = blockIdx.y;
  const int by = tile / gx, bx = tile - by * gx;
  const int i = bx * 64 + threadIdx.x;
  double v = 0.;
  if (i >= g.xg && i < g.nx - g.xg) {      // (open x boundaries: the interior columns)
#pragma unroll
    for (int r = 0; r < MS_ROWS; ++r) {
      const int j = (by * MS_ROWS + r) * 4 + threadIdx.y;
      if (j < g.ny) v += f[g.idx(i, j, k0 + k)];
    }
  }
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  if (threadIdx.x == 0) sw[threadIdx.y] = v;
  __syncthreads();
  if (threadIdx.x == 0 && threadIdx.y == 0) part[(size_t)k * gridDim.x + tile] = (sw[0] + sw[1]) + (sw[2] + sw[3]);
}
// tend += A(k) + B(k) src   (A, B indexed by the reference's k)
__global__ __launch_bounds__(256) void level_affine_kernel(Geo g, TileGrid tg, const double *__restrict__ A, const double *__restrict__ B,
                                                            const double *__restrict__ src, double *__restrict__ tend, int wrap) {
  int i, j, k;
  if (!tile_decode(g, tg, i, j, k)) return;
  const long c = g.idx(i, j, k);
  double t = tend[c] + A[k + 1];
  if (src) t = t + B[k + 1] * src[c];
  tend[c] = t;
  if (wrap && j == 0) tend[c + (long)g.sy * g.ny] = t;
}

}  // namespace

// slab averages of nf fields in one go: one all-reduce, one copy back, one synchronisation
int k_slab_averages(udc_handle *h, const int *fields, int nf, double *avg_host, int n) {
  const Geo &g = h->g;
  if (n < 1 || n > g.nz + 1) { udc_set_error("udc_slab_average: 1 <= n <= ktot+1"); return 1; }
  if (nf < 1 || nf > 16) { udc_set_error("udc_slab_averages: at most 16 fields per call"); return 1; }
  for (int q = 0; q < nf; ++q)
    if (fields[q] < 0 || fields[q] >= (int)h->fields.size() || !h->fields[fields[q]]) { udc_set_error("udc_slab_average: unknown field %d", fields[q]); return 1; }
  TileGrid tg = tile_grid(g);
  tg.tiles = tg.gx * ((g.ny + 4 * MS_ROWS - 1) / (4 * MS_ROWS));      // 64 x (4 MS_ROWS) cells per workgroup of levelsum_plain_kernel
  const size_t need = (size_t)tg.tiles * n * nf;
  if (h->lev_cap < need) {
    if (h->lev_part) HIP_OK(hipFree(h->lev_part));
    HIP_OK(hipMalloc(&h->lev_part, sizeof(double) * need));
    h->lev_cap = need;
  }
  if (!h->lev_sum16) HIP_OK(hipMalloc(&h->lev_sum16, sizeof(double) * 16 * (g.nz + 2)));
  for (int q = 0; q < nf; ++q)
    hipLaunchKernelGGL(levelsum_plain_kernel, dim3((unsigned)tg.tiles, (unsigned)n), dim3(64, 4), 0, h->stream, g, tg.gx,
                       (const double *)h->fields[fields[q]], h->lev_part + (size_t)q * n * tg.tiles);
  hipLaunchKernelGGL(levelsum_final_kernel, dim3((unsigned)(n * nf)), dim3(256), 0, h->stream, tg.tiles, h->lev_part, h->lev_sum16);
  HIP_OK(hipGetLastError());
  // immersed boundary: avexy_ibm averages over the fluid cells only (IIu / IIv / IIw / IIc, src/modthermodynamics.f90:271-301)
  if (k_ibm_levelsum_correct(h, fields, nf, n, h->lev_sum16)) return 1;
  if (comm_allreduce(h, h->lev_sum16, n * nf, 1)) return 1;
  HIP_OK(hipMemcpyAsync(h->red_host, h->lev_sum16, sizeof(double) * n * nf, hipMemcpyDeviceToHost, h->stream));
  HIP_OK(hipStreamSynchronize(h->stream));
  const double cnt = (double)(g.nx - 2 * g.xg) * (double)h->cfg.jtot;
  for (int q = 0; q < nf; ++q) {
    const std::vector<double> *fc = h->ibm_on ? &h->ibm[ibm_grid_of_field(fields[q])].fluid_cnt : nullptr;
    for (int k = 0; k < n; ++k) {
      double c = fc ? (*fc)[k + 1] : cnt;
      // a level without fluid cells (avexy_ibm, src/modmpi.f90:649-660): -999, except the floor level, which takes the
      // count of level ke (its sum is then the unmasked one: not reproduced, the floor level is never all solid here)
      if (c == 0. && k == 0 && fc) c = (*fc)[g.nz];
      avg_host[(size_t)q * n + k] = c > 0. ? h->red_host[(size_t)q * n + k] / c : -999.;
    }
  }
  return 0;
}
int k_slab_average(udc_handle *h, int field, double *avg_host, int n) { return k_slab_averages(h, &field, 1, avg_host, n); }

// level sums of one field over levels k0+1..k0+n with the immersed boundary's solid points taken out, all-reduced over the
// slabs, left in h->lev_sum16[0..n) on the device (no host round trip: for kernels that consume them)
int k_level_sums_dev(udc_handle *h, int field, int n, int k0) {
  const Geo &g = h->g;
  TileGrid tg = tile_grid(g);
  tg.tiles = tg.gx * ((g.ny + 4 * MS_ROWS - 1) / (4 * MS_ROWS));
  const size_t need = (size_t)tg.tiles * n;
  if (h->lev_cap < need) {
    if (h->lev_part) HIP_OK(hipFree(h->lev_part));
    HIP_OK(hipMalloc(&h->lev_part, sizeof(double) * need));
    h->lev_cap = need;
  }
  if (!h->lev_sum16) HIP_OK(hipMalloc(&h->lev_sum16, sizeof(double) * 16 * (g.nz + 2)));
  hipLaunchKernelGGL(levelsum_plain_kernel, dim3((unsigned)tg.tiles, (unsigned)n), dim3(64, 4), 0, h->stream, g, tg.gx,
                     (const double *)h->fields[field], h->lev_part, k0);
  hipLaunchKernelGGL(levelsum_final_kernel, dim3((unsigned)n), dim3(256), 0, h->stream, tg.tiles, h->lev_part, h->lev_sum16);
  HIP_OK(hipGetLastError());
  if (k_ibm_levelsum_correct(h, &field, 1, n, h->lev_sum16, k0)) return 1;
  return comm_allreduce(h, h->lev_sum16, n, 1);
}

int k_level_forcings(udc_handle *h, int when, bool wrap_vp) {
  const Geo &g = h->g;
  dim3 b(64, 4, 1), gr = cell_grid(g, b);
  for (const auto &f : h->level_forcings) {
    if (f.when != when) continue;
    PROF(h, "level_forcing");
    hipLaunchKernelGGL(level_affine_kernel, gr, b, 0, h->stream, g, tile_grid(g), (const double *)f.A, (const double *)f.B,
                       f.src >= 0 ? (const double *)h->fields[f.src] : nullptr, h->fields[f.tend], (wrap_vp && f.tend == UDC_VP) ? 1 : 0);
    HIP_OK(hipGetLastError());
  }
  return 0;
}

static int lev_scratch(udc_handle *h, size_t need) {
  if (h->lev_cap < need) {
    if (h->lev_part) HIP_OK(hipFree(h->lev_part));
    HIP_OK(hipMalloc(&h->lev_part, sizeof(double) * need));
    h->lev_cap = need;
  }
  return 0;
}

// thermodynamics, src/modthermodynamics.f90:57-124 with lmoist: [diagfld when none has run yet, :64] thermo(thl0, qt0)
// with the previous call's presf/exnf, diagfld (slab averages, fromztop twice, exner functions), calc_halflev + thermo
// on the half levels + calthv's thv0h, summed to thvh.  thv0h itself is recomputed where forces needs it.
int k_thermodynamics(udc_handle *h) {
  const Geo &g = h->g;
  if (!h->lmoist || !h->mt) { udc_set_error("udc_thermodynamics: moist thermodynamics are not set up (udc_set_moisture, udc_set_moist_thermo)"); return 1; }
  if ((int)h->fields.size() <= UDC_THL0 || !h->fields[UDC_THL0]) { udc_set_error("udc_thermodynamics needs the temperature equation (udc_set_tempeq)"); return 1; }
  const TileGrid tg = tile_grid(g);
  const int n2 = g.nz + 2, ke1 = g.nz + 1;
  const int gy = (g.ny + 4 * MS_ROWS - 1) / (4 * MS_ROWS), mtiles = tg.gx * gy;      // 64 x (4 MS_ROWS) cells per workgroup
  if (lev_scratch(h, (size_t)mtiles * ke1 * 3)) return 1;
  const double *thl = h->fields[UDC_THL0], *qt = h->fields[UDC_QT0];
  double *mt = h->mt, *sums = mt + udc_handle::MT_SUMS * n2;
  const double cnt = (double)(g.nx - 2 * g.xg) * (double)h->cfg.jtot;
  const DiagArgs da{g.nz, cnt, h->thls, h->qts, h->ps, h->grav};
  double *ql0 = nullptr;      // kept as a field only where something reads it: the one-equation closure's moist dthvdz
  if (h->p.sgs == UDC_SGS_ONEEQN || h->lbuoycorr) {      // (and the Vreman buoyancy correction's)
    if ((int)h->fields.size() <= UDC_QL0 || !h->fields[UDC_QL0]) {
      h->fields.resize(std::max((size_t)UDC_QL0 + 1, h->fields.size()), nullptr);
      HIP_OK(hipMalloc(&h->fields[UDC_QL0], sizeof(double) * g.n));
      HIP_OK(hipMemsetAsync(h->fields[UDC_QL0], 0, sizeof(double) * g.n, h->stream));
    }
    ql0 = h->fields[UDC_QL0];
  }
  // immersed boundary: the averages run over the fluid cells (c grid) / fluid w points only
  const udc_handle::IbmGrid *Cg = h->ibm_on ? &h->ibm[3] : nullptr, *Wg = h->ibm_on ? &h->ibm[2] : nullptr;
  if (Cg && !Cg->given) { udc_set_error("udc_thermodynamics with an immersed boundary: the c-grid point lists are needed (udc_set_ibm_points, grid 3)"); return 1; }
  PROF(h, "thermodynamics");
  for (int pass = h->mt_valid ? 1 : 0; pass < 2; ++pass) {
    const dim3 gr((unsigned)mtiles, (unsigned)ke1), b(64, 4);
    if (pass) hipLaunchKernelGGL(moist_sums_kernel<true>, gr, b, 0, h->stream, g, tg.gx, thl, qt, (const double *)(mt + udc_handle::MT_PRESF * n2),
                                 (const double *)(mt + udc_handle::MT_EXNF * n2), h->lev_part, h->lqlnr, ql0);
    else hipLaunchKernelGGL(moist_sums_kernel<false>, gr, b, 0, h->stream, g, tg.gx, thl, qt, (const double *)nullptr, (const double *)nullptr, h->lev_part, 0, (double *)nullptr);
    hipLaunchKernelGGL(levelsum_final_kernel, dim3((unsigned)(3 * ke1)), dim3(256), 0, h->stream, mtiles, h->lev_part, sums);
    if (Cg && Cg->nsolid) {
      if (pass) hipLaunchKernelGGL(ibm_moist_sums_correct_kernel<true>, dim3((unsigned)ke1), dim3(256), 0, h->stream, g, ke1, (const int *)Cg->lev_pts,
                                   (const int *)Cg->lev_off, thl, qt, (const double *)(mt + udc_handle::MT_PRESF * n2),
                                   (const double *)(mt + udc_handle::MT_EXNF * n2), sums, h->lqlnr);
      else hipLaunchKernelGGL(ibm_moist_sums_correct_kernel<false>, dim3((unsigned)ke1), dim3(256), 0, h->stream, g, ke1, (const int *)Cg->lev_pts,
                              (const int *)Cg->lev_off, thl, qt, (const double *)nullptr, (const double *)nullptr, sums, 0);
    }
    HIP_OK(hipGetLastError());
    if (comm_allreduce(h, sums, 3 * ke1, 1)) return 1;       // avexy_ibm's MPI_ALLREDUCE over the slabs
    hipLaunchKernelGGL(diagfld_kernel, dim3(1), dim3(256), sizeof(double) * 4 * n2, h->stream, da, mt, (const double *)sums, h->m.dzf, h->m.dzh, pass,
                       (const double *)(Cg ? Cg->cnt_dev : nullptr));
    HIP_OK(hipGetLastError());
    h->mt_valid = true;
  }
  double *thvh = mt + udc_handle::MT_THVH * n2;
  hipLaunchKernelGGL(thv_sums_kernel, dim3((unsigned)mtiles, (unsigned)g.nz), dim3(64, 4), 0, h->stream, g, h->m, tg.gx, thl, qt,
                     (const double *)(mt + udc_handle::MT_PRESH * n2), (const double *)(mt + udc_handle::MT_EXNH * n2), h->lev_part, h->lqlnr);
  hipLaunchKernelGGL(levelsum_final_kernel, dim3((unsigned)g.nz), dim3(256), 0, h->stream, mtiles, h->lev_part, thvh + 1);
  if (Wg && Wg->nsolid)
    hipLaunchKernelGGL(ibm_thv_correct_kernel, dim3((unsigned)g.nz), dim3(256), 0, h->stream, g, h->m, (const int *)Wg->lev_pts, (const int *)Wg->lev_off,
                       thl, qt, (const double *)(mt + udc_handle::MT_PRESH * n2), (const double *)(mt + udc_handle::MT_EXNH * n2), thvh + 1, h->lqlnr);
  HIP_OK(hipGetLastError());
  if (comm_allreduce(h, thvh + 1, g.nz, 1)) return 1;
  // (entry q of thvh + 1 is the reference's level q + 1; level kb itself is never read: forces starts at kb + 1)
  hipLaunchKernelGGL(divide_kernel, dim3((g.nz + 255) / 256), dim3(256), 0, h->stream, thvh + 1, g.nz, cnt, (const double *)(Wg ? Wg->cnt_dev + 1 : nullptr));
  HIP_OK(hipGetLastError());
  return 0;
}

static int k_buoyancy_moist(udc_handle *h) {
  const Geo &g = h->g;
  if (!h->mt || !h->mt_valid) {
    udc_set_error("moist buoyancy: call udc_thermodynamics once before the first substep (src/program.f90:120)");
    return 1;
  }
  const int n2 = g.nz + 2;
  PROF(h, "buoyancy");
  dim3 b(64, 4, 1), gr = cell_grid(g, b);
  hipLaunchKernelGGL(buoyancy_moist_kernel, gr, b, 0, h->stream, g, tile_grid(g), h->m, (const double *)h->fields[UDC_THL0],
                     (const double *)h->fields[UDC_QT0], (const double *)(h->mt + udc_handle::MT_PRESH * n2),
                     (const double *)(h->mt + udc_handle::MT_EXNH * n2), (const double *)(h->mt + udc_handle::MT_THVH * n2), h->grav,
                     h->fields[UDC_WP], h->lqlnr);
  HIP_OK(hipGetLastError());
  return 0;
}

int k_buoyancy(udc_handle *h) {
  const Geo &g = h->g;
  if (!h->lbuoyancy) return 0;
  if ((int)h->fields.size() <= UDC_THL0 || !h->fields[UDC_THL0]) { udc_set_error("buoyancy needs the temperature equation (udc_set_tempeq)"); return 1; }
  if (h->lmoist) return k_buoyancy_moist(h);
  const TileGrid tg = tile_grid(g);
  const int ltiles = tg.gx * ((g.ny + 4 * LS_ROWS - 1) / (4 * LS_ROWS));      // levelsum_kernel's workgroups per level
  const size_t need = (size_t)tg.tiles * g.nz;
  if (h->lev_cap < need) {
    if (h->lev_part) HIP_OK(hipFree(h->lev_part));
    HIP_OK(hipMalloc(&h->lev_part, sizeof(double) * need));
    h->lev_cap = need;
  }
  if (!h->lev_sum) HIP_OK(hipMalloc(&h->lev_sum, sizeof(double) * (g.nz + 2)));
  PROF(h, "buoyancy");
  const double *thl = h->fields[UDC_THL0];
  hipLaunchKernelGGL(levelsum_kernel, dim3((unsigned)ltiles, (unsigned)g.nz), dim3(64, 4), 0, h->stream, g, h->m, tg.gx, thl, h->lev_part);
  hipLaunchKernelGGL(levelsum_final_kernel, dim3((unsigned)g.nz), dim3(256), 0, h->stream, ltiles, h->lev_part, h->lev_sum);
  HIP_OK(hipGetLastError());
  const double *cntk = nullptr;
  if (h->ibm_on) {
    const udc_handle::IbmGrid &W = h->ibm[2];
    if (W.nsolid)
      hipLaunchKernelGGL(ibm_thvh_correct_kernel, dim3((unsigned)g.nz), dim3(256), 0, h->stream, g, h->m, (const int *)W.lev_pts,
                         (const int *)W.lev_off, thl, h->lev_sum);
    cntk = W.cnt_dev;
  }
  if (comm_allreduce(h, h->lev_sum, g.nz, 1)) return 1;       // avexy_ibm's MPI_ALLREDUCE over the slabs
  dim3 b(64, 4, 1), gr = cell_grid(g, b);
  hipLaunchKernelGGL(buoyancy_kernel, gr, b, 0, h->stream, g, tg, h->m, thl, h->lev_sum, (double)(g.nx - 2 * g.xg) * (double)h->cfg.jtot, cntk,
                     h->grav, h->fields[UDC_WP]);
  HIP_OK(hipGetLastError());
  return 0;
}

