// One-equation (TKE) closure, &NAMSUBGRID loneeqn: closure branch src/modsubgrid.f90:363-400, sources :415-538,
// the e120 floor ghost of `bottom` (src/modibm.f90:2012-2013).  e120 itself is transported as scalar slot 14
// (advecc_2nd + diffe, udc_scalar.hip) and clipped in the integrate kernel (udc_pois.hip).
#include "udc_internal.h"

namespace {

__device__ __forceinline__ int wrap(int i, int nx) { return i < 0 ? i + nx : (i >= nx ? i - nx : i); }

inline dim3 cell_grid(const Geo &g, dim3 b) {
  (void)b;
  return dim3((unsigned)tile_grid(g).tiles * (unsigned)g.nz, 1, 1);
}

// ---- one-equation closure -------------------------------------------------------------------------
struct TkeK { double cm, cn, ch1, ch2, ce1, ce2, grav_thvs, numol, prandtlmoli; int ldelta; };
// dthvdz of calthv for dry air (src/modthermodynamics.f90:208-222, eps1 clamp :224-232): thl may be null (neutral)
// dthvdz is STATE in the reference: calthv computes it when `thermodynamics` runs (end of every substep, after `boundary`) and the
// closure of the next substep reads it.  Nothing changes thl0 / qt0 in between, so evaluating it here from the fields is the same
// thing -- except right after a start-up, where the reference's thermodynamics runs BEFORE `boundary` has set the top ghost plane
// (src/modstartup.f90:1601 against src/program.f90:118).  TopGhost then holds that plane as calthv saw it (udc_calthv).
struct TopGhost { const double *thl, *qt; };
__device__ __forceinline__ double above(const Geo &g, const double *__restrict__ f, const double *__restrict__ top, long c, int k) {
  return (top && k == g.nz - 1) ? top[c - g.sz * (long)(k + HZ)] : f[c + g.sz];
}
__device__ __forceinline__ double dthvdz_at(const Geo &g, const Metrics &m, const double *__restrict__ thl, const TopGhost &tg, long c, int k) {
  const double eps1 = 1e-10;
  double d = 0.;
  if (thl && k >= 1) d = (above(g, thl, tg.thl, c, k) - thl[c - g.sz]) / (m.dzh[k + 2] + m.dzh[k + 1]);
  if (fabs(d) < eps1) d = copysign(eps1, d);
  return d;
}
// the same for moist air (:154-205): the unsaturated jump, or the saturated one where the mixed parcel stays saturated.
// ql0 is the reference's level-shifted field, exnf / zf the thermodynamics tables (index = reference k)
struct MoistK { const double *qt, *ql0, *exnf, *zf; TopGhost top; };
__device__ __forceinline__ double dthvdz_moist_at(const Geo &g, const Metrics &m, const double *__restrict__ thl, const MoistK &q, const TopGhost &tg, long c, int k) {
  const double eps1 = 1e-10, rd = 287.04, rv = 461.5, cp = 1004., rlv = 2.26e6, chi_half = 0.5;
  double d = 0.;
  if (k >= 1) {
    const int kf = k + 1;
    const double epsilon = rd / rv, eps_I = 1 / epsilon - 1.;
    const double a_dry = 1. + eps_I * q.qt[c], b_dry = eps_I * thl[c];
    const double dth = above(g, thl, tg.thl, c, k) - thl[c - g.sz], dq = above(g, q.qt, tg.qt, c, k) - q.qt[c - g.sz];
    const double del_thv_dry = a_dry * dth + b_dry * dq;
    double dthv = del_thv_dry;
    const double ql = q.ql0[c];
    if (ql > 0) {
      const double temp = thl[c] * q.exnf[kf] + (rlv / cp) * ql;
      const double qs = q.qt[c] - ql;
      const double a_moist = (1. - q.qt[c] + qs / epsilon * (1. + rlv / (rv * temp))) / (1. + rlv * rlv * qs / (cp * rv * (temp * temp)));
      const double b_moist = a_moist * rlv / cp - temp;
      const double c_liquid = a_dry * rlv / cp - thl[c] / epsilon;
      const double del_thv_sat = a_moist * dth + b_moist * dq;
      const double chi = 2 * chi_half * (q.zf[kf] - q.zf[kf - 1]) / (m.dzh[kf] + m.dzh[kf + 1]);
      const double chi_sat = c_liquid * ql / (del_thv_dry - del_thv_sat);
      if (chi < chi_sat) dthv = del_thv_sat;
    }
    d = dthv / (m.dzh[kf + 1] + m.dzh[kf]);
  }
  if (fabs(d) < eps1) d = copysign(eps1, d);
  return d;
}
__device__ __forceinline__ double dthvdz_any(const Geo &g, const Metrics &m, const double *__restrict__ thl, const MoistK &q, const TopGhost &tg, long c, int k) {
  return q.ql0 ? dthvdz_moist_at(g, m, thl, q, tg, c, k) : dthvdz_at(g, m, thl, tg, c, k);
}
__device__ __forceinline__ double tke_zlt(const TkeK &t, double delta, double e, double dthvdz) {
  if (t.ldelta || dthvdz <= 0) return delta;
  return fmin(delta, t.cn * e / sqrt(t.grav_thvs * fabs(dthvdz)));
}
// closure, loneeqn branch: src/modsubgrid.f90:363-400 (damp = 1)
__global__ __launch_bounds__(256) void tke_closure_kernel(Geo g, TileGrid tg, Metrics m, TkeK t, const double *__restrict__ e12,
    const double *__restrict__ thl, MoistK mq, double *__restrict__ ekm, double *__restrict__ ekh) {
  int i, j, k;
  if (!tile_decode(g, tg, i, j, k)) return;
  const long c = g.idx(i, j, k);
  const double delta = m.delta[k + 1];
  const double e = e12[c];
  const double dth = dthvdz_any(g, m, thl, mq, mq.top, c, k);
  double em, eh;
  if (t.ldelta || dth <= 0) {
    em = t.cm * delta * 1. * e;
    eh = (t.ch1 + t.ch2) * em;
  } else {
    const double zlt = fmin(delta, t.cn * e / sqrt(t.grav_thvs * fabs(dth)));
    em = t.cm * zlt * 1. * e;
    eh = (t.ch1 + t.ch2 * zlt / delta) * em;
  }
  ekm[c] = em + t.numol;
  ekh[c] = eh + t.numol * t.prandtlmoli;
}
// sources, src/modsubgrid.f90:450-497: k = kb+1..ke (the lowest level gets no source: its sb* arrays are never
// written in the reference)
__global__ __launch_bounds__(256) void tke_sources_kernel(Geo g, TileGrid tg, Metrics m, TkeK t, const double *__restrict__ u0,
    const double *__restrict__ v0, const double *__restrict__ w0, const double *__restrict__ e12, const double *__restrict__ thl,
    const double *__restrict__ ekm, const double *__restrict__ ekh, double *__restrict__ e12p, MoistK mq) {
  int i, j, k;
  if (!tile_decode(g, tg, i, j, k) || k < 1) return;
  const long r0 = g.idx(0, j, k);
  const long c = r0 + i, im = r0 + wrap(i - 1, g.nx), ip = r0 + wrap(i + 1, g.nx);
  const long sy = g.sy, sz = g.sz;
  const int kf = k + 1;
  const double dxi = m.dxi, dyi = m.dyi, dzfi = m.dzfi[kf], hk = m.dzhi[kf], hkp = m.dzhi[kf + 1];
  auto sq = [](double x) { return x * x; };
  double tdef2 = 2. * (sq((u0[ip] - u0[c]) * dxi) + sq((v0[c + sy] - v0[c]) * dyi) + sq((w0[c + sz] - w0[c]) * dzfi));
  tdef2 = tdef2 + 0.25 * (sq((w0[c + sz] - w0[im + sz]) * dxi + (u0[c + sz] - u0[c]) * hkp)
                        + sq((w0[c] - w0[im]) * dxi + (u0[c] - u0[c - sz]) * hk)
                        + sq((w0[ip] - w0[c]) * dxi + (u0[ip] - u0[ip - sz]) * hk)
                        + sq((w0[ip + sz] - w0[c + sz]) * dxi + (u0[ip + sz] - u0[ip]) * hkp));
  tdef2 = tdef2 + 0.25 * (sq((u0[c + sy] - u0[c]) * dyi + (v0[c + sy] - v0[im + sy]) * dxi)
                        + sq((u0[c] - u0[c - sy]) * dyi + (v0[c] - v0[im]) * dxi)
                        + sq((u0[ip] - u0[ip - sy]) * dyi + (v0[ip] - v0[c]) * dxi)
                        + sq((u0[ip + sy] - u0[ip]) * dyi + (v0[ip + sy] - v0[c + sy]) * dxi));
  tdef2 = tdef2 + 0.25 * (sq((v0[c + sz] - v0[c]) * hkp + (w0[c + sz] - w0[c - sy + sz]) * dyi)
                        + sq((v0[c] - v0[c - sz]) * hk + (w0[c] - w0[c - sy]) * dyi)
                        + sq((v0[c + sy] - v0[c + sy - sz]) * hk + (w0[c + sy] - w0[c]) * dyi)
                        + sq((v0[c + sy + sz] - v0[c + sy]) * hkp + (w0[c + sy + sz] - w0[c + sz]) * dyi));
  const double e = e12[c], delta = m.delta[kf];
  const double dth = dthvdz_any(g, m, thl, mq, mq.top, c, k);
  const double zlt = tke_zlt(t, delta, e, dth);
  const double sbshr = (ekm[c] - t.numol) * tdef2 / (2 * e);
  const double sbbuo = -(ekh[c] - t.numol * t.prandtlmoli) * t.grav_thvs * dth / (2 * e);
  const double sbdiss = -2. * (t.ce1 + t.ce2 * zlt / delta) * (e * e) / (2. * 1. * zlt);
  e12p[c] = e12p[c] + sbshr + sbbuo + sbdiss;
}
// `bottom`, src/modibm.f90:2012-2013: e120(kb-1) = e120(kb), e12m(kb-1) = e12m(kb), whole padded plane
__global__ void tke_floor_kernel(Geo g, double *__restrict__ e0, double *__restrict__ em) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int j = (int)blockIdx.y - HY;
  if (i >= g.nx) return;
  const long c = g.idx(i, j, 0);
  e0[c - g.sz] = e0[c];
  em[c - g.sz] = em[c];
}

}  // namespace

static TkeK tke_consts(udc_handle *h) {
  return TkeK{h->tke.cm, h->tke.cn, h->tke.ch1, h->tke.ch2, h->tke.ce1, h->tke.ce2, h->tke.grav / h->tke.thvs,
              h->p.numol, h->p.prandtlmoli, h->tke.ldelta};
}
static const double *thl_or_null(udc_handle *h) {
  return ((int)h->fields.size() > UDC_THL0 && h->fields[UDC_THL0]) ? h->fields[UDC_THL0] : nullptr;
}
// Vreman buoyancy correction for stable stratification (lbuoycorr, src/modsubgrid.f90:330-353): the closure kernel left
// the turbulent ekm; scale it with the gradient Richardson number (dthvdz: calthv's, from the same thl0), then ekh and
// the molecular parts in the reference's statement order (:355-359)
__global__ __launch_bounds__(256) void vreman_buoycorr_kernel(Geo g, TileGrid tg, Metrics m, Params pr, double grav, double rigc,
                                                              const double *__restrict__ u0, const double *__restrict__ v0,
                                                              const double *__restrict__ thl, MoistK mq,
                                                              double *__restrict__ ekm, double *__restrict__ ekh) {
  int i, j, k;
  if (!tile_decode(g, tg, i, j, k)) return;
  const long c = g.idx(i, j, k), cip = g.idx(wrap(i + 1, g.nx), j, k);
  const int kf = k + 1;
  const double dz2 = m.dzh[kf + 1] + m.dzh[kf];
  const double du0dz = 0.5 * ((u0[c + g.sz] + u0[cip + g.sz]) - (u0[c - g.sz] + u0[cip - g.sz])) / dz2;
  const double dv0dz = 0.5 * ((v0[c + g.sz] + v0[c + g.sy + g.sz]) - (v0[c - g.sz] + v0[c + g.sy - g.sz])) / dz2;
  const double rig = ((grav / thl[c]) * dthvdz_any(g, m, thl, mq, mq.top, c, k)) / (du0dz * du0dz + dv0dz * dv0dz + 1.e-10);
  double em = ekm[c] * sqrt(1.0 - fmin(fmax(rig, 0.0), rigc) / rigc);
  double eh = em * pr.prandtli;
  em = em + pr.numol;
  eh = eh + pr.numol * pr.prandtlmoli;
  ekm[c] = em;
  ekh[c] = eh;
}

// moist dthvdz inputs, or nulls for dry air; fails when the thermodynamics have not run yet (ql0, exnf undefined)
static int moist_inputs(udc_handle *h, MoistK &q) {
  const TopGhost top{h->dthv_top_on ? h->dthv_top : nullptr, (h->dthv_top_on && h->lmoist) ? h->dthv_top + (size_t)h->g.sy * h->g.py : nullptr};
  q = MoistK{nullptr, nullptr, nullptr, nullptr, top};
  if (!h->lmoist) return 0;
  if (!h->mt || !h->mt_valid || (int)h->fields.size() <= UDC_QL0 || !h->fields[UDC_QL0]) {
    udc_set_error("one-equation closure / Vreman buoyancy correction with moisture: set up udc_set_moist_thermo and call udc_thermodynamics before the first "
                  "substep (calthv's dthvdz reads ql0 and exnf, src/modthermodynamics.f90:154-205)");
    return 1;
  }
  const int n2 = h->g.nz + 2;
  q = MoistK{h->fields[UDC_QT0], h->fields[UDC_QL0], h->mt + udc_handle::MT_EXNF * n2, h->mt + udc_handle::MT_ZF * n2, top};
  return 0;
}
// udc_calthv: keep the top ghost planes of thl0 (and qt0) as they are now -- what the reference's calthv, running at this point,
// builds the top level's dthvdz from; the closures read them until the next time integration
int k_calthv_capture(udc_handle *h) {
  if ((int)h->fields.size() <= UDC_THL0 || !h->fields[UDC_THL0]) return 0;      // no temperature equation: dthvdz = 0
  const Geo &g = h->g;
  const size_t plane = (size_t)g.sy * g.py;
  if (!h->dthv_top) HIP_OK(hipMalloc(&h->dthv_top, sizeof(double) * 2 * plane));
  HIP_OK(hipMemcpyAsync(h->dthv_top, h->fields[UDC_THL0] + g.sz * (long)(g.nz + HZ), sizeof(double) * plane, hipMemcpyDeviceToDevice, h->stream));
  if (h->lmoist)
    HIP_OK(hipMemcpyAsync(h->dthv_top + plane, h->fields[UDC_QT0] + g.sz * (long)(g.nz + HZ), sizeof(double) * plane, hipMemcpyDeviceToDevice, h->stream));
  h->dthv_top_on = true;
  return 0;
}
int k_tke_closure(udc_handle *h) {
  const Geo &g = h->g;
  dim3 b(64, 4, 1), gr = cell_grid(g, b);
  MoistK mq;
  if (moist_inputs(h, mq)) return 1;
  PROF(h, "closure");
  hipLaunchKernelGGL(tke_closure_kernel, gr, b, 0, h->stream, g, tile_grid(g), h->m, tke_consts(h), h->fields[UDC_E120],
                     thl_or_null(h), mq, h->fields[UDC_EKM], h->fields[UDC_EKH]);
  HIP_OK(hipGetLastError());
  return 0;
}
int k_vreman_buoycorr(udc_handle *h) {
  const Geo &g = h->g;
  dim3 b(64, 4, 1), gr = cell_grid(g, b);
  MoistK mq;
  if (moist_inputs(h, mq)) return 1;
  PROF(h, "vreman_buoycorr");
  hipLaunchKernelGGL(vreman_buoycorr_kernel, gr, b, 0, h->stream, g, tile_grid(g), h->m, h->p, h->grav, h->rigc,
                     (const double *)h->fields[UDC_U0], (const double *)h->fields[UDC_V0], (const double *)h->fields[UDC_THL0], mq,
                     h->fields[UDC_EKM], h->fields[UDC_EKH]);
  HIP_OK(hipGetLastError());
  return 0;
}
int k_tke_sources(udc_handle *h) {
  const Geo &g = h->g;
  dim3 b(64, 4, 1), gr = cell_grid(g, b);
  MoistK mq;
  if (moist_inputs(h, mq)) return 1;
  PROF(h, "tke_sources");
  hipLaunchKernelGGL(tke_sources_kernel, gr, b, 0, h->stream, g, tile_grid(g), h->m, tke_consts(h), h->fields[UDC_U0],
                     h->fields[UDC_V0], h->fields[UDC_W0], h->fields[UDC_E120], thl_or_null(h), h->fields[UDC_EKM],
                     h->fields[UDC_EKH], h->fields[UDC_E12P], mq);
  HIP_OK(hipGetLastError());
  return 0;
}
int k_tke_floor(udc_handle *h) {
  const Geo &g = h->g;
  PROF(h, "tke_floor");
  hipLaunchKernelGGL(tke_floor_kernel, dim3((g.nx + 63) / 64, g.py), dim3(64), 0, h->stream, g, h->fields[UDC_E120], h->fields[UDC_E12M]);
  HIP_OK(hipGetLastError());
  return 0;
}

