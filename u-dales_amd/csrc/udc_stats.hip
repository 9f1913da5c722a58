// Statistics accumulation, the time-averaged 3-D set of tdump (src/modstatsdump.f90:514-1736, ltdump).
//
// Every tsample seconds of model time (on RK stage 3) the reference samples um, vm, wm, pres0, thlm, qtm, svm(1..4)
// and products of their interpolations to the cell edges / centres (:812-832, 862-927) into running time averages
//     X_t <- (X_t (T - ts) + sample ts) / T          T = tstatsdumpp, ts = tsamplep   (:1137-1213)
// and, every tstatsdump seconds, writes the means and the (co)variances  <ab> - <a><b>  (:1557-1645).  The host keeps the
// reference's two clocks (udcore/stats.py); the device holds the accumulators and does each sample in one sweep per
// group of fields (one read of the state, read-modify-write of the accumulators), so a device-resident run never
// has to bring the state to the host for statistics.  udc_stats_get returns an accumulator like udc_field_download.
#include "udc_internal.h"

namespace {

__device__ __forceinline__ int wrapx(int i, int nx) { return i < 0 ? i + nx : (i >= nx ? i - nx : i); }

struct MomStats { double *a[UDC_ST_MOM_N]; };
struct ScalStats { double *t, *tk, *wtk, *sq, *sgs; };      // sgs = nullptr for thl / qt (their sgs flux is not in tdump)

__device__ __forceinline__ void acc(double *p, long c, double w_old, double w_new, double ti, double sample) {
  p[c] = (p[c] * w_old + sample * w_new) * ti;
}

// levels kb .. ke+kh (device k = 0 .. nz): the top ghost level is sampled as well, as in the reference
__global__ __launch_bounds__(256) void stats_mom_kernel(Geo g, TileGrid tg, Metrics m, int nzp, double ts, double T,
                                                        const double *__restrict__ um, const double *__restrict__ vm,
                                                        const double *__restrict__ wm, const double *__restrict__ pres0, MomStats s) {
  int i, j, k;
  if (!tile_decode(g, tg, i, j, k)) return;
  (void)nzp;
  const int kf = k + 1;
  const long c = g.idx(i, j, k), cm = g.idx(wrapx(i - 1, g.nx), j, k), cp = g.idx(wrapx(i + 1, g.nx), j, k);
  const long sy = g.sy, sz = g.sz;
  const double dzf_k = m.dzf[kf], dzf_km = m.dzf[kf - 1], dzhi = m.dzhi[kf];
  const double u = um[c], v = vm[c], w = wm[c];
  // :816-831 (equidistant x: dxf = dx, dxhi = 1/dx)
  const double uik = 0.5 * dzhi * (u * dzf_km + um[c - sz] * dzf_k);
  const double wik = 0.5 * m.dxi * (w * m.dx + wm[cm] * m.dx);
  const double vjk = 0.5 * dzhi * (v * dzf_km + vm[c - sz] * dzf_k);
  const double wjk = 0.5 * (w + wm[c - sy]);
  const double uij = 0.5 * (u + um[c - sy]);
  const double vij = 0.5 * m.dxi * (v * m.dx + vm[cm] * m.dx);
  const double uc = 0.5 * (um[cp] + u);
  const double vc = 0.5 * (vm[c + sy] + v);
  // wc(ke+kh) = wc(ke+kh-1) (:827-831)
  const double wc = (k == g.nz) ? 0.5 * (w + wm[c - sz]) : 0.5 * (wm[c + sz] + w);
  const double wo = T - ts, ti = 1. / T;
  acc(s.a[UDC_ST_UWTIK], c, wo, ts, ti, wik * uik);
  acc(s.a[UDC_ST_VWTJK], c, wo, ts, ti, wjk * vjk);
  acc(s.a[UDC_ST_UVTIJ], c, wo, ts, ti, uij * vij);
  acc(s.a[UDC_ST_UUTC], c, wo, ts, ti, uc * uc);
  acc(s.a[UDC_ST_VVTC], c, wo, ts, ti, vc * vc);
  acc(s.a[UDC_ST_WWTC], c, wo, ts, ti, wc * wc);
  acc(s.a[UDC_ST_UTIK], c, wo, ts, ti, uik);
  acc(s.a[UDC_ST_WTIK], c, wo, ts, ti, wik);
  acc(s.a[UDC_ST_VTJK], c, wo, ts, ti, vjk);
  acc(s.a[UDC_ST_WTJK], c, wo, ts, ti, wjk);
  acc(s.a[UDC_ST_UTIJ], c, wo, ts, ti, uij);
  acc(s.a[UDC_ST_VTIJ], c, wo, ts, ti, vij);
  acc(s.a[UDC_ST_UMT], c, wo, ts, ti, u);
  acc(s.a[UDC_ST_VMT], c, wo, ts, ti, v);
  acc(s.a[UDC_ST_WMT], c, wo, ts, ti, w);
  acc(s.a[UDC_ST_UTC], c, wo, ts, ti, uc);
  acc(s.a[UDC_ST_VTC], c, wo, ts, ti, vc);
  acc(s.a[UDC_ST_WTC], c, wo, ts, ti, wc);
  acc(s.a[UDC_ST_PT], c, wo, ts, ti, pres0[c]);
}

// one transported field phi (thlm, qtm or svm(n)): phi_t, its half-level interpolation, w phi_k, phi^2 and, for the
// passive scalars, the resolved-scale SGS flux (:858-927, 1161-1211)
__global__ __launch_bounds__(256) void stats_scalar_kernel(Geo g, TileGrid tg, Metrics m, double ts, double T,
                                                           const double *__restrict__ wm, const double *__restrict__ phi,
                                                           const double *__restrict__ ekh, ScalStats s) {
  int i, j, k;
  if (!tile_decode(g, tg, i, j, k)) return;
  const int kf = k + 1;
  const long c = g.idx(i, j, k), sz = g.sz;
  const double dzf_k = m.dzf[kf], dzf_km = m.dzf[kf - 1];
  const double p0 = phi[c], pm = phi[c - sz];
  const double pk = 0.5 * m.dzhi[kf] * (p0 * dzf_km + pm * dzf_k);
  const double wo = T - ts, ti = 1. / T;
  acc(s.t, c, wo, ts, ti, p0);
  acc(s.tk, c, wo, ts, ti, pk);
  acc(s.wtk, c, wo, ts, ti, wm[c] * pk);
  acc(s.sq, c, wo, ts, ti, p0 * p0);
  if (s.sgs) {
    // sv1sgs is set on kb..ke only (:907-910); the level above (allocated, never written in the reference) is sampled as 0
    const double sg = (k < g.nz) ? 0.5 * (dzf_km * ekh[c] + dzf_k * ekh[c - sz]) * (p0 - pm) * m.dzh2i[kf] : 0.;
    acc(s.sgs, c, wo, ts, ti, sg);
  }
}

}  // namespace

static int stat_alloc(udc_handle *h, int id) {
  if ((int)h->stats.size() <= id) h->stats.resize(id + 1, nullptr);
  if (h->stats[id]) return 0;
  double *p = nullptr;
  HIP_OK(hipMalloc(&p, sizeof(double) * h->g.n));
  HIP_OK(hipMemsetAsync(p, 0, sizeof(double) * h->g.n, h->stream));
  h->stats[id] = p;
  return 0;
}

extern "C" int udc_stats_enable(udc_handle *h, int on) {
  if (!h) { udc_set_error("null handle"); return 1; }
  HIP_OK(hipSetDevice(h->device));
  if (udc_flush_pending(h)) return 1;
  if (!on) {
    HIP_OK(hipStreamSynchronize(h->stream));
    for (double *p : h->stats) if (p) hipFree(p);
    h->stats.clear();
    h->stats_on = false;
    return 0;
  }
  for (int q = 0; q < UDC_ST_MOM_N; ++q)
    if (stat_alloc(h, q)) return 1;
  for (int n : h->slots) {
    if (n == 14) continue;                                 // e12 has no tdump statistics
    if (n < 13 && n >= 4) continue;                        // the reference samples scalars 1..4 only
    const int base = n == 15 ? UDC_ST_THL : (n == 13 ? UDC_ST_QT : UDC_ST_SV + UDC_ST_SV_STRIDE * n);
    const int cnt = n >= 13 ? 4 : 5;
    for (int q = 0; q < cnt; ++q)
      if (stat_alloc(h, base + q)) return 1;
  }
  h->stats_on = true;
  return 0;
}

extern "C" int udc_stats_sample(udc_handle *h, double tsamplep, double tstatsdumpp) {
  if (!h) { udc_set_error("null handle"); return 1; }
  HIP_OK(hipSetDevice(h->device));
  if (udc_flush_pending(h)) return 1;
  if (!h->stats_on) { udc_set_error("udc_stats_sample: call udc_stats_enable first"); return 1; }
  if (!(tstatsdumpp > 0.) || tsamplep < 0. || tsamplep > tstatsdumpp) { udc_set_error("udc_stats_sample: need 0 <= tsamplep <= tstatsdumpp, tstatsdumpp > 0"); return 1; }
  const Geo &g = h->g;
  const TileGrid tg = tile_grid(g);
  const dim3 b(64, 4, 1), gr((unsigned)tg.tiles * (unsigned)(g.nz + 1), 1, 1);      // levels kb .. ke+kh
  // after RK stage 3 of a fused substep um, vm, wm are u0, v0, w0 (not copied): read those
  const int vo = h->um_alias ? UDC_U0 : UDC_UM;
  const double *um = h->fields[vo], *vm = h->fields[vo + 1], *wm = h->fields[vo + 2];
  MomStats ms;
  for (int q = 0; q < UDC_ST_MOM_N; ++q) ms.a[q] = h->stats[q];
  {
    PROF(h, "stats_mom");
    hipLaunchKernelGGL(stats_mom_kernel, gr, b, 0, h->stream, g, tg, h->m, g.nz + 1, tsamplep, tstatsdumpp, um, vm, wm,
                       (const double *)h->fields[UDC_PRES0], ms);
  }
  for (int n : h->slots) {
    if (n == 14 || (n < 13 && n >= 4)) continue;
    const int base = n == 15 ? UDC_ST_THL : (n == 13 ? UDC_ST_QT : UDC_ST_SV + UDC_ST_SV_STRIDE * n);
    ScalStats ss{h->stats[base], h->stats[base + 1], h->stats[base + 2], h->stats[base + 3], n < 13 ? h->stats[base + 4] : nullptr};
    PROF(h, "stats_scalar");
    hipLaunchKernelGGL(stats_scalar_kernel, gr, b, 0, h->stream, g, tg, h->m, tsamplep, tstatsdumpp, wm,
                       (const double *)h->fields[UDC_SVM + 3 * n], (const double *)h->fields[UDC_EKH], ss);
  }
  HIP_OK(hipGetLastError());
  return 0;
}

void stats_destroy(udc_handle *h) {
  for (double *p : h->stats) if (p) hipFree(p);
  h->stats.clear();
}

double *stats_ptr(udc_handle *h, int id) {
  if (id < 0 || id >= (int)h->stats.size()) return nullptr;
  return h->stats[id];
}
