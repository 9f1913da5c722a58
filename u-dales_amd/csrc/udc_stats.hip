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

// ---- xytdump: slab averages over the fluid points (avexy_ibm, src/modmpi.f90:623-664, lnan = .false.)
//
// Mask bits per cell (udc_stats_set_masks): 0 IIu, 1 IIv, 2 IIw, 3 IIc, 4 IIuw, 5 IIvw, 6 IIuv, levels kb..ke.  The host
// has already applied avexy_ibm's rule for a first level without fluid points (unmasked sum over the count of level ke) to
// the bits and counts it hands over, so the device only forms sum(var * bit) / count, or -999 where the count is zero.
enum { MB_U = 0, MB_V, MB_W, MB_C, MB_UW, MB_VW, MB_UV };
// one workgroup of 64 x 4 threads sums 64 x (4 XR) cells of a level before it reduces: XR = 1 spent more time in the nine
// cross-lane reductions than in the loads (256^3: 0.33 ms a sample; with 8 row groups per thread 0.1)
constexpr int XR = 8;
struct XytTiles { int gx, tiles; };
static inline XytTiles xyt_tiles(const Geo &g) {
  XytTiles t;
  t.gx = (g.nx + 63) / 64;
  t.tiles = t.gx * ((g.ny + 4 * XR - 1) / (4 * XR));
  return t;
}
enum { XS_U = 0, XS_V, XS_W, XS_THL, XS_QT, XS_P, XS_USGS, XS_VSGS, XS_THLSGS, XS_N };      // the running profiles
enum { XF_UPWP = 0, XF_WPTHLP, XF_VPWP, XF_UPVP, XF_UW, XF_WTHL, XF_UV, XF_VW, XF_WW, XF_THLPTHLP, XF_UPUP, XF_VPVP, XF_WPWP, XF_TKE, XF_N };
__constant__ int xs_mask[XS_N] = {MB_U, MB_V, MB_W, MB_C, MB_C, MB_C, MB_UW, MB_VW, MB_W};
__constant__ int xf_mask[XF_N] = {MB_UW, MB_W, MB_VW, MB_UV, MB_UW, MB_W, MB_UV, MB_VW, MB_W, MB_C, MB_C, MB_C, MB_C, MB_C};

template <int NP>
__device__ __forceinline__ void block_level_sums(double (&v)[NP], double *__restrict__ part, int tile, int tiles, int k, int nlev) {
  __shared__ double sw[NP][4];
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    double x = v[p];
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
    if (threadIdx.x == 0) sw[p][threadIdx.y] = x;
  }
  __syncthreads();
  if (threadIdx.x == 0 && threadIdx.y == 0)
    for (int p = 0; p < NP; ++p) part[((size_t)p * nlev + k) * tiles + tile] = (sw[p][0] + sw[p][1]) + (sw[p][2] + sw[p][3]);
}

__device__ __forceinline__ double bit(unsigned mb, int b) { return (double)((mb >> b) & 1u); }

// one sample (:816-846, 858-873, 1041-1054): um, vm, wm, thlm, qtm, pres0 and the SGS fluxes usgs, vsgs, thlsgs, masked,
// summed per tile and level
__global__ __launch_bounds__(256) void xyt_sample_kernel(Geo g, int gx, Metrics m, const double *__restrict__ um, const double *__restrict__ vm,
                                                         const double *__restrict__ wm, const double *__restrict__ thl, const double *__restrict__ qt,
                                                         const double *__restrict__ pres0, const double *__restrict__ ekm,
                                                         const double *__restrict__ ekh, const unsigned char *__restrict__ mask,
                                                         double *__restrict__ part) {
  const int tile = blockIdx.x, k = blockIdx.y, kf = k + 1;
  const int by = tile / gx, bx = tile - by * gx;
  const int i = bx * 64 + threadIdx.x;
  double v[XS_N];
#pragma unroll
  for (int p = 0; p < XS_N; ++p) v[p] = 0.;
  for (int r = 0; r < XR; ++r) {
    const int j = (by * XR + r) * 4 + threadIdx.y;
    if (i >= g.nx || j >= g.ny) continue;
    const long c = g.idx(i, j, k), cm = g.idx(wrapx(i - 1, g.nx), j, k), sy = g.sy, sz = g.sz;
    const unsigned mb = mask ? mask[((size_t)k * g.ny + j) * g.nx + i] : 0x7fu;
    const double dzf_k = m.dzf[kf], dzf_km = m.dzf[kf - 1], dzhi = m.dzhi[kf], dzhiq = m.dzhiq[kf];
    const double u = um[c], vv = vm[c], w = wm[c];
    double emom = (dzf_km * (ekm[c] * m.dx + ekm[cm] * m.dx) + dzf_k * (ekm[c - sz] * m.dx + ekm[cm - sz] * m.dx)) * m.dxi * dzhiq;
    const double usgs = emom * ((u - um[c - sz]) * dzhi + (w - wm[cm]) * m.dxi);
    emom = (dzf_km * (ekm[c] + ekm[c - sy]) + dzf_k * (ekm[c - sz] + ekm[c - sy - sz])) * dzhiq;
    const double vsgs = emom * ((vv - vm[c - sz]) * dzhi + (w - wm[c - sy]) * m.dyi);
    v[XS_U] += u * bit(mb, MB_U);
    v[XS_V] += vv * bit(mb, MB_V);
    v[XS_W] += w * bit(mb, MB_W);
    v[XS_P] += pres0[c] * bit(mb, MB_C);
    v[XS_USGS] += usgs * bit(mb, MB_UW);
    v[XS_VSGS] += vsgs * bit(mb, MB_VW);
    if (thl) {
      const double t0 = thl[c], tm = thl[c - sz];
      v[XS_THL] += t0 * bit(mb, MB_C);
      v[XS_THLSGS] += 0.5 * (dzf_km * ekh[c] + dzf_k * ekh[c - sz]) * (t0 - tm) * m.dzh2i[kf] * bit(mb, MB_W);
    }
    if (qt) v[XS_QT] += qt[c] * bit(mb, MB_C);
  }
  block_level_sums<XS_N>(v, part, tile, gridDim.x, k, gridDim.y);
}

struct XytAcc { const double *a[UDC_ST_MOM_N]; const double *thlt, *thltk, *wthltk, *thlthlt; };

// the table's second half (:1407-1431): slab averages of products and (co)variances of the time-averaged 3-D fields
__global__ __launch_bounds__(256) void xyt_final_kernel(Geo g, int gx, XytAcc s, const unsigned char *__restrict__ mask, double *__restrict__ part) {
  const int tile = blockIdx.x, k = blockIdx.y;
  const int by = tile / gx, bx = tile - by * gx;
  const int i = bx * 64 + threadIdx.x;
  double v[XF_N];
#pragma unroll
  for (int p = 0; p < XF_N; ++p) v[p] = 0.;
  for (int r = 0; r < XR; ++r) {
    const int j = (by * XR + r) * 4 + threadIdx.y;
    if (i >= g.nx || j >= g.ny) continue;
    const long c = g.idx(i, j, k);
    const unsigned mb = mask ? mask[((size_t)k * g.ny + j) * g.nx + i] : 0x7fu;
    const double utik = s.a[UDC_ST_UTIK][c], wtik = s.a[UDC_ST_WTIK][c], vtjk = s.a[UDC_ST_VTJK][c], wtjk = s.a[UDC_ST_WTJK][c];
    const double utij = s.a[UDC_ST_UTIJ][c], vtij = s.a[UDC_ST_VTIJ][c], wmt = s.a[UDC_ST_WMT][c];
    const double utc = s.a[UDC_ST_UTC][c], vtc = s.a[UDC_ST_VTC][c], wtc = s.a[UDC_ST_WTC][c];
    const double up2 = s.a[UDC_ST_UUTC][c] - utc * utc, vp2 = s.a[UDC_ST_VVTC][c] - vtc * vtc, wp2 = s.a[UDC_ST_WWTC][c] - wtc * wtc;
    v[XF_UW] += utik * wtik * bit(mb, MB_UW);
    v[XF_VW] += vtjk * wtjk * bit(mb, MB_VW);
    v[XF_WW] += wmt * wmt * bit(mb, MB_W);
    v[XF_UV] += utij * vtij * bit(mb, MB_UV);
    v[XF_UPWP] += (s.a[UDC_ST_UWTIK][c] - utik * wtik) * bit(mb, MB_UW);
    v[XF_VPWP] += (s.a[UDC_ST_VWTJK][c] - vtjk * wtjk) * bit(mb, MB_VW);
    v[XF_UPVP] += (s.a[UDC_ST_UVTIJ][c] - utij * vtij) * bit(mb, MB_UV);
    v[XF_UPUP] += up2 * bit(mb, MB_C);
    v[XF_VPVP] += vp2 * bit(mb, MB_C);
    v[XF_WPWP] += wp2 * bit(mb, MB_C);
    v[XF_TKE] += 0.5 * ((wp2 + vp2) + up2) * bit(mb, MB_C);
    if (s.thlt) {
      const double thltk = s.thltk[c], thlt = s.thlt[c];
      v[XF_WTHL] += wmt * thltk * bit(mb, MB_W);
      v[XF_WPTHLP] += (s.wthltk[c] - wmt * thltk) * bit(mb, MB_W);
      v[XF_THLPTHLP] += (s.thlthlt[c] - thlt * thlt) * bit(mb, MB_C);
    }
  }
  block_level_sums<XF_N>(v, part, tile, gridDim.x, k, gridDim.y);
}

// stage 2: sum the tiles' partial sums; one workgroup per (quantity, level)
__global__ __launch_bounds__(256) void xyt_tiles_kernel(int tiles, const double *__restrict__ part, double *__restrict__ S) {
  __shared__ double sw[4];
  const int q = blockIdx.x;
  double v = 0.;
  for (int t = threadIdx.x; t < tiles; t += 256) v += part[(size_t)q * tiles + t];
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  if ((threadIdx.x & 63) == 0) sw[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) S[q] = (sw[0] + sw[1]) + (sw[2] + sw[3]);
}

// sums -> masked averages -> running time averages of the nine profiles (:1088-1099)
__global__ void xyt_running_kernel(int nz, const double *__restrict__ S, const double *__restrict__ cnt, double ts, double T, double *__restrict__ prof) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= XS_N * nz) return;
  const int p = q / nz, k = q - p * nz;
  const double n = cnt[xs_mask[p] * nz + k];
  const double xy = n > 0. ? S[q] / n : -999.;
  prof[q] = (prof[q] * (T - ts) + xy * ts) * (1. / T);
}

// the table in the order of varsxyt (:1437-1460), levels kb..ke
__global__ void xyt_table_kernel(int nz, const double *__restrict__ S, const double *__restrict__ cnt, const double *__restrict__ prof,
                                 double *__restrict__ table) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= UDC_XYT_N * nz) return;
  const int row = q / nz, k = q - row * nz;
  // rows 0-5, 15-17: running profiles; the others: final slab averages
  const int from_prof[UDC_XYT_N] = {XS_U, XS_V, XS_W, XS_THL, XS_QT, XS_P, -1, -1, -1, -1, -1, -1, -1, -1, -1, XS_USGS, XS_THLSGS, XS_VSGS, -1, -1, -1, -1, -1};
  const int from_fin[UDC_XYT_N] = {-1, -1, -1, -1, -1, -1, XF_UPWP, XF_WPTHLP, XF_VPWP, XF_UPVP, XF_UW, XF_WTHL, XF_UV, XF_VW, XF_WW, -1, -1, -1,
                                   XF_THLPTHLP, XF_UPUP, XF_VPVP, XF_WPWP, XF_TKE};
  if (from_prof[row] >= 0) { table[q] = prof[from_prof[row] * nz + k]; return; }
  const int f = from_fin[row];
  const double n = cnt[xf_mask[f] * nz + k];
  table[q] = n > 0. ? S[f * nz + k] / n : -999.;
}

// ---- ytdump: averages over y (avey_ibm, src/modmpi.f90: masked sum over y / the column's fluid count, -999 where it is zero)
enum { YS_U = 0, YS_V, YS_W, YS_THL, YS_QT, YS_SV1, YS_SV2, YS_SV3, YS_USGS, YS_WSGS, YS_THLSGS, YS_QTSGS, YS_SV1SGS, YS_SV2SGS, YS_SV3SGS, YS_N };
enum { YF_UPWP = 0, YF_WPTHLP, YF_WPQTP, YF_WPSV1P, YF_WPSV2P, YF_WPSV3P, YF_UW, YF_WTHL, YF_WQT, YF_WSV1, YF_WSV2, YF_WSV3, YF_UPUP, YF_WPWP,
       YF_THLPTHLP, YF_QTPQTP, YF_SV1P, YF_SV2P, YF_SV3P, YF_N };
__constant__ int ys_mask[YS_N] = {MB_U, MB_V, MB_W, MB_C, MB_C, MB_C, MB_C, MB_C, MB_UW, MB_W, MB_W, MB_W, MB_W, MB_W, MB_W};
__constant__ int yf_mask[YF_N] = {MB_UW, MB_W, MB_W, MB_W, MB_W, MB_W, MB_UW, MB_W, MB_W, MB_W, MB_W, MB_W, MB_C, MB_C, MB_C, MB_C, MB_C, MB_C, MB_C};

struct YForced { int f[7]; };
// mask bits of a cell with the levels the caller filled for avexy_ibm's rule emptied again (avey_ibm has no such rule)
__device__ __forceinline__ unsigned ybits(const unsigned char *mask, const Geo &g, const YForced &F, int i, int j, int k) {
  unsigned mb = mask ? mask[((size_t)k * g.ny + j) * g.nx + i] : 0x7fu;
  if (k == 0 && mask)
    for (int q = 0; q < 7; ++q) if (F.f[q]) mb &= ~(1u << q);
  return mb;
}

template <int NP>
__device__ __forceinline__ void ysum_store(double (&v)[NP], double *__restrict__ S, int i, int k, int nx, int nz, bool in) {
  __shared__ double sw[NP][4][64];
#pragma unroll
  for (int p = 0; p < NP; ++p) sw[p][threadIdx.y][threadIdx.x] = v[p];
  __syncthreads();
  if (threadIdx.y == 0 && in)
    for (int p = 0; p < NP; ++p)
      S[((size_t)p * nz + k) * nx + i] = (sw[p][0][threadIdx.x] + sw[p][1][threadIdx.x]) + (sw[p][2][threadIdx.x] + sw[p][3][threadIdx.x]);
}

__global__ __launch_bounds__(256) void yt_count_kernel(Geo g, const unsigned char *__restrict__ mask, YForced F, double *__restrict__ S) {
  const int i = blockIdx.x * 64 + threadIdx.x, k = blockIdx.y;
  double v[5] = {0., 0., 0., 0., 0.};
  const bool in = i < g.nx;
  if (in)
    for (int j = threadIdx.y; j < g.ny; j += 4) {
      const unsigned mb = ybits(mask, g, F, i, j, k);
      for (int q = 0; q < 5; ++q) v[q] += bit(mb, q);
    }
  ysum_store<5>(v, S, i, k, g.nx, g.nz, in);
}

struct YtFields { const double *um, *vm, *wm, *thl, *qt, *sv[3], *ekm, *ekh; };

__global__ __launch_bounds__(256) void yt_sample_kernel(Geo g, Metrics m, YtFields f, const unsigned char *__restrict__ mask, YForced F,
                                                        double *__restrict__ S) {
  const int i = blockIdx.x * 64 + threadIdx.x, k = blockIdx.y, kf = k + 1;
  double v[YS_N];
#pragma unroll
  for (int p = 0; p < YS_N; ++p) v[p] = 0.;
  const bool in = i < g.nx;
  if (in) {
    const double dzf_k = m.dzf[kf], dzf_km = m.dzf[kf - 1], dzhi = m.dzhi[kf], dzhiq = m.dzhiq[kf], dzh2i = m.dzh2i[kf];
    const double dzfi = m.dzfi[kf], dzfi_m = m.dzfi[kf - 1];
    const int im = wrapx(i - 1, g.nx);
    for (int j = threadIdx.y; j < g.ny; j += 4) {
      const long c = g.idx(i, j, k), cm = g.idx(im, j, k), sz = g.sz;
      const unsigned mb = ybits(mask, g, F, i, j, k);
      const double u = f.um[c], w = f.wm[c];
      const double emom = (dzf_km * (f.ekm[c] * m.dx + f.ekm[cm] * m.dx) + dzf_k * (f.ekm[c - sz] * m.dx + f.ekm[cm - sz] * m.dx)) * m.dxi * dzhiq;
      const double usgs = emom * ((u - f.um[c - sz]) * dzhi + (w - f.wm[cm]) * m.dxi);
      const double wsgs = (f.ekm[c] * (f.wm[c + sz] - w) * dzfi - f.ekm[c - sz] * (w - f.wm[c - sz]) * dzfi_m) * 2. * dzhi;
      const double eh = 0.5 * (dzf_km * f.ekh[c] + dzf_k * f.ekh[c - sz]);
      v[YS_U] += u * bit(mb, MB_U);
      v[YS_V] += f.vm[c] * bit(mb, MB_V);
      v[YS_W] += w * bit(mb, MB_W);
      v[YS_USGS] += usgs * bit(mb, MB_UW);
      v[YS_WSGS] += wsgs * bit(mb, MB_W);
      if (f.thl) { const double t0 = f.thl[c]; v[YS_THL] += t0 * bit(mb, MB_C); v[YS_THLSGS] += eh * (t0 - f.thl[c - sz]) * dzh2i * bit(mb, MB_W); }
      if (f.qt) { const double q0 = f.qt[c]; v[YS_QT] += q0 * bit(mb, MB_C); v[YS_QTSGS] += eh * (q0 - f.qt[c - sz]) * dzh2i * bit(mb, MB_W); }
#pragma unroll
      for (int n = 0; n < 3; ++n)
        if (f.sv[n]) { const double s0 = f.sv[n][c]; v[YS_SV1 + n] += s0 * bit(mb, MB_C); v[YS_SV1SGS + n] += eh * (s0 - f.sv[n][c - sz]) * dzh2i * bit(mb, MB_W); }
    }
  }
  ysum_store<YS_N>(v, S, i, k, g.nx, g.nz, in);
}

struct YtAcc { const double *a[UDC_ST_MOM_N]; const double *thl[4], *qt[4], *sv[3][4]; };

__global__ __launch_bounds__(256) void yt_final_kernel(Geo g, YtAcc s, const unsigned char *__restrict__ mask, YForced F, double *__restrict__ S) {
  const int i = blockIdx.x * 64 + threadIdx.x, k = blockIdx.y;
  double v[YF_N];
#pragma unroll
  for (int p = 0; p < YF_N; ++p) v[p] = 0.;
  const bool in = i < g.nx;
  if (in)
    for (int j = threadIdx.y; j < g.ny; j += 4) {
      const long c = g.idx(i, j, k);
      const unsigned mb = ybits(mask, g, F, i, j, k);
      const double utik = s.a[UDC_ST_UTIK][c], wtik = s.a[UDC_ST_WTIK][c], wmt = s.a[UDC_ST_WMT][c];
      const double utc = s.a[UDC_ST_UTC][c], wtc = s.a[UDC_ST_WTC][c];
      v[YF_UPWP] += (s.a[UDC_ST_UWTIK][c] - utik * wtik) * bit(mb, MB_UW);
      v[YF_UW] += utik * wtik * bit(mb, MB_UW);
      v[YF_UPUP] += (s.a[UDC_ST_UUTC][c] - utc * utc) * bit(mb, MB_C);
      v[YF_WPWP] += (s.a[UDC_ST_WWTC][c] - wtc * wtc) * bit(mb, MB_C);
      // scalar-like fields: [0] t, [1] tk, [2] wtk, [3] sq
      if (s.thl[0]) {
        v[YF_WPTHLP] += (s.thl[2][c] - wmt * s.thl[1][c]) * bit(mb, MB_W); v[YF_WTHL] += wmt * s.thl[1][c] * bit(mb, MB_W);
        v[YF_THLPTHLP] += (s.thl[3][c] - s.thl[0][c] * s.thl[0][c]) * bit(mb, MB_C);
      }
      if (s.qt[0]) {
        v[YF_WPQTP] += (s.qt[2][c] - wmt * s.qt[1][c]) * bit(mb, MB_W); v[YF_WQT] += wmt * s.qt[1][c] * bit(mb, MB_W);
        v[YF_QTPQTP] += (s.qt[3][c] - s.qt[0][c] * s.qt[0][c]) * bit(mb, MB_C);
      }
#pragma unroll
      for (int n = 0; n < 3; ++n)
        if (s.sv[n][0]) {
          v[YF_WPSV1P + n] += (s.sv[n][2][c] - wmt * s.sv[n][1][c]) * bit(mb, MB_W); v[YF_WSV1 + n] += wmt * s.sv[n][1][c] * bit(mb, MB_W);
          v[YF_SV1P + n] += (s.sv[n][3][c] - s.sv[n][0][c] * s.sv[n][0][c]) * bit(mb, MB_C);
        }
    }
  ysum_store<YF_N>(v, S, i, k, g.nx, g.nz, in);
}

__global__ void yt_running_kernel(int n, const double *__restrict__ S, const double *__restrict__ cnt, double ts, double T, double *__restrict__ prof) {
  const long q = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= (long)YS_N * n) return;
  const int p = (int)(q / n);
  const long r = q - (long)p * n;
  const double c = cnt[(long)ys_mask[p] * n + r];
  const double ya = c > 0. ? S[q] / c : -999.;
  prof[q] = (prof[q] * (T - ts) + ya * ts) * (1. / T);
}

// the table in the order of varsyt (:1513-1551)
__global__ void yt_table_kernel(int n, const double *__restrict__ S, const double *__restrict__ cnt, const double *__restrict__ prof,
                                double *__restrict__ table) {
  const long q = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= (long)UDC_YT_N * n) return;
  const int row = (int)(q / n);
  const long r = q - (long)row * n;
  if (row < 8) { table[q] = prof[(long)row * n + r]; return; }                          // uyt .. sca3yt
  if (row >= 27) { table[q] = prof[(long)(YS_USGS + row - 27) * n + r]; return; }         // usgsyt .. sca3sgsyt
  const int f = row - 8;                                                                  // upwpyt .. sca3psca3pyt
  const double c = cnt[(long)yf_mask[f] * n + r];
  table[q] = c > 0. ? S[(long)f * n + r] / c : -999.;
}

// ---- xydump / ydump: the instantaneous slab / column averages of ONE sample, written every tsample (src/modstatsdump.f90:1002-1081,
// 1294-1349).  Next to the averages the running profiles are built from (xyt_sample_kernel / yt_sample_kernel above) they hold the
// products of the edge interpolations of that sample: uik wik, vjk wjk, wm thlk and the factors on their own.
enum { XI_UW = 0, XI_UIK, XI_WIK, XI_VW, XI_VJK, XI_WJK, XI_WTHL, XI_THLK, XI_N };
__constant__ int xi_mask[XI_N] = {MB_UW, MB_UW, MB_UW, MB_VW, MB_VW, MB_VW, MB_W, MB_W};
struct EdgeVals { double uik, wik, vjk, wjk, thlk, w; };
__device__ __forceinline__ EdgeVals edge_vals(const Geo &g, const Metrics &m, const double *__restrict__ um, const double *__restrict__ vm,
                                              const double *__restrict__ wm, const double *__restrict__ thl, int i, int j, int k) {
  const int kf = k + 1;
  const long c = g.idx(i, j, k), cm = g.idx(wrapx(i - 1, g.nx), j, k), sy = g.sy, sz = g.sz;
  const double dzf_k = m.dzf[kf], dzf_km = m.dzf[kf - 1], dzhi = m.dzhi[kf];
  const double w = wm[c];
  EdgeVals e;      // :816-826, 862 (as stats_mom_kernel / stats_scalar_kernel)
  e.uik = 0.5 * dzhi * (um[c] * dzf_km + um[c - sz] * dzf_k);
  e.wik = 0.5 * m.dxi * (w * m.dx + wm[cm] * m.dx);
  e.vjk = 0.5 * dzhi * (vm[c] * dzf_km + vm[c - sz] * dzf_k);
  e.wjk = 0.5 * (w + wm[c - sy]);
  e.thlk = thl ? 0.5 * dzhi * (thl[c] * dzf_km + thl[c - sz] * dzf_k) : 0.;
  e.w = w;
  return e;
}
__global__ __launch_bounds__(256) void xy_sample_kernel(Geo g, int gx, Metrics m, const double *__restrict__ um, const double *__restrict__ vm,
                                                        const double *__restrict__ wm, const double *__restrict__ thl,
                                                        const unsigned char *__restrict__ mask, double *__restrict__ part) {
  const int tile = blockIdx.x, k = blockIdx.y;
  const int by = tile / gx, bx = tile - by * gx;
  const int i = bx * 64 + threadIdx.x;
  double v[XI_N];
#pragma unroll
  for (int p = 0; p < XI_N; ++p) v[p] = 0.;
  for (int r = 0; r < XR; ++r) {
    const int j = (by * XR + r) * 4 + threadIdx.y;
    if (i >= g.nx || j >= g.ny) continue;
    const unsigned mb = mask ? mask[((size_t)k * g.ny + j) * g.nx + i] : 0x7fu;
    const EdgeVals e = edge_vals(g, m, um, vm, wm, thl, i, j, k);
    v[XI_UW] += e.uik * e.wik * bit(mb, MB_UW); v[XI_UIK] += e.uik * bit(mb, MB_UW); v[XI_WIK] += e.wik * bit(mb, MB_UW);
    v[XI_VW] += e.vjk * e.wjk * bit(mb, MB_VW); v[XI_VJK] += e.vjk * bit(mb, MB_VW); v[XI_WJK] += e.wjk * bit(mb, MB_VW);
    v[XI_WTHL] += e.w * e.thlk * bit(mb, MB_W); v[XI_THLK] += e.thlk * bit(mb, MB_W);
  }
  block_level_sums<XI_N>(v, part, tile, gridDim.x, k, gridDim.y);
}
// the 15 profiles of xydump in the order of varsxy (:1330-1344).  S9: the sample's nine masked level sums (XS_*, avexy_ibm with
// lnan = .false.: the caller's first-level rule applies), SI: the eight above (lnan = .true.: a level without fluid points is -999,
// first level included -- `forced` names the masks whose first level the caller filled).
struct XyForced { int f[7]; };
__global__ void xy_table_kernel(int nz, const double *__restrict__ S9, const double *__restrict__ SI, const double *__restrict__ cnt, XyForced F,
                                int has_thl, double *__restrict__ table) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= nz) return;
  auto a9 = [&](int p) { const double n = cnt[xs_mask[p] * nz + k]; return n > 0. ? S9[p * nz + k] / n : -999.; };
  auto ai = [&](int p) {
    const int mk = xi_mask[p];
    const double n = (k == 0 && F.f[mk]) ? 0. : cnt[mk * nz + k];
    return n > 0. ? SI[p * nz + k] / n : -999.;
  };
  // (without a temperature the reference never averages wthlxyk / thlxyk: they stay at the zero they were set to, :1060, 1072-1079)
  const double uw = ai(XI_UW), vw = ai(XI_VW), wthl = has_thl ? ai(XI_WTHL) : 0., wxy = a9(XS_W);
  double *t = table + k;
  t[0 * nz] = a9(XS_U); t[1 * nz] = a9(XS_V); t[2 * nz] = wxy; t[3 * nz] = a9(XS_THL); t[4 * nz] = a9(XS_QT); t[5 * nz] = a9(XS_P);
  t[6 * nz] = uw - ai(XI_UIK) * ai(XI_WIK);
  t[7 * nz] = has_thl ? wthl - wxy * ai(XI_THLK) : 0.;
  t[8 * nz] = vw - ai(XI_VJK) * ai(XI_WJK);
  t[9 * nz] = a9(XS_USGS); t[10 * nz] = a9(XS_THLSGS); t[11 * nz] = a9(XS_VSGS);
  t[12 * nz] = uw; t[13 * nz] = wthl; t[14 * nz] = vw;
}

enum { YI_UW = 0, YI_UIK, YI_WIK, YI_WTHL, YI_THLK, YI_N };
__constant__ int yi_mask[YI_N] = {MB_UW, MB_UW, MB_UW, MB_W, MB_W};
// column counts live in yt_cnt as [IIu, IIv, IIw, IIc, IIuw][nz][nx] (yt_count_kernel sums bits 0..4)
__global__ __launch_bounds__(256) void y_sample_kernel(Geo g, Metrics m, const double *__restrict__ um, const double *__restrict__ vm,
                                                       const double *__restrict__ wm, const double *__restrict__ thl,
                                                       const unsigned char *__restrict__ mask, YForced F, double *__restrict__ S) {
  const int i = blockIdx.x * 64 + threadIdx.x, k = blockIdx.y;
  double v[YI_N];
#pragma unroll
  for (int p = 0; p < YI_N; ++p) v[p] = 0.;
  const bool in = i < g.nx;
  if (in)
    for (int j = threadIdx.y; j < g.ny; j += 4) {
      const unsigned mb = ybits(mask, g, F, i, j, k);
      const EdgeVals e = edge_vals(g, m, um, vm, wm, thl, i, j, k);
      v[YI_UW] += e.uik * e.wik * bit(mb, MB_UW); v[YI_UIK] += e.uik * bit(mb, MB_UW); v[YI_WIK] += e.wik * bit(mb, MB_UW);
      v[YI_WTHL] += e.w * e.thlk * bit(mb, MB_W); v[YI_THLK] += e.thlk * bit(mb, MB_W);
    }
  ysum_store<YI_N>(v, S, i, k, g.nx, g.nz, in);
}
// the 14 x-z fields of ydump in the order of varsy (:1303-1316); SY: the sample's YS_* column sums, SI: the five above
__global__ void y_table_kernel(int n, const double *__restrict__ SY, const double *__restrict__ SI, const double *__restrict__ cnt,
                               double *__restrict__ table) {
  const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  auto ay = [&](int p) { const double c = cnt[(long)ys_mask[p] * n + r]; return c > 0. ? SY[(long)p * n + r] / c : -999.; };
  auto ai = [&](int p) { const double c = cnt[(long)yi_mask[p] * n + r]; return c > 0. ? SI[(long)p * n + r] / c : -999.; };
  const double uw = ai(YI_UW), wthl = ai(YI_WTHL), wy = ay(YS_W);
  const double cuw = cnt[(long)MB_UW * n + r], cw = cnt[(long)MB_W * n + r];
  double *t = table + r;
  t[0L * n] = ay(YS_U); t[1L * n] = ay(YS_V); t[2L * n] = wy; t[3L * n] = ay(YS_THL); t[4L * n] = ay(YS_QT);
  t[5L * n] = ay(YS_SV1); t[6L * n] = ay(YS_SV2); t[7L * n] = ay(YS_SV3);
  t[8L * n] = cuw > 0. ? uw - ai(YI_UIK) * ai(YI_WIK) : -999.;      // where (IIuwt == 0) upwpyik = -999 (:1026-1028)
  t[9L * n] = cw > 0. ? wthl - wy * ai(YI_THLK) : -999.;            // where (IIwt == 0) wpthlpyk = -999 (:1020-1022)
  t[10L * n] = ay(YS_USGS); t[11L * n] = ay(YS_THLSGS);
  t[12L * n] = uw; t[13L * n] = wthl;
}

}  // namespace

static int yt_counts(udc_handle *h) {
  const Geo &g = h->g;
  YForced F;
  for (int q = 0; q < 7; ++q) F.f[q] = h->yt_forced[q];
  hipLaunchKernelGGL(yt_count_kernel, dim3((unsigned)((g.nx + 63) / 64), (unsigned)g.nz), dim3(64, 4), 0, h->stream, g,
                     (const unsigned char *)h->st_mask, F, h->yt_cnt);
  HIP_OK(hipGetLastError());
  return comm_allreduce(h, h->yt_cnt, 5 * g.nz * g.nx, 1);
}

static int xyt_scratch(udc_handle *h, int nq) {
  const size_t need = (size_t)xyt_tiles(h->g).tiles * h->g.nz * nq;
  if (h->st_part_cap < need) {
    if (h->st_part) HIP_OK(hipFree(h->st_part));
    HIP_OK(hipMalloc(&h->st_part, sizeof(double) * need));
    h->st_part_cap = need;
  }
  return 0;
}

static int stat_alloc(udc_handle *h, int id) {
  if ((int)h->stats.size() <= id) h->stats.resize(id + 1, nullptr);
  if (h->stats[id]) return 0;
  double *p = nullptr;
  HIP_OK(hipMalloc(&p, sizeof(double) * h->g.n));
  HIP_OK(hipMemsetAsync(p, 0, sizeof(double) * h->g.n, h->stream));
  h->stats[id] = p;
  return 0;
}

// open x boundaries (udc_xopen.hip): the slab sums must leave the two ghost columns of the device row out -- the mask array, which is
// indexed by the device's columns, always exists there and carries zeros in them; `bits` (or null: no obstacles) is indexed by the deck's
// columns, [nz][ny][itot]
static int stats_mask_upload(udc_handle *h, const unsigned char *bits) {
  const Geo &g = h->g;
  const size_t n = (size_t)g.nz * g.ny * g.nx;
  if (!h->st_mask) HIP_OK(hipMalloc(&h->st_mask, n));
  if (!g.xg) { HIP_OK(hipMemcpy(h->st_mask, bits, n, hipMemcpyHostToDevice)); return 0; }
  std::vector<unsigned char> m(n, 0);
  const int ni = g.nx - 2 * g.xg;
  for (size_t r = 0; r < (size_t)g.nz * g.ny; ++r)
    for (int i = 0; i < ni; ++i) m[r * g.nx + g.xg + i] = bits ? bits[r * ni + i] : (unsigned char)0x7f;
  HIP_OK(hipMemcpy(h->st_mask, m.data(), n, hipMemcpyHostToDevice));
  return 0;
}

// a table [rows][row of the device] to the host as [rows][itot]: the ghost columns of an open-x handle stay behind
static int xz_table_download(udc_handle *h, const double *dev, size_t rows, double *host) {
  const Geo &g = h->g;
  const size_t ni = (size_t)(g.nx - 2 * g.xg);
  HIP_OK(hipMemcpy2DAsync(host, sizeof(double) * ni, dev + g.xg, sizeof(double) * g.nx, sizeof(double) * ni, rows, hipMemcpyDeviceToHost, h->stream));
  HIP_OK(hipStreamSynchronize(h->stream));
  return 0;
}

extern "C" int udc_stats_enable(udc_handle *h, int on) {
  if (!h) { udc_set_error("null handle"); return 1; }
  HIP_OK(hipSetDevice(h->device));
  if (udc_flush_pending(h)) return 1;
  if (!on) {
    HIP_OK(hipStreamSynchronize(h->stream));
    stats_destroy(h);
    h->stats_on = false;
    return 0;
  }
  if (on & 8) on |= 2;      // xydump's profiles are built on xytdump's sample sums, ydump's on ytdump's
  if (on & 16) on |= 4;
  if (on & 2) {        // xytdump: the nine running profiles; masks default to "no obstacles" until udc_stats_set_masks
    const int nz = h->g.nz;
    if (!h->st_prof) {
      HIP_OK(hipMalloc(&h->st_prof, sizeof(double) * XS_N * nz));
      HIP_OK(hipMalloc(&h->st_sum, sizeof(double) * XF_N * nz));
      HIP_OK(hipMalloc(&h->st_table, sizeof(double) * UDC_XYT_N * nz));
      HIP_OK(hipMalloc(&h->st_cnt, sizeof(double) * 7 * nz));
      std::vector<double> c((size_t)7 * nz, (double)(h->g.nx - 2 * h->g.xg) * (double)h->cfg.jtot);
      HIP_OK(hipMemcpy(h->st_cnt, c.data(), sizeof(double) * c.size(), hipMemcpyHostToDevice));
      if (h->xg && !h->st_mask) {
        HIP_OK(hipStreamSynchronize(h->stream));
        if (stats_mask_upload(h, nullptr)) return 1;
      }
    }
    HIP_OK(hipMemsetAsync(h->st_prof, 0, sizeof(double) * XS_N * nz, h->stream));
    h->xyt_on = true;
    if ((on & 8) && !h->xy_table) {
      HIP_OK(hipMalloc(&h->xy_table, sizeof(double) * UDC_XY_N * nz));
      HIP_OK(hipMalloc(&h->xy_sum, sizeof(double) * XI_N * nz));
      HIP_OK(hipMemsetAsync(h->xy_table, 0, sizeof(double) * UDC_XY_N * nz, h->stream));
    }
    h->xy_on = (on & 8) != 0;
  }
  if (on & 4) {        // ytdump: running y-averages and the column counts of the masks
    const size_t n = (size_t)h->g.nz * h->g.nx;
    if (h->xg && !h->st_mask) {      // (open x: the ghost columns count nothing)
      HIP_OK(hipStreamSynchronize(h->stream));
      if (stats_mask_upload(h, nullptr)) return 1;
    }
    if (!h->yt_prof) {
      HIP_OK(hipMalloc(&h->yt_prof, sizeof(double) * YS_N * n));
      HIP_OK(hipMalloc(&h->yt_cnt, sizeof(double) * 5 * n));
      HIP_OK(hipMalloc(&h->yt_sum, sizeof(double) * YF_N * n));
      HIP_OK(hipMalloc(&h->yt_table, sizeof(double) * UDC_YT_N * n));
    }
    HIP_OK(hipMemsetAsync(h->yt_prof, 0, sizeof(double) * YS_N * n, h->stream));
    h->yt_on = true;
    if ((on & 16) && !h->y_table) {
      HIP_OK(hipMalloc(&h->y_table, sizeof(double) * UDC_Y_N * n));
      HIP_OK(hipMalloc(&h->y_sum, sizeof(double) * YI_N * n));
      HIP_OK(hipMemsetAsync(h->y_table, 0, sizeof(double) * UDC_Y_N * n, h->stream));
    }
    h->y_on = (on & 16) != 0;
    if (yt_counts(h)) return 1;
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
  // (tsamplep may exceed tstatsdumpp: the reference's clocks allow it -- a sample interval that straddles a dump, with an adaptive
  //  time step -- and its running average then weighs the old value negatively, src/modstatsdump.f90:1137-1213; same formula here)
  if (!(tstatsdumpp > 0.) || tsamplep < 0.) { udc_set_error("udc_stats_sample: need tsamplep >= 0, tstatsdumpp > 0"); return 1; }
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
  if (h->xyt_on) {
    const int nz = g.nz;
    if (xyt_scratch(h, XS_N)) return 1;
    int thl = -1, qt = -1;
    for (int n : h->slots) { if (n == 15) thl = UDC_SVM + 3 * n; if (n == 13) qt = UDC_SVM + 3 * n; }
    PROF(h, "stats_xyt");
    const XytTiles xt = xyt_tiles(g);
    hipLaunchKernelGGL(xyt_sample_kernel, dim3((unsigned)xt.tiles, (unsigned)nz), dim3(64, 4), 0, h->stream, g, xt.gx, h->m, um, vm, wm,
                       thl >= 0 ? (const double *)h->fields[thl] : nullptr, qt >= 0 ? (const double *)h->fields[qt] : nullptr,
                       (const double *)h->fields[UDC_PRES0], (const double *)h->fields[UDC_EKM], (const double *)h->fields[UDC_EKH],
                       (const unsigned char *)h->st_mask, h->st_part);
    hipLaunchKernelGGL(xyt_tiles_kernel, dim3((unsigned)(XS_N * nz)), dim3(256), 0, h->stream, xt.tiles, (const double *)h->st_part, h->st_sum);
    HIP_OK(hipGetLastError());
    if (comm_allreduce(h, h->st_sum, XS_N * nz, 1)) return 1;
    hipLaunchKernelGGL(xyt_running_kernel, dim3((unsigned)((XS_N * nz + 255) / 256)), dim3(256), 0, h->stream, nz, (const double *)h->st_sum,
                       (const double *)h->st_cnt, tsamplep, tstatsdumpp, h->st_prof);
    HIP_OK(hipGetLastError());
    if (h->xy_on) {      // xydump: this sample's own profiles (st_sum still holds its nine level sums)
      hipLaunchKernelGGL(xy_sample_kernel, dim3((unsigned)xt.tiles, (unsigned)nz), dim3(64, 4), 0, h->stream, g, xt.gx, h->m, um, vm, wm,
                         thl >= 0 ? (const double *)h->fields[thl] : nullptr, (const unsigned char *)h->st_mask, h->st_part);
      hipLaunchKernelGGL(xyt_tiles_kernel, dim3((unsigned)(XI_N * nz)), dim3(256), 0, h->stream, xt.tiles, (const double *)h->st_part, h->xy_sum);
      HIP_OK(hipGetLastError());
      if (comm_allreduce(h, h->xy_sum, XI_N * nz, 1)) return 1;
      XyForced XF;
      for (int q = 0; q < 7; ++q) XF.f[q] = h->st_mask ? h->yt_forced[q] : 0;
      hipLaunchKernelGGL(xy_table_kernel, dim3((unsigned)((nz + 63) / 64)), dim3(64), 0, h->stream, nz, (const double *)h->st_sum,
                         (const double *)h->xy_sum, (const double *)h->st_cnt, XF, thl >= 0 ? 1 : 0, h->xy_table);
      HIP_OK(hipGetLastError());
    }
  }
  if (h->yt_on) {
    const size_t n = (size_t)g.nz * g.nx;
    YtFields f{um, vm, wm, nullptr, nullptr, {nullptr, nullptr, nullptr}, (const double *)h->fields[UDC_EKM], (const double *)h->fields[UDC_EKH]};
    for (int s : h->slots) {
      if (s == 15) f.thl = h->fields[UDC_SVM + 3 * s];
      if (s == 13) f.qt = h->fields[UDC_SVM + 3 * s];
      if (s < 3) f.sv[s] = h->fields[UDC_SVM + 3 * s];
    }
    YForced F;
    for (int q = 0; q < 7; ++q) F.f[q] = h->yt_forced[q];
    PROF(h, "stats_yt");
    hipLaunchKernelGGL(yt_sample_kernel, dim3((unsigned)((g.nx + 63) / 64), (unsigned)g.nz), dim3(64, 4), 0, h->stream, g, h->m, f,
                       (const unsigned char *)h->st_mask, F, h->yt_sum);
    HIP_OK(hipGetLastError());
    if (comm_allreduce(h, h->yt_sum, (int)(YS_N * n), 1)) return 1;
    hipLaunchKernelGGL(yt_running_kernel, dim3((unsigned)((YS_N * n + 255) / 256)), dim3(256), 0, h->stream, (int)n, (const double *)h->yt_sum,
                       (const double *)h->yt_cnt, tsamplep, tstatsdumpp, h->yt_prof);
    HIP_OK(hipGetLastError());
    if (h->y_on) {       // ydump: this sample's own x-z fields (yt_sum still holds its column sums)
      hipLaunchKernelGGL(y_sample_kernel, dim3((unsigned)((g.nx + 63) / 64), (unsigned)g.nz), dim3(64, 4), 0, h->stream, g, h->m, um, vm, wm,
                         f.thl, (const unsigned char *)h->st_mask, F, h->y_sum);
      HIP_OK(hipGetLastError());
      if (comm_allreduce(h, h->y_sum, (int)(YI_N * n), 1)) return 1;
      hipLaunchKernelGGL(y_table_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, (int)n, (const double *)h->yt_sum,
                         (const double *)h->y_sum, (const double *)h->yt_cnt, h->y_table);
      HIP_OK(hipGetLastError());
    }
  }
  return 0;
}

extern "C" int udc_stats_set_forced(udc_handle *h, const int *forced) {
  if (!h) { udc_set_error("null handle"); return 1; }
  HIP_OK(hipSetDevice(h->device));
  if (udc_flush_pending(h)) return 1;
  for (int q = 0; q < 7; ++q) h->yt_forced[q] = forced && forced[q] ? 1 : 0;
  if (h->yt_on && yt_counts(h)) return 1;
  return 0;
}

extern "C" int udc_stats_yt(udc_handle *h, double *table) {
  if (!h) { udc_set_error("null handle"); return 1; }
  HIP_OK(hipSetDevice(h->device));
  if (udc_flush_pending(h)) return 1;
  if (!h->yt_on || !h->stats_on) { udc_set_error("udc_stats_yt: enable the yt statistics first (udc_stats_enable with bit 4)"); return 1; }
  if (!table) { udc_set_error("udc_stats_yt: null table"); return 1; }
  const Geo &g = h->g;
  const size_t n = (size_t)g.nz * g.nx;
  auto st = [&](int id) -> const double * { return id < (int)h->stats.size() ? h->stats[id] : nullptr; };
  YtAcc s;
  for (int q = 0; q < UDC_ST_MOM_N; ++q) s.a[q] = h->stats[q];
  for (int q = 0; q < 4; ++q) {
    s.thl[q] = st(UDC_ST_THL + q); s.qt[q] = st(UDC_ST_QT + q);
    for (int m = 0; m < 3; ++m) s.sv[m][q] = st(UDC_ST_SV + UDC_ST_SV_STRIDE * m + q);
  }
  YForced F;
  for (int q = 0; q < 7; ++q) F.f[q] = h->yt_forced[q];
  hipLaunchKernelGGL(yt_final_kernel, dim3((unsigned)((g.nx + 63) / 64), (unsigned)g.nz), dim3(64, 4), 0, h->stream, g, s,
                     (const unsigned char *)h->st_mask, F, h->yt_sum);
  HIP_OK(hipGetLastError());
  if (comm_allreduce(h, h->yt_sum, (int)(YF_N * n), 1)) return 1;
  hipLaunchKernelGGL(yt_table_kernel, dim3((unsigned)((UDC_YT_N * n + 255) / 256)), dim3(256), 0, h->stream, (int)n, (const double *)h->yt_sum,
                     (const double *)h->yt_cnt, (const double *)h->yt_prof, h->yt_table);
  HIP_OK(hipGetLastError());
  return xz_table_download(h, h->yt_table, (size_t)UDC_YT_N * g.nz, table);
}

extern "C" int udc_stats_set_masks(udc_handle *h, const unsigned char *bits, const int *counts) {
  if (!h) { udc_set_error("null handle"); return 1; }
  HIP_OK(hipSetDevice(h->device));
  if (udc_flush_pending(h)) return 1;
  if (!h->xyt_on && !h->yt_on) { udc_set_error("udc_stats_set_masks: enable the xyt or yt statistics first (udc_stats_enable with bit 2 or 4)"); return 1; }
  if (!counts) { udc_set_error("udc_stats_set_masks: counts missing"); return 1; }
  const Geo &g = h->g;
  HIP_OK(hipStreamSynchronize(h->stream));
  if (bits || g.xg) {      // (bits: [nz][ny][itot] -- the deck's columns)
    if (stats_mask_upload(h, bits)) return 1;
  } else if (h->st_mask) {
    HIP_OK(hipFree(h->st_mask));
    h->st_mask = nullptr;
  }
  std::vector<double> c((size_t)7 * g.nz);
  for (size_t q = 0; q < c.size(); ++q) {
    if (counts[q] < 0) { udc_set_error("udc_stats_set_masks: negative count"); return 1; }
    c[q] = (double)counts[q];
  }
  if (h->st_cnt) HIP_OK(hipMemcpy(h->st_cnt, c.data(), sizeof(double) * c.size(), hipMemcpyHostToDevice));
  if (h->yt_on && yt_counts(h)) return 1;
  return 0;
}

extern "C" int udc_stats_xyt(udc_handle *h, double *table) {
  if (!h) { udc_set_error("null handle"); return 1; }
  HIP_OK(hipSetDevice(h->device));
  if (udc_flush_pending(h)) return 1;
  if (!h->xyt_on || !h->stats_on) { udc_set_error("udc_stats_xyt: enable the xyt statistics first (udc_stats_enable with bit 2)"); return 1; }
  if (!table) { udc_set_error("udc_stats_xyt: null table"); return 1; }
  const Geo &g = h->g;
  const XytTiles tg = xyt_tiles(g);
  const int nz = g.nz;
  if (xyt_scratch(h, XF_N)) return 1;
  XytAcc s;
  for (int q = 0; q < UDC_ST_MOM_N; ++q) s.a[q] = h->stats[q];
  const bool thl = (int)h->stats.size() > UDC_ST_THL + 3 && h->stats[UDC_ST_THL];
  s.thlt = thl ? h->stats[UDC_ST_THL] : nullptr;
  s.thltk = thl ? h->stats[UDC_ST_THL + 1] : nullptr;
  s.wthltk = thl ? h->stats[UDC_ST_THL + 2] : nullptr;
  s.thlthlt = thl ? h->stats[UDC_ST_THL + 3] : nullptr;
  hipLaunchKernelGGL(xyt_final_kernel, dim3((unsigned)tg.tiles, (unsigned)nz), dim3(64, 4), 0, h->stream, g, tg.gx, s,
                     (const unsigned char *)h->st_mask, h->st_part);
  hipLaunchKernelGGL(xyt_tiles_kernel, dim3((unsigned)(XF_N * nz)), dim3(256), 0, h->stream, tg.tiles, (const double *)h->st_part, h->st_sum);
  HIP_OK(hipGetLastError());
  if (comm_allreduce(h, h->st_sum, XF_N * nz, 1)) return 1;
  hipLaunchKernelGGL(xyt_table_kernel, dim3((unsigned)((UDC_XYT_N * nz + 255) / 256)), dim3(256), 0, h->stream, nz, (const double *)h->st_sum,
                     (const double *)h->st_cnt, (const double *)h->st_prof, h->st_table);
  HIP_OK(hipGetLastError());
  HIP_OK(hipMemcpyAsync(table, h->st_table, sizeof(double) * UDC_XYT_N * nz, hipMemcpyDeviceToHost, h->stream));
  HIP_OK(hipStreamSynchronize(h->stream));
  return 0;
}

extern "C" int udc_stats_xy(udc_handle *h, double *table) {
  if (!h) { udc_set_error("null handle"); return 1; }
  HIP_OK(hipSetDevice(h->device));
  if (udc_flush_pending(h)) return 1;
  if (!h->xy_on || !h->stats_on) { udc_set_error("udc_stats_xy: enable xydump's profiles first (udc_stats_enable with bit 8)"); return 1; }
  if (!table) { udc_set_error("udc_stats_xy: null table"); return 1; }
  HIP_OK(hipMemcpyAsync(table, h->xy_table, sizeof(double) * UDC_XY_N * h->g.nz, hipMemcpyDeviceToHost, h->stream));
  HIP_OK(hipStreamSynchronize(h->stream));
  return 0;
}

extern "C" int udc_stats_y(udc_handle *h, double *table) {
  if (!h) { udc_set_error("null handle"); return 1; }
  HIP_OK(hipSetDevice(h->device));
  if (udc_flush_pending(h)) return 1;
  if (!h->y_on || !h->stats_on) { udc_set_error("udc_stats_y: enable ydump's fields first (udc_stats_enable with bit 16)"); return 1; }
  if (!table) { udc_set_error("udc_stats_y: null table"); return 1; }
  return xz_table_download(h, h->y_table, (size_t)UDC_Y_N * h->g.nz, table);
}

void stats_destroy(udc_handle *h) {
  for (double *p : h->stats) if (p) hipFree(p);
  h->stats.clear();
  for (double **p : {&h->st_cnt, &h->st_prof, &h->st_part, &h->st_sum, &h->st_table, &h->yt_prof, &h->yt_cnt, &h->yt_sum, &h->yt_table,
                     &h->xy_table, &h->xy_sum, &h->y_table, &h->y_sum})
    if (*p) { hipFree(*p); *p = nullptr; }
  if (h->st_mask) { hipFree(h->st_mask); h->st_mask = nullptr; }
  h->st_part_cap = 0;
  h->xyt_on = false;
  h->yt_on = false;
  h->xy_on = h->y_on = false;
}

double *stats_ptr(udc_handle *h, int id) {
  if (id < 0 || id >= (int)h->stats.size()) return nullptr;
  return h->stats[id];
}
