// Transported scalars: kappa-scheme (flux-limited, kappa = 1/3) or 2nd-order central advection and eddy diffusion.
// Reference: advecc_kappa + rlim (src/modadvection.f90:316-421), advecc_2nd (:103-155), diffc (src/modsubgrid.f90:540-623).
// The reference builds each direction's fluxes into two 3-D temporaries and adds them to the
// tendency (6 zero-fills, 3 whole-array adds); here every cell evaluates its six face values
// in registers and accumulates in the reference's order ((cp + upper) + lower per direction).
#include "udc_internal.h"

namespace {

__device__ __forceinline__ int wrap(int i, int nx) { return i < 0 ? i + nx : (i >= nx ? i - nx : i); }

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

// ADV: 0 = none, 1 = kappa, 2 = cd2.  FRESH: the tendency is known to be zero on entry (fused substep) -> not read.
template <int ADV, bool DIFF, bool LES, bool FRESH>
__global__ __launch_bounds__(256) void scalar_kernel(Geo g, TileGrid tg, Metrics m, double cekh, double dfac, const double *__restrict__ u,
    const double *__restrict__ v, const double *__restrict__ w, const double *__restrict__ ekh,
    const double *__restrict__ c, double *__restrict__ cp) {
  int i, j, k;
  const bool inside_ = tile_decode(g, tg, i, j, k);
  if (!inside_) return;
  const int kf = k + 1;
  const long r0 = g.idx(0, j, k);
  const long sy = g.sy, sz = g.sz;
  const long o = r0 + i;
  const long xm1 = r0 + wrap(i - 1, g.nx), xm2 = r0 + wrap(i - 2, g.nx);
  const long xp1 = r0 + wrap(i + 1, g.nx), xp2 = r0 + wrap(i + 2, g.nx);
  const double c0 = c[o];
  const double cxm1 = c[xm1], cxp1 = c[xp1], cym1 = c[o - sy], cyp1 = c[o + sy], czm1 = c[o - sz], czp1 = c[o + sz];
  double t = FRESH ? 0. : cp[o];
  if (ADV == 2) {
    // advecc_2nd, src/modadvection.f90:127-133 and :148-151 (two statements, same order)
    const double kdzf = m.dzf[kf], kdzfm = m.dzf[kf - 1], kdzfp = m.dzf[kf + 1];
    t = t - ((u[xp1] * (cxp1 + c0) - u[o] * (cxm1 + c0)) * m.dxi5
           + (v[o + sy] * (cyp1 + c0) - v[o] * (cym1 + c0)) * m.dyi5);
    t = t - (w[o + sz] * (czp1 * kdzf + c0 * kdzfp) * m.dzhi[kf + 1]
           - w[o] * (czm1 * kdzf + c0 * kdzfm) * m.dzhi[kf]) * m.dzfi5[kf];
  }
  if (ADV == 1) {
    const double cxm2 = c[xm2], cxp2 = c[xp2];
    const double cym2 = c[o - 2 * sy], cyp2 = c[o + 2 * sy];
    const double czm2 = c[o - 2 * sz], czp2 = c[o + 2 * sz];
    const double dxi = m.dxi, dx = m.dx, dyi = m.dyi;
    {  // x: faces i (low) and i+1 (high); dxhci = dxi, dxfc = dx, dxfci = dxi on the uniform grid
      const double ul = u[o], uh = u[xp1];
      const double fl = face(ul, cxm2, cxm1, c0, cxp1, dxi, dxi, dxi, dx);
      const double fh = face(uh, cxm1, c0, cxp1, cxp2, dxi, dxi, dxi, dx);
      t = (t + (-fh * uh * dxi)) + fl * ul * dxi;
    }
    {  // y (no stretching: d's are plain differences, df = 1)
      const double vl = v[o], vh = v[o + sy];
      const double fl = face(vl, cym2, cym1, c0, cyp1, 1., 1., 1., 1.);
      const double fh = face(vh, cym1, c0, cyp1, cyp2, 1., 1., 1., 1.);
      t = (t + (-fh * vh * dyi)) + fl * vl * dyi;
    }
    {  // z: faces kb+1..ke+1 only (no flux through the floor, src/modadvection.f90:385)
      const int nzp1 = g.nz + 1;
      const double hkm1 = m.dzhi[kf - 1 < 1 ? 1 : kf - 1], hk = m.dzhi[kf], hkp1 = m.dzhi[kf + 1];
      const double hkp2 = m.dzhi[kf + 2 > nzp1 ? nzp1 : kf + 2];
      const double wl = w[o], wh = w[o + sz];
      const double dzfci = m.dzfi[kf];
      const double fh = face(wh, czm1, c0, czp1, czp2, hk, hkp1, hkp2, m.dzf[kf + 1]);
      const double upper = -fh * wh * dzfci;
      double lower = 0.;
      if (k >= 1) {
        const double fl = face(wl, czm2, czm1, c0, czp1, hkm1, hk, hkp1, m.dzf[kf]);
        lower = fl * wl * dzfci;
      }
      t = (t + upper) + lower;
    }
  }
  if (DIFF) {
    const double dzf_k = m.dzf[kf], dzf_km = m.dzf[kf - 1], dzf_kp = m.dzf[kf + 1];
    if (LES) {
      const double e0 = ekh[o], exm = ekh[xm1], exp_ = ekh[xp1], eym = ekh[o - sy], eyp = ekh[o + sy],
                   ezm = ekh[o - sz], ezp = ekh[o + sz];
      // dfac = 0.5 with ekh (diffc, src/modsubgrid.f90:569-584) or 1.0 with ekm (diffe, :649-663)
      t = t + dfac * (((exp_ + e0) * (cxp1 - c0) - (e0 + exm) * (c0 - cxm1)) * m.dx2i
                   + ((eyp + e0) * (cyp1 - c0) - (e0 + eym) * (c0 - cym1)) * m.dy2i
                   + ((dzf_kp * e0 + dzf_k * ezp) * (czp1 - c0) * m.dzh2i[kf + 1]
                    - (dzf_km * e0 + dzf_k * ezm) * (c0 - czm1) * m.dzh2i[kf]) * m.dzfi[kf]);
    } else {
      t = t + ((cekh * (cxp1 - c0) - cekh * (c0 - cxm1)) * m.dx2i
             + (cekh * (cyp1 - c0) - cekh * (c0 - cym1)) * m.dy2i
             + (cekh * (czp1 - c0) * m.dzhi[kf + 1] - cekh * (c0 - czm1) * m.dzhi[kf]) * m.dzfi[kf]);
    }
  }
  cp[o] = t;
}

inline dim3 cell_grid(const Geo &g, dim3 b) {
  (void)b;
  return dim3((unsigned)tile_grid(g).tiles * (unsigned)g.nz, 1, 1);
}

// cp(i,j,k) += src(k): thlpcar in forces, src/modforces.f90:104-110
__global__ __launch_bounds__(256) void level_source_kernel(Geo g, TileGrid tg, const double *__restrict__ src, double *__restrict__ cp) {
  int i, j, k;
  if (!tile_decode(g, tg, i, j, k)) return;
  const long c = g.idx(i, j, k);
  cp[c] = cp[c] + src[k + 1];
}

// fluxtop with a non-zero flux (src/modboundary.f90:1494-1507) on c0 and cm over the padded y extent
__global__ void top_flux_kernel(Geo g, Metrics m, const double *__restrict__ ekh, double *__restrict__ c0, double *__restrict__ cm,
                                double flux) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int j = (int)blockIdx.y - HY;
  if (i >= g.nx) return;
  const long top = g.idx(i, j, g.nz - 1), ghost = top + g.sz;
  const int ke = g.nz;
  const double d = m.dzh[ke + 1] * flux / (m.dzhi[ke + 1] * (0.5 * (m.dzf[ke] * ekh[ghost] + m.dzf[ke + 1] * ekh[top])));
  c0[ghost] = c0[top] + d;
  cm[ghost] = cm[top] + d;
}

// Buoyancy (forces with lbuoyancy, src/modforces.f90:73-84; dry air: thv0h = thl0h, src/modthermodynamics.f90:208):
//   wp(k) += grav (thl0h(i,j,k) - thvh(k)) / thvh(k),  k = kb+1..ke,
//   thl0h(k) = (thl0(k) dzf(k-1) + thl0(k-1) dzf(k)) / (2 dzh(k))          (calc_halflev, :518-524)
//   thvh(k)  = slab average of thl0h(k)                                     (thermodynamics :76, avexy_ibm)
__device__ __forceinline__ double thl_half(const Geo &g, const Metrics &m, const double *__restrict__ t, long c, int k) {
  const int kf = k + 1;
  return (t[c] * m.dzf[kf - 1] + t[c - g.sz] * m.dzf[kf]) / (2 * m.dzh[kf]);
}
// stage 1: one workgroup per (xy tile, level) -> part[level * tiles + tile]; stage 2: one workgroup per level
__global__ __launch_bounds__(256) void levelsum_kernel(Geo g, Metrics m, int gx, const double *__restrict__ thl, double *__restrict__ part) {
  __shared__ double sw[4];
  const int tile = blockIdx.x, k = blockIdx.y;
  const int by = tile / gx, bx = tile - by * gx;
  const int i = bx * 64 + threadIdx.x, j = by * 4 + threadIdx.y;
  double v = 0.;
  if (i < g.nx && j < g.ny && k >= 1) v = thl_half(g, m, thl, g.idx(i, j, k), k);
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
__global__ __launch_bounds__(256) void buoyancy_kernel(Geo g, TileGrid tg, Metrics m, const double *__restrict__ thl,
                                                        const double *__restrict__ S, double cnt, double grav, double *__restrict__ wp) {
  int i, j, k;
  if (!tile_decode(g, tg, i, j, k) || k < 1) return;
  const long c = g.idx(i, j, k);
  const double thvh = S[k] / cnt;
  wp[c] = wp[c] + grav * (thl_half(g, m, thl, c, k) - thvh) / thvh;
}

// ---- one-equation closure -------------------------------------------------------------------------
struct TkeK { double cm, cn, ch1, ch2, ce1, ce2, grav_thvs, numol, prandtlmoli; int ldelta; };
// dthvdz of calthv for dry air (src/modthermodynamics.f90:208-222, eps1 clamp :224-232): thl may be null (neutral)
__device__ __forceinline__ double dthvdz_at(const Geo &g, const Metrics &m, const double *__restrict__ thl, long c, int k) {
  const double eps1 = 1e-10;
  double d = 0.;
  if (thl && k >= 1) d = (thl[c + g.sz] - thl[c - g.sz]) / (m.dzh[k + 2] + m.dzh[k + 1]);
  if (fabs(d) < eps1) d = copysign(eps1, d);
  return d;
}
__device__ __forceinline__ double tke_zlt(const TkeK &t, double delta, double e, double dthvdz) {
  if (t.ldelta || dthvdz <= 0) return delta;
  return fmin(delta, t.cn * e / sqrt(t.grav_thvs * fabs(dthvdz)));
}
// closure, loneeqn branch: src/modsubgrid.f90:363-400 (damp = 1)
__global__ __launch_bounds__(256) void tke_closure_kernel(Geo g, TileGrid tg, Metrics m, TkeK t, const double *__restrict__ e12,
    const double *__restrict__ thl, double *__restrict__ ekm, double *__restrict__ ekh) {
  int i, j, k;
  if (!tile_decode(g, tg, i, j, k)) return;
  const long c = g.idx(i, j, k);
  const double delta = m.delta[k + 1];
  const double e = e12[c];
  const double dth = dthvdz_at(g, m, thl, c, k);
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
    const double *__restrict__ ekm, const double *__restrict__ ekh, double *__restrict__ e12p) {
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
  const double dth = dthvdz_at(g, m, thl, c, k);
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

// plain per-level slab sums (stage 1; levelsum_final_kernel is stage 2): levels k = 0..nlev-1 (device), i.e. 1..nlev
__global__ __launch_bounds__(256) void levelsum_plain_kernel(Geo g, int gx, const double *__restrict__ f, double *__restrict__ part) {
  __shared__ double sw[4];
  const int tile = blockIdx.x, k = blockIdx.y;
  const int by = tile / gx, bx = tile - by * gx;
  const int i = bx * 64 + threadIdx.x, j = by * 4 + threadIdx.y;
  double v = 0.;
  if (i < g.nx && j < g.ny) v = f[g.idx(i, j, k)];
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

int k_slab_average(udc_handle *h, int field, double *avg_host, int n) {
  const Geo &g = h->g;
  if (n < 1 || n > g.nz + 1) { udc_set_error("udc_slab_average: 1 <= n <= ktot+1"); return 1; }
  if (field < 0 || field >= (int)h->fields.size() || !h->fields[field]) { udc_set_error("udc_slab_average: unknown field %d", field); return 1; }
  const TileGrid tg = tile_grid(g);
  const size_t need = (size_t)tg.tiles * (g.nz + 1);
  if (h->lev_cap < need) {
    if (h->lev_part) HIP_OK(hipFree(h->lev_part));
    HIP_OK(hipMalloc(&h->lev_part, sizeof(double) * need));
    h->lev_cap = need;
  }
  if (!h->lev_sum) HIP_OK(hipMalloc(&h->lev_sum, sizeof(double) * (g.nz + 2)));
  hipLaunchKernelGGL(levelsum_plain_kernel, dim3((unsigned)tg.tiles, (unsigned)n), dim3(64, 4), 0, h->stream, g, tg.gx,
                     (const double *)h->fields[field], h->lev_part);
  hipLaunchKernelGGL(levelsum_final_kernel, dim3((unsigned)n), dim3(256), 0, h->stream, tg.tiles, h->lev_part, h->lev_sum);
  HIP_OK(hipGetLastError());
  if (comm_allreduce(h, h->lev_sum, n, 1)) return 1;
  HIP_OK(hipMemcpyAsync(h->red_host, h->lev_sum, sizeof(double) * n, hipMemcpyDeviceToHost, h->stream));
  HIP_OK(hipStreamSynchronize(h->stream));
  const double cnt = (double)g.nx * (double)h->cfg.jtot;
  for (int k = 0; k < n; ++k) avg_host[k] = h->red_host[k] / cnt;
  return 0;
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

static TkeK tke_consts(udc_handle *h) {
  return TkeK{h->tke.cm, h->tke.cn, h->tke.ch1, h->tke.ch2, h->tke.ce1, h->tke.ce2, h->tke.grav / h->tke.thvs,
              h->p.numol, h->p.prandtlmoli, h->tke.ldelta};
}
static const double *thl_or_null(udc_handle *h) {
  return ((int)h->fields.size() > UDC_THL0 && h->fields[UDC_THL0]) ? h->fields[UDC_THL0] : nullptr;
}
int k_tke_closure(udc_handle *h) {
  const Geo &g = h->g;
  dim3 b(64, 4, 1), gr = cell_grid(g, b);
  PROF(h, "closure");
  hipLaunchKernelGGL(tke_closure_kernel, gr, b, 0, h->stream, g, tile_grid(g), h->m, tke_consts(h), h->fields[UDC_E120],
                     thl_or_null(h), h->fields[UDC_EKM], h->fields[UDC_EKH]);
  HIP_OK(hipGetLastError());
  return 0;
}
int k_tke_sources(udc_handle *h) {
  const Geo &g = h->g;
  dim3 b(64, 4, 1), gr = cell_grid(g, b);
  PROF(h, "tke_sources");
  hipLaunchKernelGGL(tke_sources_kernel, gr, b, 0, h->stream, g, tile_grid(g), h->m, tke_consts(h), h->fields[UDC_U0],
                     h->fields[UDC_V0], h->fields[UDC_W0], h->fields[UDC_E120], thl_or_null(h), h->fields[UDC_EKM],
                     h->fields[UDC_EKH], h->fields[UDC_E12P]);
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

int k_buoyancy(udc_handle *h) {
  const Geo &g = h->g;
  if (!h->lbuoyancy) return 0;
  if ((int)h->fields.size() <= UDC_THL0 || !h->fields[UDC_THL0]) { udc_set_error("buoyancy needs the temperature equation (udc_set_tempeq)"); return 1; }
  const TileGrid tg = tile_grid(g);
  const size_t need = (size_t)tg.tiles * g.nz;
  if (h->lev_cap < need) {
    if (h->lev_part) HIP_OK(hipFree(h->lev_part));
    HIP_OK(hipMalloc(&h->lev_part, sizeof(double) * need));
    h->lev_cap = need;
  }
  if (!h->lev_sum) HIP_OK(hipMalloc(&h->lev_sum, sizeof(double) * (g.nz + 2)));
  PROF(h, "buoyancy");
  const double *thl = h->fields[UDC_THL0];
  hipLaunchKernelGGL(levelsum_kernel, dim3((unsigned)tg.tiles, (unsigned)g.nz), dim3(64, 4), 0, h->stream, g, h->m, tg.gx, thl, h->lev_part);
  hipLaunchKernelGGL(levelsum_final_kernel, dim3((unsigned)g.nz), dim3(256), 0, h->stream, tg.tiles, h->lev_part, h->lev_sum);
  HIP_OK(hipGetLastError());
  if (comm_allreduce(h, h->lev_sum, g.nz, 1)) return 1;       // avexy_ibm's MPI_ALLREDUCE over the slabs
  dim3 b(64, 4, 1), gr = cell_grid(g, b);
  hipLaunchKernelGGL(buoyancy_kernel, gr, b, 0, h->stream, g, tg, h->m, thl, h->lev_sum, (double)g.nx * (double)h->cfg.jtot,
                     h->grav, h->fields[UDC_WP]);
  HIP_OK(hipGetLastError());
  return 0;
}

int k_level_source(udc_handle *h, int slot, const double *src) {
  const Geo &g = h->g;
  dim3 b(64, 4, 1), gr = cell_grid(g, b);
  PROF(h, "level_source");
  hipLaunchKernelGGL(level_source_kernel, gr, b, 0, h->stream, g, tile_grid(g), src, h->fields[UDC_SVP + 3 * slot]);
  HIP_OK(hipGetLastError());
  return 0;
}

int k_scalar_top_flux(udc_handle *h) {
  const Geo &g = h->g;
  for (int n : h->slots) {
    if (h->slot[n].top != 1) continue;
    PROF(h, "top_flux");
    hipLaunchKernelGGL(top_flux_kernel, dim3((g.nx + 63) / 64, g.py), dim3(64), 0, h->stream, g, h->m, h->fields[UDC_EKH],
                       h->fields[UDC_SV0 + 3 * n], h->fields[UDC_SVM + 3 * n], h->slot[n].topval);
    HIP_OK(hipGetLastError());
  }
  return 0;
}

static int launch_scalar(udc_handle *h, int n, bool adv, bool diff, bool fresh = false) {
  const Geo &g = h->g;
  dim3 b(64, 4, 1), gr = cell_grid(g, b);
  const double cekh = h->p.numol * h->p.prandtlmoli;
  const double *u = h->fields[UDC_U0], *v = h->fields[UDC_V0], *w = h->fields[UDC_W0];
  const bool tke = h->slot[n].tke;
  const double dfac = tke ? 1.0 : 0.5;
  const double *ekh = h->fields[tke ? UDC_EKM : UDC_EKH], *c = h->fields[UDC_SV0 + 3 * n];
  double *cp = h->fields[UDC_SVP + 3 * n];
  const bool les = h->p.sgs != UDC_SGS_DNS;
  const bool cd2 = h->slot[n].adv == 2;
#define LS(A, D, L)                                                                                     \
  do {                                                                                                  \
    PROF(h, "scalar_" #A #D #L);                                                                        \
    if (fresh) hipLaunchKernelGGL((scalar_kernel<A, D, L, true>), gr, b, 0, h->stream, g, tile_grid(g), h->m, cekh, dfac, u, v, w, ekh, c, cp); \
    else hipLaunchKernelGGL((scalar_kernel<A, D, L, false>), gr, b, 0, h->stream, g, tile_grid(g), h->m, cekh, dfac, u, v, w, ekh, c, cp); \
  } while (0)
  if (adv && diff) {
    if (cd2) { if (les) LS(2, true, true); else LS(2, true, false); }
    else { if (les) LS(1, true, true); else LS(1, true, false); }
  } else if (adv) { if (cd2) LS(2, false, true); else LS(1, false, true); }
  else if (diff) { if (les) LS(0, true, true); else LS(0, true, false); }
#undef LS
  HIP_OK(hipGetLastError());
  return 0;
}

int k_scalar_adv(udc_handle *h, int n) { return launch_scalar(h, n, true, false); }
int k_scalar_diff(udc_handle *h, int n) { return launch_scalar(h, n, false, true); }
int k_scalar_fused(udc_handle *h, int n, bool fresh) { return launch_scalar(h, n, true, true, fresh); }
