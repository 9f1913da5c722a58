// Transported scalars: kappa-scheme (flux-limited, kappa = 1/3) or 2nd-order central advection and eddy diffusion.
// Reference: advecc_kappa + rlim (src/modadvection.f90:316-421), advecc_2nd (:103-155), diffc (src/modsubgrid.f90:540-623).
// The reference builds each direction's fluxes into two 3-D temporaries and adds them to the
// tendency (6 zero-fills, 3 whole-array adds); here every cell evaluates its six face values
// in registers and accumulates in the reference's order ((cp + upper) + lower per direction).
#include "udc_internal.h"
#include <algorithm>
#include "udc_scalar_arith.h"

namespace {

__device__ __forceinline__ int wrap(int i, int nx) { return i < 0 ? i + nx : (i >= nx ? i - nx : i); }

// direct-load accessor: every operand straight from global memory (x wraps by index)
struct GlobalAcc {
  const double *c_, *e_;
  long o, sy, sz, xm1, xm2, xp1, xp2;      // xm*/xp* are full offsets of the row's wrapped x neighbours
  __device__ __forceinline__ long off(int di, int dj, int dk) const {
    const long base = di == 0 ? o : (di == -1 ? xm1 : (di == 1 ? xp1 : (di == -2 ? xm2 : xp2)));
    return base + dj * sy + dk * sz;
  }
  __device__ __forceinline__ double c(int di, int dj, int dk) const { return c_[off(di, dj, dk)]; }
  __device__ __forceinline__ double e(int di, int dj, int dk) const { return e_[off(di, dj, dk)]; }
};

// ADV: 0 = none, 1 = kappa, 2 = cd2.  FRESH: the tendency is known to be zero on entry (fused substep) -> not read.
template <int ADV, bool DIFF, bool LES, bool FRESH>
__global__ __launch_bounds__(256) void scalar_kernel(Geo g, TileGrid tg, Metrics m, double cekh, double dfac, const double *__restrict__ u,
    const double *__restrict__ v, const double *__restrict__ w, const double *__restrict__ ekh,
    const double *__restrict__ c, double *__restrict__ cp, int gh) {
  int i, j, k;
  const bool inside_ = tile_decode(g, tg, i, j, k);
  if (!inside_) return;
  const long r0 = g.idx(0, j, k);
  const long o = r0 + i;
  const GlobalAcc A{c, ekh, o, (long)g.sy, g.sz, r0 + wrap(i - 1, g.nx), r0 + wrap(i - 2, g.nx), r0 + wrap(i + 1, g.nx), r0 + wrap(i + 2, g.nx)};
  double ul = 0., uh = 0., vl = 0., vh = 0., wl = 0., wh = 0.;
  if (ADV) { ul = u[o]; uh = u[A.xp1]; vl = v[o]; vh = v[o + g.sy]; wl = w[o]; wh = w[o + g.sz]; }
  cp[o] = scalar_tend<ADV, DIFF, LES>(A, m, ScalMetGlobal{m, k + 1, g.nz}, k, g.nz, FRESH ? 0. : cp[o], ul, uh, vl, vh, wl, wh, cekh, dfac, gh);
}

// ---- BCxs = 2: the scalars see an inlet at the low-x and a convective outlet at the high-x side of the domain while the flow
// stays periodic (src/modboundary.f90:844-861 xsi_profile, :983-996 xso_convective; `halos` then leaves their x ghosts alone,
// :99).  The fields keep the periodic, ghost-free layout; after the periodic sweep the two columns at either end get the
// difference between the tendency with the true ghost values and the one with the wrapped ones.
//   west ghosts (xsi_profile, m = 1, 2):  c(ib-1) = 2 prof(k) - c(ib),   c(ib-2) = 2 prof(k) - c(ib-1) = c(ib)
//   east ghosts:  c(ie+1) evolves by the convective condition at every `boundary`, c(ie+2) keeps what the start gave it
struct BcxAcc {
  const double *c_, *e_, *east, *prof;      // east: [2][pz][py]
  Geo g;
  int i, j, k;
  __device__ __forceinline__ double c(int di, int dj, int dk) const {
    const int ii = i + di, jj = j + dj, kk = k + dk;
    if (ii >= 0 && ii < g.nx) return c_[g.idx(ii, jj, kk)];
    if (ii == -1) return 2. * prof[kk + 1] - c_[g.idx(0, jj, kk)];
    if (ii == -2) return c_[g.idx(0, jj, kk)];
    const long q = (long)(kk + HZ) * g.py + (jj + HY);
    return ii == g.nx ? east[q] : east[(long)g.pz * g.py + q];
  }
  __device__ __forceinline__ double e(int di, int dj, int dk) const {      // ekh: periodic in x as before
    const int ii = i + di;
    return e_[g.idx(ii < 0 ? ii + g.nx : (ii >= g.nx ? ii - g.nx : ii), j + dj, k + dk)];
  }
};

template <int ADV, bool DIFF, bool LES>
__global__ __launch_bounds__(64) void scalar_bcx_edge_kernel(Geo g, Metrics m, double cekh, double dfac, const double *__restrict__ u,
    const double *__restrict__ v, const double *__restrict__ w, const double *__restrict__ ekh, const double *__restrict__ c,
    const double *__restrict__ east, const double *__restrict__ prof, double *__restrict__ cp, int gh) {
  const int col = blockIdx.x & 3, j = (blockIdx.x >> 2) * 64 + threadIdx.x, k = blockIdx.y;
  if (j >= g.ny) return;
  const int i = col < 2 ? col : g.nx - 4 + col;      // 0, 1, nx-2, nx-1
  if (g.nx < 4 || i < 0) return;
  const long r0 = g.idx(0, j, k), o = r0 + i;
  const GlobalAcc A{c, ekh, o, (long)g.sy, g.sz, r0 + wrap(i - 1, g.nx), r0 + wrap(i - 2, g.nx), r0 + wrap(i + 1, g.nx), r0 + wrap(i + 2, g.nx)};
  const BcxAcc B{c, ekh, east, prof, g, i, j, k};
  double ul = 0., uh = 0., vl = 0., vh = 0., wl = 0., wh = 0.;
  if (ADV) { ul = u[o]; uh = u[A.xp1]; vl = v[o]; vh = v[o + g.sy]; wl = w[o]; wh = w[o + g.sz]; }
  const ScalMetGlobal lm{m, k + 1, g.nz};
  const double periodic = scalar_tend<ADV, DIFF, LES>(A, m, lm, k, g.nz, 0., ul, uh, vl, vh, wl, wh, cekh, dfac, gh);
  const double bounded = scalar_tend<ADV, DIFF, LES>(B, m, lm, k, g.nz, 0., ul, uh, vl, vh, wl, wh, cekh, dfac, gh);
  cp[o] = cp[o] + (bounded - periodic);
}

// xso_convective: c(ie+1) <- c(ie+1) - (c(ie+1) - c(ie)) dxi rk3coef uouttot, on every row and level the plane holds
__global__ void scalar_bcx_outlet_kernel(Geo g, double fac0, const double *__restrict__ uout, const double *__restrict__ c,
                                         double *__restrict__ east) {
  const int jj = blockIdx.x * blockDim.x + threadIdx.x, kk = blockIdx.y;      // padded indices
  if (jj >= g.py) return;
  const long q = (long)kk * g.py + jj;
  const double fac = fac0 * uout[0];
  const double e1 = east[q];
  east[q] = e1 - (e1 - c[g.idx(g.nx - 1, jj - HY, kk - HZ)]) * fac;
}

// uouttot without a prescribed volume flow (src/modboundary.f90:143-156): sum_k u0av(k) dzf(k) / (zh(ke+1) - zh(kb+1)) with
// diagfld's slab averages over the fluid u points; S = level sums (solid points taken out), cnt = fluid counts or nullptr
__global__ void scalar_bcx_uout_kernel(int nz, const double *__restrict__ S, const double *__restrict__ cnt, double ncell,
                                       const double *__restrict__ wlev, double *__restrict__ out) {
  // the terms in parallel (one wave stages them in LDS), the sum by one lane in the reference's order (a chain of nz dependent global
  // loads cost 47 us at 256 levels: profiles/r06/open_x_kernel_stats_256.csv, first version)
  __shared__ double term[1024];
  if (blockIdx.x) return;
  double u = 0.;
  for (int k0 = 0; k0 < nz; k0 += 1024) {
    const int n = min(1024, nz - k0);
    for (int q = threadIdx.x; q < n; q += blockDim.x) {
      const int k = k0 + q;
      double c = cnt ? cnt[k + 1] : ncell;
      if (c == 0. && k == 0) c = cnt[nz];      // avexy_ibm's rule for a level without fluid cells (src/modmpi.f90:646-660)
      term[q] = (c > 0. ? S[k] / c : -999.) * wlev[k];
    }
    __syncthreads();
    if (threadIdx.x == 0)
      for (int q = 0; q < n; ++q) u = u + term[q];      // (the reference's order: sum over k of u0av dzf, then the division)
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = u;
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

}  // namespace

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
  const int gh = h->slot[n].kappa_ghosts;
#define LS(A, D, L)                                                                                     \
  do {                                                                                                  \
    PROF(h, "scalar_" #A #D #L);                                                                        \
    if (fresh) hipLaunchKernelGGL((scalar_kernel<A, D, L, true>), gr, b, 0, h->stream, g, tile_grid(g), h->m, cekh, dfac, u, v, w, ekh, c, cp, gh); \
    else hipLaunchKernelGGL((scalar_kernel<A, D, L, false>), gr, b, 0, h->stream, g, tile_grid(g), h->m, cekh, dfac, u, v, w, ekh, c, cp, gh); \
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

// the end columns of a passive scalar under BCxs = 2 (after any periodic sweep, LDS or plain)
int k_scalar_bcx_edges(udc_handle *h, int n, bool adv, bool diff) {
  if (h->scal_bcx != 2 || n >= 13 || !h->bcx_east[n]) return 0;
  const Geo &g = h->g;
  const double cekh = h->p.numol * h->p.prandtlmoli;
  const double *u = h->fields[UDC_U0], *v = h->fields[UDC_V0], *w = h->fields[UDC_W0];
  const double *ekh = h->fields[UDC_EKH], *c = h->fields[UDC_SV0 + 3 * n];
  double *cp = h->fields[UDC_SVP + 3 * n];
  const bool les = h->p.sgs != UDC_SGS_DNS;
  const double *prof = h->bcx_prof + (size_t)n * (g.nz + 2);
  const dim3 gr((unsigned)(4 * ((g.ny + 63) / 64)), (unsigned)g.nz), b(64);
  PROF(h, "scalar_bcx_edges");
#define LE(A, D, L) hipLaunchKernelGGL((scalar_bcx_edge_kernel<A, D, L>), gr, b, 0, h->stream, g, h->m, cekh, 0.5, u, v, w, ekh, c, \
                                       (const double *)h->bcx_east[n], prof, cp, 0)
  if (adv && diff) { if (les) LE(1, true, true); else LE(1, true, false); }
  else if (adv) LE(1, false, true);
  else if (diff) { if (les) LE(0, true, true); else LE(0, true, false); }
#undef LE
  HIP_OK(hipGetLastError());
  return 0;
}

int k_scalar_bcx_outlet(udc_handle *h) {
  if (h->scal_bcx != 2) return 0;
  const Geo &g = h->g;
  const double fac = h->m.dxi * h->bcx_rk3coef;
  for (int n : h->slots) {
    if (n >= 13 || !h->bcx_east[n]) continue;
    hipLaunchKernelGGL(scalar_bcx_outlet_kernel, dim3((unsigned)((g.py + 63) / 64), (unsigned)g.pz), dim3(64), 0, h->stream, g, fac,
                       (const double *)h->bcx_uout_dev, (const double *)h->fields[UDC_SV0 + 3 * n], h->bcx_east[n]);
  }
  HIP_OK(hipGetLastError());
  return 0;
}

// the outlet's speed from the state the substep starts from (= what diagfld left at the end of the previous one)
int k_scalar_bcx_uout(udc_handle *h) {
  if ((h->scal_bcx != 2 && !h->xg) || !h->bcx_uout_avg) return 0;      // (BCxs = 2, or BCxm = 2: udc_xopen.hip)
  if (h->xg && h->xo_hold) { h->xo_hold = false; return 0; }
  const Geo &g = h->g;
  if (k_level_sums_dev(h, UDC_U0, g.nz)) return 1;
  const double *cnt = h->ibm_on ? h->ibm[0].cnt_dev : nullptr;
  hipLaunchKernelGGL(scalar_bcx_uout_kernel, dim3(1), dim3(64), 0, h->stream, g.nz, (const double *)h->lev_sum16, cnt,
                     (double)(g.nx - 2 * g.xg) * (double)h->jtot, (const double *)h->bcx_wlev, h->bcx_uout_dev);
  HIP_OK(hipGetLastError());
  return 0;
}

// an upload of sv0 whose host array carries the x ghost columns (the reference's c-arrays do): keep the two east ones
int k_scalar_bcx_capture(udc_handle *h, int n, const double *host, const int lb[3], const int ub[3]) {
  if (h->scal_bcx != 2 || n >= 13 || !h->bcx_east[n]) return 0;
  const Geo &g = h->g;
  if (lb[0] > g.nx + 1 || ub[0] < g.nx + 2) return 0;      // no east ghost columns in this array
  const long hnx = ub[0] - lb[0] + 1, hny = ub[1] - lb[1] + 1;
  std::vector<double> e((size_t)2 * g.pz * g.py);
  HIP_OK(hipMemcpy(e.data(), h->bcx_east[n], sizeof(double) * e.size(), hipMemcpyDeviceToHost));
  for (int k = std::max(lb[2], 1 - HZ); k <= std::min(ub[2], g.nz + HZ); ++k)
    for (int j = std::max(lb[1], 1 - HY); j <= std::min(ub[1], g.ny + HY); ++j)
      for (int q = 0; q < 2; ++q)
        e[(size_t)q * g.pz * g.py + (size_t)(k - 1 + HZ) * g.py + (j - 1 + HY)] =
            host[(long)(g.nx + 1 + q - lb[0]) + hnx * ((long)(j - lb[1]) + hny * (long)(k - lb[2]))];
  HIP_OK(hipMemcpy(h->bcx_east[n], e.data(), sizeof(double) * e.size(), hipMemcpyHostToDevice));
  return 0;
}

// ... and a download of sv0 into such an array gets them back: the east columns from the stored planes, the inlet ghosts by
// xsi_profile's rule on levels kb .. ke+1 (elsewhere the periodic images udc_field_download put there stay)
int k_scalar_bcx_fill_host(udc_handle *h, int n, double *host, const int lb[3], const int ub[3]) {
  if (h->scal_bcx != 2 || n >= 13 || !h->bcx_east[n]) return 0;
  const Geo &g = h->g;
  const long hnx = ub[0] - lb[0] + 1, hny = ub[1] - lb[1] + 1;
  std::vector<double> e((size_t)2 * g.pz * g.py);
  HIP_OK(hipMemcpy(e.data(), h->bcx_east[n], sizeof(double) * e.size(), hipMemcpyDeviceToHost));
  const double *prof = h->bcx_prof_host.data() + (size_t)n * (g.nz + 2);
  for (int k = std::max(lb[2], 1 - HZ); k <= std::min(ub[2], g.nz + HZ); ++k)
    for (int j = std::max(lb[1], 1 - HY); j <= std::min(ub[1], g.ny + HY); ++j) {
      double *row = host + hnx * ((long)(j - lb[1]) + hny * (long)(k - lb[2])) - lb[0];      // row[i], i = the reference's index
      for (int q = 0; q < 2; ++q)
        if (g.nx + 1 + q <= ub[0]) row[g.nx + 1 + q] = e[(size_t)q * g.pz * g.py + (size_t)(k - 1 + HZ) * g.py + (j - 1 + HY)];
      if (k >= 1 && k <= g.nz + 1 && j >= 1 && j <= g.ny) {
        if (lb[0] <= 0) row[0] = 2. * prof[k] - row[1];
        if (lb[0] <= -1) row[-1] = 2. * prof[k] - row[0];
      }
    }
  return 0;
}

// Open lid (BCtopm = 3) with an immersed boundary: advecc_kappa's loop over the faces kb+1 .. ke+1 also leaves
// duml(ke+1) = cf w0(ke+1) dzfci(ke+1) in the out-of-domain plane svp(ke+1) (src/modadvection.f90:386-404; zero under a closed lid, where
// w0(ke+1) = 0), and ibmnorm's `solid` averages that plane into a solid cell of level ke whose upper neighbour counts as fluid
// (src/modibm.f90:796-800).  The sweeps stop at level ke: this plane kernel writes it (same face value as the top level's high face).
namespace {
__global__ void lid_kappa_plane_kernel(Geo g, Metrics m, const double *__restrict__ c, const double *__restrict__ w, double *__restrict__ cp) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y;
  if (i >= g.nx) return;
  const int nz = g.nz;
  const long o = g.idx(i, j, nz);                    // the cell (i, j, ke+1)
  const double wt = w[o];
  // face ke+1: cells ke-1, ke | ke+1, ke+2; dzhci(ke), dzhci(ke+1), dzhci(ke+2) = dzhci(ke+1) (src/modglobal.f90:857-859); dzfc(ke+1)
  const double f = face(wt, c[o - 2 * g.sz], c[o - g.sz], c[o], c[o + g.sz], m.dzhi[nz], m.dzhi[nz + 1], m.dzhi[nz + 1], m.dzf[nz + 1]);
  cp[o] = f * wt * m.dzfi[nz + 1];
}
}  // namespace
static int lid_kappa_plane(udc_handle *h, int n) {
  if (h->p.bctopm != UDC_TOP_PRESSURE || !h->ibm_on || n >= h->cfg.nsv || h->slot[n].adv != 1) return 0;
  const Geo &g = h->g;
  hipLaunchKernelGGL(lid_kappa_plane_kernel, dim3((g.nx + 63) / 64, g.ny), dim3(64), 0, h->stream, g, h->m, (const double *)h->fields[UDC_SV0 + 3 * n],
                     (const double *)h->fields[UDC_W0], h->fields[UDC_SVP + 3 * n]);
  HIP_OK(hipGetLastError());
  return 0;
}

int k_scalar_adv(udc_handle *h, int n) { return launch_scalar(h, n, true, false) || k_scalar_bcx_edges(h, n, true, false) || lid_kappa_plane(h, n); }
int k_scalar_diff(udc_handle *h, int n) { return launch_scalar(h, n, false, true) || k_scalar_bcx_edges(h, n, false, true); }
bool k_scalar_fused_lds(udc_handle *h, int n, bool fresh, int *rc);      // udc_scalar_lds.hip
int k_scalar_fused(udc_handle *h, int n, bool fresh) {
  int rc = 0;
  if (k_scalar_fused_lds(h, n, fresh, &rc)) return rc || k_scalar_bcx_edges(h, n, true, true) || lid_kappa_plane(h, n);
  return launch_scalar(h, n, true, true, fresh) || k_scalar_bcx_edges(h, n, true, true) || lid_kappa_plane(h, n);
}

// two slots in one sweep where that applies (udc_scalar_lds.hip, k_scalar_pair_lds): 0 done, -1 not applicable, 1 error
bool k_scalar_pair_lds(udc_handle *h, int na, int nb, bool fresh, int *rc);
int k_scalar_fused_pair(udc_handle *h, int na, int nb, bool fresh) {
  int rc = 0;
  if (!k_scalar_pair_lds(h, na, nb, fresh, &rc)) return -1;
  if (rc || k_scalar_bcx_edges(h, na, true, true) || k_scalar_bcx_edges(h, nb, true, true)) return 1;
  return 0;
}

// scalsource (src/modscalsource.f90:379-483): the sources are constant in time, the host evaluated them once
// (udcore/sources.py, udc_set_scalar_source); svp += source over the box that holds it
namespace {
__global__ __launch_bounds__(256) void box_add_kernel(Geo g, const double *__restrict__ src, int i0, int j0, int k0, int ni, int nj, int nk,
                                                       double *__restrict__ svp) {
  const int i = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y, k = blockIdx.z;
  if (i >= ni || j >= nj || k >= nk) return;
  const long c = g.idx(i0 + i, j0 + j, k0 + k);
  svp[c] = svp[c] + src[((size_t)k * nj + j) * ni + i];
}
}  // namespace

int k_scalsource(udc_handle *h) {
  const Geo &g = h->g;
  for (int n = 0; n < h->cfg.nsv; ++n) {
    const udc_handle::ScalarSource &s = h->svsrc[n];
    if (!s.d) continue;
    const int ni = s.hi[0] - s.lo[0] + 1, nj = s.hi[1] - s.lo[1] + 1, nk = s.hi[2] - s.lo[2] + 1;
    PROF(h, "scalsource");
    hipLaunchKernelGGL(box_add_kernel, dim3((unsigned)((ni + 63) / 64), (unsigned)((nj + 3) / 4), (unsigned)nk), dim3(64, 4), 0, h->stream,
                       g, (const double *)s.d, s.lo[0], s.lo[1], s.lo[2], ni, nj, nk, h->fields[UDC_SVP + 3 * n]);
    HIP_OK(hipGetLastError());
  }
  return 0;
}


// chem, src/modchem.f90:27-73 (IIc = 1 without obstacles): on RK stage 3, after sv0 = svm + rk3coef svp and before
// svm = sv0 -- here both have been written already, so both get the new value
namespace {
__global__ __launch_bounds__(256) void chem_kernel(Geo g, TileGrid tg, double k1, double J, double dt, double *__restrict__ a0,
    double *__restrict__ b0, double *__restrict__ c0, double *__restrict__ am, double *__restrict__ bm, double *__restrict__ cm) {
  int i, j, k;
  if (!tile_decode(g, tg, i, j, k)) return;
  const long c = g.idx(i, j, k);
  const double dNO = 1 * a0[c] / 30.006, dNO2 = 1 * b0[c] / 46.005, dO3 = 1 * c0[c] / 47.997;
  const double r = (dt * (-k1 * dNO * dO3 + J * dNO2)) / (1. + ((dNO + dO3) * k1 + J) * dt);
  const double na = 30.006 * ((a0[c] / 30.006) + r), nb = 46.005 * ((b0[c] / 46.005) - r), nc = 47.997 * ((c0[c] / 47.997) + r);
  a0[c] = na; b0[c] = nb; c0[c] = nc;
  am[c] = na; bm[c] = nb; cm[c] = nc;
}
}  // namespace

int k_chem(udc_handle *h, double dt) {
  const Geo &g = h->g;
  if (!h->lchem) return 0;
  PROF(h, "chem");
  hipLaunchKernelGGL(chem_kernel, dim3((unsigned)tile_grid(g).tiles * (unsigned)g.nz), dim3(64, 4), 0, h->stream, g, tile_grid(g),
                     h->chem_k1, h->chem_jno2, dt, h->fields[UDC_SV0], h->fields[UDC_SV0 + 3], h->fields[UDC_SV0 + 6],
                     h->fields[UDC_SVM], h->fields[UDC_SVM + 3], h->fields[UDC_SVM + 6]);
  HIP_OK(hipGetLastError());
  return 0;
}
