// Immersed boundary, sparse corrections (src/modibm.f90): the solid / fluid-boundary point lists of the u, v, w and c
// grids (solid_*.txt, fluid_boundary_*.txt), the masks initibm derives from them (:150-186), and
//   ibmwallfun without facet wall functions (:1216-1218, 1262-1264): diffu_corr (:990-1030), diffv_corr (:1033-1072),
//       diffw_corr (:1075-1117), diffc_corr (:1120-1164) -- the subgrid fluxes through solid faces are taken back out
//   ibmnorm (:697-745): solid (:748-826) -- velocities and tendencies zeroed inside the solid, scalars set to the mean of
//       their fluid neighbours (zero-flux condition)
// One thread per listed point.  The masks are only ever read at the listed points, so commit() evaluates them on the
// host into a few flag bits per point and the device never holds a mask array.
#include "udc_internal.h"
#include "udc_scalar_arith.h"
#include <cstring>

namespace {

// flags of a fluid-boundary point: which neighbours (of the same grid) are solid
//   u grid: 0 j+1, 1 j-1, 2 k+1, 3 k-1      v grid: 0 i+1, 1 i-1, 2 k+1, 3 k-1      w grid: 0 i+1, 1 i-1, 2 j+1, 3 j-1
//   c grid: 0 i+1, 1 i-1, 2 j+1, 3 j-1, 4 k+1, 5 k-1
// flags of a solid c point: which neighbours are fluid, in the reference's summation order 0 j+1, 1 j-1, 2 k+1, 3 k-1, 4 i+1, 5 i-1
__device__ __forceinline__ int wrapx(int i, int nx) { return i < 0 ? i + nx : (i >= nx ? i - nx : i); }

__global__ void ibm_diffu_corr_kernel(Geo g, Metrics m, int n, const int *__restrict__ pt, const unsigned char *__restrict__ fl,
                                      const double *__restrict__ u0, const double *__restrict__ ekm, double *__restrict__ up) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n) return;
  const int i = pt[3 * q], j = pt[3 * q + 1], k = pt[3 * q + 2], kf = k + 1;
  const int im = wrapx(i - 1, g.nx);
  const long c = g.idx(i, j, k), cm = g.idx(im, j, k), sy = g.sy, sz = g.sz;
  const unsigned f = fl[q];
  double t = up[c];
  if (f & 1u) {
    const double empo = 0.25 * ((ekm[c] + ekm[c + sy]) + (ekm[cm] + ekm[cm + sy]));
    t = t - empo * (u0[c + sy] - u0[c]) * m.dy2i;
  }
  if (f & 2u) {
    const double emmo = 0.25 * ((ekm[c] + ekm[c - sy]) + (ekm[cm - sy] + ekm[cm]));
    t = t + emmo * (u0[c] - u0[c - sy]) * m.dy2i;
  }
  if (f & 4u) {
    const double emop = (m.dzf[kf + 1] * (ekm[c] + ekm[cm]) + m.dzf[kf] * (ekm[c + sz] + ekm[cm + sz])) * m.dzhiq[kf + 1];
    t = t - emop * (u0[c + sz] - u0[c]) * m.dzhi[kf + 1] * m.dzfi[kf];
  }
  if (f & 8u) {
    const double emom = (m.dzf[kf - 1] * (ekm[c] + ekm[cm]) + m.dzf[kf] * (ekm[c - sz] + ekm[cm - sz])) * m.dzhiq[kf];
    t = t + emom * (u0[c] - u0[c - sz]) * m.dzhi[kf] * m.dzfi[kf];
  }
  up[c] = t;
}

__global__ void ibm_diffv_corr_kernel(Geo g, Metrics m, int n, const int *__restrict__ pt, const unsigned char *__restrict__ fl,
                                      const double *__restrict__ v0, const double *__restrict__ ekm, double *__restrict__ vp) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n) return;
  const int i = pt[3 * q], j = pt[3 * q + 1], k = pt[3 * q + 2], kf = k + 1;
  const long c = g.idx(i, j, k), cp = g.idx(wrapx(i + 1, g.nx), j, k), cm = g.idx(wrapx(i - 1, g.nx), j, k), sy = g.sy, sz = g.sz;
  const unsigned f = fl[q];
  double t = vp[c];
  if (f & 1u) {
    const double epmo = 0.25 * (ekm[c] + ekm[c - sy] + ekm[cp - sy] + ekm[cp]);
    t = t - epmo * (v0[cp] - v0[c]) * m.dx2i;
  }
  if (f & 2u) {
    const double emmo = 0.25 * (ekm[c] + ekm[c - sy] + ekm[cm - sy] + ekm[cm]);
    t = t + emmo * (v0[c] - v0[cm]) * m.dx2i;
  }
  if (f & 4u) {
    const double eomp = (m.dzf[kf + 1] * (ekm[c] + ekm[c - sy]) + m.dzf[kf] * (ekm[c + sz] + ekm[c - sy + sz])) * m.dzhiq[kf + 1];
    t = t - eomp * (v0[c + sz] - v0[c]) * m.dzhi[kf + 1] * m.dzfi[kf];
  }
  if (f & 8u) {
    const double eomm = (m.dzf[kf - 1] * (ekm[c] + ekm[c - sy]) + m.dzf[kf] * (ekm[c - sz] + ekm[c - sy - sz])) * m.dzhiq[kf];
    t = t + eomm * (v0[c] - v0[c - sz]) * m.dzhi[kf] * m.dzfi[kf];
  }
  vp[c] = t;
}

__global__ void ibm_diffw_corr_kernel(Geo g, Metrics m, int n, const int *__restrict__ pt, const unsigned char *__restrict__ fl,
                                      const double *__restrict__ w0, const double *__restrict__ ekm, double *__restrict__ wp) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n) return;
  const int i = pt[3 * q], j = pt[3 * q + 1], k = pt[3 * q + 2], kf = k + 1;
  const long c = g.idx(i, j, k), cp = g.idx(wrapx(i + 1, g.nx), j, k), cm = g.idx(wrapx(i - 1, g.nx), j, k), sy = g.sy, sz = g.sz;
  const unsigned f = fl[q];
  const double dzf_m = m.dzf[kf - 1], dzf_k = m.dzf[kf], q4 = m.dzhiq[kf];
  double t = wp[c];
  if (f & 1u) {
    const double epom = (dzf_m * (ekm[c] + ekm[cp]) + dzf_k * (ekm[c - sz] + ekm[cp - sz])) * q4;
    t = t - epom * (w0[cp] - w0[c]) * m.dx2i;
  }
  if (f & 2u) {
    const double emom = (dzf_m * (ekm[c] + ekm[cm]) + dzf_k * (ekm[c - sz] + ekm[cm - sz])) * q4;
    t = t + emom * (w0[c] - w0[cm]) * m.dx2i;
  }
  if (f & 4u) {
    const double eopm = (dzf_m * (ekm[c] + ekm[c + sy]) + dzf_k * (ekm[c - sz] + ekm[c + sy - sz])) * q4;
    t = t - eopm * (w0[c + sy] - w0[c]) * m.dy2i;
  }
  if (f & 8u) {
    const double eomm = (dzf_m * (ekm[c] + ekm[c - sy]) + dzf_k * (ekm[c - sz] + ekm[c - sy - sz])) * q4;
    t = t + eomm * (w0[c] - w0[c - sy]) * m.dy2i;
  }
  wp[c] = t;
}

__global__ void ibm_diffc_corr_kernel(Geo g, Metrics m, int n, const int *__restrict__ pt, const unsigned char *__restrict__ fl,
                                      const double *__restrict__ var, const double *__restrict__ ekh, double *__restrict__ rhs) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n) return;
  const int i = pt[3 * q], j = pt[3 * q + 1], k = pt[3 * q + 2], kf = k + 1;
  const long c = g.idx(i, j, k), cp = g.idx(wrapx(i + 1, g.nx), j, k), cm = g.idx(wrapx(i - 1, g.nx), j, k), sy = g.sy, sz = g.sz;
  const unsigned f = fl[q];
  double t = rhs[c];
  if (f & 1u) t = t - 0.5 * (ekh[cp] + ekh[c]) * (var[cp] - var[c]) * m.dx2i;
  if (f & 2u) t = t + 0.5 * (ekh[c] + ekh[cm]) * (var[c] - var[cm]) * m.dx2i;
  if (f & 4u) t = t - 0.5 * (ekh[c + sy] + ekh[c]) * (var[c + sy] - var[c]) * m.dy2i;
  if (f & 8u) t = t + 0.5 * (ekh[c] + ekh[c - sy]) * (var[c] - var[c - sy]) * m.dy2i;
  if (f & 16u) t = t - 0.5 * (m.dzf[kf + 1] * ekh[c] + m.dzf[kf] * ekh[c + sz]) * (var[c + sz] - var[c]) * m.dzh2i[kf + 1] * m.dzfi[kf];
  if (f & 32u) t = t + 0.5 * (m.dzf[kf - 1] * ekh[c] + m.dzf[kf] * ekh[c - sz]) * (var[c] - var[c - sz]) * m.dzh2i[kf] * m.dzfi[kf];
  rhs[c] = t;
}

// solid without a mask: var = 0, rhs = 0 at the listed points (velocities).
// comp >= 0 (pressure-total form of the fused substep, udc_plan.h): the sweep left -grad pres0 out everywhere, the reference's zero at
// a solid point includes it -- so the tendency there becomes + the gradient's component (u, v, w: 0, 1, 2; advecu/v/w's own
// differences, src/modadvection.f90:187,245,309), which the projection of pres0 + p then takes back out.
__global__ void ibm_solid_zero_kernel(Geo g, int n, const int *__restrict__ pt, double *__restrict__ var, double *__restrict__ rhs,
                                      int comp, const double *__restrict__ pres0, Metrics m) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n) return;
  const int i = pt[3 * q], j = pt[3 * q + 1], k = pt[3 * q + 2];
  const long c = g.idx(i, j, k);
  var[c] = 0.;
  double t = 0.;
  if (comp == 0) t = (pres0[c] - pres0[g.idx(i == 0 ? g.nx - 1 : i - 1, j, k)]) * m.dxi;
  else if (comp == 1) t = (pres0[c] - pres0[c - g.sy]) * m.dyi;
  else if (comp == 2 && k >= 1) t = (pres0[c] - pres0[c - g.sz]) * m.dzhi[k + 1];
  rhs[c] = t;
}

// solid with the c mask: value and tendency become the mean over the fluid neighbours (or val / 0 without any).  A neighbour
// beyond a lateral edge of the domain (flagged "fluid" where the reference's masks are not wrapped: udc_set_ibm_mask_wrap) is a
// ghost cell there: its value is the periodic image; its tendency (src/modibm.f90:748-826 reads rhs(i+-1, j+-1) all the same, and
// tendency ghosts are not exchanged) is whatever this substep's routines left there -- nothing (zero), except under the kappa
// scheme, whose face loops run to ie + 1 / je + 1 and credit each face's flux to both cells it borders (src/modadvection.f90:335-377):
// the ghost cell beyond the high edge holds + (flux through the domain's edge face) / dx, the one beyond the low edge - that.
// kc / u0 / v0: the kappa-advected field and the velocities (null: no such term).  j0 / jtot: this slab's first global row / the
// domain's rows.
__global__ void ibm_solid_mean_kernel(Geo g, int n, const int *__restrict__ pt, const unsigned char *__restrict__ fl, const double *__restrict__ valp,
                                      double *__restrict__ var, double *__restrict__ rhs, int j0, int jtot, const double *__restrict__ kc,
                                      const double *__restrict__ u0, const double *__restrict__ v0, double dx, double dxi, double dyi) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n) return;
  const double val = valp ? *valp : 0.;
  const int i = pt[3 * q], j = pt[3 * q + 1], k = pt[3 * q + 2];
  const long c = g.idx(i, j, k), sy = g.sy, sz = g.sz;
  const long nb[6] = {c + sy, c - sy, c + sz, c - sz, g.idx(wrapx(i + 1, g.nx), j, k), g.idx(wrapx(i - 1, g.nx), j, k)};
  const unsigned f = fl[q];
  const bool ghost[6] = {j0 + j + 1 >= jtot, j0 + j - 1 < 0, false, false, i + 1 >= g.nx, i - 1 < 0};
  double v = val, r = 0., count = 0.;
#pragma unroll
  for (int b = 0; b < 6; ++b)
    if (f & (1u << b)) {
      count += 1.; v = v + var[nb[b]];
      if (!ghost[b]) r = r + rhs[nb[b]];
      else if (kc) {
        if (b >= 4) {      // the edge face in x: between columns nx-1 and 0
          const long r0 = g.idx(0, j, k);
          const double vel = u0[r0];
          const double fx = face(vel, kc[r0 + g.nx - 2], kc[r0 + g.nx - 1], kc[r0], kc[r0 + 1], dxi, dxi, dxi, dx) * vel;
          r = r + (b == 4 ? fx * dxi : -fx * dxi);
        } else {           // the edge face in y: the low face of row jj = ny (b = 0: its ghost copy) or 0 (b = 1)
          const long r0 = g.idx(i, b == 0 ? j + 1 : j, k);
          const double vel = v0[r0];
          const double fy = face(vel, kc[r0 - 2 * sy], kc[r0 - sy], kc[r0], kc[r0 + sy], 1., 1., 1., 1.) * vel;
          r = r + (b == 0 ? fy * dyi : -fy * dyi);
        }
      }
    }
  if (count > 0.) { v = (v - val) / count; r = r / count; }
  var[c] = v;
  rhs[c] = r;
}

// the value `solid` gives thl where an obstacle cell has no fluid neighbour (ibmnorm, src/modibm.f90:715):
// sum(thl0av(kb:ke) dzf(kb:ke)) / zh(ke+1) with thl0av the slab average over the fluid cells; S = masked level sums
__global__ void ibm_thl_val_kernel(int nz, const double *__restrict__ S, const double *__restrict__ cnt, const double *__restrict__ dzf,
                                   double zsize, double *__restrict__ out) {
  // a level without fluid cells: avexy_ibm's rule (src/modmpi.f90:646-660, as k_slab_averages applies it on the host): -999,
  // except the floor level, which takes the count of level ke
  double v = 0.;
  for (int k = 1; k <= nz; ++k) {
    double c = cnt[k];
    if (c == 0. && k == 1) c = cnt[nz];
    v += (c > 0. ? S[k - 1] / c : -999.) * dzf[k];
  }
  *out = v / zsize;
}

// advecc2nd_corr_conservative (src/modibm.f90:889-933) / advecc2nd_corr_liberal (:936-987) at the fluid-boundary points of
// the c grid, for a field advected by advecc_2nd (thl, qt).  fl bit b: mask_c of neighbour b (i+1, i-1, j+1, j-1, k+1,
// k-1) is solid; fl2 bit b: the momentum mask on the face towards neighbour b is solid (mask_u(i+1), mask_u(i),
// mask_v(j+1), mask_v(j), mask_w(k+1), mask_w(k)).
template <bool CONS>
__global__ void ibm_advecc_corr_kernel(Geo g, Metrics m, int n, const int *__restrict__ pt, const unsigned char *__restrict__ fl,
                                       const unsigned char *__restrict__ fl2, const double *__restrict__ u0, const double *__restrict__ v0,
                                       const double *__restrict__ w0, const double *__restrict__ var, double *__restrict__ rhs) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n) return;
  const int i = pt[3 * q], j = pt[3 * q + 1], k = pt[3 * q + 2], kf = k + 1;
  const long c = g.idx(i, j, k), cp = g.idx(wrapx(i + 1, g.nx), j, k), cm = g.idx(wrapx(i - 1, g.nx), j, k), sy = g.sy, sz = g.sz;
  const unsigned f = CONS ? (unsigned)(fl[q] | fl2[q]) : (unsigned)fl[q];
  const double vc = var[c];
  double t = rhs[c];
  if (CONS) {
    if (f & 1u)  t = t + u0[cp] * (var[cp] + vc) * m.dxi5;
    if (f & 2u)  t = t - u0[c] * (var[cm] + vc) * m.dxi5;
    if (f & 4u)  t = t + v0[c + sy] * (var[c + sy] + vc) * m.dyi5;
    if (f & 8u)  t = t - v0[c] * (var[c - sy] + vc) * m.dyi5;
    if (f & 16u) t = t + w0[c + sz] * (var[c + sz] * m.dzf[kf] + vc * m.dzf[kf + 1]) * m.dzhi[kf + 1] * m.dzfi5[kf];
    if (f & 32u) t = t - w0[c] * (var[c - sz] * m.dzf[kf] + vc * m.dzf[kf - 1]) * m.dzhi[kf] * m.dzfi5[kf];
  } else {
    if (f & 1u)  t = t + u0[cp] * (var[cp] + vc) * m.dxi5 - u0[cp] * (vc + vc) * m.dxi5;
    if (f & 2u)  t = t - u0[c] * (var[cm] + vc) * m.dxi5 + u0[c] * (vc + vc) * m.dxi5;
    if (f & 4u)  t = t + v0[c + sy] * (var[c + sy] + vc) * m.dyi5 - v0[c + sy] * (vc + vc) * m.dyi5;
    if (f & 8u)  t = t - v0[c] * (var[c - sy] + vc) * m.dyi5 + v0[c] * (vc + vc) * m.dyi5;
    if (f & 16u) t = t + w0[c + sz] * (var[c + sz] * m.dzf[kf] + vc * m.dzf[kf + 1]) * m.dzhi[kf + 1] * m.dzfi5[kf]
                       - w0[c + sz] * (vc * m.dzf[kf] + vc * m.dzf[kf + 1]) * m.dzhi[kf + 1] * m.dzfi5[kf];
    if (f & 32u) t = t - w0[c] * (var[c - sz] * m.dzf[kf] + vc * m.dzf[kf - 1]) * m.dzhi[kf] * m.dzfi5[kf]
                       + w0[c] * (vc * m.dzf[kf] + vc * m.dzf[kf - 1]) * m.dzhi[kf] * m.dzfi5[kf];
  }
  rhs[c] = t;
}

// masked slab sums (avexy_ibm, src/modmpi.f90:623-664): the sum over the fluid cells of a level is the sum over all
// cells minus the sum over the listed solid points.  One workgroup per (level, field); deterministic tree reduction.
__global__ __launch_bounds__(256) void ibm_levelsum_kernel(Geo g, const int *__restrict__ pts, const int *__restrict__ off,
                                                           const double *__restrict__ f, double *__restrict__ S, int k0) {
  __shared__ double sw[4];
  const int k = k0 + blockIdx.x;
  double v = 0.;
  for (int q = off[k] + threadIdx.x; q < off[k + 1]; q += 256) v += f[g.idx(pts[3 * q], pts[3 * q + 1], pts[3 * q + 2])];
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  if ((threadIdx.x & 63) == 0) sw[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) S[blockIdx.x] = S[blockIdx.x] - ((sw[0] + sw[1]) + (sw[2] + sw[3]));
}
// the same with a per-level weight, summed over all levels (masscorr's volume averages): one workgroup
__global__ __launch_bounds__(1024) void ibm_flowsum_kernel(Geo g, int n, const int *__restrict__ pts, const double *__restrict__ a,
                                                           const double *__restrict__ b, const double *__restrict__ wlev, double *__restrict__ S,
                                                           int only_i) {
  __shared__ double sa_[16], sb_[16];
  double sa = 0., sb = 0.;
  for (int q = threadIdx.x; q < n; q += 1024) {
    if (only_i >= 0 && pts[3 * q] != only_i) continue;
    const int k = pts[3 * q + 2];
    const long c = g.idx(pts[3 * q], pts[3 * q + 1], k);
    const double w = wlev[k + 1];
    sa += a[c] * w;
    if (b) sb += b[c] * w;
  }
  for (int o = 32; o > 0; o >>= 1) { sa += __shfl_xor(sa, o, 64); sb += __shfl_xor(sb, o, 64); }
  if ((threadIdx.x & 63) == 0) { sa_[threadIdx.x >> 6] = sa; sb_[threadIdx.x >> 6] = sb; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double ta = 0., tb = 0.;
    for (int q = 0; q < 16; ++q) { ta += sa_[q]; tb += sb_[q]; }
    S[0] = S[0] - ta;
    S[1] = S[1] - tb;
  }
}

inline unsigned blocks(int n) { return (unsigned)((n + 127) / 128); }

}  // namespace

// ------------------------------------------------------------------------------------------------ host side
extern "C" int udc_set_ibm_points(udc_handle *h, int grid, const int *solid, int nsolid, const int *bound, int nbound) {
  if (!h) { udc_set_error("null handle"); return 1; }
  if (grid < 0 || grid > 3) { udc_set_error("udc_set_ibm_points: grid 0 (u), 1 (v), 2 (w) or 3 (c)"); return 1; }
  if (nsolid < 0 || nbound < 0 || (nsolid && !solid) || (nbound && !bound)) { udc_set_error("udc_set_ibm_points: bad list"); return 1; }
  // A point whose (i, j) lies outside the domain belongs to no rank: the reference's reader keeps the points of its own pencil and
  // drops the rest without a word (read_sparse_ijk, src/readinput.f90:90-100 -- the lists of its own tests/cases/526 cover 256 x 128
  // columns of a 128 x 64 domain).  Same here; a level outside kb..ke is an error (the reference would index out of bounds).
  if (h->xg && grid == 3 && h->cfg.nsv == 0 && !((int)h->fields.size() > UDC_THL0 && h->fields[UDC_THL0])) {
    udc_set_error("udc_set_ibm_points: open x boundaries without a scalar field: the c lists have no use (the reference does not read them either, src/modibm.f90:181)");
    return 1;
  }
  const int ext[3] = {h->g.nx - 2 * h->g.xg, h->jtot, h->g.nz};      // (open x boundaries: the deck's itot)
  udc_handle::IbmGrid &G = h->ibm[grid];
  G.solid_g.clear(); G.bound_g.clear();
  for (int pass = 0; pass < 2; ++pass) {
    const int *p = pass ? bound : solid;
    const int n = pass ? nbound : nsolid;
    std::vector<int> &dst = pass ? G.bound_g : G.solid_g;
    dst.reserve((size_t)3 * n);
    for (int q = 0; q < n; ++q) {
      if (p[3 * q + 2] < 1 || p[3 * q + 2] > ext[2]) {
        udc_set_error("udc_set_ibm_points: %s point %d of grid %d outside the levels of the domain (%d %d %d)", pass ? "boundary" : "solid", q + 1,
                      grid, p[3 * q], p[3 * q + 1], p[3 * q + 2]);
        return 1;
      }
      if (p[3 * q] < 1 || p[3 * q] > ext[0] || p[3 * q + 1] < 1 || p[3 * q + 1] > ext[1]) continue;
      dst.insert(dst.end(), p + 3 * q, p + 3 * q + 3);
    }
  }
  G.given = true;
  h->ibm_on = false;      // until udc_ibm_commit
  return 0;
}

static int upload_points(udc_handle *h, const std::vector<int> &pts, const std::vector<unsigned char> &fl, int **dpt, unsigned char **dfl) {
  if (*dpt) { HIP_OK(hipFree(*dpt)); *dpt = nullptr; }
  if (*dfl) { HIP_OK(hipFree(*dfl)); *dfl = nullptr; }
  if (pts.empty()) return 0;
  HIP_OK(hipMalloc(dpt, sizeof(int) * pts.size()));
  HIP_OK(hipMemcpy(*dpt, pts.data(), sizeof(int) * pts.size(), hipMemcpyHostToDevice));
  HIP_OK(hipMalloc(dfl, fl.size()));
  HIP_OK(hipMemcpy(*dfl, fl.data(), fl.size(), hipMemcpyHostToDevice));
  (void)h;
  return 0;
}

extern "C" int udc_set_ibm_mask_wrap(udc_handle *h, int wrapx, int wrapy) {
  if (!h) { udc_set_error("null handle"); return 1; }
  h->ibm_wrap_x = wrapx != 0;
  h->ibm_wrap_y = wrapy != 0;
  return 0;
}

extern "C" int udc_ibm_commit(udc_handle *h) {
  if (!h) { udc_set_error("null handle"); return 1; }
  HIP_OK(hipSetDevice(h->device));
  if (udc_flush_pending(h)) return 1;
  for (int gq = 0; gq < 3; ++gq)
    if (!h->ibm[gq].given) { udc_set_error("udc_ibm_commit: the u, v and w point lists are needed (udc_set_ibm_points)"); return 1; }
  const bool have_c = h->ibm[3].given;
  // tables built from the point lists are rebuilt on their next use: masscorr's outlet-area weights (luoutflowr) and level weights
  if (h->outlet_w) { HIP_OK(hipStreamSynchronize(h->stream)); HIP_OK(hipFree(h->outlet_w)); h->outlet_w = nullptr; }
  if (!h->slots.empty() && !have_c) { udc_set_error("udc_ibm_commit: transported scalars need the c point lists"); return 1; }
  // thl (slot 15) and qt (13): ibmnorm / diffc_corr as for the scalars plus advecc2nd_corr; wall fluxes of heat and moisture:
  // wallfunheat (udc_ibm_wf.hip; udc_set_ibm_wallheat / udc_set_ibm_wallmoist), else adiabatic / impermeable walls.  The moist
  // thermodynamics' slab averages run over the fluid cells (udc_thermo.hip).  Not built: the one-equation closure's e12 next to obstacles.
  for (int n : h->slots)
    if (n == 14) { udc_set_error("udc_ibm_commit: the one-equation closure (e12) is not available with immersed boundaries"); return 1; }
  if (h->lmoist && h->mt && !have_c) { udc_set_error("udc_ibm_commit: the moist thermodynamics average over the fluid cells: the c point lists are needed"); return 1; }
  // (open x boundaries: nx is the deck's itot, the lists' i; device column = i - 1 + xg)
  const int xg = h->g.xg, nx = h->g.nx - 2 * xg, ny = h->jtot, nz = h->g.nz, j0 = h->cfg.rank * h->g.ny, nyl = h->g.ny;
  // masks as initibm builds them (src/modibm.f90:150-186): 1 = fluid; planes k = 0 (kb-1) .. nz+1.  Beyond a lateral
  // boundary of the domain: the periodic image, or "fluid" where the reference's exchange_halo_z would not have wrapped
  // (a direction held by one rank; udc_set_ibm_mask_wrap)
  const size_t plane = (size_t)nx * ny, msz = plane * (nz + 2);
  std::vector<std::vector<unsigned char>> mask(4);
  for (int gq = 0; gq < 4; ++gq) {
    if (gq == 3 && !have_c) break;
    mask[gq].assign(msz, 1);
    std::memset(mask[gq].data(), 0, plane);                                  // k = kb-1
    if (gq == 2) std::memset(mask[gq].data() + plane, 0, plane);             // mask_w(:,:,kb) = 0
    const std::vector<int> &s = h->ibm[gq].solid_g;
    for (size_t q = 0; q < s.size() / 3; ++q)
      mask[gq][(size_t)(s[3 * q] - 1) + (size_t)nx * (s[3 * q + 1] - 1) + plane * s[3 * q + 2]] = 0;
  }
  const bool wrapx = h->ibm_wrap_x, wrapy = h->ibm_wrap_y;
  auto at = [&](int gq, int i, int j, int k) -> unsigned char {      // 1-based i, j; k = 0..nz+1
    if ((i < 1 || i > nx) && !wrapx) return 1;
    if ((j < 1 || j > ny) && !wrapy) return 1;
    i = (i - 1 + nx) % nx; j = (j - 1 + ny) % ny;
    return mask[gq][(size_t)i + (size_t)nx * j + plane * k];
  };
  for (int gq = 0; gq < 4; ++gq) {
    udc_handle::IbmGrid &G = h->ibm[gq];
    if (gq == 3 && !have_c) { G.nsolid = G.nbound = 0; continue; }
    std::vector<int> sp, bp;
    std::vector<unsigned char> sf, bf, bf2;
    for (size_t q = 0; q < G.solid_g.size() / 3; ++q) {
      const int i = G.solid_g[3 * q], j = G.solid_g[3 * q + 1], k = G.solid_g[3 * q + 2];
      if (j <= j0 || j > j0 + nyl) continue;
      sp.push_back(i - 1 + xg); sp.push_back(j - 1 - j0); sp.push_back(k - 1);
      unsigned f = 0;
      if (gq == 3) {
        const int ni[6] = {i, i, i, i, i + 1, i - 1}, nj[6] = {j + 1, j - 1, j, j, j, j}, nk[6] = {k, k, k + 1, k - 1, k, k};
        for (int b = 0; b < 6; ++b)
          if (at(3, ni[b], nj[b], nk[b]) == 1) f |= 1u << b;
      }
      sf.push_back((unsigned char)f);
    }
    for (size_t q = 0; q < G.bound_g.size() / 3; ++q) {
      const int i = G.bound_g[3 * q], j = G.bound_g[3 * q + 1], k = G.bound_g[3 * q + 2];
      if (j <= j0 || j > j0 + nyl) continue;
      bp.push_back(i - 1 + xg); bp.push_back(j - 1 - j0); bp.push_back(k - 1);
      unsigned f = 0;
      int ni[6], nj[6], nk[6], nb = 4;
      for (int b = 0; b < 6; ++b) { ni[b] = i; nj[b] = j; nk[b] = k; }
      if (gq == 0) { nj[0] = j + 1; nj[1] = j - 1; nk[2] = k + 1; nk[3] = k - 1; }
      else if (gq == 1) { ni[0] = i + 1; ni[1] = i - 1; nk[2] = k + 1; nk[3] = k - 1; }
      else if (gq == 2) { ni[0] = i + 1; ni[1] = i - 1; nj[2] = j + 1; nj[3] = j - 1; }
      else { ni[0] = i + 1; ni[1] = i - 1; nj[2] = j + 1; nj[3] = j - 1; nk[4] = k + 1; nk[5] = k - 1; nb = 6; }
      for (int b = 0; b < nb; ++b)
        if (at(gq, ni[b], nj[b], nk[b]) == 0) f |= 1u << b;
      bf.push_back((unsigned char)f);
      if (gq == 3) {      // advecc2nd_corr_conservative: mask_u(i+1), mask_u(i), mask_v(j+1), mask_v(j), mask_w(k+1), mask_w(k)
        unsigned f2 = 0;
        if (at(0, i + 1, j, k) == 0) f2 |= 1u;
        if (at(0, i, j, k) == 0) f2 |= 2u;
        if (at(1, i, j + 1, k) == 0) f2 |= 4u;
        if (at(1, i, j, k) == 0) f2 |= 8u;
        if (at(2, i, j, k + 1) == 0) f2 |= 16u;
        if (at(2, i, j, k) == 0) f2 |= 32u;
        bf2.push_back((unsigned char)f2);
      }
    }
    G.nsolid = (int)(sp.size() / 3); G.nbound = (int)(bp.size() / 3);
    if (upload_points(h, sp, sf, &G.solid, &G.solid_fl)) return 1;
    if (upload_points(h, bp, bf, &G.bound, &G.bound_fl)) return 1;
    if (G.bound_fl2) { HIP_OK(hipFree(G.bound_fl2)); G.bound_fl2 = nullptr; }
    if (gq == 3 && !bf2.empty()) {
      HIP_OK(hipMalloc(&G.bound_fl2, bf2.size()));
      HIP_OK(hipMemcpy(G.bound_fl2, bf2.data(), bf2.size(), hipMemcpyHostToDevice));
    }
    // solid points by level (counting sort), fluid cells per level of the whole domain (IIus ... of createmasks; the w mask
    // also excludes the floor level itself, src/modibm.f90:2179)
    std::vector<int> off(nz + 3, 0), lp(sp.size());
    for (size_t q = 0; q < sp.size() / 3; ++q) ++off[sp[3 * q + 2] + 1];
    for (int k = 0; k <= nz + 1; ++k) off[k + 1] += off[k];
    {
      std::vector<int> pos(off.begin(), off.end() - 1);
      for (size_t q = 0; q < sp.size() / 3; ++q) {
        const int d = pos[sp[3 * q + 2]]++;
        lp[3 * d] = sp[3 * q]; lp[3 * d + 1] = sp[3 * q + 1]; lp[3 * d + 2] = sp[3 * q + 2];
      }
    }
    if (G.lev_pts) { HIP_OK(hipFree(G.lev_pts)); G.lev_pts = nullptr; }
    if (G.lev_off) { HIP_OK(hipFree(G.lev_off)); G.lev_off = nullptr; }
    if (!lp.empty()) {
      HIP_OK(hipMalloc(&G.lev_pts, sizeof(int) * lp.size()));
      HIP_OK(hipMemcpy(G.lev_pts, lp.data(), sizeof(int) * lp.size(), hipMemcpyHostToDevice));
    }
    HIP_OK(hipMalloc(&G.lev_off, sizeof(int) * off.size()));
    HIP_OK(hipMemcpy(G.lev_off, off.data(), sizeof(int) * off.size(), hipMemcpyHostToDevice));
    G.fluid_cnt.assign(nz + 2, (double)nx * (double)ny);
    for (size_t q = 0; q < G.solid_g.size() / 3; ++q) {
      const int k = G.solid_g[3 * q + 2];
      if (!(gq == 2 && k == 1)) G.fluid_cnt[k] -= 1.;
    }
    if (gq == 2) G.fluid_cnt[1] = 0.;
    if (G.cnt_dev) { HIP_OK(hipFree(G.cnt_dev)); G.cnt_dev = nullptr; }
    HIP_OK(hipMalloc(&G.cnt_dev, sizeof(double) * (nz + 2)));
    HIP_OK(hipMemcpy(G.cnt_dev, G.fluid_cnt.data(), sizeof(double) * (nz + 2), hipMemcpyHostToDevice));
  }
  if (!h->ibm_val) HIP_OK(hipMalloc(&h->ibm_val, sizeof(double)));
  if (h->ibm_wlev) { HIP_OK(hipFree(h->ibm_wlev)); h->ibm_wlev = nullptr; }
  h->ibm_on = true;
  return 0;
}

void ibm_destroy(udc_handle *h) {
  if (h->ibm_wlev) hipFree(h->ibm_wlev);
  for (auto &G : h->ibm) {
    if (G.solid) hipFree(G.solid);
    if (G.solid_fl) hipFree(G.solid_fl);
    if (G.bound) hipFree(G.bound);
    if (G.bound_fl) hipFree(G.bound_fl);
    if (G.lev_pts) hipFree(G.lev_pts);
    if (G.lev_off) hipFree(G.lev_off);
    if (G.cnt_dev) hipFree(G.cnt_dev);
    if (G.bound_fl2) hipFree(G.bound_fl2);
  }
  if (h->ibm_val) hipFree(h->ibm_val);
}

int ibm_grid_of_field(int field) {
  if (field == UDC_U0 || field == UDC_UM || field == UDC_UP) return 0;
  if (field == UDC_V0 || field == UDC_VM || field == UDC_VP) return 1;
  if (field == UDC_W0 || field == UDC_WM || field == UDC_WP) return 2;
  return 3;
}

int k_ibm_levelsum_correct(udc_handle *h, const int *fields, int nf, int n, double *S, int k0) {
  if (!h->ibm_on) return 0;
  for (int q = 0; q < nf; ++q) {
    const udc_handle::IbmGrid &G = h->ibm[ibm_grid_of_field(fields[q])];
    if (!G.nsolid) continue;
    hipLaunchKernelGGL(ibm_levelsum_kernel, dim3((unsigned)n), dim3(256), 0, h->stream, h->g, G.lev_pts, G.lev_off,
                       (const double *)h->fields[fields[q]], S + (size_t)q * n, k0);
  }
  HIP_OK(hipGetLastError());
  return 0;
}

int k_ibm_flowsum_correct(udc_handle *h, int grid, const double *a, const double *b, const double *wlev, double *S, int only_i) {
  const udc_handle::IbmGrid &G = h->ibm[grid];
  if (!h->ibm_on || !G.nsolid) return 0;
  hipLaunchKernelGGL(ibm_flowsum_kernel, dim3(1), dim3(1024), 0, h->stream, h->g, G.nsolid, G.solid, a, b, wlev, S, only_i);
  HIP_OK(hipGetLastError());
  return 0;
}

// ibmwallfun without facet wall functions: the diffusion corrections (additive, so equally valid on pup = up + um/rk3coef)
int k_ibm_wallfun(udc_handle *h) {
  if (!h->ibm_on) return 0;
  const Geo &g = h->g;
  PROF(h, "ibm_wallfun");
  const udc_handle::IbmGrid &U = h->ibm[0], &V = h->ibm[1], &W = h->ibm[2], &C = h->ibm[3];
  const double *ekm = h->fields[UDC_EKM], *ekh = h->fields[UDC_EKH];
  if (k_ibm_facet_begin(h)) return 1;     // lwritefac on an RK stage 3: this substep's per-facet sums (udc_ibm_wf.hip)
  if (k_ibm_wallfunmom(h)) return 1;      // iwallmom > 1: the facet wall functions first (src/modibm.f90:1183-1194)
  if (U.nbound) hipLaunchKernelGGL(ibm_diffu_corr_kernel, dim3(blocks(U.nbound)), dim3(128), 0, h->stream, g, h->m, U.nbound, U.bound, U.bound_fl,
                                   (const double *)h->fields[UDC_U0], ekm, h->fields[UDC_UP]);
  if (V.nbound) hipLaunchKernelGGL(ibm_diffv_corr_kernel, dim3(blocks(V.nbound)), dim3(128), 0, h->stream, g, h->m, V.nbound, V.bound, V.bound_fl,
                                   (const double *)h->fields[UDC_V0], ekm, h->fields[UDC_VP]);
  if (W.nbound) hipLaunchKernelGGL(ibm_diffw_corr_kernel, dim3(blocks(W.nbound)), dim3(128), 0, h->stream, g, h->m, W.nbound, W.bound, W.bound_fl,
                                   (const double *)h->fields[UDC_W0], ekm, h->fields[UDC_WP]);
  if (k_ibm_wallfunheat(h)) return 1;     // iwalltemp = 2: the heat wall function on thlp (:1220-1231), then diffc_corr
  if (k_ibm_facet_end(h)) return 1;
  if (C.nbound)
    for (int n : h->slots)
      hipLaunchKernelGGL(ibm_diffc_corr_kernel, dim3(blocks(C.nbound)), dim3(128), 0, h->stream, g, h->m, C.nbound, C.bound, C.bound_fl,
                         (const double *)h->fields[UDC_SV0 + 3 * n], ekh, h->fields[UDC_SVP + 3 * n]);
  HIP_OK(hipGetLastError());
  return 0;
}

// ibmnorm: solid velocities (um, and the tendency -- or pup, which must vanish with both) to zero; scalars to the mean
// of their fluid neighbours
int k_ibm_norm(udc_handle *h) {
  if (!h->ibm_on) return 0;
  const Geo &g = h->g;
  PROF(h, "ibm_norm");
  for (int q = 0; q < 3; ++q) {
    const udc_handle::IbmGrid &G = h->ibm[q];
    if (G.nsolid) hipLaunchKernelGGL(ibm_solid_zero_kernel, dim3(blocks(G.nsolid)), dim3(128), 0, h->stream, g, G.nsolid, G.solid,
                                     h->fields[UDC_UM + q], h->fields[UDC_UP + q], h->ptotal_now ? q : -1,
                                     (const double *)h->fields[UDC_PRES0], h->m);
  }
  const udc_handle::IbmGrid &C = h->ibm[3];
  for (int n : h->slots) {
    const double *valp = nullptr;
    if (n == 15) {
      // thl: the volume mean of thl0av (slab averages over the fluid cells, as the last `thermodynamics` left them: thl0
      // has not changed since) -- level sums with the solid points taken out, all-reduced over the slabs, on the device
      if (k_level_sums_dev(h, UDC_THL0, g.nz)) return 1;
      hipLaunchKernelGGL(ibm_thl_val_kernel, dim3(1), dim3(1), 0, h->stream, g.nz, (const double *)h->lev_sum16, (const double *)C.cnt_dev,
                         h->m.dzf, h->zsize, h->ibm_val);
      valp = h->ibm_val;
    }
    if (C.nsolid)
      hipLaunchKernelGGL(ibm_solid_mean_kernel, dim3(blocks(C.nsolid)), dim3(128), 0, h->stream, g, C.nsolid, C.solid, C.solid_fl, valp,
                         h->fields[UDC_SVM + 3 * n], h->fields[UDC_SVP + 3 * n], h->cfg.rank * g.ny, h->jtot,
                         (const double *)(h->slot[n].adv == 2 ? nullptr : h->fields[UDC_SV0 + 3 * n]), (const double *)h->fields[UDC_U0],
                         (const double *)h->fields[UDC_V0], h->m.dx, h->m.dxi, h->m.dyi);
    // fields advected by advecc_2nd (thl unless iadv_thl = 7, qt): src/modibm.f90:716-722, 727-731
    if (C.nbound && n >= 13 && h->slot[n].adv == 2) {
      const double *u0 = h->fields[UDC_U0], *v0 = h->fields[UDC_V0], *w0 = h->fields[UDC_W0];
      if (h->ibm_conservative)
        hipLaunchKernelGGL(ibm_advecc_corr_kernel<true>, dim3(blocks(C.nbound)), dim3(128), 0, h->stream, g, h->m, C.nbound, C.bound, C.bound_fl,
                           C.bound_fl2, u0, v0, w0, (const double *)h->fields[UDC_SV0 + 3 * n], h->fields[UDC_SVP + 3 * n]);
      else
        hipLaunchKernelGGL(ibm_advecc_corr_kernel<false>, dim3(blocks(C.nbound)), dim3(128), 0, h->stream, g, h->m, C.nbound, C.bound, C.bound_fl,
                           C.bound_fl2, u0, v0, w0, (const double *)h->fields[UDC_SV0 + 3 * n], h->fields[UDC_SVP + 3 * n]);
    }
  }
  HIP_OK(hipGetLastError());
  return 0;
}

extern "C" int udc_set_ibm_conservative(udc_handle *h, int lconservativeibm) {
  if (!h) { udc_set_error("null handle"); return 1; }
  h->ibm_conservative = lconservativeibm != 0;
  return 0;
}
