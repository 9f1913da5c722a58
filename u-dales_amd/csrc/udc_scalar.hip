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
    const double *__restrict__ c, double *__restrict__ cp, int gh) {
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
    double czm2 = c[o - 2 * sz], czp2 = c[o + 2 * sz];
    double czm1 = c[o - sz], czp1 = c[o + sz];      // (shadow the diffusion operands: those stay thl0's own ghosts)
    if (gh) {
      // kappa on thl (iadv_thl = 7) runs on the reference's separate copy thl0c, whose vertical ghost planes are not
      // thl0's: nothing ever writes the two below the floor (zero), and at the top `boundary` copies level ke upwards
      // for a flux condition (src/modboundary.f90:211-213) and leaves them untouched (zero) for a value condition
      const double top = gh == 1 ? 1. : 0.;
      if (k == 0) czm1 = 0.;
      if (k <= 1) czm2 = 0.;
      if (k == g.nz - 1) { czp1 = top * c0; czp2 = top * c0; }
      if (k == g.nz - 2) czp2 = top * czp1;
    }
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

int k_scalar_adv(udc_handle *h, int n) { return launch_scalar(h, n, true, false); }
int k_scalar_diff(udc_handle *h, int n) { return launch_scalar(h, n, false, true); }
int k_scalar_fused(udc_handle *h, int n, bool fresh) { return launch_scalar(h, n, true, true, fresh); }

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
