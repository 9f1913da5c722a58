// Facet wall functions of the immersed boundary: wallfunmom (src/modibm.f90:1286-1433).
//
// Every facet section -- a piece of wall of given area seen from a fluid-boundary cell of the u, v or w grid -- takes the
// velocity at the cell (interp_velocity_*, :1738-1777) or, where the log law would be ill-defined that close to the wall,
// at a reconstruction point further out along the facet normal (trilinear_interp_var, :1610-1680; the point and its cells come
// from the host, initibmwallfun :273-644 = udcore/facets.py), turns it into the wall-parallel speed (local_coords, :1833),
// gets the stress from the log law (neutral :1907, or with the Uno et al. 1995 stability functions on the facet temperature
// :1856) and takes its component along the grid's direction out of the cell's tendency.  One thread per boundary cell walks
// that cell's sections in the file's order, so the sum into the tendency is the reference's sum.
#include "udc_internal.h"
#include <algorithm>
#include <numeric>

namespace {

struct WfArgs {
  int ncell, grid, iwallmom, j0;
  const int *cell, *off, *comprec, *recids;
  const double *area, *dist, *norm, *z0, *z0h, *tsurf, *recpt, *tmask;
  const double *u0, *v0, *w0, *thl0, *zf, *zh;      // zf, zh: entry 0 = reference index 1
  double *rhs;
  double prt, fkar;      // &WALLS prandtlturb, fkar (src/modglobal.f90:304, 317)
  // latent part of wallfunheat (c grid, lmoist): 0 off, 1 prescribed flux per section (in qwall), 2 moist_flux on the facets' humidity
  int iwallmoist;
  const int *lgr;
  const double *qwall, *hurel, *resc, *ress, *qt0;
  double *rhsq;
  // lwritefac on an RK stage 3: facet of every section and this substep's sums (fa: tau of this grid's direction; fb, fc: htc, cth)
  const int *fac;
  double *fa, *fb, *fc;
};

__device__ __forceinline__ int wx(int i, int nx) { return i < 0 ? i + nx : (i >= nx ? i - nx : i); }
// field value at the reference's global (i, j, k), ghost ring included
__device__ __forceinline__ double at(const Geo &g, const double *f, int j0, int i, int j, int k) {
  return f[g.idx(wx(i - 1 + g.xg, g.nx), j - 1 - j0, k - 1)];      // (xg: open x boundaries, the device row starts at ib - 1)
}

// coordinates of the staggered grids (equidistant x, y): xh(i) = (i-1) dx, xf(i) = (i - 1/2) dx
__device__ __forceinline__ double trilinear(const Geo &g, const Metrics &m, const double *f, int j0, const int *cell, int sx, int sy_, const double *zg,
                                            const double *p) {
  const int i = cell[0], j = cell[1], k = cell[2];
  const double c000 = at(g, f, j0, i, j, k), c100 = at(g, f, j0, i + 1, j, k), c010 = at(g, f, j0, i, j + 1, k), c110 = at(g, f, j0, i + 1, j + 1, k);
  const double c001 = at(g, f, j0, i, j, k + 1), c101 = at(g, f, j0, i + 1, j, k + 1), c011 = at(g, f, j0, i, j + 1, k + 1),
               c111 = at(g, f, j0, i + 1, j + 1, k + 1);
  // sx / sy_: 1 = the half-level (h) grid, 0 = the full-level (f) grid
  const double x0 = sx ? (i - 1) * m.dx : (i - 0.5) * m.dx, x1 = sx ? i * m.dx : (i + 0.5) * m.dx;
  const double y0 = sy_ ? (j - 1) * m.dy : (j - 0.5) * m.dy, y1 = sy_ ? j * m.dy : (j + 0.5) * m.dy;
  const double z0 = zg[k - 1], z1 = zg[k];
  const double xd = (p[0] - x0) / (x1 - x0), yd = (p[1] - y0) / (y1 - y0), zd = (p[2] - z0) / (z1 - z0);
  return c000 * (1 - xd) * (1 - yd) * (1 - zd) + c100 * xd * (1 - yd) * (1 - zd) + c010 * (1 - xd) * yd * (1 - zd) + c110 * xd * yd * (1 - zd) +
         c001 * (1 - xd) * (1 - yd) * zd + c101 * xd * (1 - yd) * zd + c011 * (1 - xd) * yd * zd + c111 * xd * yd * zd;
}

// mom_transfer_coef_stability, :1856-1904
__device__ __forceinline__ double ctm_stability(double utan, double dist, double z0, double z0h, double Tair, double Tsurf, double prt, double fkar) {
  const double b1 = 9.4, b2 = 4.7, dm = 7.4, dh = 5.3, grav = 9.81;
  const double dT = Tair - Tsurf;
  const double Ribl0 = grav * dist * dT / (Tsurf * (utan * utan));
  const double logdz = log(dist / z0), logzh = log(z0 / z0h), sqdz = sqrt(dist / z0), fkar2 = fkar * fkar;
  double Fm, Fh;
  if (Ribl0 > 0.) {
    Fm = 1. / ((1. + b2 * Ribl0) * (1. + b2 * Ribl0));
    Fh = Fm;
  } else {
    const double cm = (dm * fkar2) / (logdz * logdz) * b1 * sqdz, ch = (dh * fkar2) / (logdz * logdz) * b1 * sqdz;
    Fm = 1. - (b1 * Ribl0) / (1. + cm * sqrt(fabs(Ribl0)));
    Fh = 1. - (b1 * Ribl0) / (1. + ch * sqrt(fabs(Ribl0)));
  }
  const double M = prt * logdz * sqrt(Fm) / Fh;
  const double Ribl1 = Ribl0 - Ribl0 * prt * logzh / (prt * logzh + M);
  if (Ribl1 > 0.) {
    Fm = 1. / ((1. + b2 * Ribl1) * (1. + b2 * Ribl1));
  } else {
    const double cm = (dm * fkar2) / (logdz * logdz) * b1 * sqdz;
    Fm = 1. - (b1 * Ribl1) / (1. + cm * sqrt(fabs(Ribl1)));
  }
  return fkar2 / (logdz * logdz) * Fm;
}

__global__ void ibm_wallfunmom_kernel(Geo g, Metrics m, WfArgs a) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= a.ncell) return;
  const int i = a.cell[3 * q], j = a.cell[3 * q + 1], k = a.cell[3 * q + 2], j0 = a.j0;
  const long c = g.idx(i - 1 + g.xg, j - 1 - j0, k - 1);
  const double eps1 = 1.e-10, fkar = a.fkar;
  const double vol = m.dx * m.dy * m.dzf[k];
  double t = a.rhs[c];
  for (int s = a.off[q]; s < a.off[q + 1]; ++s) {
    const double nrm[3] = {a.norm[3 * s], a.norm[3 * s + 1], a.norm[3 * s + 2]};
    const double z0 = a.z0[s];
    double uv[3], Tair = 0., dist;
    const bool comp = a.comprec[s] != 0;
    if (comp) {
      if (a.grid == 0) {      // interp_velocity_u
        uv[0] = at(g, a.u0, j0, i, j, k);
        uv[1] = 0.25 * (at(g, a.v0, j0, i, j, k) + at(g, a.v0, j0, i, j + 1, k) + at(g, a.v0, j0, i - 1, j, k) + at(g, a.v0, j0, i - 1, j + 1, k));
        uv[2] = 0.25 * (at(g, a.w0, j0, i, j, k) + at(g, a.w0, j0, i, j, k + 1) + at(g, a.w0, j0, i - 1, j, k) + at(g, a.w0, j0, i - 1, j, k + 1));
      } else {                // interp_velocity_v and _w are the same expressions
        uv[0] = 0.25 * (at(g, a.u0, j0, i, j, k) + at(g, a.u0, j0, i + 1, j, k) + at(g, a.u0, j0, i, j - 1, k) + at(g, a.u0, j0, i + 1, j - 1, k));
        uv[1] = at(g, a.v0, j0, i, j, k);
        uv[2] = 0.25 * (at(g, a.w0, j0, i, j, k) + at(g, a.w0, j0, i, j, k + 1) + at(g, a.w0, j0, i, j - 1, k) + at(g, a.w0, j0, i, j - 1, k + 1));
      }
      if (a.iwallmom == 2) {  // interp_temperature_*: the fluid one of the two cells, their mean when both are
        const int di = a.grid == 0 ? 1 : 0, dj = a.grid == 1 ? 1 : 0, dk = a.grid == 2 ? 1 : 0;
        const double m0 = a.tmask[2 * s], m1 = a.tmask[2 * s + 1];
        Tair = 0.5 * (at(g, a.thl0, j0, i, j, k) * m0 * (2. - m1) + at(g, a.thl0, j0, i - di, j - dj, k - dk) * m1 * (2. - m0));
      }
      dist = a.dist[s];
    } else {
      const double *p = a.recpt + 3 * s;
      const int *r = a.recids + 12 * s;
      uv[0] = trilinear(g, m, a.u0, j0, r, 1, 0, a.zf, p);
      uv[1] = trilinear(g, m, a.v0, j0, r + 3, 0, 1, a.zf, p);
      uv[2] = trilinear(g, m, a.w0, j0, r + 6, 0, 0, a.zh, p);
      if (a.iwallmom == 2) Tair = trilinear(g, m, a.thl0, j0, r + 9, 0, 0, a.zf, p);
      const double ex = p[0] - (i - 0.5) * m.dx, ey = p[1] - (j - 0.5) * m.dy, ez = p[2] - a.zf[k - 1];
      dist = a.dist[s] + sqrt(ex * ex + ey * ey + ez * ez);
    }
    if (log(dist / z0) <= 1.) continue;
    if (fabs(uv[0]) < eps1 && fabs(uv[1]) < eps1 && fabs(uv[2]) < eps1) continue;
    // local_coords: span = norm x u, strm = span x norm
    double sp[3] = {nrm[1] * uv[2] - nrm[2] * uv[1], nrm[2] * uv[0] - nrm[0] * uv[2], nrm[0] * uv[1] - nrm[1] * uv[0]};
    if (fabs(sp[0]) < eps1 && fabs(sp[1]) < eps1 && fabs(sp[2]) < eps1) continue;
    const double sn = sqrt(sp[0] * sp[0] + sp[1] * sp[1] + sp[2] * sp[2]);
    sp[0] /= sn; sp[1] /= sn; sp[2] /= sn;
    const double st[3] = {sp[1] * nrm[2] - sp[2] * nrm[1], sp[2] * nrm[0] - sp[0] * nrm[2], sp[0] * nrm[1] - sp[1] * nrm[0]};
    const double utan = uv[0] * st[0] + uv[1] * st[1] + uv[2] * st[2];
    double ctm;
    if (a.iwallmom == 2) ctm = ctm_stability(utan, dist, z0, a.z0h[s], Tair, a.tsurf[s], a.prt, a.fkar);
    else { const double l = fkar / log(dist / z0); ctm = l * l; }
    const double stress = ctm * (utan * utan);
    const double a_is = st[a.grid];                       // dot(dir, strm)
    double sd;
    if (comp) sd = a_is * stress;
    else {
      const double sx_ = a_is * nrm[0] * stress, sy_ = a_is * nrm[1] * stress, sz_ = a_is * nrm[2] * stress;
      sd = sqrt(sx_ * sx_ + sy_ * sy_ + sz_ * sz_);
    }
    sd = copysign(fabs(sd), uv[a.grid]);                  // sign(stress_dir, dot(uvec, dir))
    t = t - sd * a.area[s] / vol;
    if (a.fa) atomicAdd(&a.fa[a.fac[s] - 1], sd * a.area[s]);      // fac_tau_loc(fac), :1413
  }
  a.rhs[c] = t;
}

// heat_transfer_coef_flux, :1920-1986 -> flux [K m/s]; htc = flux / (|utan| dT) where that is defined, else 0 (:1975-1979)
__device__ __forceinline__ double heat_flux(double utan, double dist, double z0, double z0h, double Tair, double Tsurf, double prt, double fkar, double &htc,
                                            double &cth_out) {
  const double b1 = 9.4, b2 = 4.7, dm = 7.4, dh = 5.3, grav = 9.81;
  const double dT = Tair - Tsurf;
  const double Ribl0 = grav * dist * dT / (Tsurf * (utan * utan));
  const double logdz = log(dist / z0), logzh = log(z0 / z0h), sqdz = sqrt(dist / z0), fkar2 = fkar * fkar;
  double Fm, Fh;
  auto F = [&](double Ri) {
    if (Ri > 0.) {
      Fm = 1. / ((1. + b2 * Ri) * (1. + b2 * Ri));
      Fh = Fm;
    } else {
      const double cm = (dm * fkar2) / (logdz * logdz) * b1 * sqdz, ch = (dh * fkar2) / (logdz * logdz) * b1 * sqdz;
      Fm = 1. - (b1 * Ri) / (1. + cm * sqrt(fabs(Ri)));
      Fh = 1. - (b1 * Ri) / (1. + ch * sqrt(fabs(Ri)));
    }
  };
  F(Ribl0);
  double M = prt * logdz * sqrt(Fm) / Fh;
  const double Ribl1 = Ribl0 - Ribl0 * prt * logzh / (prt * logzh + M);
  F(Ribl1);
  M = prt * logdz * sqrt(Fm) / Fh;
  const double dTrough = dT * 1. / (prt * logzh / M + 1.);
  const double cth = fkar2 / (logdz * logdz) * Fh / prt;
  const double flux = fabs(utan) * cth * dTrough;
  htc = fabs(fabs(utan) * dT) > 0. ? flux / (fabs(utan) * dT) : 0.;
  cth_out = cth;
  return flux;
}

// wallfunheat (:1436-1607): the c-grid sections.  Sensible part from the facet temperatures (iwalltemp = 2) or prescribed
// (iwalltemp = 1); latent part on the vegetated facets (faclGR, :1556-1600): prescribed (iwallmoist = 1) or moist_flux (:1989) of the
// air's humidity against the facet's saturation humidity through the aerodynamic + canopy / soil resistances (iwallmoist = 2)
__global__ void ibm_wallfunheat_kernel(Geo g, Metrics m, WfArgs a) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= a.ncell) return;
  const int i = a.cell[3 * q], j = a.cell[3 * q + 1], k = a.cell[3 * q + 2], j0 = a.j0;
  const long c = g.idx(i - 1 + g.xg, j - 1 - j0, k - 1);
  const double eps1 = 1.e-10;
  const double vol = m.dx * m.dy * m.dzh[k];
  double t = a.rhs[c];
  const bool moist = a.iwallmoist > 0;
  double tq = moist ? a.rhsq[c] : 0.;
  for (int s = a.off[q]; s < a.off[q + 1]; ++s) {
    const double nrm[3] = {a.norm[3 * s], a.norm[3 * s + 1], a.norm[3 * s + 2]};
    const double z0 = a.z0[s];
    double uv[3], Tair, dist, qtair = 0.;
    if (a.comprec[s]) {      // interp_velocity_c, :1780-1791
      uv[0] = 0.5 * (at(g, a.u0, j0, i, j, k) + at(g, a.u0, j0, i + 1, j, k));
      uv[1] = 0.5 * (at(g, a.v0, j0, i, j, k) + at(g, a.v0, j0, i, j + 1, k));
      uv[2] = 0.5 * (at(g, a.w0, j0, i, j, k) + at(g, a.w0, j0, i, j, k + 1));
      Tair = at(g, a.thl0, j0, i, j, k);
      if (moist) qtair = at(g, a.qt0, j0, i, j, k);
      dist = a.dist[s];
    } else {
      const double *p = a.recpt + 3 * s;
      const int *r = a.recids + 12 * s;
      uv[0] = trilinear(g, m, a.u0, j0, r, 1, 0, a.zf, p);
      uv[1] = trilinear(g, m, a.v0, j0, r + 3, 0, 1, a.zf, p);
      uv[2] = trilinear(g, m, a.w0, j0, r + 6, 0, 0, a.zh, p);
      Tair = trilinear(g, m, a.thl0, j0, r + 9, 0, 0, a.zf, p);
      if (moist) qtair = trilinear(g, m, a.qt0, j0, r + 9, 0, 0, a.zf, p);
      const double ex = p[0] - (i - 0.5) * m.dx, ey = p[1] - (j - 0.5) * m.dy, ez = p[2] - a.zf[k - 1];
      dist = a.dist[s] + sqrt(ex * ex + ey * ey + ez * ez);
    }
    if (log(dist / z0) <= 1.) continue;
    if (fabs(uv[0]) < eps1 && fabs(uv[1]) < eps1 && fabs(uv[2]) < eps1) continue;
    double sp[3] = {nrm[1] * uv[2] - nrm[2] * uv[1], nrm[2] * uv[0] - nrm[0] * uv[2], nrm[0] * uv[1] - nrm[1] * uv[0]};
    if (fabs(sp[0]) < eps1 && fabs(sp[1]) < eps1 && fabs(sp[2]) < eps1) continue;
    const double sn = sqrt(sp[0] * sp[0] + sp[1] * sp[1] + sp[2] * sp[2]);
    sp[0] /= sn; sp[1] /= sn; sp[2] /= sn;
    const double st[3] = {sp[1] * nrm[2] - sp[2] * nrm[1], sp[2] * nrm[0] - sp[0] * nrm[2], sp[0] * nrm[1] - sp[1] * nrm[0]};
    const double utan = uv[0] * st[0] + uv[1] * st[1] + uv[2] * st[2];
    // iwalltemp = 1: the prescribed flux of the facet's direction rides in the slot of the facet temperature
    double htc = 0., cth = 0.;
    double flux = a.iwallmom == 1 ? a.tsurf[s] : heat_flux(utan, dist, z0, a.z0h[s], Tair, a.tsurf[s], a.prt, a.fkar, htc, cth);
    t = t - flux * a.area[s] / vol;
    if (a.fb && a.iwallmom == 2) { atomicAdd(&a.fc[a.fac[s] - 1], cth * a.area[s]); atomicAdd(&a.fb[a.fac[s] - 1], htc * a.area[s]); }      // :1540-1541
    if (moist && a.lgr[s]) {
      if (a.iwallmoist == 1) flux = a.qwall[s];
      else if (fabs(htc * fabs(utan)) > 0.) {      // (else the reference's `flux` still holds the sensible one, which is then zero)
        const double resa = 1. / (htc * fabs(utan)), cveg = 0.8;
        const double qw = a.qwall[s];
        flux = fmin(0., cveg * (qtair - qw) / (resa + a.resc[s]) + (1 - cveg) * (qtair - qw * a.hurel[s]) / (resa + a.ress[s]));
      }
      tq = tq - flux * a.area[s] / vol;
    }
  }
  a.rhs[c] = t;
  if (moist) a.rhsq[c] = tq;
}

// lwritefac: fac_pres_loc / fac_pres2_loc of wallfunheat (:1475-1476) -- pres0 of the boundary cell of EVERY c section, skipped ones included
__global__ void ibm_fac_pres_kernel(Geo g, int n, int j0, const int *__restrict__ cell, const int *__restrict__ fac, const double *__restrict__ area,
                                    const double *__restrict__ pres0, double *__restrict__ fp, double *__restrict__ fp2) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n) return;
  const double p = pres0[g.idx(cell[3 * q] - 1 + g.xg, cell[3 * q + 1] - 1 - j0, cell[3 * q + 2] - 1)];
  atomicAdd(&fp[fac[q] - 1], p * area[q]);
  atomicAdd(&fp2[fac[q] - 1], p * p * area[q]);
}
// ... / faca (:1418, 1596-1599) and the running time integrals of ibmwallfun (:1247-1254): av += dt * value
__global__ void ibm_fac_finish_kernel(int n, double dt, const double *__restrict__ now, const double *__restrict__ faca, double *__restrict__ av) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= 7 * n) return;
  av[q] = av[q] + dt * (now[q] / faca[q % n]);
}

template <class T>
int upload(T **dst, const std::vector<T> &v) {
  if (*dst) { hipFree(*dst); *dst = nullptr; }
  if (v.empty()) return 0;
  HIP_OK(hipMalloc(dst, sizeof(T) * v.size()));
  HIP_OK(hipMemcpy(*dst, v.data(), sizeof(T) * v.size(), hipMemcpyHostToDevice));
  return 0;
}

}  // namespace

extern "C" int udc_set_ibm_wallfun(udc_handle *h, int iwallmom, double prandtlturb, const double *zf, const double *zh) {
  if (!h) { udc_set_error("null handle"); return 1; }
  HIP_OK(hipSetDevice(h->device));
  if (udc_flush_pending(h)) return 1;
  if (iwallmom < 1 || iwallmom > 3) { udc_set_error("udc_set_ibm_wallfun: iwallmom must be 1 (none), 2 (stability functions) or 3 (neutral)"); return 1; }
  if (!zf || !zh || !(prandtlturb > 0.)) { udc_set_error("udc_set_ibm_wallfun: zf, zh (levels 1..ktot+1) and prandtlturb > 0 are needed"); return 1; }
  if (iwallmom == 2 && ((int)h->fields.size() <= UDC_THL0 || !h->fields[UDC_THL0])) {
    udc_set_error("udc_set_ibm_wallfun: iwallmom = 2 judges the stability on the air temperature: call udc_set_tempeq first (or use iwallmom = 3)");
    return 1;
  }
  h->ibm_iwallmom = iwallmom;
  h->ibm_prt = prandtlturb;
  {
    const int n = h->g.nz + 1;
    std::vector<double> z(zf, zf + n);
    z.insert(z.end(), zh, zh + n);
    HIP_OK(hipStreamSynchronize(h->stream));
    if (upload(&h->ibm_zgrid, z)) return 1;
  }
  return 0;
}

extern "C" int udc_set_ibm_sections(udc_handle *h, int grid, int n, const int *cell, const double *area, const double *dist, const double *norm,
                                    const double *z0, const double *z0h, const double *tsurf, const int *comprec, const double *recpt,
                                    const int *recids, const double *tmask) {
  if (!h) { udc_set_error("null handle"); return 1; }
  HIP_OK(hipSetDevice(h->device));
  if (udc_flush_pending(h)) return 1;
  if (grid < 0 || grid > 3) { udc_set_error("udc_set_ibm_sections: grid 0 (u), 1 (v), 2 (w) or 3 (c, for the heat wall function)"); return 1; }
  if (n < 0 || (n && (!cell || !area || !dist || !norm || !z0 || !z0h || !tsurf || !comprec || !recpt || !recids || !tmask))) {
    udc_set_error("udc_set_ibm_sections: null array"); return 1;
  }
  const Geo &g = h->g;
  const int j0 = h->cfg.rank * g.ny;
  // this slab's sections, grouped by cell, the file's order kept inside a cell and between first appearances of cells
  std::vector<int> mine;
  std::vector<char> fallback(n, 0);
  for (int s = 0; s < n; ++s) {
    const int i = cell[3 * s], j = cell[3 * s + 1], k = cell[3 * s + 2];
    if (i < 1 || i > g.nx - 2 * g.xg || j < 1 || j > h->jtot || k < 1 || k > g.nz) { udc_set_error("udc_set_ibm_sections: section %d: cell (%d %d %d) outside the domain", s + 1, i, j, k); return 1; }
    if (!(z0[s] > 0.) || !(dist[s] > 0.)) { udc_set_error("udc_set_ibm_sections: section %d: z0 and the wall distance must be positive", s + 1); return 1; }
    const bool is_mine = j > j0 && j <= j0 + g.ny;
    if (!comprec[s] && is_mine)
      for (int q = 0; q < 4; ++q) {
        const int *r = recids + 12 * s + 3 * q;
        if (r[2] < 1 || r[2] > g.nz) {
          udc_set_error("udc_set_ibm_sections: section %d: reconstruction cell (%d %d %d) outside the levels of the domain", s + 1, r[0], r[1], r[2]);
          return 1;
        }
        // a reconstruction cell beyond the rows this slab can read (its own and one ghost row either side): the reference then
        // falls back to the boundary point itself -- lcomprec_loc = .true., src/modibm.f90:606-622 -- and so does this slab
        if (r[0] < 0 || r[0] > g.nx - 2 * g.xg || r[1] < j0 || r[1] > j0 + g.ny) fallback[s] = 1;
      }
    if (is_mine) mine.push_back(s);
  }
  {
    // the reference's `p` counter and warning (src/modibm.f90:608-630).  Its bound is one row wider at the slab's upper end
    // (zend + 1; the interpolation then reads row zend + 2): here a reconstruction cell must lie where the cell AND its +1
    // neighbour are within the one ghost row the velocities exchange ("facet sections next to a slab boundary")
    long nover = 0;
    for (int s : mine) nover += fallback[s] ? 1 : 0;
    if (nover > 0)
      fprintf(stderr, " WARNING libudcore udc_set_ibm_sections: rank %d overrode %ld facet section(s) of grid %d to simple reconstruction "
                      "because a reconstruction cell falls outside the rows this slab can read.\n", h->cfg.rank, nover, grid);
    h->ibm_sec[grid].noverride = (int)nover;
  }
  auto key = [&](int s) { return ((long)cell[3 * s + 2] * h->jtot + cell[3 * s + 1]) * g.nx + cell[3 * s]; };
  std::stable_sort(mine.begin(), mine.end(), [&](int x, int y) { return key(x) < key(y); });
  std::vector<int> cells, off, comp, rid;
  std::vector<double> ar, di, nr, zz, zh_, ts, rp, tm;
  for (size_t q = 0; q < mine.size(); ++q) {
    const int s = mine[q];
    if (q == 0 || key(s) != key(mine[q - 1])) {
      off.push_back((int)q);
      cells.insert(cells.end(), cell + 3 * s, cell + 3 * s + 3);
    }
    comp.push_back(fallback[s] ? 1 : comprec[s]);
    rid.insert(rid.end(), recids + 12 * s, recids + 12 * s + 12);
    ar.push_back(area[s]); di.push_back(dist[s]); zz.push_back(z0[s]); zh_.push_back(z0h[s]); ts.push_back(tsurf[s]);
    nr.insert(nr.end(), norm + 3 * s, norm + 3 * s + 3);
    rp.insert(rp.end(), recpt + 3 * s, recpt + 3 * s + 3);
    tm.insert(tm.end(), tmask + 2 * s, tmask + 2 * s + 2);
  }
  off.push_back((int)mine.size());
  udc_handle::IbmSections &S = h->ibm_sec[grid];
  HIP_OK(hipStreamSynchronize(h->stream));
  S.ncell = (int)cells.size() / 3;
  S.nsec = (int)mine.size();
  S.order = mine; S.nglobal = n;
  if (upload(&S.cell, cells) || upload(&S.off, off) || upload(&S.comprec, comp) || upload(&S.recids, rid) || upload(&S.area, ar) ||
      upload(&S.dist, di) || upload(&S.norm, nr) || upload(&S.z0, zz) || upload(&S.z0h, zh_) || upload(&S.tsurf, ts) || upload(&S.recpt, rp) ||
      upload(&S.tmask, tm))
    return 1;
  return 0;
}

// wallfunmom on the u, v, w tendencies (ibmwallfun, src/modibm.f90:1183-1194), before the diffusion corrections
int k_ibm_wallfunmom(udc_handle *h) {
  if (h->ibm_iwallmom <= 1) return 0;
  const Geo &g = h->g;
  for (int q = 0; q < 3; ++q) {
    const udc_handle::IbmSections &S = h->ibm_sec[q];
    if (!S.ncell) continue;
    WfArgs a;
    a.ncell = S.ncell; a.grid = q; a.iwallmom = h->ibm_iwallmom; a.j0 = h->cfg.rank * g.ny;
    a.cell = S.cell; a.off = S.off; a.comprec = S.comprec; a.recids = S.recids;
    a.area = S.area; a.dist = S.dist; a.norm = S.norm; a.z0 = S.z0; a.z0h = S.z0h; a.tsurf = S.tsurf; a.recpt = S.recpt; a.tmask = S.tmask;
    a.u0 = h->fields[UDC_U0]; a.v0 = h->fields[UDC_V0]; a.w0 = h->fields[UDC_W0];
    a.thl0 = h->ibm_iwallmom == 2 ? h->fields[UDC_THL0] : nullptr;
    a.zf = h->ibm_zgrid; a.zh = h->ibm_zgrid + (g.nz + 1);
    a.rhs = h->fields[UDC_UP + q];
    a.prt = h->ibm_prt; a.fkar = h->fkar; a.iwallmoist = 0;
    const bool smp = h->fac_sample_dt >= 0. && h->fac_now && S.fac;
    a.fac = smp ? S.fac : nullptr; a.fa = smp ? h->fac_now + (size_t)q * h->fac_n : nullptr; a.fb = a.fc = nullptr;
    hipLaunchKernelGGL(ibm_wallfunmom_kernel, dim3((unsigned)((S.ncell + 127) / 128)), dim3(128), 0, h->stream, g, h->m, a);
  }
  HIP_OK(hipGetLastError());
  return 0;
}

extern "C" int udc_set_ibm_wallheat(udc_handle *h, int iwalltemp) {
  if (!h) { udc_set_error("null handle"); return 1; }
  HIP_OK(hipSetDevice(h->device));
  if (udc_flush_pending(h)) return 1;
  if (iwalltemp < 0 || iwalltemp > 2) { udc_set_error("udc_set_ibm_wallheat: 0 (off: adiabatic walls), 1 (prescribed fluxes in the sections' tsurf) or 2 (facet temperatures)"); return 1; }
  if (iwalltemp && ((int)h->fields.size() <= UDC_THL0 || !h->fields[UDC_THL0])) { udc_set_error("udc_set_ibm_wallheat: call udc_set_tempeq first"); return 1; }
  if (iwalltemp && !h->ibm_zgrid) { udc_set_error("udc_set_ibm_wallheat: call udc_set_ibm_wallfun first (level coordinates, prandtlturb)"); return 1; }
  h->ibm_iwalltemp = iwalltemp;
  return 0;
}

// Latent part of wallfunheat (src/modibm.f90:1556-1600): per c-grid section, in the order udc_set_ibm_sections(grid 3) was given them:
// lgr -- the facet is vegetated (faclGR; only those exchange moisture); iwallmoist = 1: qwall = the prescribed flux of the facet's
// direction (bcqfxm ...); iwallmoist = 2: qwall = facqsat, hurel = fachurel, resc / ress = facf(:, 4) / facf(:, 5)
extern "C" int udc_set_ibm_wallmoist(udc_handle *h, int iwallmoist, int n, const int *lgr, const double *qwall, const double *hurel,
                                     const double *resc, const double *ress) {
  if (!h) { udc_set_error("null handle"); return 1; }
  HIP_OK(hipSetDevice(h->device));
  if (udc_flush_pending(h)) return 1;
  if (iwallmoist < 0 || iwallmoist > 2) { udc_set_error("udc_set_ibm_wallmoist: 0 (off: impermeable walls), 1 (prescribed fluxes) or 2 (moist_flux on the facets' humidity)"); return 1; }
  udc_handle::IbmSections &S = h->ibm_sec[3];
  if (!iwallmoist) { h->ibm_iwallmoist = 0; return 0; }
  if (!h->lmoist) { udc_set_error("udc_set_ibm_wallmoist: call udc_set_moisture first"); return 1; }
  if (iwallmoist == 2 && h->ibm_iwalltemp != 2) {
    udc_set_error("udc_set_ibm_wallmoist: iwallmoist = 2 takes its aerodynamic resistance from the heat transfer coefficient of iwalltemp = 2 "
                  "(undefined in the reference otherwise, src/modibm.f90:1571-1574): call udc_set_ibm_wallheat(2) first");
    return 1;
  }
  if (h->ibm_iwalltemp < 1) { udc_set_error("udc_set_ibm_wallmoist: the latent flux is part of wallfunheat: call udc_set_ibm_wallheat first"); return 1; }
  if (n != S.nglobal) { udc_set_error("udc_set_ibm_wallmoist: %d sections, udc_set_ibm_sections(grid 3) was given %d", n, S.nglobal); return 1; }
  if (n && (!lgr || !qwall || (iwallmoist == 2 && (!hurel || !resc || !ress)))) { udc_set_error("udc_set_ibm_wallmoist: null array"); return 1; }
  std::vector<int> lg; std::vector<double> qw, hu, rc, rs;
  for (int s : S.order) {
    lg.push_back(lgr[s]); qw.push_back(qwall[s]);
    hu.push_back(iwallmoist == 2 ? hurel[s] : 0.); rc.push_back(iwallmoist == 2 ? resc[s] : 0.); rs.push_back(iwallmoist == 2 ? ress[s] : 0.);
  }
  HIP_OK(hipStreamSynchronize(h->stream));
  if (upload(&S.lgr, lg) || upload(&S.qwall, qw) || upload(&S.hurel, hu) || upload(&S.resc, rc) || upload(&S.ress, rs)) return 1;
  h->ibm_iwallmoist = iwallmoist;
  return 0;
}

// wallfunheat on thlp (ibmwallfun, src/modibm.f90:1220-1231), after the momentum corrections and before diffc_corr
int k_ibm_wallfunheat(udc_handle *h) {
  if (h->ibm_iwalltemp < 1) return 0;
  const udc_handle::IbmSections &S = h->ibm_sec[3];
  if (!S.ncell) return 0;
  const Geo &g = h->g;
  WfArgs a;
  a.ncell = S.ncell; a.grid = 3; a.iwallmom = h->ibm_iwalltemp; a.j0 = h->cfg.rank * g.ny;      // (iwallmom carries iwalltemp here)
  a.cell = S.cell; a.off = S.off; a.comprec = S.comprec; a.recids = S.recids;
  a.area = S.area; a.dist = S.dist; a.norm = S.norm; a.z0 = S.z0; a.z0h = S.z0h; a.tsurf = S.tsurf; a.recpt = S.recpt; a.tmask = S.tmask;
  a.u0 = h->fields[UDC_U0]; a.v0 = h->fields[UDC_V0]; a.w0 = h->fields[UDC_W0]; a.thl0 = h->fields[UDC_THL0];
  a.zf = h->ibm_zgrid; a.zh = h->ibm_zgrid + (g.nz + 1);
  a.rhs = h->fields[UDC_THLP];
  a.prt = h->ibm_prt; a.fkar = h->fkar;
  a.iwallmoist = (h->ibm_iwallmoist && S.lgr) ? h->ibm_iwallmoist : 0;
  a.lgr = S.lgr; a.qwall = S.qwall; a.hurel = S.hurel; a.resc = S.resc; a.ress = S.ress;
  a.qt0 = a.iwallmoist ? h->fields[UDC_QT0] : nullptr; a.rhsq = a.iwallmoist ? h->fields[UDC_QTP] : nullptr;
  const bool smp = h->fac_sample_dt >= 0. && h->fac_now && S.fac;
  a.fac = smp ? S.fac : nullptr; a.fa = nullptr;
  a.fb = smp ? h->fac_now + (size_t)5 * h->fac_n : nullptr; a.fc = smp ? h->fac_now + (size_t)6 * h->fac_n : nullptr;
  hipLaunchKernelGGL(ibm_wallfunheat_kernel, dim3((unsigned)((S.ncell + 127) / 128)), dim3(128), 0, h->stream, g, h->m, a);
  HIP_OK(hipGetLastError());
  return 0;
}

// ---- lwritefac: facet output of ibmwallfun (src/modibm.f90:196-243, 1246-1282, 1413-1430, 1475-1476, 1540-1541, 1595-1605)
extern "C" int udc_set_ibm_facet_output(udc_handle *h, int nfcts, const double *faca, const int *nsec, const int *fac_u, const int *fac_v,
                                        const int *fac_w, const int *fac_c, int npres, const int *pcell, const double *parea, const int *pfac) {
  if (!h) { udc_set_error("null handle"); return 1; }
  HIP_OK(hipSetDevice(h->device));
  if (udc_flush_pending(h)) return 1;
  if (nfcts < 1 || !faca || !nsec) { udc_set_error("udc_set_ibm_facet_output: facets and their areas are needed"); return 1; }
  const int *facs[4] = {fac_u, fac_v, fac_w, fac_c};
  HIP_OK(hipStreamSynchronize(h->stream));
  for (int q = 0; q < 4; ++q) {
    udc_handle::IbmSections &S = h->ibm_sec[q];
    if (!S.nglobal && !nsec[q]) continue;
    if (nsec[q] != S.nglobal || (nsec[q] && !facs[q])) { udc_set_error("udc_set_ibm_facet_output: grid %d: %d facet ids, udc_set_ibm_sections was given %d sections", q, nsec[q], S.nglobal); return 1; }
    std::vector<int> f;
    for (int s_ : S.order) {
      if (facs[q][s_] < 1 || facs[q][s_] > nfcts) { udc_set_error("udc_set_ibm_facet_output: facet %d of %d", facs[q][s_], nfcts); return 1; }
      f.push_back(facs[q][s_]);
    }
    if (upload(&S.fac, f)) return 1;
  }
  // the c sections that carry the pressure: this slab's
  const Geo &g = h->g;
  const int j0 = h->cfg.rank * g.ny;
  std::vector<int> pc, pf; std::vector<double> pa;
  for (int s_ = 0; s_ < npres; ++s_) {
    const int i = pcell[3 * s_], j = pcell[3 * s_ + 1], k = pcell[3 * s_ + 2];
    if (i < 1 || i > g.nx - 2 * g.xg || j < 1 || j > h->jtot) continue;      // (a boundary point no rank owns)
    if (k < 1 || k > g.nz || pfac[s_] < 1 || pfac[s_] > nfcts) { udc_set_error("udc_set_ibm_facet_output: pressure section %d out of range", s_ + 1); return 1; }
    if (j <= j0 || j > j0 + g.ny) continue;
    pc.insert(pc.end(), pcell + 3 * s_, pcell + 3 * s_ + 3); pf.push_back(pfac[s_]); pa.push_back(parea[s_]);
  }
  h->fac_npres = (int)pf.size();
  if (upload(&h->fac_pcell, pc) || upload(&h->fac_pfac, pf) || upload(&h->fac_parea, pa)) return 1;
  std::vector<double> fa(faca, faca + nfcts), z((size_t)7 * nfcts, 0.);
  if (upload(&h->fac_area, fa) || upload(&h->fac_now, z) || upload(&h->fac_av, z)) return 1;
  h->fac_n = nfcts;
  h->fac_sample_dt = -1.;
  return 0;
}
// the next udc_ibmwallfun is an RK stage 3 with time step dt: it samples the facets (once)
extern "C" int udc_ibm_facet_sample(udc_handle *h, double dt) {
  if (!h || !h->fac_now) { udc_set_error("udc_ibm_facet_sample: udc_set_ibm_facet_output first"); return 1; }
  h->fac_sample_dt = dt;      // (no flush: with deferred execution the recorded ibmwallfun of this substep consumes it)
  return 0;
}
// the running time integrals [7][nfcts] (tau_x, tau_y, tau_z, pres, pres2, htc, cth), and back to zero if asked
extern "C" int udc_ibm_facet_get(udc_handle *h, double *out, int reset) {
  if (!h || !out || !h->fac_av) { udc_set_error("udc_ibm_facet_get: udc_set_ibm_facet_output first"); return 1; }
  HIP_OK(hipSetDevice(h->device));
  if (udc_flush_pending(h)) return 1;
  const size_t nb = sizeof(double) * 7 * h->fac_n;
  HIP_OK(hipMemcpyAsync(out, h->fac_av, nb, hipMemcpyDeviceToHost, h->stream));
  if (reset) HIP_OK(hipMemsetAsync(h->fac_av, 0, nb, h->stream));
  HIP_OK(hipStreamSynchronize(h->stream));
  return 0;
}
// around the wall functions of one ibmwallfun: zero this substep's sums / the pressure sections, the reduction over the slabs, the integrals
int k_ibm_facet_begin(udc_handle *h) {
  if (h->fac_sample_dt < 0. || !h->fac_now) return 0;
  HIP_OK(hipMemsetAsync(h->fac_now, 0, sizeof(double) * 7 * h->fac_n, h->stream));
  return 0;
}
int k_ibm_facet_end(udc_handle *h) {
  if (h->fac_sample_dt < 0. || !h->fac_now) return 0;
  const Geo &g = h->g;
  if (h->fac_npres)
    hipLaunchKernelGGL(ibm_fac_pres_kernel, dim3((unsigned)((h->fac_npres + 127) / 128)), dim3(128), 0, h->stream, g, h->fac_npres, h->cfg.rank * g.ny,
                       (const int *)h->fac_pcell, (const int *)h->fac_pfac, (const double *)h->fac_parea, (const double *)h->fields[UDC_PRES0],
                       h->fac_now + (size_t)3 * h->fac_n, h->fac_now + (size_t)4 * h->fac_n);
  HIP_OK(hipGetLastError());
  if (comm_allreduce(h, h->fac_now, 7 * h->fac_n, 1)) return 1;      // MPI_ALLREDUCE(..., MPI_SUM) over the slabs, :1419, 1601-1604
  hipLaunchKernelGGL(ibm_fac_finish_kernel, dim3((unsigned)((7 * h->fac_n + 127) / 128)), dim3(128), 0, h->stream, h->fac_n, h->fac_sample_dt,
                     (const double *)h->fac_now, (const double *)h->fac_area, h->fac_av);
  HIP_OK(hipGetLastError());
  h->fac_sample_dt = -1.;
  return 0;
}

void ibm_wf_destroy(udc_handle *h) {
  for (double **p : {&h->fac_now, &h->fac_av, &h->fac_area, &h->fac_parea}) if (*p) { hipFree(*p); *p = nullptr; }
  for (int **p : {&h->fac_pcell, &h->fac_pfac}) if (*p) { hipFree(*p); *p = nullptr; }
  for (auto &S : h->ibm_sec) {
    for (int **p : {&S.cell, &S.off, &S.comprec, &S.recids, &S.lgr, &S.fac}) if (*p) { hipFree(*p); *p = nullptr; }
    for (double **p : {&S.area, &S.dist, &S.norm, &S.z0, &S.z0h, &S.tsurf, &S.recpt, &S.tmask, &S.qwall, &S.hurel, &S.resc, &S.ress}) if (*p) { hipFree(*p); *p = nullptr; }
    S.ncell = S.nsec = 0;
  }
  if (h->ibm_zgrid) { hipFree(h->ibm_zgrid); h->ibm_zgrid = nullptr; }
}
