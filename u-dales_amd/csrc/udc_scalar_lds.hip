// Transported scalars, LDS-staged and marching in k: the fused advection + diffusion sweep of one scalar
// (advecc_kappa / advecc_2nd + diffc, arithmetic in udc_scalar_arith.h, identical to the direct-load kernel's).
//
// The direct-load kernel (udc_scalar.hip) fetches 26 values per cell from global memory (13 of c, 7 of the diffusivity,
// 6 velocities); most hit L1/L2, but every one costs address arithmetic and a vector-memory issue slot, and with the
// kappa scheme it evaluates the limiter six times per cell (0.34-0.36 of the HBM roofline at 512x512x256).  Here a
// workgroup owns a 32 x 8 column of cells and walks KC levels upward like the momentum sweep (udc_mom_lds.hip): per
// level it stages the (32+4) x (8+4) tile of c of plane k+3 and the (32+2) x (8+2) tile of the diffusivity of plane
// k+2 into LDS (6 and 4 rotating plane buffers: c is read at k-2..k+2, the diffusivity at k-1..k+1), prefetched into
// registers one level ahead.  Two kernels: scalar_lds_kernel (2nd-order central advection: thl, qt) takes the six face
// velocities from global memory; scalar_kappa_faces_kernel (kappa scheme) in addition evaluates every face once and
// shares it between the two cells it borders.  Measured (MI355X): cd2 0.229 -> 0.215 ms at 256^3, kappa 1.135 ->
// 0.925 ms at 512x512x256 (0.355 -> 0.435 of the roofline).  UDC_SCALAR_LDS=0 selects the direct-load kernel.  thl and qt (both
// advecc_2nd, both diffused with ekh) share one launch of scalar_lds_kernel (NS = 2; UDC_SCALAR_PAIR=0: one by one).
#include "udc_internal.h"
#include "udc_scalar_arith.h"
#include <cstdlib>

namespace {

constexpr int MX = 32, MY = 8, NT = MX * MY;
constexpr int CX = MX + 4, CY = MY + 4, CN = CX * CY;      // c tile with a halo of 2: 36 x 12 = 432
constexpr int EX = MX + 2, EY = MY + 2, EN = EX * EY;      // diffusivity tile with a halo of 1: 34 x 10 = 340
constexpr int NCB = 6, NEB = 4;                            // rotating plane buffers
static_assert(CN <= 2 * NT && EN <= 2 * NT, "at most two staged elements per thread and field");

struct LdsAcc {
  const double *cb[5];      // c planes k-2 .. k+2, pointing at this thread's centre element
  const double *eb[3];      // diffusivity planes k-1 .. k+1
  __device__ __forceinline__ double c(int di, int dj, int dk) const { return cb[dk + 2][dj * CX + di]; }
  __device__ __forceinline__ double e(int di, int dj, int dk) const { return eb[dk + 1][dj * EX + di]; }
};

// NS: scalars per launch.  Two fields that share the advecting velocities and the diffusivity (thl and qt, both advecc_2nd + diffc
// with ekh) go through one sweep: u0, v0, w0 and ekh are fetched once for both, 64 B per cell instead of 2 x 48.
template <int ADV, bool LES, bool FRESH, int NS>
__global__ __launch_bounds__(NT, 3) void scalar_lds_kernel(Geo g, int gx, int tiles, Metrics m, double cekh, double dfac,
    const double *__restrict__ u, const double *__restrict__ v, const double *__restrict__ w, const double *__restrict__ ekh,
    const double *__restrict__ c_a, double *__restrict__ cp_a, const double *__restrict__ c_b, double *__restrict__ cp_b, int gh, int kc) {
  __shared__ double sc[NS][NCB][CN];
  __shared__ double se[LES ? NEB : 1][LES ? EN : 1];
  __shared__ double smet[2][NSCALMET + 1];       // level metrics of levels k and k+1 (udc_scalar_arith.h)
  const double *const cs[2] = {c_a, c_b};
  double *const cps[2] = {cp_a, cp_b};
  const unsigned Lb = blockIdx.x;
  const int chunk = Lb / tiles;
  const unsigned lp = Lb - (unsigned)chunk * tiles;
  unsigned tt = lp;
  if ((tiles & 7) == 0) tt = (lp & 7u) * (tiles >> 3) + (lp >> 3);      // XCD-aware, as tile_decode
  const int by = tt / gx, bx = tt - by * gx;
  const int i0 = bx * MX, j0 = by * MY;
  const int tx = threadIdx.x, ty = threadIdx.y, tid = ty * MX + tx;
  const int i = i0 + tx, j = j0 + ty;
  const bool inside = i < g.nx && j < g.ny;
  const int k0 = chunk * kc, k1 = min(k0 + kc, g.nz);
  const bool met_thread = tid < NSCALMET;
  double mreg = 0.;
  if (met_thread) {
    smet[k0 & 1][tid] = scalmet_load(m, tid, k0 + 1, g.nz);
    if (k0 + 1 < k1) mreg = scalmet_load(m, tid, k0 + 2, g.nz);
  }

  // the (up to) two elements of each tile this thread stages: element e -> (lx, ly) row-major in the tile
  long coff[2], eoff[2];
  int cl[2], el[2];
  bool chas[2], ehas[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int e = tid + q * NT;
    chas[q] = e < CN;
    {
      const int ly = e / CX, lx = e - ly * CX;
      int gi = (i0 - 2 + lx) % g.nx; if (gi < 0) gi += g.nx;
      const int gj = min(j0 - 2 + ly, g.ny + HY - 1);
      coff[q] = (long)gi + (long)g.sy * (gj + HY);
      cl[q] = e;
    }
    ehas[q] = LES && e < EN;
    {
      const int ly = e / EX, lx = e - ly * EX;
      int gi = (i0 - 1 + lx) % g.nx; if (gi < 0) gi += g.nx;
      const int gj = min(j0 - 1 + ly, g.ny + HY - 1);
      eoff[q] = (long)gi + (long)g.sy * (gj + HY);
      el[q] = e;
    }
  }
  // planes below -HZ / above nz-1+HZ do not exist: c is read at k-2 >= -2 and k+2 <= nz+1, both inside the padding
  double rc[NS][2], re[2];
  auto load_c = [&](int k) {
    const long pb = g.sz * (long)(k + HZ);
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
      for (int q = 0; q < 2; ++q) rc[s][q] = chas[q] ? cs[s][pb + coff[q]] : 0.;
  };
  auto commit_c = [&](int buf) {
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
      for (int q = 0; q < 2; ++q) if (chas[q]) sc[s][buf][cl[q]] = rc[s][q];
  };
  auto load_e = [&](int k) {
    if (!LES) return;
    const long pb = g.sz * (long)(k + HZ);
#pragma unroll
    for (int q = 0; q < 2; ++q) re[q] = ehas[q] ? ekh[pb + eoff[q]] : 0.;
  };
  auto commit_e = [&](int buf) {
    if (!LES) return;
#pragma unroll
    for (int q = 0; q < 2; ++q) if (ehas[q]) se[buf][el[q]] = re[q];
  };
  // prologue: c planes k0-2 .. k0+2 -> buffers 0..4, diffusivity planes k0-1 .. k0+1 -> buffers 0..2
  for (int d = 0; d < 5; ++d) { load_c(k0 - 2 + d); commit_c(d); }
  for (int d = 0; d < 3; ++d) { load_e(k0 - 1 + d); commit_e(d); }
  if (k0 + 1 < k1) { load_c(k0 + 3); load_e(k0 + 2); }
  int cb0 = 0, eb0 = 0;      // buffer holding plane k-2 (c) / k-1 (diffusivity)
  const int own_c = (ty + 2) * CX + (tx + 2), own_e = (ty + 1) * EX + (tx + 1);
  const long own = (long)(i % g.nx) + (long)g.sy * (min(j, g.ny + HY - 1) + HY);
  const long xp1 = (long)((i + 1) % g.nx) + (long)g.sy * (min(j, g.ny + HY - 1) + HY);
  double wl = inside ? w[g.sz * (long)(k0 + HZ) + own] : 0.;
  for (int k = k0; k < k1; ++k) {
    __syncthreads();
    if (k + 1 < k1) {
      commit_c((cb0 + 5) % NCB);
      commit_e((eb0 + 3) % NEB);
      if (met_thread) smet[(k + 1) & 1][tid] = mreg;
      if (k + 2 < k1) {
        if (met_thread) mreg = scalmet_load(m, tid, k + 3, g.nz);
        load_c(k + 4); load_e(k + 3);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    const ScalMetLds lm{smet[k & 1]};
    if (inside) {
      const long pb = g.sz * (long)(k + HZ);
      const double ul = u[pb + own], uh = u[pb + xp1], vl = v[pb + own], vh = v[pb + own + g.sy];
      const double wh = w[pb + own + g.sz];
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        LdsAcc A;
#pragma unroll
        for (int d = 0; d < 5; ++d) A.cb[d] = sc[s][(cb0 + d) % NCB] + own_c;
#pragma unroll
        for (int d = 0; d < 3; ++d) A.eb[d] = LES ? se[(eb0 + d) % NEB] + own_e : nullptr;
        const double t0 = FRESH ? 0. : cps[s][pb + own];
        NT_STORE((scalar_tend<ADV, true, LES>(A, m, lm, k, g.nz, t0, ul, uh, vl, vh, wl, wh, cekh, dfac, gh)), &cps[s][pb + own]);
      }
      wl = wh;
    }
    cb0 = (cb0 + 1) % NCB;
    eb0 = (eb0 + 1) % NEB;
  }
}

// kappa scheme with every face value evaluated once.  The limiter (two additions, a division and a min/max chain per
// face, src/modadvection.f90:410-421) dominates this kernel, and the face between two cells is the same number for
// both: face(vel, c-2, c-1, c0, c+1) on the low side of cell i is what cell i-1 needs on its high side.  Each thread
// evaluates the low x- and low y-face of its own cell and leaves them in LDS; 40 lanes of one wave add the tile's 33rd
// column of x-faces and its 9th row of y-faces in one pass; the high z-face is carried to the next level as
// its low face.  3 (+1/4) limiter evaluations per cell instead of 6, bit-identical operands and operation order.
// One barrier per level: faces of level k go to the flux buffers of parity k&1, planes are committed two levels
// before they are first read, and the diffusion (which reads the diffusivity planes) is evaluated before the barrier.
// (A fifth wave owning the extra faces was built and measured slower, profiles/r03/kappa_w5_ab.json: not kept.)
template <bool LES, bool FRESH>
__global__ __launch_bounds__(NT, 4) void scalar_kappa_faces_kernel(Geo g, int gx, int tiles, Metrics m, double cekh, double dfac,
    const double *__restrict__ u, const double *__restrict__ v, const double *__restrict__ w, const double *__restrict__ ekh,
    const double *__restrict__ c, double *__restrict__ cp, int gh, int kc, const double *__restrict__ svm, double *svm_out, double rk3coef) {
  // svm != null (round 5): the RK3 update rides in the sweep -- cp receives svm + rk3coef * tendency, i.e. the NEW value of the scalar
  // (the caller swaps the two arrays afterwards; svm_out != null on stage 3: svm takes it too), and the integration skips this scalar:
  // 16 B per cell less (the tendency is not written and read back).  Only where nothing else touches the tendency of this scalar.
  __shared__ double sc[NCB][CN];
  __shared__ double se[LES ? NEB : 1][LES ? EN : 1];
  __shared__ double sfx[2][MY][MX + 1];
  __shared__ double sfy[2][MY + 1][MX];
  __shared__ double smet[2][NSCALMET + 1];       // level metrics of levels k and k+1
  const unsigned Lb = blockIdx.x;
  const int chunk = Lb / tiles;
  const unsigned lp = Lb - (unsigned)chunk * tiles;
  unsigned tt = lp;
  if ((tiles & 7) == 0) tt = (lp & 7u) * (tiles >> 3) + (lp >> 3);      // XCD-aware, as tile_decode
  const int by = tt / gx, bx = tt - by * gx;
  const int i0 = bx * MX, j0 = by * MY;
  constexpr int NTS = NT;                         // threads that stage planes
  const int tx = threadIdx.x, ty = threadIdx.y, tid = ty * MX + tx;
  constexpr bool own = true;                      // every thread has a cell of the tile
  const int i = i0 + tx, j = j0 + (own ? ty : 0);
  const bool inside = own && i < g.nx && j < g.ny;
  const int k0 = chunk * kc, k1 = min(k0 + kc, g.nz);
  const int jmax = g.ny + HY - 1;
  const bool met_thread = tid < NSCALMET;
  double mreg = 0.;
  if (met_thread) {
    smet[k0 & 1][tid] = scalmet_load(m, tid, k0 + 1, g.nz);
    if (k0 + 1 < k1) mreg = scalmet_load(m, tid, k0 + 2, g.nz);
  }

  long coff[2], eoff[2];
  bool chas[2], ehas[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int e = tid + q * NTS;
    chas[q] = e < CN;
    {
      const int ly = e / CX, lx = e - ly * CX;
      int gi = (i0 - 2 + lx) % g.nx; if (gi < 0) gi += g.nx;
      coff[q] = (long)gi + (long)g.sy * (min(j0 - 2 + ly, jmax) + HY);
    }
    ehas[q] = LES && e < EN;
    {
      const int ly = e / EX, lx = e - ly * EX;
      int gi = (i0 - 1 + lx) % g.nx; if (gi < 0) gi += g.nx;
      eoff[q] = (long)gi + (long)g.sy * (min(j0 - 1 + ly, jmax) + HY);
    }
  }
  double rc[2], re[2];
  double *const scf_ = &sc[0][0];
  double *const sef_ = &se[0][0];
  auto load_c = [&](int k) {
    const long pb = g.sz * (long)(k + HZ);
#pragma unroll
    for (int q = 0; q < 2; ++q) rc[q] = chas[q] ? c[pb + coff[q]] : 0.;
  };
  auto commit_c = [&](int buf) {
#pragma unroll
    for (int q = 0; q < 2; ++q) if (chas[q]) sc[buf][tid + q * NTS] = rc[q];
  };
  auto commit_co = [&](int off) {
#pragma unroll
    for (int q = 0; q < 2; ++q) if (chas[q]) scf_[off + tid + q * NTS] = rc[q];
  };
  auto commit_eo = [&](int off) {
    if (!LES) return;
#pragma unroll
    for (int q = 0; q < 2; ++q) if (ehas[q]) sef_[off + tid + q * NTS] = re[q];
  };
  auto load_e = [&](int k) {
    if (!LES) return;
    const long pb = g.sz * (long)(k + HZ);
#pragma unroll
    for (int q = 0; q < 2; ++q) re[q] = ehas[q] ? ekh[pb + eoff[q]] : 0.;
  };
  auto commit_e = [&](int buf) {
    if (!LES) return;
#pragma unroll
    for (int q = 0; q < 2; ++q) if (ehas[q]) se[buf][tid + q * NTS] = re[q];
  };
  for (int d = 0; d < 5; ++d) { load_c(k0 - 2 + d); commit_c(d); }
  for (int d = 0; d < 3; ++d) { load_e(k0 - 1 + d); commit_e(d); }
  if (k0 + 1 < k1) { load_c(k0 + 3); load_e(k0 + 2); }
  // plane buffers as rotating element offsets into sc / se: co[d] holds plane k-2+d (co[5] is the one being filled),
  // eo[d] holds plane k-1+d (eo[3] being filled); rotated by register moves, no modulo in the loop
  int co[NCB], eo[NEB];
#pragma unroll
  for (int d = 0; d < NCB; ++d) co[d] = d * CN;
#pragma unroll
  for (int d = 0; d < NEB; ++d) eo[d] = d * EN;
  double *const scf = &sc[0][0];
  double *const sef = &se[0][0];
  const int oty = own ? ty : 0;
  const int own_c = (oty + 2) * CX + (tx + 2), own_e = (oty + 1) * EX + (tx + 1);
  const long owng = (long)(i % g.nx) + (long)g.sy * (min(j, jmax) + HY);
  // the extra faces: wave xw, lanes 0..MY-1 -> x-face MX of row lane; wave yw, lanes 0..MX-1 -> y-face MY of column lane
  // (the two waves rotate with the block so that no SIMD always hosts the longer ones)
  // ONE wave per workgroup (rotating with the block) evaluates all 40 of them in a single pass: lanes 0..MX-1 the y-faces,
  // lanes MX..MX+MY-1 the x-faces (two passes in two waves cost a whole wave's issue time each, for 8 and 32 lanes)
  const int wave = tid >> 6, lane = tid & 63;
  const bool erole = wave == (int)(Lb & 3) && lane < MX + MY;
  const bool ex = lane >= MX;                      // this lane's extra face is an x-face (row lane - MX), else a y-face (column lane)
  const int el = ex ? lane - MX : lane;
  const int er_c = ex ? (el + 2) * CX + (MX + 2) : (MY + 2) * CX + (el + 2);      // the cell on the high side of the face
  const int er_s = ex ? 1 : CX;                                                    // stride towards the low side
  const long er_g = ex ? (long)((i0 + MX) % g.nx) + (long)g.sy * (min(j0 + el, jmax) + HY)
                       : (long)((i0 + el) % g.nx) + (long)g.sy * (min(j0 + MY, jmax) + HY);
  const double dxi = m.dxi, dx = m.dx, dyi = m.dyi;
  const double top = gh == 1 ? 1. : 0.;

  __syncthreads();
  // what crosses a face is (face value) x (face velocity): the product is shared, so each thread needs the velocities
  // on its own low faces only; those of the next level are fetched one level ahead like the staged planes
  const double wl0 = own ? w[g.sz * (long)(k0 + HZ) + owng] : 0.;
  double pzl = 0.;      // face value x w on the low z side of the own cell at level k
  if (own && k0 >= 1) {
    const int k = k0, kf = k + 1;
    double kzm2 = sc[0][own_c], kzm1 = sc[1][own_c], kzp1 = sc[3][own_c];
    const double c0 = sc[2][own_c];
    if (gh) {
      if (k <= 1) kzm2 = 0.;
      if (k == g.nz - 1) kzp1 = top * c0;
    }
    pzl = face(wl0, kzm2, kzm1, c0, kzp1, m.dzhi[kf - 1 < 1 ? 1 : kf - 1], m.dzhi[kf], m.dzhi[kf + 1], m.dzf[kf]) * wl0;
  }
  double nu = 0., nv = 0., nw = 0., ne_ = 0.;
  const double *const evel = ex ? u : v;
  auto load_vel = [&](int k) {
    const long pb = g.sz * (long)(k + HZ);
    if (own) { nu = u[pb + owng]; nv = v[pb + owng]; nw = w[pb + owng + g.sz]; }
    if (erole) ne_ = evel[pb + er_g];
  };
  load_vel(k0);
  for (int k = k0; k < k1; ++k) {
    const int fb = k & 1;
    const double ul = nu, vl = nv, wh = nw, ve = ne_;
    if (k + 1 < k1) {
      commit_co(co[5]);
      commit_eo(eo[3]);
      if (met_thread) smet[(k + 1) & 1][tid] = mreg;
      if (k + 2 < k1) {
        if (met_thread) mreg = scalmet_load(m, tid, k + 3, g.nz);
        load_c(k + 4); load_e(k + 3);
      }
      load_vel(k + 1);
    }
    __builtin_amdgcn_sched_barrier(0);
    const ScalMetLds lm{smet[k & 1]};
    const long pb = g.sz * (long)(k + HZ);
    LdsAcc A;
#pragma unroll
    for (int d = 0; d < 5; ++d) A.cb[d] = scf + co[d] + own_c;
#pragma unroll
    for (int d = 0; d < 3; ++d) A.eb[d] = LES ? sef + eo[d] + own_e : nullptr;
    const double *pl = scf + co[2];
    const double t0 = (FRESH || !inside) ? 0. : cp[pb + owng];
    const double sm = (svm && inside) ? NT_LOAD(&svm[pb + owng]) : 0.;
    double pxl = 0., pyl = 0., pzh = 0., dif = 0.;
    if (own) {
    const double c0 = pl[own_c];
    pxl = face(ul, pl[own_c - 2], pl[own_c - 1], c0, pl[own_c + 1], dxi, dxi, dxi, dx) * ul;
    pyl = face(vl, pl[own_c - 2 * CX], pl[own_c - CX], c0, pl[own_c + CX], 1., 1., 1., 1.) * vl;
    sfx[fb][ty][tx] = pxl;
    sfy[fb][ty][tx] = pyl;
    }
    if (erole) {
      const double hh = ex ? dxi : 1., df = ex ? dx : 1.;
      const double f = face(ve, pl[er_c - 2 * er_s], pl[er_c - er_s], pl[er_c], pl[er_c + er_s], hh, hh, hh, df) * ve;
      if (ex) sfx[fb][el][MX] = f; else sfy[fb][MY][el] = f;
    }
    if (own) {
    const double c0 = pl[own_c];
    // z: the high face (faces kb+1..ke+1; no flux through the floor, src/modadvection.f90:385); operands as scalar_tend
    double kzm1 = A.c(0, 0, -1), kzp1 = A.c(0, 0, 1), kzp2 = A.c(0, 0, 2);
    if (gh) {
      if (k == 0) kzm1 = 0.;
      if (k == g.nz - 1) { kzp1 = top * c0; kzp2 = top * c0; }
      if (k == g.nz - 2) kzp2 = top * kzp1;
    }
    pzh = face(wh, kzm1, c0, kzp1, kzp2, lm.get(4), lm.get(5), lm.get(6), lm.get(2)) * wh;
    dif = scalar_tend<0, true, LES>(A, m, lm, k, g.nz, 0., 0., 0., 0., 0., 0., 0., cekh, dfac, 0);
    }
    __syncthreads();
    if (own) {
    const double pxh = sfx[fb][ty][tx + 1], pyh = sfy[fb][ty + 1][tx];
    double t = t0;
    t = (t + (-pxh * dxi)) + pxl * dxi;
    t = (t + (-pyh * dyi)) + pyl * dyi;
    {
      const double dzfci = lm.get(7);
      const double upper = -pzh * dzfci;
      const double lower = k >= 1 ? pzl * dzfci : 0.;
      t = (t + upper) + lower;
    }
    t = t + dif;
    if (svm) {
      t = sm + rk3coef * t;
      if (svm_out && inside) NT_STORE(t, &svm_out[pb + owng]);
    }
    if (inside) NT_STORE(t, &cp[pb + owng]);      // written once, read by the integration from memory
    }
    pzl = pzh;
    {
      const int c0_ = co[0];
#pragma unroll
      for (int d = 0; d + 1 < NCB; ++d) co[d] = co[d + 1];
      co[NCB - 1] = c0_;
      const int e0_ = eo[0];
#pragma unroll
      for (int d = 0; d + 1 < NEB; ++d) eo[d] = eo[d + 1];
      eo[NEB - 1] = e0_;
    }
  }
}

}  // namespace

// fused advection + diffusion of scalar slot n; false when this kernel does not apply (the caller falls back)
bool k_scalar_fused_lds(udc_handle *h, int n, bool fresh, int *rc) {
  *rc = 0;
  const bool inl = h->sv_inline[n];      // the RK3 update rides in the sweep (substep_fused decided; kappa kernel only)
  const Geo &g = h->g;
  if (g.nx < MX || g.ny < 4) return false;
  const int gx = (g.nx + MX - 1) / MX, gy = (g.ny + MY - 1) / MY, tiles = gx * gy;
  // k-chunk: amortise the 5-plane prologue while keeping 256 CUs x 3 workgroups busy (as pick_kc of the momentum sweep)
  int kc = g.nz < 8 ? g.nz : 8;
  {
    const long slots = 256L * 4;      // 4 workgroups per CU (LDS- and VGPR-limited alike)
    double best = 1e300;
    for (int q = 8; q <= g.nz; ++q) {
      const long blocks = (long)tiles * ((g.nz + q - 1) / q);
      const double cost = (double)((blocks + slots - 1) / slots) * (q + 5);
      if (cost < best - 1e-9) { best = cost; kc = q; }
    }
  }
  const int chunks = (g.nz + kc - 1) / kc;
  const dim3 b(MX, MY, 1), gr((unsigned)tiles * (unsigned)chunks, 1, 1);
  const double cekh = h->p.numol * h->p.prandtlmoli;
  const double *u = h->fields[UDC_U0], *v = h->fields[UDC_V0], *w = h->fields[UDC_W0];
  const bool tke = h->slot[n].tke;
  const double dfac = tke ? 1.0 : 0.5;
  const double *ekh = h->fields[tke ? UDC_EKM : UDC_EKH], *c = h->fields[UDC_SV0 + 3 * n];
  double *cp = h->fields[UDC_SVP + 3 * n];
  const bool les = h->p.sgs != UDC_SGS_DNS, cd2 = h->slot[n].adv == 2;
  const int gh = h->slot[n].kappa_ghosts;
  if (inl && (cd2 || !fresh)) { udc_set_error("scalar sweep: the in-sweep update is the fresh kappa sweep's only"); *rc = 1; return true; }
  const double *svm_in = inl ? h->fields[UDC_SVM + 3 * n] : nullptr;
  double *svm_out = (inl && h->sv_inline_last) ? h->fields[UDC_SVM + 3 * n] : nullptr;
  const double rk = h->sv_inline_rk3coef;
#define LF(L, F) hipLaunchKernelGGL((scalar_kappa_faces_kernel<L, F>), gr, b, 0, h->stream, g, gx, tiles, h->m, cekh, dfac, u, v, w, ekh, c, cp, gh, kc, svm_in, svm_out, rk)
#define LS(A, L, F) hipLaunchKernelGGL((scalar_lds_kernel<A, L, F, 1>), gr, b, 0, h->stream, g, gx, tiles, h->m, cekh, dfac, u, v, w, ekh, c, cp, c, cp, gh, kc)
  {
    PROF(h, cd2 ? "scalar_lds_cd2" : "scalar_kappa_faces");
    if (cd2) { if (les) { if (fresh) LS(2, true, true); else LS(2, true, false); } else { if (fresh) LS(2, false, true); else LS(2, false, false); } }
    else     { if (les) { if (fresh) LF(true, true); else LF(true, false); } else { if (fresh) LF(false, true); else LF(false, false); } }
  }
#undef LS
#undef LF
  if (hipGetLastError() != hipSuccess) { udc_set_error("scalar_lds_kernel launch failed"); *rc = 1; }
  return true;
}

// advection + diffusion of TWO scalar slots in one sweep (see scalar_lds_kernel's NS): both advected by advecc_2nd, both diffused
// with ekh, same vertical ghost rule -- thl and qt.  false when that does not apply (the caller launches them one by one).
bool k_scalar_pair_lds(udc_handle *h, int na, int nb, bool fresh, int *rc) {
  *rc = 0;
  if (!h->sw.scalar_pair) return false;
  const Geo &g = h->g;
  if (g.nx < MX || g.ny < 4) return false;
  const udc_handle::Slot &A = h->slot[na], &B = h->slot[nb];
  if (A.adv != 2 || B.adv != 2 || A.tke || B.tke || A.kappa_ghosts != B.kappa_ghosts) return false;
  const int gx = (g.nx + MX - 1) / MX, gy = (g.ny + MY - 1) / MY, tiles = gx * gy;
  int kc = g.nz < 8 ? g.nz : 8;
  {
    const long slots = 256L * 3;      // 52 KB of LDS: three workgroups per CU
    double best = 1e300;
    for (int q = 8; q <= g.nz; ++q) {
      const long blocks = (long)tiles * ((g.nz + q - 1) / q);
      const double cost = (double)((blocks + slots - 1) / slots) * (q + 5);
      if (cost < best - 1e-9) { best = cost; kc = q; }
    }
  }
  const int chunks = (g.nz + kc - 1) / kc;
  const dim3 b(MX, MY, 1), gr((unsigned)tiles * (unsigned)chunks, 1, 1);
  const double cekh = h->p.numol * h->p.prandtlmoli, dfac = 0.5;
  const double *u = h->fields[UDC_U0], *v = h->fields[UDC_V0], *w = h->fields[UDC_W0], *ekh = h->fields[UDC_EKH];
  const double *ca = h->fields[UDC_SV0 + 3 * na], *cb = h->fields[UDC_SV0 + 3 * nb];
  double *cpa = h->fields[UDC_SVP + 3 * na], *cpb = h->fields[UDC_SVP + 3 * nb];
  const bool les = h->p.sgs != UDC_SGS_DNS;
  const int gh = A.kappa_ghosts;
#define LP(L, F) hipLaunchKernelGGL((scalar_lds_kernel<2, L, F, 2>), gr, b, 0, h->stream, g, gx, tiles, h->m, cekh, dfac, u, v, w, ekh, ca, cpa, cb, cpb, gh, kc)
  {
    PROF(h, "scalar_lds_cd2_pair");
    if (les) { if (fresh) LP(true, true); else LP(true, false); } else { if (fresh) LP(false, true); else LP(false, false); }
  }
#undef LP
  if (hipGetLastError() != hipSuccess) { udc_set_error("scalar_lds_kernel (pair) launch failed"); *rc = 1; }
  return true;
}

