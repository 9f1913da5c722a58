// Closure + momentum in ONE k-marching sweep (single slab, Smagorinsky / Vreman without the buoyancy correction).
//
// The two-kernel version (udc_mom_lds.hip) writes ekm, ekh (16 B per cell) and reads ekm straight back (8 B), and both kernels
// stage u0, v0, w0 (2 x 24 B): 128 B of algorithmic traffic per cell where 80 are compulsory (u0, v0, w0, pres0 in; um, vm, wm
// in; pup, pvp, pwp out).  Here a workgroup owns 32 x 8 cells and marches upward holding
//   * u0, v0, w0 on the tile extended by TWO cells (36 x 12) for planes k-1 .. k+2 (+ k+3 arriving in registers),
//   * ekm on the tile extended by ONE cell (34 x 10) for planes k-1, k, k+1, evaluated here from the staged velocities
//     (closure_arith, the expressions of src/modsubgrid.f90:208-360; ghost planes by closurebc's rules,
//     src/modboundary.f90:447-465) -- 1.33 evaluations per cell instead of 1, out of LDS,
//   * pres0 on 34 x 10 for planes k-1, k.
// Per level: barrier, commit plane k+3, issue the loads of plane k+4, evaluate ekm(k+1), barrier, momentum stencil of level k
// (mom_arith: advecu/v/w_2nd, diffu/v/w, forces; src/modadvection.f90:158-314, src/modsubgrid.f90:672-997).  x and y wrap by
// index (the slab is the whole periodic domain), so no ghost row or column of any field is read.
// ekm / ekh go to memory only when somebody else reads them in this substep (EMIT: scalars, immersed boundary, RK stage 3 for
// the time-step maxima, statistics and restart files) -- then with the ghost rows and planes closurebc gives them.
#include "udc_internal.h"
#include "udc_mom_arith.h"
#include "udc_closure_arith.h"
#include <cstdlib>

namespace {

constexpr int FX = 32, FY = 8, NT = FX * FY;                  // own cells = threads
constexpr int TXL = FX + 2, TYL = FY + 2, TN = TXL * TYL;     // ekm / pres0 tile: 34 x 10
constexpr int EXL = FX + 4, EYL = FY + 4, EN = EXL * EYL;     // velocity tile: 36 x 12
constexpr int NVH = EN - NT, NTH = TN - NT;                   // halo elements: 176 of the velocity tile, 84 of the ekm tile
static_assert(NVH <= NT && NTH <= NT, "one halo element per thread");
constexpr int NMET = NLEVMET + NCLOSMET;                      // 21 level entries staged per level
constexpr size_t FUSED_LDS = sizeof(double) * (15 * EN + 6 * TN + 2 * (NMET + 3));

struct FusedArgs {
  const double *u, *v, *w, *p;
  double *up, *vp, *wp;
  const double *um, *vm, *wm;
  double *ekm, *ekh;            // written when emit
  double rk3coefi;
  int wrap_vp, um_is_u0, emit;
  int dbg;                      // timing experiments only (UDC_FUSED_DBG bits: 1 no halo evaluations, 2 no closure at all, 4 no second barrier, 8 no stores)
};

// velocity-tile halo element e (0 .. NVH-1) -> tile coordinates
__device__ __forceinline__ void vhalo_coords(int e, int &lx, int &ly) {
  if (e < 2 * EXL) { ly = e / EXL; lx = e - ly * EXL; }
  else if (e < 4 * EXL) { const int r = e - 2 * EXL; const int q = r / EXL; ly = EYL - 2 + q; lx = r - q * EXL; }
  else { const int r = e - 4 * EXL; ly = 2 + (r >> 2); const int q = r & 3; lx = q < 2 ? q : EXL - 4 + q; }
}
// ekm-tile halo element e (0 .. NTH-1) -> tile coordinates
__device__ __forceinline__ void thalo_coords(int e, int &lx, int &ly) {
  if (e < TXL) { ly = 0; lx = e; }
  else if (e < 2 * TXL) { ly = TYL - 1; lx = e - TXL; }
  else { const int r = e - 2 * TXL; ly = 1 + (r >> 1); lx = (r & 1) ? TXL - 1 : 0; }
}

struct VAcc {      // neighbour access into three staged velocity planes (pointers to the centre element, stride EXL)
  const double *um, *uc, *up, *vm, *vc, *vp, *wm, *wc, *wp;
  __device__ __forceinline__ double u(int di, int dj, int dk) const { return (dk == 0 ? uc : (dk < 0 ? um : up))[dj * EXL + di]; }
  __device__ __forceinline__ double v(int di, int dj, int dk) const { return (dk == 0 ? vc : (dk < 0 ? vm : vp))[dj * EXL + di]; }
  __device__ __forceinline__ double w(int di, int dj, int dk) const { return (dk == 0 ? wc : (dk < 0 ? wm : wp))[dj * EXL + di]; }
};

struct FusedMetLds {      // the level block in LDS: entries 0 .. NLEVMET-1 = LevelMet of level k, NLEVMET .. = closure metrics of level k+1
  const double *p;
  __device__ __forceinline__ double get(int t) const {
    union { double d; int i[2]; } u;
    u.d = p[t];
    u.i[0] = __builtin_amdgcn_readfirstlane(u.i[0]);
    u.i[1] = __builtin_amdgcn_readfirstlane(u.i[1]);
    return u.d;
  }
};
struct FusedClosMetLds {
  const double *p;
  __device__ __forceinline__ double get(int t) const { return FusedMetLds{p + NLEVMET}.get(t); }
};

template <int SGS, bool FORCES>
__global__ __launch_bounds__(NT, 2) void mom_closure_kernel(Geo g, TileGrid tg, Metrics m, Params pr, FusedArgs a, int kc) {
  // 68.5 KB of LDS (two workgroups per CU): more than the static limit, so dynamic (hipFuncAttributeMaxDynamicSharedMemorySize)
  extern __shared__ double lds_fused[];
  double (*V)[3][EN] = reinterpret_cast<double (*)[3][EN]>(lds_fused);                        // [5]: velocity planes, slots rotate
  double (*E)[TN] = reinterpret_cast<double (*)[TN]>(lds_fused + 15 * EN);                      // [3]: ekm planes k-1, k, k+1
  double (*P)[TN] = reinterpret_cast<double (*)[TN]>(lds_fused + 15 * EN + 3 * TN);             // [3]: pres0 planes k-1, k, filling
  double (*smet)[NMET + 3] = reinterpret_cast<double (*)[NMET + 3]>(lds_fused + 15 * EN + 6 * TN);   // [2]

  const unsigned L = blockIdx.x;
  const int chunk = L / tg.tiles;
  const unsigned lp = L - (unsigned)chunk * tg.tiles;
  unsigned tt = lp;
  if ((tg.tiles & 7) == 0) tt = (lp & 7u) * (tg.tiles >> 3) + (lp >> 3);
  const int by = tt / tg.gx, bx = tt - by * tg.gx;
  const int i0 = bx * FX, j0 = by * FY;
  const int tx = threadIdx.x, ty = threadIdx.y, tid = ty * FX + tx;
  const int i = i0 + tx, j = j0 + ty;
  const bool inside = i < g.nx && j < g.ny;
  const int k0 = chunk * kc;
  const int k1 = min(k0 + kc, g.nz);

  auto wrapi = [&](int q) { q %= g.nx; return q < 0 ? q + g.nx : q; };
  auto wrapj = [&](int q) { q %= g.ny; return q < 0 ? q + g.ny : q; };
  // this thread's elements: own cell (velocity tile (tx+2, ty+2), ekm tile (tx+1, ty+1)), one velocity halo element, one ekm halo element
  const long own_off = (long)wrapi(i) + (long)g.sy * (wrapj(j) + HY);
  const int own_v = (ty + 2) * EXL + (tx + 2), own_t = (ty + 1) * TXL + (tx + 1);
  const bool has_vh = tid < NVH, has_th = tid < NTH;
  int vlx = 0, vly = 0, hlx = 0, hly = 0;
  if (has_vh) vhalo_coords(tid, vlx, vly);
  if (has_th) thalo_coords(tid, hlx, hly);
  const long vh_off = (long)wrapi(i0 - 2 + vlx) + (long)g.sy * (wrapj(j0 - 2 + vly) + HY);
  const int vh_l = vly * EXL + vlx;
  const long th_off = (long)wrapi(i0 - 1 + hlx) + (long)g.sy * (wrapj(j0 - 1 + hly) + HY);
  const int th_t = hly * TXL + hlx;                        // ekm-tile index of the halo element ...
  const int th_v = (hly + 1) * EXL + (hlx + 1);            // ... and the velocity-tile index of the same cell
  const bool p_halo = has_th && (hlx == 0 || hly == 0);    // pres0 is read at (c, i-1, j-1) only

  const double *fld[3] = {a.u, a.v, a.w};
  double sc[3], sh[3];                                     // a velocity plane between load and commit
  auto load_v = [&](int k) {
    const long pb = g.sz * (long)(k + HZ);
#pragma unroll
    for (int f = 0; f < 3; ++f) {
      sc[f] = fld[f][pb + own_off];
      sh[f] = has_vh ? fld[f][pb + vh_off] : 0.0;
    }
  };
  auto commit_v = [&](int slot) {
#pragma unroll
    for (int f = 0; f < 3; ++f) {
      V[slot][f][own_v] = sc[f];
      if (has_vh) V[slot][f][vh_l] = sh[f];
    }
  };
  double pc = 0., ph = 0.;
  auto load_p = [&](int k) {
    const long pb = g.sz * (long)(k + HZ);
    pc = a.p[pb + own_off];
    ph = p_halo ? a.p[pb + th_off] : 0.0;
  };
  auto commit_p = [&](int slot) {
    P[slot][own_t] = pc;
    if (p_halo) P[slot][th_t] = ph;
  };

  // ekm of one tile element from three velocity planes (slots sm, s0, sp = planes q-1, q, q+1), or by closurebc's ghost rules
  const double nm = pr.numol, nh = pr.numol * pr.prandtlmoli;
  auto eval = [&](int vidx, int sm, int s0, int sp, int q, auto lm, double &em, double &eh) {
    VAcc A{V[sm][0] + vidx, V[s0][0] + vidx, V[sp][0] + vidx, V[sm][1] + vidx, V[s0][1] + vidx, V[sp][1] + vidx,
           V[sm][2] + vidx, V[s0][2] + vidx, V[sp][2] + vidx};
    closure_arith<SGS>(A, m, lm, pr, q, em, eh);
  };
  // closurebc folded in, as closure_lds_kernel does it: the thread that owns a cell writes its periodic image rows and the
  // ghost planes below the floor / above the top
  auto emit_own = [&](int q, double em, double eh) {
    if (!a.emit || !inside) return;
    const long c = g.sz * (long)(q + HZ) + own_off;
    a.ekm[c] = em; a.ekh[c] = eh;
    const long up_row = (long)g.sy * g.ny, dn_row = -(long)g.sy * g.ny;
    const long wr = (j == 0) ? up_row : ((j == g.ny - 1) ? dn_row : 0);
    if (wr) { a.ekm[c + wr] = em; a.ekh[c + wr] = eh; }
    if (q == 0) {
      const double gm = 2. * nm - em, gh = (2. * nh) - eh;
      a.ekm[c - g.sz] = gm; a.ekh[c - g.sz] = gh;
      if (wr) { a.ekm[c - g.sz + wr] = gm; a.ekh[c - g.sz + wr] = gh; }
    }
    if (q == g.nz - 1) {
      const double gm = pr.bctopm == UDC_TOP_NOSLIP ? 2. * nm - em : em;
      const double gh = pr.bctopm == UDC_TOP_NOSLIP ? (2. * nh) - eh : eh;
      a.ekm[c + g.sz] = gm; a.ekh[c + g.sz] = gh;
      if (wr) { a.ekm[c + g.sz + wr] = gm; a.ekh[c + g.sz + wr] = gh; }
    }
  };

  // ---- prologue: velocity planes k0-2 .. k0+2 into slots 0 .. 4; ekm(k0-1), ekm(k0); pres0 planes k0-1, k0
  for (int q = 0; q < 5; ++q) { load_v(max(k0 - 2 + q, -HZ)); commit_v(q); }
  load_p(k0 - 1); commit_p(0);
  load_p(k0);     commit_p(1);
  const bool met_thread = tid < NMET;
  // entry t of the block of level k: LevelMet entry t of level k, or closure metric t - NLEVMET of level k + 1 (clamped: the
  // closure of level nz is never evaluated, its ghost plane follows from level nz - 1)
  const int mt = met_thread ? tid : 0;
  const double *mp = (mt < NLEVMET ? levmet_src(m, mt) : closmet_src(m, mt - NLEVMET)) + 1;
  const int mshift = mt < NLEVMET ? 0 : 1;
  auto met_at = [&](int k) { return mp[min(k + mshift, g.nz - 1)]; };
  double mreg = 0.;
  if (met_thread) {
    smet[k0 & 1][tid] = met_at(k0);
    if (k0 + 1 < k1) mreg = met_at(k0 + 1);
  }
  __syncthreads();
  double em_own = 0., eh_own = 0., em_halo = 0., eh_halo = 0.;      // this thread's latest evaluations (for the ghost planes)
  {
    // ekm(k0): slots 1, 2, 3 hold planes k0-1, k0, k0+1
    const ClosMetGlobal lm0{m, k0 + 1};
    eval(own_v, 1, 2, 3, k0, lm0, em_own, eh_own);
    E[1][own_t] = em_own;
    emit_own(k0, em_own, eh_own);
    if (has_th) { eval(th_v, 1, 2, 3, k0, lm0, em_halo, eh_halo); E[1][th_t] = em_halo; }
    // ekm(k0-1): below the floor the mirror value (src/modboundary.f90:452-453), else from planes k0-2, k0-1, k0
    if (k0 == 0) {
      E[0][own_t] = 2. * nm - em_own;
      if (has_th) E[0][th_t] = 2. * nm - em_halo;
    } else {
      const ClosMetGlobal lmm{m, k0};
      double e1, e2;
      eval(own_v, 0, 1, 2, k0 - 1, lmm, e1, e2); E[0][own_t] = e1;
      if (has_th) { eval(th_v, 0, 1, 2, k0 - 1, lmm, e1, e2); E[0][th_t] = e1; }
    }
  }
  // plane k0+3 into registers (committed after the first barrier of the loop into the slot of plane k0-2)
  if (k0 + 3 <= k1 + 1) load_v(k0 + 3);
  if (k0 + 1 < k1) load_p(k0 + 1);
  // slots of velocity planes k-1, k, k+1, k+2 and the free one; ekm slots k-1, k, k+1(free); pres0 slots k-1, k, free
  int vm_ = 1, vc_ = 2, vp_ = 3, vq_ = 4, vf_ = 0;
  int em_ = 0, ec_ = 1, ep_ = 2;
  int qm_ = 0, qc_ = 1, qf_ = 2;

  for (int k = k0; k < k1; ++k) {
    // this level's direct operands, requested before the barrier
    const long cc = g.sz * (long)(k + HZ) + own_off;
    double pum = 0., pvm = 0., pwm = 0.;
    if (inside && !a.um_is_u0) { pum = a.um[cc]; pvm = a.vm[cc]; pwm = a.wm[cc]; }
    __syncthreads();                                           // B1: level k-1 is finished everywhere
    if (k + 3 <= k1 + 1) commit_v(vf_);                        // plane k+3 (first read by the closure of level k+2)
    if (k + 1 < k1) {
      commit_p(qf_);                                           // pres0 plane k+1
      if (met_thread) smet[(k + 1) & 1][tid] = mreg;           // block of level k+1
      if (met_thread && k + 2 < k1) mreg = met_at(k + 2);
      if (k + 2 < k1) load_p(k + 2);
    }
    if (k + 4 <= k1 + 1) load_v(k + 4);                        // in flight while this level is computed
    __builtin_amdgcn_sched_barrier(0);
    // ---- ekm(k+1) on the 34 x 10 tile
    {
      double em, eh;
      if (k + 1 < g.nz && !(a.dbg & 2)) {
        const FusedClosMetLds lmc{smet[k & 1]};
        eval(own_v, vc_, vp_, vq_, k + 1, lmc, em, eh);
        if (k + 1 < k1) emit_own(k + 1, em, eh);
        em_own = em; eh_own = eh;
        E[ep_][own_t] = em;
        if (has_th && !(a.dbg & 1)) { eval(th_v, vc_, vp_, vq_, k + 1, lmc, em, eh); em_halo = em; E[ep_][th_t] = em; }
      } else {      // above the top: zero gradient, or the mirror value under a no-slip lid (src/modboundary.f90:455-465)
        const bool ns = pr.bctopm == UDC_TOP_NOSLIP;
        E[ep_][own_t] = ns ? 2. * nm - em_own : em_own;
        if (has_th) E[ep_][th_t] = ns ? 2. * nm - em_halo : em_halo;
      }
    }
    if (!(a.dbg & 4)) __syncthreads();                         // B2: ekm(k+1) is complete
    // ---- momentum stencil of level k
    if (inside) {
      const FusedMetLds lm{smet[k & 1]};
      const double *um = V[vm_][0] + own_v, *uc = V[vc_][0] + own_v, *up = V[vp_][0] + own_v;
      const double *vm = V[vm_][1] + own_v, *vc = V[vc_][1] + own_v, *vp = V[vp_][1] + own_v;
      const double *wm = V[vm_][2] + own_v, *wc = V[vc_][2] + own_v, *wp = V[vp_][2] + own_v;
      MomVals qq;
      qq.u_c = uc[0]; qq.u_xm = uc[-1]; qq.u_xp = uc[1]; qq.u_ym = uc[-EXL]; qq.u_yp = uc[EXL];
      qq.u_zm = um[0]; qq.u_zp = up[0]; qq.u_xp_ym = uc[1 - EXL]; qq.u_xp_zm = um[1];
      qq.v_c = vc[0]; qq.v_xm = vc[-1]; qq.v_xp = vc[1]; qq.v_ym = vc[-EXL]; qq.v_yp = vc[EXL];
      qq.v_zm = vm[0]; qq.v_zp = vp[0]; qq.v_xm_yp = vc[-1 + EXL]; qq.v_yp_zm = vm[EXL];
      qq.w_c = wc[0]; qq.w_xm = wc[-1]; qq.w_xp = wc[1]; qq.w_ym = wc[-EXL]; qq.w_yp = wc[EXL];
      qq.w_zm = wm[0]; qq.w_zp = wp[0]; qq.w_xm_zp = wp[-1]; qq.w_ym_zp = wp[-EXL];
      const int o = own_t;
      qq.p_c = P[qc_][o]; qq.p_xm = P[qc_][o - 1]; qq.p_ym = P[qc_][o - TXL]; qq.p_zm = P[qm_][o];
      const double *em = E[em_], *ec = E[ec_], *ep = E[ep_];
      qq.e_c = ec[o]; qq.e_xm = ec[o - 1]; qq.e_xp = ec[o + 1]; qq.e_ym = ec[o - TXL]; qq.e_yp = ec[o + TXL];
      qq.e_zm = em[o]; qq.e_zp = ep[o];
      qq.e_xm_yp = ec[o - 1 + TXL]; qq.e_xm_ym = ec[o - 1 - TXL]; qq.e_xm_zm = em[o - 1]; qq.e_xm_zp = ep[o - 1];
      qq.e_ym_zm = em[o - TXL]; qq.e_ym_zp = ep[o - TXL]; qq.e_xp_ym = ec[o + 1 - TXL];
      qq.e_yp_zm = em[o + TXL]; qq.e_xp_zm = em[o + 1];
      double tu = 0., tv = 0., tw = 0.;
      mom_arith<true, true, true, FORCES>(qq, m, lm, k, pr.numol, tu, tv, tw);
      if (a.um_is_u0) { pum = qq.u_c; pvm = qq.v_c; pwm = qq.w_c; }
      tu = tu + pum * a.rk3coefi;
      tv = tv + pvm * a.rk3coefi;
      tw = (k == 0) ? 0. : tw + pwm * a.rk3coefi;
      if (!(a.dbg & 8) || tu == 1.2345e300) { a.up[cc] = tu; a.vp[cc] = tv; a.wp[cc] = tw; }
      if (a.wrap_vp && j == 0) a.vp[cc + (long)g.sy * g.ny] = tv;
    }
    { const int t = vf_; vf_ = vm_; vm_ = vc_; vc_ = vp_; vp_ = vq_; vq_ = t; }
    { const int t = em_; em_ = ec_; ec_ = ep_; ep_ = t; }
    { const int t = qm_; qm_ = qc_; qc_ = qf_; qf_ = t; }
  }
}

}  // namespace

bool fused_closure_possible(const udc_handle *h) {
  return !h->no_fused_closure && !h->slab && (h->p.sgs == UDC_SGS_SMAGORINSKY || h->p.sgs == UDC_SGS_VREMAN) && !h->lbuoycorr && !h->p.bare &&
         h->g.ny >= 4 && h->g.nx >= 4;
}

// the fused sweep: pup = (advection + diffusion + forces) + um / rk3coef, fresh tendencies; emit: ekm, ekh also go to memory
int k_momentum_closure(udc_handle *h, bool forces, double rk3coefi, bool um_is_u0, bool emit) {
  const Geo &g = h->g;
  TileGrid tg; tg.gx = (g.nx + FX - 1) / FX; tg.gy = (g.ny + FY - 1) / FY; tg.tiles = tg.gx * tg.gy;
  // chunk length: two workgroups per CU (68 KB of LDS each); each pays a five-plane prologue with two closure evaluations
  const long slots = 256L * 2;
  int kc = g.nz;
  if (getenv("UDC_FUSED_KC") && atoi(getenv("UDC_FUSED_KC")) >= 1) kc = atoi(getenv("UDC_FUSED_KC")) < g.nz ? atoi(getenv("UDC_FUSED_KC")) : g.nz;
  else {
    double best = 1e300;
    for (int c = (g.nz < 8 ? g.nz : 8); c <= g.nz; ++c) {
      const long blocks = (long)tg.tiles * ((g.nz + c - 1) / c);
      const long rounds = (blocks + slots - 1) / slots;
      const double cost = (double)rounds * (c + 6);
      if (cost < best - 1e-9) { best = cost; kc = c; }
    }
  }
  const int chunks = (g.nz + kc - 1) / kc;
  dim3 b(FX, FY, 1), gr((unsigned)tg.tiles * (unsigned)chunks, 1, 1);
  FusedArgs a{h->fields[UDC_U0], h->fields[UDC_V0], h->fields[UDC_W0], h->fields[UDC_PRES0],
              h->fields[UDC_UP], h->fields[UDC_VP], h->fields[UDC_WP],
              h->fields[UDC_UM], h->fields[UDC_VM], h->fields[UDC_WM], h->fields[UDC_EKM], h->fields[UDC_EKH],
              rk3coefi, 1, um_is_u0 ? 1 : 0, emit ? 1 : 0, getenv("UDC_FUSED_DBG") ? atoi(getenv("UDC_FUSED_DBG")) : 0};
  static bool attr_set = false;
  if (!attr_set) {
    HIP_OK(hipFuncSetAttribute((const void *)mom_closure_kernel<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FUSED_LDS));
    HIP_OK(hipFuncSetAttribute((const void *)mom_closure_kernel<1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FUSED_LDS));
    HIP_OK(hipFuncSetAttribute((const void *)mom_closure_kernel<2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FUSED_LDS));
    HIP_OK(hipFuncSetAttribute((const void *)mom_closure_kernel<2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FUSED_LDS));
    attr_set = true;
  }
  PROF(h, "mom_closure");
  const bool smag = h->p.sgs == UDC_SGS_SMAGORINSKY;
  if (smag) {
    if (forces) hipLaunchKernelGGL((mom_closure_kernel<1, true>), gr, b, FUSED_LDS, h->stream, g, tg, h->m, h->p, a, kc);
    else hipLaunchKernelGGL((mom_closure_kernel<1, false>), gr, b, FUSED_LDS, h->stream, g, tg, h->m, h->p, a, kc);
  } else {
    if (forces) hipLaunchKernelGGL((mom_closure_kernel<2, true>), gr, b, FUSED_LDS, h->stream, g, tg, h->m, h->p, a, kc);
    else hipLaunchKernelGGL((mom_closure_kernel<2, false>), gr, b, FUSED_LDS, h->stream, g, tg, h->m, h->p, a, kc);
  }
  HIP_OK(hipGetLastError());
  return 0;
}
