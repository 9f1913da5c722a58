// Fused momentum sweep, LDS-staged and marching in k ("2.5-D blocking").
//
// A workgroup owns a 64(x) x 4(y) column of cells and walks KC levels upward.  For every level it
// stages the (64+2) x (4+2) tile of u0, v0, w0 (and ekm) of plane k+1 into LDS, keeping planes
// k-1, k, k+1 resident (3 rotating buffers), so each value is fetched from global memory once per
// workgroup instead of once per neighbouring cell (the direct-load kernel issues ~50 global loads
// per cell; this one ~8 + 4 for pres0 + 3 for the tendencies).  The next plane is loaded into
// registers BEFORE the current level is computed and written to LDS after it (async-stage split),
// so HBM latency hides behind the ~400 flops of the stencil.  pres0 needs only (c, i-1, j-1, k-1) and
// is read directly.  x is periodic: tile column c holds global i = (i0 - 1 + c) mod nx.
#include "udc_internal.h"
#include "udc_mom_arith.h"
#include "udc_closure_arith.h"
#include <cstdlib>

namespace {

// Tile of the two stencil kernels: 32 x 8 rather than the 64 x 4 of the streaming kernels -- (34 x 10)/(32 x 8)
// = 1.33 halo overhead instead of 1.55, and 19 field-planes of it fit three workgroups into a CU's 160 KB.
#ifndef MOM_TX
#define MOM_TX 32
#define MOM_TY 8
#endif
constexpr int MX = MOM_TX, MY = MOM_TY;
constexpr int LX = MX + 2, LY = MY + 2, LN = LX * LY;   // 34 x 10 = 340 doubles per field-plane
constexpr int NT = MX * MY;                               // 256 threads
static_assert((NT == 256 || NT == 512) && LN - NT <= NT, "one own cell and at most one halo cell per thread");
constexpr int MOM_WAVES = tune::MOM_PER_CU;
static TileGrid lds_tile_grid(const Geo &g) {
  TileGrid t; t.gx = (g.nx + MX - 1) / MX; t.gy = (g.ny + MY - 1) / MY; t.tiles = t.gx * t.gy; return t;
}

struct MomArgs {
  const double *u, *v, *w, *p, *ek;
  double *up, *vp, *wp;
  const double *um, *vm, *wm;   // only read in PUP mode
  double rk3coefi;
  int wrap_vp;                  // single slab: also store row 0 of vp into ghost row ny (bcpup's cyclic pvp)
  int um_is_u0;                 // RK stage 1 after an aliased stage 3: um == u0, already staged in LDS
  int kbeg, kend;               // levels [kbeg, kend) of this launch (the whole column unless the sweep is pipelined with the solve)
};

template <int NF, int CPT = 1>
struct Stage {            // values of one plane held in registers between "load" and "commit"
  double c[NF][CPT];      // this thread's own cells
  double h[NF];           // one halo element (threads 0..LN-NT-1)
};

// tile element e (0..LN-1) -> offset from the plane's row base; halo elements are e >= NT
__device__ __forceinline__ void halo_coords(int e, int &lx, int &ly) {
  // enumerate the halo cells: rows 0 and LY-1 fully (2*LX), then columns 0 and LX-1 of rows 1..LY-2 (2*MY)
  if (e < LX) { ly = 0; lx = e; }
  else if (e < 2 * LX) { ly = LY - 1; lx = e - LX; }
  else { const int r = e - 2 * LX; ly = 1 + (r >> 1); lx = (r & 1) ? LX - 1 : 0; }
}

// FRESH: the tendencies are known to be zero on entry (fused substep) -> not read at all.
// PUP:   store the predicted velocity pup = up + um/rk3coef (fillps, src/modpois.f90:942-944, with
//        pwp(kb) = 0 of bcpup) instead of the bare tendency, so that the divergence and the projection
//        read 3 arrays instead of 6 (um,vm,wm are read here once instead of twice downstream).
// CPT:   cells per thread (rows ty and ty + MY/CPT of the tile).  2 halves the threads of a workgroup and gives every
//        wave two independent stencils to interleave: measured 0.454 ms against 0.369 ms for CPT = 1 at 256^3 (181
//        VGPRs, 2 waves/SIMD), so only CPT = 1 is instantiated.
// PGRAD = false (pressure-total form, udc_plan.h): pres0 is neither staged nor differenced.
template <bool ADV, bool DIFF, bool LES, bool FORCES, bool FRESH, bool PUP, int CPT, bool PGRAD = true>
__global__ __launch_bounds__(NT / CPT, CPT == 1 ? MOM_WAVES : 2) void mom_lds_kernel(Geo g, TileGrid tg, Metrics m, MomArgs a, double numol, int kc) {
  constexpr int NF = (DIFF && LES) ? 4 : 3;
  constexpr int RY = MY / CPT;        // rows of threads
  constexpr int NTH = NT / CPT;
  static_assert(LN - NT <= NTH, "one halo cell per thread");
  __shared__ double s[4][NF][LN];     // 4 rotating plane buffers: one barrier per level is enough
  constexpr bool PRS = ADV && PGRAD;                 // pres0 takes part
  __shared__ double sp[PRS ? 3 : 1][PRS ? LN : 1];   // pres0: planes k-1, k and the one being filled (k+1)
  __shared__ double smet[2][NLEVMET + 4];            // LevelMet of levels k and k+1 (see udc_mom_arith.h)

  // workgroup -> (tile, k-chunk); XCD-aware like tile_decode but with chunks instead of planes
  const unsigned L = blockIdx.x;
  const int chunk = L / tg.tiles;
  const unsigned lp = L - (unsigned)chunk * tg.tiles;
  int byl, bx;
  xcd_tile(tg, lp, byl, bx);
  const int by = tile_row(tg, byl);
  const int i0 = bx * MX, j0 = by * MY;
  const int tx = threadIdx.x, ty = threadIdx.y, tid = ty * MX + tx;
  const int i = i0 + tx;
  const int k0 = a.kbeg + chunk * kc;
  const int k1 = min(k0 + kc, a.kend);

  const double *fld[4] = {a.u, a.v, a.w, a.ek};

  // global column offsets (within a plane) of the elements this thread stages
  const int ic = i % g.nx;                               // partial tiles: columns beyond nx hold the periodic images
  int jj[CPT], own_l[CPT];
  long own_off[CPT];
  bool inside[CPT];
#pragma unroll
  for (int c = 0; c < CPT; ++c) {
    jj[c] = j0 + ty + c * RY;
    inside[c] = i < g.nx && jj[c] < g.ny;
    own_off[c] = (long)ic + (long)g.sy * (min(jj[c], g.ny + HY - 1) + HY);
    own_l[c] = (ty + c * RY + 1) * LX + (tx + 1);
  }
  int hlx = 0, hly = 0;
  const bool has_halo = tid < LN - NT;
  if (has_halo) halo_coords(tid, hlx, hly);
  int hi = i0 - 1 + hlx;
  hi %= g.nx; if (hi < 0) hi += g.nx;
  const int hj = min(j0 - 1 + hly, g.ny + HY - 1);
  const long halo_off = (long)hi + (long)g.sy * (hj + HY);
  const int halo_l = hly * LX + hlx;

  auto load_plane = [&](int k, Stage<NF, CPT> &st) {
    const long pb = g.sz * (long)(k + HZ);
#pragma unroll
    for (int f = 0; f < NF; ++f) {
#pragma unroll
      for (int c = 0; c < CPT; ++c) st.c[f][c] = fld[f][pb + own_off[c]];
      st.h[f] = has_halo ? fld[f][pb + halo_off] : 0.0;
    }
  };
  auto commit_plane = [&](int buf, const Stage<NF, CPT> &st) {
#pragma unroll
    for (int f = 0; f < NF; ++f) {
#pragma unroll
      for (int c = 0; c < CPT; ++c) s[buf][f][own_l[c]] = st.c[f][c];
      if (has_halo) s[buf][f][halo_l] = st.h[f];
    }
  };
  // pres0 is only needed at (c, i-1, j-1, k-1): its planes run one level behind the velocity planes
  double pst_c[CPT], pst_h = 0.;
  const bool p_halo = has_halo && (hlx == 0 || hly == 0);
  auto load_p = [&](int k) {
    const long pb = g.sz * (long)(k + HZ);
#pragma unroll
    for (int c = 0; c < CPT; ++c) pst_c[c] = a.p[pb + own_off[c]];
    pst_h = p_halo ? a.p[pb + halo_off] : 0.0;
  };
  auto commit_p = [&](int buf) {
#pragma unroll
    for (int c = 0; c < CPT; ++c) sp[buf][own_l[c]] = pst_c[c];
    if (p_halo) sp[buf][halo_l] = pst_h;
  };

  // level metrics: thread t < NLEVMET stages entry t, one level ahead of its use, riding on the plane prefetch's wait
  const bool met_thread = tid < NLEVMET;
  const double *mp = levmet_src(m, met_thread ? tid : 0) + 1;      // entry of level k = mp[k]
  double mreg = 0.;
  if (met_thread) {
    smet[k0 & 1][tid] = mp[k0];
    if (k0 + 1 < k1) mreg = mp[k0 + 1];
  }
  // prologue: planes k0-1, k0, k0+1 into buffers 0..2, plane k0+2 into registers
  Stage<NF, CPT> st;
  load_plane(k0 - 1, st); commit_plane(0, st);
  load_plane(k0, st);     commit_plane(1, st);
  load_plane(k0 + 1, st); commit_plane(2, st);
  if (PRS) {
    load_p(k0 - 1); commit_p(0);
    load_p(k0);     commit_p(1);
  }
  if (k0 + 1 < k1) { load_plane(k0 + 2, st); if (PRS) load_p(k0 + 1); }
  int bm = 0, bc = 1, bp = 2, bn = 3;      // buffers holding planes k-1, k, k+1 and the one being filled (k+2)
  int qm = 0, qc = 1, qn = 2;              // pres0 buffers: planes k-1, k and the one being filled (k+1)

  for (int k = k0; k < k1; ++k) {
    // this level's direct operands (tendencies, um) are requested before the barrier so that their
    // latency overlaps the barrier wait and the LDS traffic
    long cc[CPT];
    double tu[CPT], tv[CPT], tw[CPT], pum[CPT], pvm[CPT], pwm[CPT];
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
      cc[c] = g.sz * (long)(k + HZ) + own_off[c];
      tu[c] = tv[c] = tw[c] = pum[c] = pvm[c] = pwm[c] = 0.;
      if (inside[c]) {
        // read-once operands: nontemporal, so that they do not push the tile halos of u0, v0, w0, ekm (re-read by the neighbouring
        // workgroups) out of the XCD's L2
        if (!FRESH) { tu[c] = NT_LOAD(&a.up[cc[c]]); tv[c] = NT_LOAD(&a.vp[cc[c]]); tw[c] = NT_LOAD(&a.wp[cc[c]]); }
        if (PUP && !a.um_is_u0) { pum[c] = NT_LOAD(&a.um[cc[c]]); pvm[c] = NT_LOAD(&a.vm[cc[c]]); pwm[c] = NT_LOAD(&a.wm[cc[c]]); }
      }
    }
    // everyone has finished level k-1 (last readers of buffer bn) and committed plane k+1
    __syncthreads();
    if (k + 1 < k1) {
      commit_plane(bn, st);                                // plane k+2, read from level k+1 on
      if (PRS) commit_p(qn);                               // pres0 plane k+1
      if (met_thread) smet[(k + 1) & 1][tid] = mreg;       // metrics of level k+1 (visible after the next barrier)
      if (k + 2 < k1) {
        if (met_thread) mreg = mp[k + 2];
        load_plane(k + 3, st); if (PRS) load_p(k + 2);     // in flight while this level is computed
      }
    }
    const LevelMetLds lm{smet[k & 1]};
    const double *um_ = s[bm][0], *uc_ = s[bc][0], *up_ = s[bp][0];
    const double *vm_ = s[bm][1], *vc_ = s[bc][1], *vp_ = s[bp][1];
    const double *wm_ = s[bm][2], *wc_ = s[bc][2], *wp_ = s[bp][2];
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
      if (!inside[c]) continue;
      const int o = own_l[c];
      MomVals qq;
      qq.u_c = uc_[o]; qq.u_xm = uc_[o - 1]; qq.u_xp = uc_[o + 1]; qq.u_ym = uc_[o - LX]; qq.u_yp = uc_[o + LX];
      qq.u_zm = um_[o]; qq.u_zp = up_[o]; qq.u_xp_ym = uc_[o + 1 - LX]; qq.u_xp_zm = um_[o + 1];
      qq.v_c = vc_[o]; qq.v_xm = vc_[o - 1]; qq.v_xp = vc_[o + 1]; qq.v_ym = vc_[o - LX]; qq.v_yp = vc_[o + LX];
      qq.v_zm = vm_[o]; qq.v_zp = vp_[o]; qq.v_xm_yp = vc_[o - 1 + LX]; qq.v_yp_zm = vm_[o + LX];
      qq.w_c = wc_[o]; qq.w_xm = wc_[o - 1]; qq.w_xp = wc_[o + 1]; qq.w_ym = wc_[o - LX]; qq.w_yp = wc_[o + LX];
      qq.w_zm = wm_[o]; qq.w_zp = wp_[o]; qq.w_xm_zp = wp_[o - 1]; qq.w_ym_zp = wp_[o - LX];
      if (PRS) { qq.p_c = sp[qc][o]; qq.p_xm = sp[qc][o - 1]; qq.p_ym = sp[qc][o - LX]; qq.p_zm = sp[qm][o]; }
      if (DIFF && LES) {
        const double *em_ = s[bm][NF - 1], *ec_ = s[bc][NF - 1], *ep_ = s[bp][NF - 1];
        qq.e_c = ec_[o]; qq.e_xm = ec_[o - 1]; qq.e_xp = ec_[o + 1]; qq.e_ym = ec_[o - LX]; qq.e_yp = ec_[o + LX];
        qq.e_zm = em_[o]; qq.e_zp = ep_[o];
        qq.e_xm_yp = ec_[o - 1 + LX]; qq.e_xm_ym = ec_[o - 1 - LX]; qq.e_xm_zm = em_[o - 1]; qq.e_xm_zp = ep_[o - 1];
        qq.e_ym_zm = em_[o - LX]; qq.e_ym_zp = ep_[o - LX]; qq.e_xp_ym = ec_[o + 1 - LX];
        qq.e_yp_zm = em_[o + LX]; qq.e_xp_zm = em_[o + 1];
      }
      mom_arith<ADV, DIFF, LES, FORCES, PGRAD>(qq, m, lm, k, numol, tu[c], tv[c], tw[c]);
      if (PUP) {
        if (a.um_is_u0) { pum[c] = qq.u_c; pvm[c] = qq.v_c; pwm[c] = qq.w_c; }
        tu[c] = tu[c] + pum[c] * a.rk3coefi;
        tv[c] = tv[c] + pvm[c] * a.rk3coefi;
        tw[c] = (k == 0) ? 0. : tw[c] + pwm[c] * a.rk3coefi;
      }
      NT_STORE(tu[c], &a.up[cc[c]]); NT_STORE(tv[c], &a.vp[cc[c]]); NT_STORE(tw[c], &a.wp[cc[c]]);
      if (a.wrap_vp && jj[c] == 0) a.vp[cc[c] + (long)g.sy * g.ny] = tv[c];
    }
    const int t = bm; bm = bc; bc = bp; bp = bn; bn = t;
    const int t2 = qm; qm = qc; qc = qn; qn = t2;
  }
}

// -------------------------------------------------------------------------------- closure
struct LdsAcc {      // neighbour access from the three staged planes (pointers to the tile centre element)
  const double *um, *uc, *up, *vm, *vc, *vp, *wm, *wc, *wp;
  __device__ __forceinline__ double u(int di, int dj, int dk) const {
    return (dk == 0 ? uc : (dk < 0 ? um : up))[dj * LX + di];
  }
  __device__ __forceinline__ double v(int di, int dj, int dk) const {
    return (dk == 0 ? vc : (dk < 0 ? vm : vp))[dj * LX + di];
  }
  __device__ __forceinline__ double w(int di, int dj, int dk) const {
    return (dk == 0 ? wc : (dk < 0 ? wm : wp))[dj * LX + di];
  }
};

// Same marching/staging scheme as mom_lds_kernel for u0, v0, w0; writes ekm, ekh.
// EKH = false: ekh is not written (nothing reads it in this substep: no transported scalar, not the stage whose fields the
// time-step maxima / statistics / restart files see) -- 8 of the kernel's 40 B per cell.
template <int SGS, bool EKH>
__global__ __launch_bounds__(NT) void closure_lds_kernel(Geo g, TileGrid tg, Metrics m, Params pr,
    const double *__restrict__ gu, const double *__restrict__ gv, const double *__restrict__ gw,
    double *__restrict__ ekm, double *__restrict__ ekh, int kc, int ghosts) {
  constexpr int NF = 3;
  __shared__ double s[4][NF][LN];
  const unsigned L = blockIdx.x;
  const int chunk = L / tg.tiles;
  const unsigned lp = L - (unsigned)chunk * tg.tiles;
  int byl, bx;
  xcd_tile(tg, lp, byl, bx);
  const int by = tile_row(tg, byl);
  const int i0 = bx * MX, j0 = by * MY;
  const int tx = threadIdx.x, ty = threadIdx.y, tid = ty * MX + tx;
  const int i = i0 + tx, j = j0 + ty;
  const bool inside = i < g.nx && j < g.ny;
  const int k0 = chunk * kc;
  const int k1 = min(k0 + kc, g.nz);
  const double *fld[3] = {gu, gv, gw};
  const int ic = i % g.nx;
  const int jc = min(j, g.ny + HY - 1);
  const long own_off = (long)ic + (long)g.sy * (jc + HY);
  int hlx = 0, hly = 0;
  const bool has_halo = tid < LN - NT;
  if (has_halo) halo_coords(tid, hlx, hly);
  int hi = i0 - 1 + hlx;
  hi %= g.nx; if (hi < 0) hi += g.nx;
  const int hj = min(j0 - 1 + hly, g.ny + HY - 1);
  const long halo_off = (long)hi + (long)g.sy * (hj + HY);
  const int own_l = (ty + 1) * LX + (tx + 1);
  const int halo_l = hly * LX + hlx;
  auto load_plane = [&](int k, Stage<NF> &st) {
    const long pb = g.sz * (long)(k + HZ);
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      st.c[f][0] = fld[f][pb + own_off];
      st.h[f] = has_halo ? fld[f][pb + halo_off] : 0.0;
    }
  };
  auto commit_plane = [&](int buf, const Stage<NF> &st) {
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      s[buf][f][own_l] = st.c[f][0];
      if (has_halo) s[buf][f][halo_l] = st.h[f];
    }
  };
  // level metrics: lane t < NCLOSMET of every wave holds entry t, loaded one level ahead together with the plane
  // prefetch (its wait is the commit's); the stencil reads them with v_readlane into scalar registers.  No LDS: the
  // four plane buffers alone are 32 640 B, so five workgroups fit a CU
  const int lane = tid & 63;
  const bool met_thread = lane < NCLOSMET;
  const double *mp = closmet_src(m, met_thread ? lane : 0) + 1;    // entry of level k = mp[k]
  double mcur = met_thread ? mp[k0] : 0.;
  double mnext = (met_thread && k0 + 1 < k1) ? mp[k0 + 1] : 0.;
  Stage<NF> st;
  load_plane(k0 - 1, st); commit_plane(0, st);
  load_plane(k0, st);     commit_plane(1, st);
  load_plane(k0 + 1, st); commit_plane(2, st);
  if (k0 + 1 < k1) load_plane(k0 + 2, st);
  int bm = 0, bc = 1, bp = 2, bn = 3;
  for (int k = k0; k < k1; ++k) {
    __syncthreads();
    if (k + 1 < k1) {
      commit_plane(bn, st);
      if (k + 2 < k1) load_plane(k + 3, st);
    }
    const ClosMetLane lm{mcur};
    mcur = mnext;                                            // (arrived: the commit above waited for it)
    if (met_thread && k + 2 < k1) mnext = mp[k + 2];
    __builtin_amdgcn_sched_barrier(0);      // the prefetch is issued here, not wherever it shortens live ranges
    if (inside) {
      const int o = own_l;
      LdsAcc A{s[bm][0] + o, s[bc][0] + o, s[bp][0] + o, s[bm][1] + o, s[bc][1] + o, s[bp][1] + o,
               s[bm][2] + o, s[bc][2] + o, s[bp][2] + o};
      double em, eh;
      closure_arith<SGS>(A, m, lm, pr, k, em, eh);
      const long c = g.sz * (long)(k + HZ) + own_off;
      ekm[c] = em;                      // (re-read by the momentum sweep that follows: kept cacheable)
      if (EKH) NT_STORE(eh, &ekh[c]);
      if (ghosts) {
        // closurebc (src/modboundary.f90:447-500) folded in when the slab is the whole domain in y:
        // periodic ghost rows and the bottom/top ghost planes are written by the owning thread.
        const long up_row = (long)g.sy * g.ny, dn_row = -(long)g.sy * g.ny;
        const long wr = (j == 0) ? up_row : ((j == g.ny - 1) ? dn_row : 0);   // ny >= 4, so at most one
        if (wr) { ekm[c + wr] = em; if (EKH) ekh[c + wr] = eh; }
        const double nm = pr.numol, nh = pr.numol * pr.prandtlmoli;
        if (k == 0) {
          const double gm = 2. * nm - em, gh = (2. * nh) - eh;
          ekm[c - g.sz] = gm; if (EKH) ekh[c - g.sz] = gh;
          if (wr) { ekm[c - g.sz + wr] = gm; if (EKH) ekh[c - g.sz + wr] = gh; }
        }
        if (k == g.nz - 1) {
          const double gm = pr.bctopm == UDC_TOP_NOSLIP ? 2. * nm - em : em;
          const double gh = pr.bctopm == UDC_TOP_NOSLIP ? (2. * nh) - eh : eh;
          ekm[c + g.sz] = gm; if (EKH) ekh[c + g.sz] = gh;
          if (wr) { ekm[c + g.sz + wr] = gm; if (EKH) ekh[c + g.sz + wr] = gh; }
        }
      }
    }
    const int t = bm; bm = bc; bc = bp; bp = bn; bn = t;
  }
}

}  // namespace

// k-chunk length: each workgroup pays a 3-plane prologue, and the chip runs `slots` workgroups at a
// time (256 CUs x per_cu, register/LDS-limited), so pick the chunk that minimises rounds x (kc + 3).
static int pick_kc(const Geo &g, const TileGrid &tg, int per_cu) {
  const long slots = 256L * per_cu;
  int best = g.nz < 4 ? g.nz : 4;
  double best_cost = 1e300;
  for (int kc = 4; kc <= g.nz; ++kc) {
    const long blocks = (long)tg.tiles * ((g.nz + kc - 1) / kc);
    const long rounds = (blocks + slots - 1) / slots;
    const double cost = (double)rounds * (kc + 3);
    if (cost < best_cost - 1e-9) { best_cost = cost; best = kc; }
  }
  return best;
}

int closure_lds_tile_rows(const Geo &g) { return lds_tile_grid(g).gy; }

int k_closure_lds(udc_handle *h, bool ghosts, bool write_ekh, int rows) {
  const Geo &g = h->g;
  // rows 1 / 2: the tile rows next to the neighbouring ranks / the rows in between (k_ek_ghosts_overlapped exchanges in between)
  const TileGrid tg = rows == 0 ? lds_tile_grid(g) : tile_rows(lds_tile_grid(g), 1, rows == 1);
  // 90 VGPRs, 32 640 B LDS: five workgroups fit a CU.  Measured: 512x512x256 0.634 -> 0.592 ms, 1024x512x512 2.60 -> 2.38 ms
  // when the chunking fills them; at 256^3 (256 tiles) five shorter chunks per tile lose to four (0.19 against 0.172 ms)
  int kc = pick_kc(g, tg, tune::closure_per_cu(tg.tiles));
  const int chunks = (g.nz + kc - 1) / kc;
  dim3 b(MX, MY, 1), gr((unsigned)tg.tiles * (unsigned)chunks, 1, 1);
  double *u = h->fields[UDC_U0], *v = h->fields[UDC_V0], *w = h->fields[UDC_W0];
  double *ekm = h->fields[UDC_EKM], *ekh = h->fields[UDC_EKH];
  PROF(h, rows == 1 ? "closure_edge" : "closure");
  const int gh = ghosts ? 1 : 0;
#define CLOSURE_LAUNCH(S, E) hipLaunchKernelGGL((closure_lds_kernel<S, E>), gr, b, 0, h->stream, g, tg, h->m, h->p, u, v, w, ekm, ekh, kc, gh)
  const bool smag = h->p.sgs == UDC_SGS_SMAGORINSKY;
  if (smag) { if (write_ekh) CLOSURE_LAUNCH(1, true); else CLOSURE_LAUNCH(1, false); }
  else      { if (write_ekh) CLOSURE_LAUNCH(2, true); else CLOSURE_LAUNCH(2, false); }
#undef CLOSURE_LAUNCH
  HIP_OK(hipGetLastError());
  return 0;
}

int momentum_lds_tile_rows(const Geo &g) { return lds_tile_grid(g).gy; }
int momentum_lds_tile_height() { return MY; }

// part: a piece of the sweep -- tile rows [r0, r1) (r1 <= 0: all), levels [kbeg, kend) (kend <= 0: all); `more`: not the last piece
// of this substep (profiled under "<name>_edge", which bench.py folds into <name>)
int k_momentum_lds(udc_handle *h, bool adv, bool diff, bool forces, bool fresh, double rk3coefi, bool um_is_u0, const MomPart *part, bool pgrad) {
  const bool pup = fresh && rk3coefi != 0.;
  if (!pgrad && !(pup && adv && diff)) { udc_set_error("k_momentum_lds: the pressure-total form is the fused substep's only"); return 1; }
  const Geo &g = h->g;
  MomArgs a{h->fields[UDC_U0], h->fields[UDC_V0], h->fields[UDC_W0], h->fields[UDC_PRES0],
            h->fields[UDC_EKM], h->fields[UDC_UP], h->fields[UDC_VP], h->fields[UDC_WP],
            h->fields[UDC_UM], h->fields[UDC_VM], h->fields[UDC_WM], rk3coefi, (fresh && !h->slab) ? 1 : 0, um_is_u0 ? 1 : 0, 0, g.nz};
  TileGrid tg = lds_tile_grid(g);
  bool more = false;
  if (part) {
    if (part->r1 > 0) tg = tile_range(tg, part->r0, part->r1);
    if (part->kend > 0) { a.kbeg = part->kbeg; a.kend = part->kend; }
    more = part->more;
  }
  // k-chunk: long enough to amortise the 2-plane prologue, short enough to fill 256 CUs x 4 workgroups
  Geo gl = g;
  gl.nz = a.kend - a.kbeg;
  int kc = pick_kc(gl, tg, tune::MOM_PER_CU);
  const int chunks = (gl.nz + kc - 1) / kc;
  dim3 b(MX, MY, 1), gr((unsigned)tg.tiles * (unsigned)chunks, 1, 1);
  const bool les = h->p.sgs != UDC_SGS_DNS;
  const double nu = h->p.numol;
#define LAUNCH(A, D, L, F)                                                                         \
  do {                                                                                             \
    PROF(h, more ? "mom_" #A #D #L #F "_edge" : "mom_" #A #D #L #F);                               \
    if (pup && !pgrad) hipLaunchKernelGGL((mom_lds_kernel<A, D, L, F, true, true, 1, false>), gr, b, 0, h->stream, g, tg, h->m, a, nu, kc);    \
    else if (pup) hipLaunchKernelGGL((mom_lds_kernel<A, D, L, F, true, true, 1>), gr, b, 0, h->stream, g, tg, h->m, a, nu, kc);    \
    else if (fresh) hipLaunchKernelGGL((mom_lds_kernel<A, D, L, F, true, false, 1>), gr, b, 0, h->stream, g, tg, h->m, a, nu, kc);  \
    else hipLaunchKernelGGL((mom_lds_kernel<A, D, L, F, false, false, 1>), gr, b, 0, h->stream, g, tg, h->m, a, nu, kc);            \
  } while (0)
  if (adv && diff) {
    if (les) { if (forces) LAUNCH(true, true, true, true); else LAUNCH(true, true, true, false); }
    else     { if (forces) LAUNCH(true, true, false, true); else LAUNCH(true, true, false, false); }
  } else if (adv) {
    LAUNCH(true, false, true, false);
  } else if (diff) {
    if (les) LAUNCH(false, true, true, false); else LAUNCH(false, true, false, false);
  }
#undef LAUNCH
  HIP_OK(hipGetLastError());
  return 0;
}
