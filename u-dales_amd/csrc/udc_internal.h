// Internal declarations of libudcore (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <rocfft/rocfft.h>
#include <cstdint>
#include <cstdio>
#include <map>
#include <string>
#include <vector>
#include "../../include/udcore.h"
#include "udc_plan.h"
#include "udc_tuning.h"

// ---------------------------------------------------------------------------------------
// Device layout of every 3-D field (one geometry for all of them, so that one offset
// serves u0,v0,w0,pres0,ekm,up,... in a fused kernel):
//
//   a[ i + nx * ( (j + HY) + py * (k + HZ) ) ],  i = 0..nx-1, j = -HY..nyl-1+HY, k = -HZ..nz-1+HZ
//
// * no ghost columns in x: with the y-slab decomposition x is always whole and periodic on
//   a rank, so the kernels wrap the i index instead (the reference's x ghosts are, at every
//   point of use, periodic images: src/modboundary.f90:508-539, 1249-1255, 1365-1375);
//   rows are therefore nx*8 bytes, contiguous and 128-byte aligned for nx % 16 == 0.
// * HY = HZ = 2 ghost rows / planes (kappa scalars need 2, momentum 1).
// Reference index (if, jf, kf) (1-based interior) <-> device (if-1, jf-1, kf-1).
// ---------------------------------------------------------------------------------------
constexpr int HY = 2;
constexpr int HZ = 2;

// read-once / write-once operands of a sweep: nontemporal accesses (`nt` on the global_load / global_store), so that they do not push
// the lines neighbouring workgroups are about to re-read (tile halos, the planes above and below) out of the XCD's L2.  Measured at
// 256^3: momentum sweep 0.293 -> 0.285 ms, project + integrate 0.219 -> 0.205 ms (profiles/r03/nontemporal_ab.txt); where the next kernel
// reads the stored array straight back (the divergence's p into the FFT, the Thomas solution, ekm into the momentum sweep) they cost a
// little instead and are not used.  -DUDC_NO_NT compiles them back to plain accesses.
#ifndef UDC_NO_NT
#define NT_LOAD(p) __builtin_nontemporal_load(p)
#define NT_STORE(v, p) __builtin_nontemporal_store(v, p)
#else
#define NT_LOAD(p) (*(p))
#define NT_STORE(v, p) (*(p) = (v))
#endif
#if defined(__HIPCC__)
typedef double udc_v2d __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double2 nt_load2(const double2 *p) {
  const udc_v2d t = NT_LOAD(reinterpret_cast<const udc_v2d *>(p));
  return make_double2(t.x, t.y);
}
__device__ __forceinline__ void nt_store2(double2 v, double2 *p) {
  udc_v2d t; t.x = v.x; t.y = v.y;
  NT_STORE(t, reinterpret_cast<udc_v2d *>(p));
}
#endif

struct Geo {
  int nx, ny, nz;      // local interior (ny = rows of this y-slab)
  int py, pz;          // padded extents ny+2HY, nz+2HZ
  int sy;              // row stride   (nx, or nx + 16 for large power-of-two rows: udc_create)
  long sz;             // plane stride (= sy*py)
  long n;              // total elements
  int xg = 0;          // open x boundaries (udc_create_open_x): ghost columns kept at either end of a row -- the interior is
                       // i = xg .. nx - xg - 1 (udc_xopen.hip); 0: x periodic by index wrap
  __host__ __device__ inline long idx(int i, int j, int k) const {
    return (long)i + (long)sy * (j + HY) + sz * (long)(k + HZ);
  }
};

// ---------------------------------------------------------------------------------------
// Cell kernels use 64x4-thread tiles (one wave per x-row segment).  They are launched as a 1-D
// grid of tiles_per_plane * nz workgroups and decode (tile, k) themselves, XCD-aware: the
// hardware deals workgroup b to XCD b % 8, so within a plane XCD c is given a contiguous band of
// y-tiles (its private 4 MiB L2 then serves the j+-1 rows shared by neighbouring tiles), and the
// same (x,y) tile of plane k+1 lands on the same XCD (k+-1 reuse).  Placement only affects speed.
// ---------------------------------------------------------------------------------------
constexpr int TX = 64, TY = 4;
// A launch may cover a subset of the tile rows (y-slabs: the rows next to the neighbouring ranks first, their ghost-row exchange
// then runs beside the launch over the interior rows, udc_halo.hip): gy / tiles count the rows of THIS launch, and local tile row
// b stands for row  y0 + b + (b >= ysplit ? yjump : 0)  of the slab.
struct TileGrid { int gx, gy, tiles; int y0 = 0, ysplit = 1 << 30, yjump = 0; };
inline TileGrid tile_grid(const Geo &g) {
  TileGrid t; t.gx = (g.nx + TX - 1) / TX; t.gy = (g.ny + TY - 1) / TY; t.tiles = t.gx * t.gy; return t;
}
// rows of a split launch: edge = the first and last `e` tile rows of `full`, otherwise the rows in between (full.gy > 2 e)
inline TileGrid tile_range(const TileGrid &full, int r0, int r1) {      // the tile rows [r0, r1)
  TileGrid t = full;
  t.gy = r1 - r0; t.y0 = r0; t.tiles = t.gx * t.gy;
  return t;
}
inline TileGrid tile_rows(const TileGrid &full, int e, bool edge) {
  TileGrid t = full;
  if (edge) { t.gy = 2 * e; t.y0 = 0; t.ysplit = e; t.yjump = full.gy - 2 * e; }
  else { t.gy = full.gy - 2 * e; t.y0 = e; }
  t.tiles = t.gx * t.gy;
  return t;
}
__host__ __device__ __forceinline__ int tile_row(const TileGrid &t, int b) { return t.y0 + b + (b >= t.ysplit ? t.yjump : 0); }
// Launch index within a plane / k-chunk (lp) -> local tile row and column.  The hardware deals workgroup b to XCD b % 8: XCD c is
// given a contiguous run of tiles, row by row.  (Round 4 tried sub-blocks of 4 / 8 / 16 columns x all rows of the XCD's band instead,
// so that the tile above / below a tile is a few dispatch slots away: the sweeps' traffic and time did not move,
// profiles/r04/tile_order_ab.txt -- not kept.)
__host__ __device__ __forceinline__ void xcd_tile(const TileGrid &t, unsigned lp, int &byl, int &bx) {
  unsigned tt = lp;
  if ((t.tiles & 7) == 0) tt = (lp & 7u) * (t.tiles >> 3) + (lp >> 3);
  byl = tt / t.gx; bx = tt - byl * t.gx;
}
#if defined(__HIPCC__)
__device__ __forceinline__ bool tile_decode(const Geo &g, const TileGrid &t, int &i, int &j, int &k) {
  const unsigned L = blockIdx.x;
  k = L / t.tiles;
  const unsigned lp = L - (unsigned)k * t.tiles;
  unsigned tt = lp;
  if ((t.tiles & 7) == 0) tt = (lp & 7u) * (t.tiles >> 3) + (lp >> 3);
  const int byl = tt / t.gx, bx = tt - byl * t.gx;
  const int by = tile_row(t, byl);
  i = bx * TX + threadIdx.x;
  j = by * TY + threadIdx.y;
  return i < g.nx && j < g.ny;
}
#endif

// per-level metrics in constant-like global memory, indexed by the reference's k (0..nz+1)
struct Metrics {
  const double *dzf, *dzfi, *dzfi5, *dzfiq, *dzf2;   // 0..nz+1
  const double *dzhi, *dzhiq, *dzh2i, *dzh;          // 1..nz+1 (entry 0 = copy of entry 1)
  const double *dpdxl, *dpdyl;                       // 0..nz+1 (forcing, zero if unset)
  const double *mlen;                                // 0..nz+1 csz*delta(k), src/modsubgrid.f90:215
  const double *delta;                               // 0..nz+1 delta(k) = (dx dy dzf(k))^(1/3), src/modglobal.f90:793-797
  double dxi, dyi, dxiq, dyiq, dx2i, dy2i, dxi5, dyi5, dx2, dy2, dx, dy;
};

struct Params {
  double numol, prandtlmoli, prandtli, c_vreman, csz, uinf, vinf;
  int sgs, bctopm;
  int lbottom;     // floor wall function (src/modibm.f90:2021)
  double z0;
  int bare = 0;    // closure kernels leave the turbulent part of ekm only (ekh unwritten): the Vreman buoyancy correction
                   // (k_vreman_buoycorr) scales it, derives ekh and adds the molecular parts in the reference's order
};

struct ProfEntry { hipEvent_t a, b; int name; bool own_a; };

// Every environment switch of the library.  Read ONCE, in udc_create (udc_read_switches, udc_api.hip); no launcher reads the
// environment.  -1 / 0 = "not set: the library's own choice".  DESIGN.md section 7 lists what each one is for.
struct Switches {
  // order of a slab substep (A/B switches, each order is tested against the others)
  int force_slab = 0;        // UDC_FORCE_SLAB=1: one rank through the slab layout
  int force_comm = 0;        // UDC_FORCE_COMM=1: a real one-rank RCCL communicator under the forced slab path
  int halo_overlap = 1;      // UDC_HALO_OVERLAP=0: every ghost-row exchange in line
  int mom_pipe = 2;          // UDC_MOM_PIPE=0: momentum sweep not cut along the solve's k-chunks; 1: tile row 0 first, then the k-chunks of the
                             // other rows; 2 (default): the k-chunks of all rows, vp's ghost row travelling chunk by chunk
  int a2a_chunks = 0;        // UDC_A2A_CHUNKS: k-chunks of the transposes (0: the library's choice, pois_slab_init)
  int fft_fused = 1;         // UDC_FFT_FUSED=0: rocFFT + transpose kernels on the slab path
  int own_fwd = -1;          // UDC_OWN_FWD=0/1: single-slab forward half in own kernels
  int div_in_fft = 1;        // UDC_DIV_IN_FFT=0: separate divergence kernel on the slab path
  int sv_inline = 1;         // UDC_SV_INLINE=0: no scalar takes its RK3 update inside its sweep
  int p_transpose = 1;       // UDC_P_TRANSPOSE=0: p's ghost rows travel in an exchange of their own, not inside the backward transpose
  int ptotal = 1;            // UDC_PTOTAL=0: the fused substep keeps pres0 and p apart like the reference (single slab: no pressure-total form)
  int no_fold = 0, no_alias = 0;      // UDC_NO_FOLD / UDC_NO_ALIAS = 1
  int ek_always = 0;         // UDC_EK_ALWAYS=1: every substep writes ekm / ekh
  int scalar_pair = 1;       // UDC_SCALAR_PAIR=0: thl and qt swept one by one
  // tridiagonal solve
  int thomas = -1;           // UDC_THOMAS=0: the streaming kernel (one thread per mode) instead of register-resident segments
  int thomas_pair = 1;       // UDC_THOMAS_PAIR=0: rows ky and ny - ky not solved together (one GPU: rows of spec; slab ranks: mirrored runs of a line)
  int thomas_mirror_min = 256;      // slab ranks: shortest line solved in mirrored pairs (UDC_THOMAS_MIRROR_MIN; tests set 16)
  // variants of the line transforms (A/B switches; the launch shapes themselves are constants: udc_tuning.h)
  int nat_reg = 1, slab_yreg = 1;      // UDC_NAT_REG / UDC_SLAB_YREG = 0: the Stockham y passes instead of 16 x N2 in registers
};
void udc_read_switches(Switches &sw);

struct udc_handle {
  Switches sw;
  udc_config cfg;
  Geo g;
  Metrics m;
  Params p;
  int device;
  hipStream_t stream = nullptr;
  std::vector<double *> fields;         // device arrays, UDC_FIELD ids
  double *metrics_dev = nullptr;        // backing store for Metrics arrays
  // Poisson
  double *spec = nullptr;               // complex (nkx, ny, nz) spectral work array
  double *ztab = nullptr;               // Thomas pivot table z(kx,ky,k)
  bool thomas_lds = false, thomas_lds_slab = false;   // which kernel solves (and hence how its table is laid out)
  double *ev = nullptr;                 // eigenvalue xrt(kx)+yrt(ky)
  double *tri = nullptr;                // a,b,c (3*(nz+2))
  int nkx = 0, nkxp = 0;                // r2c modes in x and the padded row pitch of `spec`
  double btopD = 0.;                    // Dirichlet top coefficient of the singular mode
  rocfft_plan plan_fwd = nullptr, plan_bwd = nullptr;
  rocfft_execution_info info_fwd = nullptr, info_bwd = nullptr;
  void *fft_work = nullptr;
  double *rbuf = nullptr;               // compact real (nx,ny,nz) staging, only if rocFFT rejects strides
  bool fwd_compact = false, bwd_compact = false;
  // reductions
  double *red = nullptr;                // small device scratch
  double *red_host = nullptr;           // pinned
  size_t red_cap = 0;                   // doubles in red / red_host
  double *partials = nullptr;           // per-workgroup partial results of the two-stage reductions
  size_t partials_cap = 0;
  double *chk_host = nullptr, *chk_dev = nullptr;      // checksim's four numbers (udc_checksim_begin / _end)
  hipEvent_t ev_chk = nullptr;
  bool chk_pending = false;
  // profiling
  // transported scalars: passive scalars occupy slots 0..nsv-1 (kappa scheme, zero-flux top and floor); the
  // temperature equation (ltempeq, udc_set_tempeq) occupies slot 15.  `slots` lists the active ones.
  struct Slot {
    int adv = 1;          // 1 = kappa (advecc_kappa), 2 = cd2 (advecc_2nd)
    int top = 0;          // 0 = zero-flux copy, 1 = fluxtop with topval = flux (src/modboundary.f90:1494), 2 = valuetop
    double topval = 0.;
    double floorflux = 0.;   // wtsurf in bottom's Neumann floor (src/modibm.f90:2035-2047); 0 for passive scalars
    int kappa_ghosts = 0;    // kappa on thl: vertical ghosts of the reference's thl0c copy (1 flux top, 2 value top), else 0
    bool tke = false;        // e120: diffused with ekm (diffe), clipped at e12min, own floor/top ghosts, no floor flux
  };
  Slot slot[16];
  std::vector<int> slots;
  // passive scalars with an inflow / outflow in x (BCxs = 2; udc_set_scalar_bcx): the fields stay ghost-free and periodic by
  // index; the two east ghost columns sv0(ie+1), sv0(ie+2) live here as planes [pz][py] per scalar, the inflow profile per level
  int scal_bcx = 1;
  double bcx_uout = 0., bcx_rk3coef = 0.;
  double *bcx_uout_dev = nullptr;        // the outlet's convection speed uouttot (device scalar)
  double *bcx_wlev = nullptr;            // without a prescribed volume flow: uouttot = sum_k wlev(k) u0av(k) of the previous substep's end
  bool bcx_uout_avg = false;
  double *bcx_east[13] = {nullptr};      // [2][pz][py] per passive scalar: ie+1, ie+2
  double *bcx_prof = nullptr;            // [nsv][nz+2], indexed by the reference's k
  std::vector<double> bcx_prof_host;
  // one-equation closure constants (udc_set_tke)
  struct Tke { double cm = 0., cn = 0., ch1 = 0., ch2 = 0., ce1 = 0., ce2 = 0., e12min = 5e-5, grav = 9.81, thvs = 0.; int ldelta = 0; } tke;
  // udc_set_level_forcing: A and B share one device block; `stage` is its pinned host copy and `copied` marks the last upload done
  struct LevelForcing { int tend = -1, src = -1, when = 0; double *A = nullptr, *B = nullptr, *stage = nullptr; hipEvent_t copied = nullptr; };
  std::vector<LevelForcing> level_forcings;
  int coriolis_mode = 0;       // 0 off, 1 lcoriol, 2 lprofforc (src/modforces.f90:600-717)
  double om22 = 0., om23 = 0.;
  double *ug = nullptr;        // [nz+2] geostrophic wind profile (lprofforc)
  // constant scalar sources (udc_set_scalar_source): a dense box per scalar, local device indices [lo, hi]
  struct ScalarSource { double *d = nullptr; int lo[3] = {0, 0, 0}, hi[3] = {-1, -1, -1}; };
  ScalarSource svsrc[16];
  int lchem = 0;               // udc_set_chem
  double chem_k1 = 0., chem_jno2 = 0.;
  // shiftedPBCs (udc_set_shifted_pbc): a, sinx[nx], u0av[nz] on the device
  double shift_a = 0.;
  double *shift_tab = nullptr;
  // floor wall function choice (udc_set_floor_wf): BCbotm 3 neutral / 2 wfuno, BCbotT 1 flux / 2 wfuno
  int floor_bcbotm = 3, floor_bcbott = 1;
  double floor_thls = 0., floor_z0h = 0., floor_prt = 0.71;
  double fkar = 0.41;          // von Karman constant of every wall function (udc_set_fkar; &WALLS fkar, src/modglobal.f90:317)
  // temperature equation off: the reference's thl0 keeps prof.inp's values for ever and wfuno still reads its first level
  bool floor_thl_air_on = false;
  double floor_thl_air = 0.;
  bool lmoist = false;         // qt transported in slot 13 (udc_set_moisture)
  int lbuoyancy = 0;           // forces' buoyancy term (dry air), needs the temperature equation
  int lbuoycorr = 0;           // Vreman buoyancy correction (udc_set_buoycorr)
  double rigc = 0.25;
  double *dthv_top = nullptr;  // udc_calthv: the top ghost planes of thl0, qt0 as the last explicit thermodynamics call saw them
  bool dthv_top_on = false;    // ... in force until the next time integration (the closures' dthvdz at level ke)
  double grav = 9.81;
  double *lev_part = nullptr, *lev_sum = nullptr;   // per-level slab sums (thvh)
  double *lev_sum16 = nullptr;                      // udc_slab_averages: up to 16 fields x (nz+2)
  size_t lev_cap = 0;
  // moist thermodynamics (udc_set_moist_thermo): MT_N tables of [nz+2] indexed by the reference's k, kept between
  // thermodynamics calls (presf/exnf feed the next call's `thermo`, presh/exnh/thvh the next forces)
  enum { MT_PRESF, MT_PRESH, MT_EXNF, MT_EXNH, MT_THVH, MT_THL0AV, MT_QT0AV, MT_QL0AV, MT_TH0AV, MT_STATE_N,
         MT_ZF = MT_STATE_N, MT_ZH, MT_SUMS, MT_N = MT_SUMS + 3 };
  double *mt = nullptr;
  bool mt_valid = false;       // diagfld has run at least once
  double thls = 0., qts = 0., ps = 0.;
  int lqlnr = 0;               // condensate by Newton-Raphson (src/modthermodynamics.f90:37)
  double *thlpcar = nullptr;   // [nz+2] radiative heating profile added by forces (src/modforces.f90:104-110), or null
  // masscorr (src/modforces.f90:328): prescribed volume-flow rates
  int luvolflowr = 0, lvvolflowr = 0;     // luvolflowr: 1 = volume flow (luvolflowr), 2 = flow through the outlet plane (luoutflowr)
  int uvol_req = 0, uout_req = 0;         // what udc_set_masscorr / udc_set_masscorr_outflow asked for; luvolflowr is derived from both
  double uvol_rate = 0., uout_rate = 0.;
  double *outlet_w = nullptr;             // luoutflowr: dy dzf(k) / outlet area, [nz+2] indexed by the reference's k
  double uflowrate = 0., vflowrate = 0., zsize = 0.;
  double dzhi_top = 0.;      // dzhi(ke+1)
  // inflow / outflow in x (&BC BCxm = 2, udc_create_open_x; udc_xopen.hip): the device grid is the deck's plus one ghost column at
  // either end (g.xg = 1), swept as if periodic; the columns are set where the reference sets its x ghosts
  int xg = 0;
  double *xo_prof = nullptr;          // uprof, vprof: [2][nz+2], indexed by the reference's k
  double *xo_east = nullptr;          // v0, w0, vm, wm at i = ie+1: [4][pz][py] (the convective outlet's own state)
  double *xo_west = nullptr;          // u0, v0, w0, um, vm, wm at i = ib-1 as the last `boundary` left them: [6][pz][py] (the integration
                                      // runs over ib:ie in the reference; here its result in that column is put back)
  // BCxm = 3: the inlet from a precursor run's planes (udc_set_open_x_inlet) instead of the profile: u0, um, v0, vm, w0, wm at the inlet,
  // [6][pz][py] each -- `now` is what the last `boundary` applied (bcpup reads it), `next` what the coming one will
  int xo_driver = 0;
  double *xo_inlet_now = nullptr, *xo_inlet_next = nullptr;
  bool xo_inlet_fresh = false;
  // the temperature on such a handle (&BC BCxT = 2: xTi_profile, xTo_convective; udc_set_open_x_thl): the inflow profile [nz+2] by k,
  // thl0 / thlm at ie+1 and at ib-1 as the last `boundary` left them ([2][pz][py] each)
  double *xo_thl_prof = nullptr, *xo_thl_east = nullptr, *xo_thl_west = nullptr, *xo_thl_top = nullptr;      // (top: k_xo_thl_top's row [py])
  // ... the total water likewise (&BC BCxq = 2: xqi_profile mirrors the ghost about the profile, xqo_convective; udc_set_open_x_qt)
  double *xo_qt_prof = nullptr, *xo_qt_east = nullptr, *xo_qt_west = nullptr;
  // BCxT / BCxq / BCxs = 3: the inlet's ghost columns from the planes of a precursor run (xTi_driver, xqi_driver, xsi_driver;
  // udc_set_open_x_inlet_scalar): f0driver, fmdriver, [2][pz][py] each; slot 0 thl, 1 qt, 2 + n scalar n; `now` / `next` as for the flow
  double *xo_sc_in_now[15] = {nullptr}, *xo_sc_in_next[15] = {nullptr};
  bool xo_sc_fresh[15] = {false};
  // passive scalars there (&BC BCxs = 2: xsi_profile, xso_convective; udc_set_open_x_scalars; xg = 2): the inflow profiles [nsv][nz+2], and
  // per scalar the four ghost columns ib-2, ib-1, ie+1, ie+2 of sv0 and svm as the last `boundary` left them ([8][pz][py])
  double *xo_sv_prof = nullptr, *xo_sv_cols[13] = {nullptr};
  bool xo_sample_gap = false;         // udc_set_open_x_sample_gap: the next fused stage-3 substep ends ahead of `boundary` (one-shot)
  bool xo_boundary_owed = false;      // a fused stage-3 substep ended before `boundary` (statistics on: they sample in between); the next substep
                                      // runs it first if no udc_boundary came
  bool xo_rhs_mirrored = false;       // the divergence kernel has written the right-hand side into the solver's doubled row itself
  bool xo_hold = false;               // the next refresh of uouttot is skipped (udc_set_open_x_outflow, hold_first)
  udc_handle *xpois = nullptr;        // the pressure solve's own periodic domain: the row and its mirror image, 2 itot wide
  const udc_handle *xo_src = nullptr; // (that handle, during a solve whose x transform evaluates the divergence of the owner's tendencies itself)
  bool poisson_only = false;          // (that handle: p and the solver's arrays only)
  hipEvent_t xo_ev[2] = {nullptr, nullptr};
  int bczp = 1;              // &BC BCzp: 1 tridiagonal solve in z, 2 the cosine transform's solution (udc_set_poisson_bczp)
  bool um_alias = false;                // um,vm,wm are logically equal to u0,v0,w0 (after RK stage 3 of a fused
                                        // substep); the UM buffers are stale until stage 1 rotates the pointers
  bool no_alias = false;                // UDC_NO_ALIAS=1: always copy (A/B switch)
  bool tend_scratch = false;            // up,vp,wp hold leftovers of a fused substep (logically zero)
  bool no_fold = false;                 // UDC_NO_FOLD=1: keep separate ghost-row kernels on a single slab (A/B switch)
  bool ekh_stale = false;               // the last closure wrote ekm only (no reader of ekh in that substep)
  bool ek_stale = false;                // the last fused substep kept ekm / ekh in LDS only: the arrays hold an older substep's values
  bool ek_always = false;               // UDC_EK_ALWAYS=1: every substep writes ekm / ekh
  bool no_div_in_fft = false;           // UDC_DIV_IN_FFT=0: slab path with a separate divergence kernel (A/B switch)
  // immersed boundary (udc_ibm.hip): per grid (u, v, w, c) the global point lists as given, and this slab's points
  // (local 0-based i, j, k triplets) with their neighbour flags on the device
  struct IbmGrid {
    std::vector<int> solid_g, bound_g;
    bool given = false;
    int nsolid = 0, nbound = 0;
    int *solid = nullptr, *bound = nullptr;
    unsigned char *solid_fl = nullptr, *bound_fl = nullptr;
    // this slab's solid points sorted by level (for the masked slab sums of avexy_ibm): points of device level k are
    // lev_pts[3 lev_off[k] .. 3 lev_off[k+1]), k = 0..nz; fluid_cnt[k] = fluid cells of the whole level (all slabs)
    int *lev_pts = nullptr, *lev_off = nullptr;
    std::vector<double> fluid_cnt;
    double *cnt_dev = nullptr;          // fluid_cnt on the device ([nz+2], indexed by the reference's k)
    unsigned char *bound_fl2 = nullptr; // c grid: the momentum masks on the faces of a fluid-boundary cell (advecc2nd_corr_conservative)
  };
  IbmGrid ibm[4];
  // facet sections of the wall functions (udc_ibm_wf.hip): this slab's sections grouped by boundary cell in file order
  struct IbmSections {
    int ncell = 0, nsec = 0;
    int *cell = nullptr, *off = nullptr, *comprec = nullptr, *recids = nullptr;     // cell: global 1-based i, j, k per row
    double *area = nullptr, *dist = nullptr, *norm = nullptr, *z0 = nullptr, *z0h = nullptr, *tsurf = nullptr, *recpt = nullptr, *tmask = nullptr;
    std::vector<int> order;             // position in the caller's list of each kept section (later per-section tables follow it)
    int nglobal = 0;                    // sections the caller listed (all slabs)
    int noverride = 0;                  // sections of this slab whose reconstruction cell is out of reach (simple reconstruction instead)
    int *lgr = nullptr;                 // c grid, latent wall flux (udc_set_ibm_wallmoist): vegetated facet; saturation humidity /
    double *qwall = nullptr, *hurel = nullptr, *resc = nullptr, *ress = nullptr;      // prescribed flux, humidity, resistances
    int *fac = nullptr;                 // facet (1-based) of every kept section (udc_set_ibm_facet_output: lwritefac)
  };
  IbmSections ibm_sec[4];               // u, v, w (wallfunmom), c (wallfunheat)
  // lwritefac (src/modibm.f90:1246-1282): per-facet tau_x, tau_y, tau_z, pres, pres2, htc, cth -- this substep's area-weighted sums
  // [7][nfcts] and their running time integrals; the c sections of this slab that carry the pressure (every one, skipped or not)
  int fac_n = 0;
  double *fac_now = nullptr, *fac_av = nullptr, *fac_area = nullptr;
  int fac_npres = 0;
  int *fac_pcell = nullptr, *fac_pfac = nullptr;
  double *fac_parea = nullptr;
  double fac_sample_dt = -1.;           // >= 0: the next ibmwallfun samples the facets (RK stage 3) with this dt, once
  int ibm_iwallmom = 1;                 // 1: no wall functions; 2: Uno et al. stability functions; 3: neutral log law
  int ibm_iwallmoist = 0;               // wallfunheat's latent part: 0 off (impermeable walls); 1 prescribed per section; 2 moist_flux
  int ibm_iwalltemp = 0;                // wallfunheat: 0 off (adiabatic walls); 1 prescribed fluxes per section; 2 from the facet temperatures
  double ibm_prt = 0.71;
  double *ibm_zgrid = nullptr;          // zf(1 : nz+1), zh(1 : nz+1)
  double *bottom_diag[3] = {nullptr, nullptr, nullptr};      // tau_x, tau_y, thl_flux planes [ny_l][nx] (udc_bottom_diagnostics)
  bool ibm_on = false;
  bool ibm_conservative = false;        // lconservativeibm: which advecc2nd_corr ibmnorm applies to thl, qt
  double *ibm_val = nullptr;            // device scalar: the value solid() gives thl inside obstacles (volume mean of thl0av)
  bool ibm_wrap_x = true, ibm_wrap_y = true;     // mask look-ups across the domain's lateral boundaries (udc_set_ibm_mask_wrap)
  double *ibm_wlev = nullptr;           // masscorr's per-level weights with the masks, u then v ([2][nz+2])
  // statistics accumulators (udc_stats.hip), UDC_ST_* ids
  std::vector<double *> stats;
  bool stats_on = false;
  // xytdump (udc_stats.hip): fluid masks of createmasks as bits per cell [nz][ny][nx] (nullptr = no obstacles), the
  // counts avexy_ibm divides by [7][nz], the nine running slab-average profiles [9][nz], scratch for the level sums
  bool xyt_on = false;
  unsigned char *st_mask = nullptr;
  double *st_cnt = nullptr, *st_prof = nullptr, *st_part = nullptr, *st_sum = nullptr, *st_table = nullptr;
  double *xy_table = nullptr, *xy_sum = nullptr, *y_table = nullptr, *y_sum = nullptr;      // xydump / ydump: the last sample's own tables
  bool xy_on = false, y_on = false;
  size_t st_part_cap = 0;
  // ytdump: running y-averages [15][nz][nx], column counts of the masks IIu, IIv, IIw, IIc, IIuw [5][nz][nx], scratch, table
  bool yt_on = false;
  double *yt_prof = nullptr, *yt_cnt = nullptr, *yt_sum = nullptr, *yt_table = nullptr;
  int yt_forced[7] = {0, 0, 0, 0, 0, 0, 0};      // masks whose first level udc_stats_set_masks' caller filled for avexy_ibm's rule
  // deferred execution (udc_set_deferred): the tendency routines of one RK3 substep are recorded instead of launched;
  // udc_tstep_integrate then runs the recorded sequence -- as the fused substep when it is the reference's own
  // (src/program.f90:142-197), routine by routine otherwise.  pend holds OP_* bits in call order.
  bool deferred = false;
  std::vector<unsigned> pend;
  int pend_rk = 0;                      // rk3step / dt handed to the recorded masscorr / poisson (0 = none recorded)
  double pend_dt = 0.;
  // what the last whole substep (or udc_halos / udc_boundary / udc_thermodynamics) left valid; any change of the
  // prognostic fields clears them, so that a driver's own halos / boundary / thermodynamics calls after a fused substep
  // (which already did them) cost nothing
  bool halos_fresh = false, boundary_fresh = false, thermo_fresh = false;
  long n_fused = 0, n_unfused = 0;      // deferred substeps that ran fused / routine by routine (udc_deferred_stats)
  bool prof = false;
  bool prof_focus_on = false;           // only launches whose name starts with prof_focus are timed
  std::string prof_focus;
  int prof_every = 1;                   // focus mode: the launches of every prof_every-th fused substep only (udc_profile_every)
  bool sv_inline[16] = {false};         // this fused substep: the scalar's RK3 update rides in its kappa sweep (the integration skips it)
  int last_inline_scalars = 0;          // ... how many in the last fused substep (udc_last_plan)
  bool sv_inline_last = false;          // ... on RK stage 3 (svm takes the new value too)
  double sv_inline_rk3coef = 0.;
  bool ptotal_now = false;              // inside a fused substep in the pressure-total form (k_ibm_norm: solid tendencies = + grad pres0)
  long substep_seq = 0, prof_phase = 0; // fused substeps run so far; ... when udc_profile_every was called
  std::vector<ProfEntry> prof_events;
  std::vector<hipEvent_t> prof_pool;
  hipEvent_t prof_chain = nullptr;      // end marker of the previous profiled launch (start marker of the next)
  std::vector<std::string> prof_names;
  std::map<std::string, int> prof_ids;
  std::map<int, std::pair<double, int>> prof_acc;
  // multi-GPU (y-slabs): RCCL communicator or in-process local group (udc_comm.hip)
  void *nccl = nullptr;
  // exchange bookkeeping (udc_comm_stats) and the timing-only mode (udc_comm_dry_run)
  struct CommStats { double a2a_ops = 0, a2a_block_bytes = 0, a2a_bytes = 0, halo_ops = 0, halo_prev = 0, halo_next = 0, red_ops = 0, red_doubles = 0; } cstat;
  struct CommTimed { hipEvent_t a, b; int kind; };
  std::vector<CommTimed> ctimed;
  bool comm_timing = false, comm_dry = false;
  Plan last_plan{};                     // the order the last fused substep ran in (udc_last_plan)
  bool have_plan = false;
  void *local_group = nullptr;
  void *shm_group = nullptr;            // test build only: the inter-process test transport (udc_comm_init_shm)
  bool slab = false;                    // distributed Poisson layout in use (nranks > 1 or UDC_FORCE_SLAB)
  int jtot = 0;                         // global number of rows
  double *halo_buf[4] = {nullptr, nullptr, nullptr, nullptr};   // to_prev, to_next, from_prev, from_next
  size_t halo_cap = 0;
  // slab Poisson: x-spectral rows (all kx, local rows), transposed block (local kx, all rows)
  int cx = 0;                           // kx chunk per rank = ceil(nkx / nranks)
  int nch = 1;                          // k-chunks of the all-to-all pipeline
  hipStream_t comm_stream = nullptr;    // all-to-all exchanges run here, overlapped with rocFFT on `stream`
  hipEvent_t ev_ready[16] = {}, ev_done[16] = {};
  hipEvent_t ev_halo_ready = nullptr, ev_halo_done = nullptr;      // k_halo_y_begin / _join
  hipEvent_t ev_vp[16] = {};            // vp's ghost row of k-chunk c has arrived (pipelined momentum sweep, UDC_MOM_PIPE=2)
  bool halo_async_pending = false;      // a k_halo_y_begin has not been joined yet (k_halo_y joins it before touching the shared buffers)
  // the momentum sweep pipelined with the slab solve (substep_fused, k_momentum_pipe_stage): tile row 0 is swept first over all
  // levels, the other rows level range by level range ahead of the x forward transform of the same k-chunk
  // rows_all: every piece covers all tile rows and hands vp's first row of its own levels on (UDC_MOM_PIPE=2, the default); else tile
  // row 0 was swept first over all levels and its row is already travelling (UDC_MOM_PIPE=1)
  struct MomPipe { bool active = false, forces = false, um_is_u0 = false, bottom = false, pgrad = true, rows_all = false; double rk3coefi = 0.; } mom_pipe;
  bool no_mom_pipe = false;             // UDC_MOM_PIPE=0
  bool p_scratch = false;               // the last fused substep ran in the pressure-total form: UDC_P holds the previous pres0 (download refused)
  bool p_ghost_in_transpose = false;    // this solve's backward blocks carry p's two ghost rows (substep_fused asked; k_poisson_solve_slab)
  bool vp_halo_pending = false;         // vp's ghost row is travelling (k_halo_y_begin): the x forward transform joins before its last row group
  bool no_halo_overlap = false;         // UDC_HALO_OVERLAP=0: every ghost-row exchange in line on the compute stream
  double *specA = nullptr, *specB = nullptr, *a2a_send = nullptr, *a2a_recv = nullptr;
  double *ev_slab = nullptr, *ztab_slab = nullptr;
  rocfft_plan plan_xf = nullptr, plan_xb = nullptr, plan_yf = nullptr, plan_yb = nullptr;
  rocfft_execution_info info_x = nullptr, info_y = nullptr;
  void *fft_work_slab = nullptr;
  // own line FFTs with the all-to-all packing fused in (udc_fft.hip; power-of-two nx, jtot), else rocFFT + transposes
  bool fft_fused = false;
  double *fft_tw = nullptr;             // twiddle tables
  int fft_L = 0, fft_C = 0;             // x rows / y columns per workgroup
  bool slab_yreg = false;               // slab path: y transforms as 16 x N2 in registers (ny = 128, 256, 512)
  bool own_fwd = false;                 // UDC_OWN_FWD=1, one GPU: divergence + x transform + y pass of udc_fft.hip instead of div_rhs + rocFFT's forward plan
  int nat_L = 0, nat_C = 0;
  bool nat_reg16 = false;               // ny = 256: the y pass as 16 x 16 in registers (UDC_NAT_REG=0: the Stockham kernel)
  bool div_in_fft = false;              // this solve: the x forward transform evaluates fillps' divergence itself
};

void udc_set_error(const char *fmt, ...);

// entry points that are not offered on a handle with open x boundaries (udc_create_open_x): said, not silently wrong
#define NO_OPEN_X(h, who)                                                                                              \
  do {                                                                                                                 \
    if ((h) && (h)->xg) { udc_set_error("%s: not offered with open x boundaries (udc_create_open_x) yet", who); return 1; } \
  } while (0)

#define HIP_OK(expr)                                                                  \
  do {                                                                                \
    hipError_t e_ = (expr);                                                           \
    if (e_ != hipSuccess) {                                                           \
      udc_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__,  \
                    __LINE__);                                                        \
      return 1;                                                                       \
    }                                                                                 \
  } while (0)

#define FFT_OK(expr)                                                                  \
  do {                                                                                \
    rocfft_status s_ = (expr);                                                        \
    if (s_ != rocfft_status_success) {                                                \
      udc_set_error("%s failed: rocfft status %d (%s:%d)", #expr, (int)s_, __FILE__,  \
                    __LINE__);                                                        \
      return 1;                                                                       \
    }                                                                                 \
  } while (0)

// kernel-launch bracket used for the optional HIP-event profile
hipEvent_t prof_take(udc_handle *h);
struct ProfScope {
  udc_handle *h; int id; hipEvent_t a, b; bool own_a = true;
  ProfScope(udc_handle *h_, const char *name);
  ~ProfScope();
};
#define PROF(h, name) ProfScope prof_scope_(h, name)

// ---- kernels (udc_mom.hip, udc_pois.hip, udc_scalar.hip, udc_halo.hip)
int k_closure(udc_handle *h);
int k_closure_lds(udc_handle *h, bool ghosts, bool write_ekh = true, int rows = 0);   // ghosts: closurebc folded in (single slab); write_ekh false: ekm only
int k_ek_ghosts(udc_handle *h, bool exchange = true);
int closure_lds_tile_rows(const Geo &g);      // tile rows of k_closure_lds over the slab
struct MomPart { int r0, r1, kbeg, kend; bool more; };      // a piece of the momentum sweep: tile rows [r0, r1), levels [kbeg, kend)
int k_momentum_lds(udc_handle *h, bool adv, bool diff, bool forces, bool fresh, double rk3coefi, bool um_is_u0 = false, const MomPart *part = nullptr,
                   bool pgrad = true);      // pgrad false: the pressure-total form (no gradient of pres0; fused substep only)
int momentum_lds_tile_rows(const Geo &g);
int momentum_lds_tile_height();
int k_momentum_pipe_stage(udc_handle *h, int c, bool row = true);
int k_momentum_pipe_row(udc_handle *h, int c);      // the sweep's level range that feeds k-chunk c of the slab solve (udc_api.hip)  // rk3coefi != 0: PUP mode   // LDS-staged k-marching version (default)
int k_level_sums_dev(udc_handle *h, int field, int n, int k0 = 0);      // udc_thermo.hip: masked, all-reduced sums of levels k0 .. k0 + n - 1 left on the device
int k_scalar_adv(udc_handle *h, int n);
int k_scalar_bcx_outlet(udc_handle *h);
int k_scalar_bcx_uout(udc_handle *h);
int k_scalar_bcx_capture(udc_handle *h, int n, const double *host, const int lb[3], const int ub[3]);
int k_scalar_bcx_fill_host(udc_handle *h, int n, double *host, const int lb[3], const int ub[3]);
int k_scalar_diff(udc_handle *h, int n);
int k_scalar_fused(udc_handle *h, int n, bool fresh);
int k_scalar_fused_pair(udc_handle *h, int na, int nb, bool fresh);      // 0 done, -1 not applicable, 1 error          // advection + diffusion in one sweep (same accumulation order)
int k_forces(udc_handle *h);
int k_coriolis(udc_handle *h, bool wrap_vp);                     // coriolis: lcoriol / lprofforc
int k_masscorr(udc_handle *h, double rk3coef, bool pup_mode, bool wrap_vp);   // masscorr, volume-flow branches
int k_bottom(udc_handle *h, bool wrap_vp, int jbeg = 0, int jend = -1);      // rows [jbeg, jend) (jend < 0: all)       // floor wall function; wrap_vp: also refresh vp's ghost row ny (bcpup)
int k_divergence_rhs(udc_handle *h, double rk3coef, bool pup);
int k_poisson_solve(udc_handle *h);
int k_project(udc_handle *h);                       // tderive: up,vp,wp -= grad p ; pres0 += p
// the open lid (BCtopm = 3): bcpup's, tderive's and tstep_integrate's row w(ke+1) (src/modboundary.f90:1234-1243, src/modpois.f90:1058-1069,
// src/modtstep.f90:270-286); pup: the tendency arrays hold the predicted velocity; wrap: periodic ghost rows written here
int k_lid_bcpup(udc_handle *h, double rk3coef, bool pup, bool ptotal = false);
bool k_lid_masked(const udc_handle *h);
int k_lid_tderive(udc_handle *h);
int k_lid_integrate(udc_handle *h, int rk3step, double dt, bool pup, bool zero, bool wrap);
int k_integrate(udc_handle *h, int rk3step, double dt);
int k_project_integrate(udc_handle *h, int rk3step, double dt, bool zero_tend, bool pup, bool ghosts,
                        bool write_um = true, bool out_to_um = false, int rows = 0, int r0 = 0, int r1 = 0, bool ptotal = false);   // fused tderive + tstep_integrate
// rows (k_closure_lds, k_project_integrate): 0 all tile rows of the slab, 1 only the tile rows next to the neighbouring ranks,
// 2 only the rows in between; k_project_integrate also 3 / 4: the tile rows [r0, r1) (3: profiled with the edge launch)
// dirs: HALO_TO_PREV (this slab's first rows -> the previous rank's upper ghost rows; the upper ghost rows here arrive from the next
// rank), HALO_TO_NEXT (the last rows -> the next rank's lower ghost rows), or both
enum { HALO_TO_PREV = 1, HALO_TO_NEXT = 2, HALO_BOTH = 3 };
int k_halo_y(udc_handle *h, const int *fields, int nf, int width, int dirs = HALO_BOTH);
// the same exchange beside the compute stream: _begin queues pack, exchange and unpack on the communication stream behind what
// the compute stream holds so far; _join makes the compute stream wait for it.  `ptrs` (optional): the arrays, where the caller
// knows better than h->fields (pointer rotation in flight)
int k_halo_y_begin(udc_handle *h, const int *fields, int nf, int width, double *const *ptrs = nullptr, int dirs = HALO_BOTH, int klev0 = -1, int nlev = 0);
int k_halo_y_join(udc_handle *h);
bool halo_overlap(const udc_handle *h, int tile_rows_y);      // y-slabs with enough tile rows for an edge / interior split
int k_top_bottom(udc_handle *h);
int k_top_rows_after_closure(udc_handle *h);
int k_scalar_top_flux(udc_handle *h);              // fluxtop with a non-zero flux: re-imposed after closure (reassure_fluxtop_boundary)
int k_level_source(udc_handle *h, int slot, const double *src);
int k_buoyancy(udc_handle *h);
int k_scalsource(udc_handle *h);
int k_chem(udc_handle *h, double dt);
int k_shifted_pbcs(udc_handle *h, bool wrap_vp);
int k_thermodynamics(udc_handle *h);
int k_slab_average(udc_handle *h, int field, double *avg_host, int n);
int k_slab_averages(udc_handle *h, const int *fields, int nf, double *avg_host, int n);
int k_level_forcings(udc_handle *h, int when, bool wrap_vp);
int k_tke_closure(udc_handle *h);                  // closure, loneeqn branch
int k_tke_sources(udc_handle *h);                  // sources: e12p += shear + buoyancy + dissipation
int k_calthv_capture(udc_handle *h);
int k_vreman_buoycorr(udc_handle *h);            // ekm *= sqrt(1 - min(max(Rig,0),Rigc)/Rigc), then ekh and the molecular parts
int k_ibm_wallfun(udc_handle *h);                // diffu/v/w/c_corr at the fluid-boundary points
int k_ibm_norm(udc_handle *h);
int ibm_grid_of_field(int field);                // 0 u, 1 v, 2 w, 3 c: the mask a field's slab sums use (src/modthermodynamics.f90:271-301)
// S[q n + k] -= sum of fields[q] over the solid points of device level k (q < nf, k < n); no-op without IBM
int k_ibm_levelsum_correct(udc_handle *h, const int *fields, int nf, int n, double *S, int k0 = 0);      // (levels k0 .. k0 + n - 1)
// S[0] -= sum a w(k), S[1] -= sum b w(k) over the solid points of `grid` (b may be null)
int k_ibm_flowsum_correct(udc_handle *h, int grid, const double *a, const double *b, const double *wlev, double *S, int only_i = -1);                   // solid: velocities zeroed, scalars to the mean of their fluid neighbours
void ibm_destroy(udc_handle *h);
void stats_destroy(udc_handle *h);
int k_ibm_wallfunmom(udc_handle *h);
int k_ibm_facet_begin(udc_handle *h);          // lwritefac: around the wall functions of an ibmwallfun that samples the facets
int k_ibm_facet_end(udc_handle *h);
int k_ibm_wallfunheat(udc_handle *h);
void ibm_wf_destroy(udc_handle *h);
int udc_flush_pending(udc_handle *h);
int k_tke_floor(udc_handle *h);                    // e120(kb-1) = e120(kb), e12m likewise (`bottom`)                     // wp += grav (thv0h - thvh)/thvh, src/modforces.f90:73-84   // cp(i,j,k) += src(k)
int k_maxima(udc_handle *h, double dt, double *cour, double *diffn, bool checksim = false);
int k_divergence_check(udc_handle *h, double *divmax, double *divtot);
int k_checksim_begin(udc_handle *h, double dtmn);
int k_checksim_end(udc_handle *h, double out[4]);
// udc_xopen.hip: inflow / outflow in x
int k_xo_ek_ghosts(udc_handle *h);                                  // closurebc's ekm(ib-1) = ekm(ib), ekm(ie+1) = ekm(ie)
int k_xo_bcpup(udc_handle *h, double rk3coef, bool pup, bool ptotal = false);            // bcpup's BCxm_profile branch
int k_xo_thl_top(udc_handle *h, bool fix);         // BCxT = 2: reassure_fluxtop_boundary's zero-flux row above the inlet's first column
int k_xo_after_integrate(udc_handle *h, int rk3step, bool boundary_follows = false);               // v, w at ie+1 back from the outlet's planes (vm = v0 at stage 3)
int k_xo_halos(udc_handle *h);                                      // xT_periodic / xq_periodic where those stay periodic beside the open flow
int k_xo_boundary(udc_handle *h, int merged_stage3 = -1);                                   // xmi_profile, xmo_convective (+ bcp's pres0 columns)
int k_xo_poisson(udc_handle *h);                                    // the solve on the mirrored row
int xo_init(udc_handle *h, const double *uprof, const double *vprof);
void xo_destroy(udc_handle *h);
int xo_capture_east(udc_handle *h, int field, const double *host, const int lb[3], const int ub[3]);
int pois_init(udc_handle *h);
int pois_slab_init(udc_handle *h);
int k_poisson_solve_slab(udc_handle *h);
bool fft_fused_possible(const udc_handle *h);
int fft_fused_init(udc_handle *h);
int fft_nat_init(udc_handle *h);
int fft_nat_forward(udc_handle *h);
int fft_x_fwd_pack(udc_handle *h, int k0, int nzc, double *send, int g0 = 0, int g1 = 0, hipStream_t st = nullptr);      // row groups [g0, g1); g1 <= 0: all
int fft_x_row_groups(const udc_handle *h);
int fft_x_bwd_unpack(udc_handle *h, int k0, int nzc, const double *recv, bool ghost = false);
int fft_y_fwd_unpack(udc_handle *h, int k0, int nzc, const double *recv);
int fft_y_bwd_pack(udc_handle *h, int k0, int nzc, double *send, bool ghost = false);
// udc_comm.hip
int comm_neighbours(udc_handle *h, const double *to_prev, const double *to_next, double *from_prev,
                    double *from_next, size_t count, hipStream_t st = nullptr, int dirs = 3);      // st: the stream it runs on (default h->stream); dirs: HALO_*
int comm_alltoall(udc_handle *h, const double *send, double *recv, size_t block, hipStream_t st);
int comm_allreduce(udc_handle *h, double *buf, int n, int op);
void comm_destroy(udc_handle *h);
void pois_destroy(udc_handle *h);
