// Pressure correction: divergence RHS (fillps), rocFFT (x,y) + batched Thomas solve in z
// (solmpj), projection (tderive), RK3 update (tstep_integrate), plus the two reductions
// (adaptive-dt maxima, chkdiv).
//
// Spectral layout.  The reference keeps real "half-complex" lines and transforms x then y
// with 8 pencil transposes (src/modpois.f90:459-702).  Here the x and y transforms are one
// batched rocFFT 2-D real-to-complex transform per k-plane, read straight from the padded
// p field (strides), producing complex modes (kx = 0..nx/2, ky = 0..ny-1); every mode is an
// independent real tridiagonal system in z, so Re and Im are solved with the same
// coefficients.  Mode (kx,ky) has eigenvalue xrt(kx)+yrt(min(ky,ny-ky)), the same values the
// reference assigns to the half-complex slots (:100-107,124-131); the two 1/sqrt(n) scalings
// per direction (:490,534,623,677) collapse into one factor 1/(nx*ny).
#include "udc_internal.h"
#include <atomic>
#include <mutex>
#include <cmath>
#include <cstdlib>

namespace {

__device__ __forceinline__ int wrapm(int i, int nx) { return i == 0 ? nx - 1 : i - 1; }
__device__ __forceinline__ int wrapp(int i, int nx) { return i == nx - 1 ? 0 : i + 1; }

inline dim3 cell_grid(const Geo &g, dim3 b) {
  (void)b;
  return dim3((unsigned)tile_grid(g).tiles * (unsigned)g.nz, 1, 1);
}

// fillps + bcpup, src/modpois.f90:939-973, src/modboundary.f90:1227-1255,1309-1315:
// p = d(pup)/dx + d(pvp)/dy + d(pwp)/dz with pup = up + um/rk3coef (not materialised),
// pwp(kb) = pwp(ke+1) = 0, x cyclic by index wrap, y cyclic through the vp/vm ghost row.
template <bool PUP>
__global__ __launch_bounds__(256) void div_rhs_kernel(Geo g, TileGrid tg, Metrics m, double r,
    const double *__restrict__ up, const double *__restrict__ vp, const double *__restrict__ wp,
    const double *__restrict__ um, const double *__restrict__ vm, const double *__restrict__ wm,
    double *__restrict__ p, int lid, double *__restrict__ p2, int sy2, long sz2) {
  int i, j, k;
  const bool inside_ = tile_decode(g, tg, i, j, k);
  if (!inside_) return;
  const long r0 = g.idx(0, j, k);
  const long c = r0 + i, xp = r0 + wrapp(i, g.nx);
  double pu_c, pu_p, pv_c, pv_p, pw_c, pw_p;
  const bool closed_top = k == g.nz - 1 && !lid;      // (open lid, BCtopm = 3: lid_bcpup_kernel has filled the plane ke+1 of wp)
  if (PUP) {   // up,vp,wp already hold pup,pvp,pwp (momentum sweep in PUP mode)
    pu_c = up[c]; pu_p = up[xp];
    pv_c = vp[c]; pv_p = vp[c + g.sy];
    pw_c = wp[c];
    pw_p = closed_top ? 0. : wp[c + g.sz];
  } else {
    pu_c = up[c] + um[c] * r; pu_p = up[xp] + um[xp] * r;
    pv_c = vp[c] + vm[c] * r; pv_p = vp[c + g.sy] + vm[c + g.sy] * r;
    pw_c = (k == 0) ? 0. : wp[c] + wm[c] * r;
    pw_p = closed_top ? 0. : wp[c + g.sz] + wm[c + g.sz] * r;
  }
  const double v = (pu_p - pu_c) * m.dxi + (pv_p - pv_c) * m.dyi + (pw_p - pw_c) * m.dzfi[k + 1];
  if (p2) {
    // open x boundaries (udc_xopen.hip): the right-hand side goes straight to the solver's doubled row -- the interior columns and their
    // mirror image; what the ghost columns' cells would hold is nobody's
    const int n = g.nx - 2 * g.xg, ii = i - g.xg;
    if (ii < 0 || ii >= n) return;
    const long b = (long)sy2 * (j + HY) + sz2 * (long)(k + HZ);
    p2[b + ii] = v;
    p2[b + 2 * n - 1 - ii] = v;
    return;
  }
  p[c] = v;
}

// ---- Thomas table: the pivots z(m,k) = 1/(b_k + e_m - a_k d_{k-1}) of solmpj (src/modpois.f90:1120-1139)
// do not depend on the RHS; d(m,k) = c_k z(m,k) is rebuilt from it with the reference's own multiply.
// Two layouts: [level][mode] for the streaming kernel (a wave reads 64 consecutive modes of one level), and blocks of
// ZB modes [mode / ZB][level][mode % ZB] for the LDS kernel, whose workgroup owns ZB modes and walks the levels: its
// chunk of the table is then one contiguous run instead of 64-byte pieces of 128-byte lines shared with the neighbour.
constexpr int ZB = 8;
__host__ __device__ __forceinline__ size_t ztab_index(bool blocked, int nmodes, int nz, int lev, int mo) {
  return blocked ? ((size_t)(mo / ZB) * (size_t)(nz - 1) + lev) * ZB + (mo % ZB) : (size_t)lev * nmodes + mo;
}
__global__ void thomas_table_kernel(int nmodes, int nz, const double *__restrict__ ev,
                                    const double *__restrict__ tri, double btopD, double *__restrict__ ztab, int blocked) {
  const int mo = blockIdx.x * blockDim.x + threadIdx.x;
  if (mo >= nmodes) return;
  const double *a = tri, *b = tri + (nz + 2), *c = tri + 2 * (nz + 2);
  const double e = ev[mo];
  double z = 1. / (b[1] + e);
  double d = c[1] * z;
  // (round 5: the blocked table used to be followed by a second one holding -(c_k z_k), the back substitution's coefficient -- a
  // sixth of the solve's traffic; the register kernel now forms it from z and the level's c, the same multiply)
  if (nz >= 2) ztab[ztab_index(blocked, nmodes, nz, 0, mo)] = z;
  for (int k = 2; k <= nz - 1; ++k) {
    const double bbk = b[k] + e;
    z = 1. / (bbk - a[k] * d);
    d = c[k] * z;
    ztab[ztab_index(blocked, nmodes, nz, k - 1, mo)] = z;
  }
  (void)btopD;
}

// solmpj, src/modpois.f90:1107-1166, one thread per (kx,ky) mode, complex data.
// scale = 1/(nx*ny) carries the reference's four 1/sqrt(n) factors.
// The recurrence is sequential in k but its loads are not: each thread fetches TU levels ahead
// (independent 16-B loads in flight) before running the dependent arithmetic on them, which is what
// keeps HBM busy with only nmodes/64 waves on the chip.
#ifndef THOMAS_TU
#define THOMAS_TU 8
#endif
constexpr int TU = THOMAS_TU;
__global__ __launch_bounds__(64) void thomas_kernel(int nmodes, int nz, double scale,
    const double *__restrict__ ev, const double *__restrict__ tri, double btopD,
    const double *__restrict__ ztab, double2 *__restrict__ x) {
  const int mo = blockIdx.x * blockDim.x + threadIdx.x;
  if (mo >= nmodes) return;
  const double *a = tri, *b = tri + (nz + 2), *c = tri + 2 * (nz + 2);
  const double e = ev[mo];
  const size_t st = (size_t)nmodes;
  double2 *xm = x + mo;                 // xm[(k-1)*st] = level k
  double z = 1. / (b[1] + e);
  double d = c[1] * z;
  double2 xp = xm[0];
  xp.x = (xp.x * scale) * z; xp.y = (xp.y * scale) * z;
  xm[0] = xp;
  // forward elimination, levels 2 .. nz-1
  for (int k0 = 2; k0 <= nz - 1; k0 += TU) {
    double2 buf[TU];
#pragma unroll
    for (int u = 0; u < TU; ++u)
      if (k0 + u <= nz - 1) buf[u] = xm[(size_t)(k0 + u - 1) * st];
#pragma unroll
    for (int u = 0; u < TU; ++u) {
      const int k = k0 + u;
      if (k <= nz - 1) {
        const double bbk = b[k] + e;
        z = 1. / (bbk - a[k] * d);
        d = c[k] * z;
        double2 xc = buf[u];
        const double gz = -(a[k] * z);
        xc.x = __builtin_fma(gz, xp.x, (xc.x * scale) * z);
        xc.y = __builtin_fma(gz, xp.y, (xc.y * scale) * z);
        buf[u] = xc;
        xp = xc;
      }
    }
#pragma unroll
    for (int u = 0; u < TU; ++u)
      if (k0 + u <= nz - 1) xm[(size_t)(k0 + u - 1) * st] = buf[u];
  }
  {
    // the singular (0,0) mode gets a Dirichlet condition across the top cell (:209-220)
    const double bbk = (e == 0.) ? btopD : b[nz] + e;
    const double ak = a[nz];
    z = bbk - ak * d;
    double2 xc = xm[(size_t)(nz - 1) * st];
    xc.x = (xc.x * scale - ak * xp.x) / z;
    xc.y = (xc.y * scale - ak * xp.y) / z;
    xm[(size_t)(nz - 1) * st] = xc;
    xp = xc;
  }
  // back substitution, levels nz-1 .. 1
  const double *dm = ztab + mo;
  for (int k0 = nz - 1; k0 >= 1; k0 -= TU) {
    double2 buf[TU];
    double dk[TU];
#pragma unroll
    for (int u = 0; u < TU; ++u)
      if (k0 - u >= 1) { buf[u] = xm[(size_t)(k0 - u - 1) * st]; dk[u] = dm[(size_t)(k0 - u - 1) * st]; }
#pragma unroll
    for (int u = 0; u < TU; ++u)
      if (k0 - u >= 1) {
        double2 xc = buf[u];
        const double dd = -(c[k0 - u] * dk[u]);
        xc.x = __builtin_fma(dd, xp.x, xc.x);
        xc.y = __builtin_fma(dd, xp.y, xc.y);
        buf[u] = xc;
        xp = xc;
      }
#pragma unroll
    for (int u = 0; u < TU; ++u)
      if (k0 - u >= 1) xm[(size_t)(k0 - u - 1) * st] = buf[u];
  }
}

// Register-resident segments (round 4).  Rounds 2-3 kept a workgroup's columns in LDS and let one wave walk them (plain and
// wave-specialised variants, 64 dependent chains per CU): half of the roofline whatever fed them (profiles/r04/thomas_variants_ab.txt;
// removed).  Here nothing is resident anywhere but in registers: thread (mode mm of the workgroup's ZB modes, segment s) owns SL consecutive levels of
// one complex column -- SL independent 16-B loads of x and SL pivots, all issued before the first use (the coefficients -(a z) and
// -(c z) are formed from the pivot and the level's a, c, which sit in LDS: round 5, the table of -(c z) was a sixth of the traffic)
// -- and the two sweeps, first-order linear recurrences  v_l = g_l v_prev + t_l,  are solved by partition:
//   1. zero-inflow recurrence over the own segment -> summary (P = prod g, y) -> LDS, one barrier;
//   2. the summaries of the segments before (forward) / above (back) the own one, composed: v_in (thomas_scan_in);
//   3. the own segment again, from v_in, in the reference's order (src/modpois.f90:1120-1166) -- so only the inflow
//      value carries the partition's rounding;
//   the top level (Dirichlet row of the singular mode, :209-220) is closed by the thread that owns it between the sweeps.
// A wave is 8 modes x 8 segments: every load instruction fetches eight full 128-B lines of x (64-B runs of the blocked
// table).  Three barriers per workgroup, no LDS traffic but a, c and the summaries; x and the table cross the bus exactly once.
// NP = 2 (one GPU, spectral layout spec[k][ky][kx]): the eigenvalue of mode (kx, ky) is xrt(kx) + yrt(ky) with yrt(ky) = yrt(ny - ky)
// bit for bit (pois_init), so rows ky and ny - ky have the same matrix: the workgroup takes the same eight kx of both rows, and the
// pivot table -- a fifth of the solve's traffic -- is read for one of them only (40 -> 36 B per complex mode and level); rows 0
// and ny / 2 are their own mirror images and run with the second system switched off.
// MIR (NP = 2, the slab ranks' layout specB[k][kx_l][y], round 5): the mirror image of the run y = 8 b .. 8 b + 7 of a line is the
// run ny - 8 b - 7 .. ny - 8 b -- contiguous, lane-reversed and one element off the 128-B alignment.  A workgroup takes block b of
// line kx_l as system 0 and that run as system 1 (lane mm <-> y' = ny - y), so that here too the tables are read for one row of each
// pair; blocks b = 0 .. ny/16 - 1 cover y < ny/2 and their images, lane 0 of block 0 (y = 0) has no partner, and the modes y = ny/2
// of eight lines share one more workgroup (one system, every lane on its own line and table block).  The straddled 128-B line of a
// mirrored run is shared with the neighbouring block of the same line: the workgroups are dealt to the XCDs in contiguous runs (as
// xcd_tile does; the grid is padded to a multiple of eight for that), so that it is fetched once per L2.
// Step 2 of the partition: the value that enters a thread's segment, from the summaries (P, y) of the segments before it (FWD) or
// above it (back substitution) -- each summary the affine map v -> P v + y, applied in the order of the sweep to v = 0.
// Round 5, second session: every thread used to chain all of them out of LDS itself (seg iterations of two LDS reads and two
// dependent FMAs: at nz = 512 up to 63, 32 on average -- 2.6 us of a workgroup's 10 during which nothing was in flight, and with
// 512-thread workgroups only one fits a CU).  Now: an exclusive scan of the affine maps over the eight segments a wave holds (lanes
// 8 apart; Kogge-Stone, three rounds of lane shuffles, compositions f o f_before = (P P', P y' + y)), the composition of a whole wave
// through LDS (one barrier, as before), at most NW - 1 chained wave summaries, one FMA for the own prefix.  The association differs
// from the chain's: the inflow value carries the partition's rounding, as it did.
template <int NP, int NW, bool FWD>
__device__ __forceinline__ void thomas_scan_in(double P, const double2 (&y)[NP], double2 (&v)[NP], double (*sP)[ZB], double2 (*sY)[NW][ZB], int tid) {
  const int lane = tid & 63, w = tid >> 6, mm = tid & (ZB - 1), sl = lane >> 3;
  auto from = [&](double a, int d) { return FWD ? __shfl_up(a, d, 64) : __shfl_down(a, d, 64); };
  const int pos = FWD ? sl : 7 - sl;       // segments of this wave that come before this one in the sweep
  // the map of the segment just before (exclusive scan: the first of the wave starts from the identity)
  double Pe = from(P, 8);
  double2 ye[NP];
#pragma unroll
  for (int s = 0; s < NP; ++s) { ye[s].x = from(y[s].x, 8); ye[s].y = from(y[s].y, 8); }
  if (pos == 0) {
    Pe = 1.;
#pragma unroll
    for (int s = 0; s < NP; ++s) ye[s] = make_double2(0., 0.);
  }
#pragma unroll
  for (int d = 1; d < 8; d *= 2) {
    const double Pp = from(Pe, 8 * d);
    double2 yp[NP];
#pragma unroll
    for (int s = 0; s < NP; ++s) { yp[s].x = from(ye[s].x, 8 * d); yp[s].y = from(ye[s].y, 8 * d); }
    if (pos >= d) {
#pragma unroll
      for (int s = 0; s < NP; ++s) { ye[s].x = __builtin_fma(Pe, yp[s].x, ye[s].x); ye[s].y = __builtin_fma(Pe, yp[s].y, ye[s].y); }
      Pe *= Pp;
    }
  }
  // the wave as a whole: the last segment of the sweep composes its own map with its prefix
  if (pos == 7) {
    sP[w][mm] = P * Pe;
#pragma unroll
    for (int s = 0; s < NP; ++s) sY[s][w][mm] = make_double2(__builtin_fma(P, ye[s].x, y[s].x), __builtin_fma(P, ye[s].y, y[s].y));
  }
  __syncthreads();
#pragma unroll
  for (int s = 0; s < NP; ++s) v[s] = make_double2(0., 0.);
  if (NW > 1) {
    if (FWD) {
      for (int q = 0; q < w; ++q) {
        const double Pq = sP[q][mm];
#pragma unroll
        for (int s = 0; s < NP; ++s) {
          const double2 yq = sY[s][q][mm];
          v[s].x = __builtin_fma(Pq, v[s].x, yq.x); v[s].y = __builtin_fma(Pq, v[s].y, yq.y);
        }
      }
    } else {
      for (int q = NW - 1; q > w; --q) {
        const double Pq = sP[q][mm];
#pragma unroll
        for (int s = 0; s < NP; ++s) {
          const double2 yq = sY[s][q][mm];
          v[s].x = __builtin_fma(Pq, v[s].x, yq.x); v[s].y = __builtin_fma(Pq, v[s].y, yq.y);
        }
      }
    }
  }
#pragma unroll
  for (int s = 0; s < NP; ++s) { v[s].x = __builtin_fma(Pe, v[s].x, ye[s].x); v[s].y = __builtin_fma(Pe, v[s].y, ye[s].y); }
}

template <int SL, int NT, int W, int NP, bool MIR>
__global__ __launch_bounds__(NT, W) void thomas_reg_kernel(int nmodes, int nz, double scale,
    const double *__restrict__ ev, const double *__restrict__ tri, double btopD,
    const double *__restrict__ ztab, double2 *__restrict__ x, int nkb, int ny) {
  static_assert(!MIR || NP == 2, "the mirrored layout is a paired solve");
  constexpr int M = ZB;                    // modes per workgroup and system = lanes per segment
  constexpr int NW = NT / 64;              // waves: eight segments of the eight modes each
  __shared__ double sP[2][NW][M];          // summaries of whole waves only (see scan_in below)
  __shared__ double2 sY[2][NP][NW][M];
  constexpr int NLEV = SL * (NT / M);      // levels the workgroup's segments cover (>= nz)
  __shared__ double sA[NLEV + 1], sC[NLEV + 1];      // a_k, c_k of the levels (entry k = reference level k, 1 .. nz)
  const int tid = threadIdx.x, mm = tid & (M - 1), seg = tid >> 3;
  // block of eight modes of system 0 (whose tables are read) and of system 1
  int blk0 = blockIdx.x, blk1 = blockIdx.x;
  bool on1 = false;                        // this lane solves a second system
  int mo = 0;                              // this lane's mode of system 0
  bool mok = true;
  int tlane = mm;                          // lane's entry within its table block, and how many table blocks it lies beyond blk0
  long tskip = 0, xskip = 0;               // (MIR, the workgroups of the modes y = ny/2: doubles / elements from blk0's)
  long base1 = 0;                          // MIR: first element (level 0) of system 1's run
  int lane1 = mm;                          // lane's element within system 1's run
  if (NP == 2 && !MIR) {
    const int kyh = blockIdx.x / nkb, kxb = blockIdx.x - kyh * nkb;
    on1 = kyh != 0 && 2 * kyh != ny;
    blk1 = on1 ? (ny - kyh) * nkb + kxb : blk0;
  }
  if (MIR) {
    unsigned v = blockIdx.x;
    v = (v & 7u) * (gridDim.x >> 3) + (v >> 3);                                         // (gridDim.x is a multiple of eight)
    const int lines = nmodes / ny;
    const unsigned npair = (unsigned)lines * (unsigned)nkb;                             // nkb = ny/16 paired blocks per line
    if (v >= npair + (unsigned)((lines + 7) >> 3)) return;                              // padding
    if (v < npair) {
      const int kxl = (int)(v / (unsigned)nkb), b = (int)(v - (unsigned)kxl * nkb);
      const int y = 8 * b + mm;
      blk0 = kxl * (ny >> 3) + b;
      mo = blk0 * M + mm;
      xskip = mm;
      on1 = y >= 1;
      base1 = (long)kxl * ny + (ny - 8 * b - 7);
      lane1 = 7 - (y >= 1 ? mm : 1);       // (y = 0: its neighbour's address, loaded and never stored)
    } else {
      const int l0_ = 8 * (int)(v - npair), kxl = min(l0_ + mm, lines - 1);
      mok = l0_ + mm < lines;
      blk0 = (int)(((long)l0_ * ny + (ny >> 1)) >> 3);
      mo = kxl * ny + (ny >> 1);
      tlane = 0;
      tskip = (long)(kxl - l0_) * (ny >> 3) * (long)(nz - 1) * M;
      xskip = (long)(kxl - l0_) * ny;
    }
  } else {
    mo = blk0 * M + mm;
    mok = mo < nmodes;
  }
  const int moc = mok ? mo : (MIR ? mo : nmodes - 1);
  const double *a = tri, *b = tri + (nz + 2), *c = tri + 2 * (nz + 2);
  const size_t st = (size_t)nmodes;
  const int l0 = seg * SL;
  // addresses: buffer descriptors over the whole of x and of the pivot table (kernel arguments: wave-uniform, so the compiler keeps them
  // in scalar registers), one 32-bit byte offset per lane and stream, the level offset j in the instruction's scalar offset (x) or its
  // immediate (table): the loads in flight cost no address registers.  (Round 5; the pointer forms cost a 64-bit pair per load in
  // flight.  The launcher sends arrays of 4 GiB and more to the streaming kernel.)
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(x, 0, (int)(unsigned)((size_t)nmodes * (size_t)nz * sizeof(double2)), 0x00020000);
  const __amdgpu_buffer_rsrc_t rz = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(ztab), 0,
      (int)(unsigned)((size_t)((nmodes + ZB - 1) / ZB) * ZB * (size_t)(nz > 1 ? nz - 1 : 1) * sizeof(double)), 0x00020000);
  const size_t zrun = (size_t)blk0 * (size_t)(nz - 1) * M;      // [lev][M] run of this workgroup in the table
  const unsigned zoff = MIR ? (unsigned)((zrun + (size_t)l0 * M + (size_t)tlane + (size_t)tskip) * sizeof(double))
                            : (unsigned)((zrun + (size_t)l0 * M + mm) * sizeof(double));
  unsigned xo_[NP];                        // lane offsets of level l0 in the two systems
  xo_[0] = MIR ? (unsigned)(((size_t)blk0 * M + (size_t)l0 * st + (size_t)xskip) * sizeof(double2))
               : (unsigned)(((size_t)l0 * st + (size_t)moc) * sizeof(double2));
  if (NP == 2) xo_[NP - 1] = MIR ? (unsigned)(((size_t)base1 + (size_t)l0 * st + (size_t)lane1) * sizeof(double2))
                                 : (unsigned)(((size_t)blk1 * M + (size_t)l0 * st + (size_t)(moc - blk0 * M)) * sizeof(double2));
  const unsigned xlev = (unsigned)(st * sizeof(double2));      // one level up in x
  auto ldx = [&](int s, int j) {
    const auto r = __builtin_amdgcn_raw_buffer_load_b128(rx, (int)xo_[s], (int)(j * xlev), 0);
    union { decltype(__builtin_amdgcn_raw_buffer_load_b128(rx, 0, 0, 0)) v; double2 d; } u;
    u.v = r;
    return u.d;
  };
  auto ldz = [&](int j) {
    const auto r = __builtin_amdgcn_raw_buffer_load_b64(rz, (int)(zoff + (unsigned)(j * M * sizeof(double))), 0, 0);
    union { decltype(__builtin_amdgcn_raw_buffer_load_b64(rz, 0, 0, 0)) v; double d; } u;
    u.v = r;
    return u.d;
  };
  double2 t[NP][SL];
  double g[SL], cz[SL];      // cz: the pivots z until the back substitution turns them into -(c z)
  double2 xt[NP];                          // (x s) of the top level, kept by its owner
  // a wave whose eight segments all lie below the top level loads without a test (every load of the kernel is then issued
  // before the first wait); the wave(s) around the top level test each level
  const bool full = __all(l0 + SL <= nz - 1) != 0;
  if (full) {
#pragma unroll
    for (int j = 0; j < SL; ++j) {
#pragma unroll
      for (int s = 0; s < NP; ++s) t[s][j] = ldx(s, j);
      cz[j] = ldz(j);
    }
  } else {
#pragma unroll
    for (int j = 0; j < SL; ++j) {
      const int lev = l0 + j;
#pragma unroll
      for (int s = 0; s < NP; ++s) t[s][j] = lev < nz ? ldx(s, j) : make_double2(0., 0.);
      cz[j] = lev < nz - 1 ? ldz(j) : 1.;       // levels 1 .. nz-1 of solmpj take part in the recurrences
    }
  }
  // the levels' a and c go through LDS (8 KB at nz = 512) instead of sitting in 32 registers per thread while x and the pivots
  // travel; their loads (cache hits) behind the big ones, so that nothing waits before those are issued
  // (entries from the top level on: a = -1, c = 0, which with z = 1 there make the coefficients g = 1 and -(c z) = -0 of the levels
  // that only pass the carry through -- no test per level in the sweeps)
#pragma unroll
  for (int it = 0; it < NLEV / NT; ++it) {      // (NLEV = (SL / 8) NT: entries 1 .. NLEV, no test)
    const int k = tid + 1 + it * NT, li_ = min(k, nz);
    const double av = a[li_], cv = c[li_];
    sA[k] = k < nz ? av : -1.; sC[k] = k < nz ? cv : 0.;
  }
  __syncthreads();
  const double *zt = reinterpret_cast<const double *>(reinterpret_cast<const char *>(ztab) + zoff) - (size_t)l0 * M;      // this lane's column of the pivot table
  const bool own_top = l0 <= nz - 1 && nz - 1 < l0 + SL;
  double zl = 0., etop = 0.;
  if (own_top) { zl = zt[(size_t)(nz - 2) * M]; etop = ev[moc]; }
  // forward: t = (x s) z, g = -(a z); levels >= nz-1 pass the carry through (g = 1, t = 0)
  double P = 1.;
  double2 y[NP];
#pragma unroll
  for (int s = 0; s < NP; ++s) { y[s] = make_double2(0., 0.); xt[s] = make_double2(0., 0.); }
#pragma unroll
  for (int j = 0; j < SL; ++j) {
    const int lev = l0 + j;
    const double zz = cz[j];
    const bool rec = lev < nz - 1;
    g[j] = -(sA[lev + 1] * zz);
#pragma unroll
    for (int s = 0; s < NP; ++s) {
      double2 v = t[s][j];
      v.x = (v.x * scale) * zz; v.y = (v.y * scale) * zz;
      if (lev == nz - 1) xt[s] = v;
      t[s][j] = rec ? v : make_double2(0., 0.);
      y[s].x = __builtin_fma(g[j], y[s].x, t[s][j].x); y[s].y = __builtin_fma(g[j], y[s].y, t[s][j].y);
    }
    P *= g[j];
  }
  double2 v[NP];
  thomas_scan_in<NP, NW, true>(P, y, v, sP[0], sY[0], tid);
#pragma unroll
  for (int j = 0; j < SL; ++j) {
#pragma unroll
    for (int s = 0; s < NP; ++s) {
      v[s].x = __builtin_fma(g[j], v[s].x, t[s][j].x); v[s].y = __builtin_fma(g[j], v[s].y, t[s][j].y);
      t[s][j] = v[s];
    }
  }
  if (own_top) {
    // v = x'_{nz-1} (the levels from nz-1 on passed it through); the singular (0,0) mode gets a Dirichlet condition
    // across the top cell (:209-220)
    const double bbk = (etop == 0.) ? btopD : b[nz] + etop;
    const double ak = a[nz];
    const double d = c[nz - 1] * zl;
    const double z = bbk - ak * d;
#pragma unroll
    for (int s = 0; s < NP; ++s) {
      const double2 xc = make_double2((xt[s].x - ak * v[s].x) / z, (xt[s].y - ak * v[s].y) / z);
#pragma unroll
      for (int j = 0; j < SL; ++j) if (l0 + j == nz - 1) t[s][j] = xc;
    }
  }
#pragma unroll
  for (int j = 0; j < SL; ++j)
    if (l0 + j > nz - 1) {
#pragma unroll
      for (int s = 0; s < NP; ++s) t[s][j] = make_double2(0., 0.);
    }
  // back substitution  x_l = x'_l - (c z)_l x_{l+1}  downwards; the top level has coefficient 0 and x' = its solution
  P = 1.;
#pragma unroll
  for (int s = 0; s < NP; ++s) y[s] = make_double2(0., 0.);
#pragma unroll
  for (int j = SL - 1; j >= 0; --j) {
    cz[j] = -(sC[l0 + j + 1] * cz[j]);
#pragma unroll
    for (int s = 0; s < NP; ++s) { y[s].x = __builtin_fma(cz[j], y[s].x, t[s][j].x); y[s].y = __builtin_fma(cz[j], y[s].y, t[s][j].y); }
    P *= cz[j];
  }
  thomas_scan_in<NP, NW, false>(P, y, v, sP[1], sY[1], tid);      // (segments above the top level are (P, y) = (0, 0): they start the chain at 0)
#pragma unroll
  for (int j = SL - 1; j >= 0; --j) {
#pragma unroll
    for (int s = 0; s < NP; ++s) {
      v[s].x = __builtin_fma(cz[j], v[s].x, t[s][j].x); v[s].y = __builtin_fma(cz[j], v[s].y, t[s][j].y);
      if (l0 + j < nz && mok && (s == 0 || on1)) {
        union { decltype(__builtin_amdgcn_raw_buffer_load_b128(rx, 0, 0, 0)) q; double2 d; } u;
        u.d = v[s];
        __builtin_amdgcn_raw_buffer_store_b128(u.q, rx, (int)xo_[s], (int)(j * xlev), 0);
      }
    }
  }
}

// Which solve.  The table's layout follows the kernel, so the choice is made once per table (udc_create reads UDC_THOMAS into the
// handle: 0 = the streaming kernel everywhere).  Default: register-resident segments for nz <= 1024 and arrays below 4 GiB (every
// deck there is), the streaming kernel otherwise.
// (32-bit byte offsets into x: the whole array, and eight levels from any element of it)
static bool thomas_reg_fits(long nmodes, int nz) { return nz >= 3 && nz <= 1024 && (size_t)nmodes * (size_t)(nz > 8 ? nz : 8) * 16 < ((size_t)1 << 32); }
static bool thomas_wants_lds(const udc_handle *h, long nmodes, int nz) {      // -> the blocked tables of the register kernel
  return h->sw.thomas != 0 && thomas_reg_fits(nmodes, nz);
}
static size_t ztab_doubles(long nmodes, int nz) { return (size_t)((nmodes + ZB - 1) / ZB) * ZB * (size_t)(nz > 1 ? nz - 1 : 1); }

template <int NT, int W, int NP, bool MIR, int SL = 8>
static void launch_thomas_reg(udc_handle *h, long nmodes, int nz, double scale, const double *ev, const double *ztab, double2 *x, int nkb, int ny) {
  const unsigned blocks = MIR ? (unsigned)(((nmodes / ny) * nkb + (nmodes / ny + 7) / 8 + 7) / 8 * 8) : (NP == 2 ? (unsigned)((ny / 2 + 1) * nkb) : (unsigned)((nmodes + ZB - 1) / ZB));
  hipLaunchKernelGGL((thomas_reg_kernel<SL, NT, (W > NT / 256 ? W : NT / 256), NP, MIR>), dim3(blocks), dim3(NT), 0, h->stream,
                     (int)nmodes, nz, scale, ev, h->tri, h->btopD, ztab, x, nkb, ny);
}

// nkb, pair_ny > 0: the caller's modes are spec[k][ky][kx] rows of nkb blocks of eight, ky = 0 .. pair_ny - 1 (one GPU): rows ky and
// ny - ky are solved together (UDC_THOMAS_PAIR=0: not).  mirror_ny > 0: the caller's modes are lines specB[k][kx_l][y] of mirror_ny
// elements (the slab ranks): y and ny - y of a line are solved together (thomas_reg_kernel, MIR).
static int launch_thomas(udc_handle *h, bool blocked, long nmodes, int nz, double scale, const double *ev, const double *ztab, double2 *x,
                         int nkb = 0, int pair_ny = 0, int mirror_ny = 0) {
  if (blocked) {
    // eight levels per thread; workgroups of 8 x ceil(nz / 8) threads.  Measured (profiles/r04/thomas_variants_ab.txt, thomas_pair_ab.txt):
    // 16 levels per thread spill, 4 leave too little in flight per thread; compiled for 3 waves per SIMD (2 with two systems per thread)
    const int nt = 8 * ((nz + 7) / 8);
    const bool pairs_fit = h->sw.thomas_pair && nt <= 512;      // (two systems per thread do not fit 1024-thread workgroups' 128 VGPRs)
    const bool pair = pair_ny > 0 && pairs_fit;
    // measured (profiles/r05/thomas_mirror_ab.txt, one rank's spectral slab of 1024 x 512 x 512 on eight ranks): lines of 512 -> 210 us
    // against 228 unpaired; lines of 64 (the forced-slab stand-in 1024 x 64 x 512) -> 251 against 206: the straddled lines of the
    // mirrored runs are fetched twice there (PMC) -- paired from lines of 256 on
    const bool mirror = mirror_ny >= h->sw.thomas_mirror_min && mirror_ny >= 16 && mirror_ny % 16 == 0 && nmodes % mirror_ny == 0 && pairs_fit;
#define UDC_TR(Wv, NPv, MIRv)                                                                                      \
    do {                                                                                                           \
      if (nt <= 64) launch_thomas_reg<64, Wv, NPv, MIRv>(h, nmodes, nz, scale, ev, ztab, x, nkb, ny_);             \
      else if (nt <= 128) launch_thomas_reg<128, Wv, NPv, MIRv>(h, nmodes, nz, scale, ev, ztab, x, nkb, ny_);      \
      else if (nt <= 256) launch_thomas_reg<256, Wv, NPv, MIRv>(h, nmodes, nz, scale, ev, ztab, x, nkb, ny_);      \
      else if (nt <= 512) launch_thomas_reg<512, Wv, NPv, MIRv>(h, nmodes, nz, scale, ev, ztab, x, nkb, ny_);      \
      else launch_thomas_reg<1024, Wv, NPv, MIRv>(h, nmodes, nz, scale, ev, ztab, x, nkb, ny_);                    \
    } while (0)
    int ny_ = pair_ny;
    if (mirror) { ny_ = mirror_ny; nkb = mirror_ny / 16; UDC_TR(2, 2, true); }
    else if (pair) UDC_TR(2, 2, false);
    else if (nz > tune::THOMAS_SL16_ABOVE) {
      // tall columns, one system per thread: sixteen levels per thread (t, z, g: 128 registers of 250) halve the workgroup -- at
      // nz = 512 two of 256 threads share a CU and one loads while the other solves (with eight levels the one 512-thread
      // workgroup a CU holds does the two in turn); above 512 levels no 1024-thread workgroup capped at 128 registers
      if (nz <= 512) launch_thomas_reg<256, 2, 1, false, 16>(h, nmodes, nz, scale, ev, ztab, x, nkb, ny_);
      else launch_thomas_reg<512, 2, 1, false, 16>(h, nmodes, nz, scale, ev, ztab, x, nkb, ny_);
    }
    else UDC_TR(3, 1, false);
#undef UDC_TR
    return 0;
  }
  hipLaunchKernelGGL(thomas_kernel, dim3((unsigned)((nmodes + 63) / 64)), dim3(64), 0, h->stream, (int)nmodes, nz, scale,
                     ev, h->tri, h->btopD, ztab, x);
  return 0;
}

// compact (nx,ny,nz) <-> padded field interior (only used when rocFFT rejects the padded layout)
template <bool TO_FIELD>
__global__ __launch_bounds__(256) void real_copy_kernel(Geo g, TileGrid tg, double *__restrict__ field, double *__restrict__ buf) {
  int i, j, k;
  const bool inside_ = tile_decode(g, tg, i, j, k);
  if (!inside_) return;
  const long c = g.idx(i, j, k);
  const long q = (long)i + (long)g.nx * (j + (long)g.ny * k);
  if (TO_FIELD) field[c] = buf[q]; else buf[q] = field[c];
}

// ======================================================================= y-slab Poisson path
// Layouts (complex = interleaved double2):
//   specA[row][kx]            row = (j+HY) + py*k over the p field's padded rows, kx = 0..nkx-1
//   a2a blocks [d][k][kxl][j] kxl = 0..cx-1 (kx = d*cx + kxl), j = local row   (j fastest)
//   specB[k][kxl][y]          y = global row 0..jtot-1                          (y fastest)
// The x transform needs no communication; one all-to-all makes y whole with kx split, where the
// y transform AND the tridiagonal solve are local; one all-to-all goes back.  (The reference's
// pencil scheme does 8 transposes per solve, src/modpois.f90:459-702.)

// specA -> send blocks (transposes kx-fastest rows into j-fastest runs through an LDS tile)
// All four kernels work on one k-chunk [k0, k0+nzc): blocks are [d][k-k0][kxl][j] inside the chunk's
// slice of the send/recv buffers, so that chunks can be exchanged while others are transformed.
__global__ __launch_bounds__(256) void slab_pack_fwd_kernel(Geo g, int nkx, int pitch, int cx, int P, int k0, int nzc,
    const double2 *__restrict__ specA, double2 *__restrict__ send) {
  __shared__ double2 tile[16][17];
  const int kc = blockIdx.z, k = k0 + kc;
  const int kx0 = blockIdx.x * 16, j0 = blockIdx.y * 16;
  const int tx = threadIdx.x, ty = threadIdx.y;
  {
    const int kx = kx0 + tx, j = j0 + ty;
    double2 v = make_double2(0., 0.);
    if (kx < nkx && j < g.ny) v = specA[((size_t)(j + HY) + (size_t)g.py * k) * pitch + kx];
    tile[ty][tx] = v;
  }
  __syncthreads();
  {
    const int kx = kx0 + ty, j = j0 + tx;
    if (kx < cx * P && j < g.ny) {
      const int d = kx / cx, kxl = kx - d * cx;
      send[(((size_t)d * nzc + kc) * cx + kxl) * g.ny + j] = tile[tx][ty];
    }
  }
}

// received blocks -> specB (contiguous runs of ny_local)
__global__ __launch_bounds__(256) void slab_unpack_fwd_kernel(Geo g, int cx, int P, int jtot, int k0, int nzc,
    const double2 *__restrict__ recv, double2 *__restrict__ specB) {
  // grid (ceil(ny/256), cx, P*nzc): the run index comes from the block coordinates, no per-element 64-bit divisions
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= g.ny) return;
  const int kxl = blockIdx.y, s = blockIdx.z / nzc, kc = blockIdx.z - s * nzc;
  (void)P;
  specB[((size_t)(k0 + kc) * cx + kxl) * jtot + (size_t)s * g.ny + j] = recv[(((size_t)s * nzc + kc) * cx + kxl) * g.ny + j];
}

__global__ __launch_bounds__(256) void slab_pack_bwd_kernel(Geo g, int cx, int P, int jtot, int k0, int nzc,
    const double2 *__restrict__ specB, double2 *__restrict__ send) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= g.ny) return;
  const int kxl = blockIdx.y, d = blockIdx.z / nzc, kc = blockIdx.z - d * nzc;
  (void)P;
  send[(((size_t)d * nzc + kc) * cx + kxl) * g.ny + j] = specB[((size_t)(k0 + kc) * cx + kxl) * jtot + (size_t)d * g.ny + j];
}

__global__ __launch_bounds__(256) void slab_unpack_bwd_kernel(Geo g, int nkx, int pitch, int cx, int P, int k0, int nzc,
    const double2 *__restrict__ recv, double2 *__restrict__ specA) {
  __shared__ double2 tile[16][17];
  const int kc = blockIdx.z, k = k0 + kc;
  const int kx0 = blockIdx.x * 16, j0 = blockIdx.y * 16;
  const int tx = threadIdx.x, ty = threadIdx.y;
  {
    const int kx = kx0 + ty, j = j0 + tx;
    double2 v = make_double2(0., 0.);
    if (kx < cx * P && j < g.ny) {
      const int s = kx / cx, kxl = kx - s * cx;
      v = recv[(((size_t)s * nzc + kc) * cx + kxl) * g.ny + j];
    }
    tile[ty][tx] = v;
  }
  __syncthreads();
  {
    const int kx = kx0 + tx, j = j0 + ty;
    if (kx < nkx && j < g.ny) specA[((size_t)(j + HY) + (size_t)g.py * k) * pitch + kx] = tile[tx][ty];
  }
}

// tderive, src/modpois.f90:1046-1056,1096-1102 (p ghosts: x by wrap, y by ghost row)
__global__ __launch_bounds__(256) void project_kernel(Geo g, TileGrid tg, Metrics m, const double *__restrict__ p,
    double *__restrict__ up, double *__restrict__ vp, double *__restrict__ wp, double *__restrict__ pres0) {
  int i, j, k;
  const bool inside_ = tile_decode(g, tg, i, j, k);
  if (!inside_) return;
  const long r0 = g.idx(0, j, k);
  const long c = r0 + i, xm = r0 + wrapm(i, g.nx);
  const double pc = p[c];
  up[c] = up[c] - (pc - p[xm]) * m.dxi;
  vp[c] = vp[c] - (pc - p[c - g.sy]) * m.dyi;
  if (k >= 1) wp[c] = wp[c] - (pc - p[c - g.sz]) * m.dzhi[k + 1];
  pres0[c] = pres0[c] + pc;
}

struct IntArgs {
  double *u0, *v0, *w0, *um, *vm, *wm, *up, *vp, *wp;
  double *sv0[16], *svm[16], *svp[16];
  double clip[16];      // lower bound (e12min for the TKE slot, src/modtstep.f90:210-211) or a negative value = none
  int nsv;
};

// tstep_integrate, src/modtstep.f90:219-230,322-338; PROJECT fuses tderive in front of it.
// ZERO: write zeros to the tendencies as the reference does; the fused substep skips that (its next
// momentum sweep does not read them).
// PUP:  up,vp,wp hold pup = up + um/rk3coef, so u0 = rk3coef*(pup - grad p) and um is not read
//       (algebraically the reference's um + rk3coef*(up - grad p); differs by one rounding of um).
// PTOT: pressure-total form of the fused substep (udc_plan.h): the momentum sweep left the gradient of pres0 out, the solve returned
//       pres0 + p in `p`, which is projected here as a whole and IS the new pres0 (the caller swaps the two arrays): pres0 is neither read
//       nor written -- 16 of the kernel's 72 B per cell; `pres0` then points at the same array as `p`, for its two ghost rows.
template <bool PROJECT, bool ZERO, bool PUP, bool PTOT = false>
__global__ __launch_bounds__(256) void integrate_kernel(Geo g, TileGrid tg, Metrics m, IntArgs a, const double *__restrict__ p,
                                                         double *pres0, double rk3coef, int last,
                                                         int ghosts, Params pr) {
  int i, j, k;
  const bool inside_ = tile_decode(g, tg, i, j, k);
  if (!inside_) return;
  const long r0 = g.idx(0, j, k);
  const long c = r0 + i;
  // ghosts != 0 (the slab is the whole domain in y): `halos` and `boundary` (src/modboundary.f90:67-109,
  // 163-178) are folded in -- p(j-1) wraps by index, and the owning thread also writes the periodic
  // ghost row and the top ghost plane of what it updates.
  const long wr = !ghosts ? 0 : ((j == 0) ? (long)g.sy * g.ny : ((j == g.ny - 1) ? -(long)g.sy * g.ny : 0));
  double tu = NT_LOAD(&a.up[c]), tv = NT_LOAD(&a.vp[c]), tw = NT_LOAD(&a.wp[c]);
  double pr0 = 0.;
  if (PROJECT) {
    const long xm = r0 + wrapm(i, g.nx);
    const long ym = (ghosts && j == 0) ? c + (long)g.sy * (g.ny - 1) : c - g.sy;
    const double pc = p[c];
    tu = tu - (pc - p[xm]) * m.dxi;
    tv = tv - (pc - p[ym]) * m.dyi;
    if (k >= 1) tw = tw - (pc - p[c - g.sz]) * m.dzhi[k + 1];
    if (PTOT) {
      if (wr) pres0[c + wr] = pc;      // (ghost rows of the array itself: read by nothing in this launch)
    } else {
      pr0 = NT_LOAD(&pres0[c]) + pc;
      NT_STORE(pr0, &pres0[c]);
      if (wr) pres0[c + wr] = pr0;
    }
  }
  double u, v, w;
  if (PUP) { u = rk3coef * tu; v = rk3coef * tv; w = rk3coef * tw; }
  else { u = a.um[c] + rk3coef * tu; v = a.vm[c] + rk3coef * tv; w = a.wm[c] + rk3coef * tw; }
  if (ghosts && k == 0) w = 0.;                     // boundary: w(kb) = 0
  NT_STORE(u, &a.u0[c]); NT_STORE(v, &a.v0[c]); NT_STORE(w, &a.w0[c]);
  if (ZERO) { a.up[c] = 0.; a.vp[c] = 0.; a.wp[c] = 0.; }
  const bool last_m = (last & 1) != 0, last_s = (last & 2) != 0;   // write um.. / write svm..
  if (last_m) { NT_STORE(u, &a.um[c]); NT_STORE(v, &a.vm[c]); NT_STORE(w, &a.wm[c]); }
  if (ghosts) {
    if (wr) {
      a.u0[c + wr] = u; a.v0[c + wr] = v; a.w0[c + wr] = w;
      if (last_m) { a.um[c + wr] = u; a.vm[c + wr] = v; a.wm[c + wr] = w; }
    }
    if (k == g.nz - 1) {
      const bool ns = pr.bctopm == UDC_TOP_NOSLIP;
      const double ut = ns ? 2 * pr.uinf - u : u, vt = ns ? 2 * pr.vinf - v : v;
      const long t = c + g.sz;
      const bool wtop = pr.bctopm != UDC_TOP_PRESSURE;      // (open lid: w(ke+1) is lid_integrate_kernel's, `boundary` leaves it alone)
      a.u0[t] = ut; a.v0[t] = vt; if (wtop) a.w0[t] = 0.;
      if (last_m) { a.um[t] = ut; a.vm[t] = vt; if (wtop) a.wm[t] = 0.; }
      if (wr) {
        a.u0[t + wr] = ut; a.v0[t + wr] = vt; if (wtop) a.w0[t + wr] = 0.;
        if (last_m) { a.um[t + wr] = ut; a.vm[t + wr] = vt; if (wtop) a.wm[t + wr] = 0.; }
      }
    }
  }
  for (int s = 0; s < a.nsv; ++s) {
    double sv = NT_LOAD(&a.svm[s][c]) + rk3coef * NT_LOAD(&a.svp[s][c]);
    if (a.clip[s] > 0.) {
      sv = fmax(a.clip[s], sv);
      if (!last_s) a.svm[s][c] = fmax(a.clip[s], a.svm[s][c]);
    }
    NT_STORE(sv, &a.sv0[s][c]);
    if (ZERO) a.svp[s][c] = 0.;        // the fused substep's scalar sweep does not read svp either
    if (last_s) a.svm[s][c] = sv;
  }
}

// ---- BCzp = 2 (src/modpois.f90:179-191, 559-590): cosine transform in z on equidistant levels instead of the tridiagonal solve.  The
// cosine modes diagonalise exactly the matrix solmpj solves (a = c = dzi^2, Neumann rows at the floor and the lid), so every horizontal
// mode with a non-zero eigenvalue has the same unique solution either way; the singular mode (kx = ky = 0) is closed by a Dirichlet row at
// the lid there (:209-220) and by dropping its kz = 0 coefficient here -- two solutions of the same compatible system that differ by a
// constant: the one with zero mean over the levels is the cosine solve's.  One small kernel behind the Thomas solve, on the rank that holds
// kx = 0: x[k stride] -= mean_k x.
__global__ __launch_bounds__(256) void zero_mode_mean_kernel(int nz, size_t stride, double2 *__restrict__ x) {
  __shared__ double sr[4], si[4];
  double ar = 0., ai = 0.;
  for (int k = threadIdx.x; k < nz; k += 256) { const double2 v = x[(size_t)k * stride]; ar += v.x; ai += v.y; }
  for (int o = 32; o > 0; o >>= 1) { ar += __shfl_xor(ar, o, 64); ai += __shfl_xor(ai, o, 64); }
  if ((threadIdx.x & 63) == 0) { sr[threadIdx.x >> 6] = ar; si[threadIdx.x >> 6] = ai; }
  __syncthreads();
  const double mr = ((sr[0] + sr[1]) + (sr[2] + sr[3])) / nz, mi = ((si[0] + si[1]) + (si[2] + si[3])) / nz;
  for (int k = threadIdx.x; k < nz; k += 256) { double2 v = x[(size_t)k * stride]; v.x -= mr; v.y -= mi; x[(size_t)k * stride] = v; }
}

// ---- the open lid, BCtopm = 3 (BCtopm_pressure, src/modglobal.f90:142: "vertical velocity can vary according to pressure gradient")
// w(ke+1) is a prognostic plane: `boundary` leaves it alone (src/modboundary.f90:191-200), bcpup gives it the predicted velocity
//   pwp(ke+1) = wm(ke+1) / rk3coef + 2 <pres0>(ke) dzhi(ke+1),   wp(ke+1) = pwp(ke+1) - wm(ke+1) / rk3coef   (:1234-1243),
// fillps' divergence of level ke reads it, tderive adds 2 <p>(ke) dzhi(ke+1) (src/modpois.f90:1058-1069) and tstep_integrate steps
// it like any other level (src/modtstep.f90:270-286).  <.> = avexy_ibm: the mean of the level over the fluid c cells of the whole
// domain (src/modmpi.f90:623-664; -999 for a level without any).  S = that level's masked, all-reduced sum (k_level_sums_dev).
// Planes of nx x ny cells: three small kernels beside the sweeps, which keep their closed-lid code.
__device__ __forceinline__ double lid_mean(const double *S, double cnt) { return cnt > 0. ? S[0] / cnt : -999.; }
// pup != 0: the tendency arrays hold the predicted velocity (the fused substep's form): wp(ke+1) takes pwp(ke+1)
// S null: the pressure-total form -- the row carries no term of pres0 (planner, udc_plan.h: the solve then returns pres0 + p with the
// Dirichlet row of the zero mode applied to the sum, and lid_tderive_kernel's 2 <pres0 + p>(ke) dzhi(ke+1) is the whole of it)
__global__ void lid_bcpup_kernel(Geo g, double r, double dzhi_top, double cnt, const double *__restrict__ S,
                                 const double *__restrict__ wm, double *__restrict__ wp, int pup) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y;
  if (i >= g.nx) return;
  const long c = g.idx(i, j, g.nz);
  const double pres0ij = S ? lid_mean(S, cnt) : 0.;
  const double pwp = wm[c] * r + 2 * pres0ij * dzhi_top;
  wp[c] = pup ? pwp : pwp - wm[c] * r;
}
__global__ void lid_tderive_kernel(Geo g, double dzhi_top, double cnt, const double *__restrict__ S, double *__restrict__ wp) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y;
  if (i >= g.nx) return;
  const long c = g.idx(i, j, g.nz);
  wp[c] = wp[c] + 2 * lid_mean(S, cnt) * dzhi_top;
}
// w0(ke+1) = wm(ke+1) + rk3coef wp(ke+1) (pup: rk3coef pwp(ke+1)); zero: wp(ke+1) = 0 as the reference's `wp = 0.`; last: wm = w0;
// wrap: the slab is the whole domain in y and the periodic ghost rows are written here (`halos` folded in)
__global__ void lid_integrate_kernel(Geo g, double rk3coef, int pup, int zero, int last, int wrap, const double *__restrict__ wm,
                                     double *__restrict__ wp, double *__restrict__ w0, double *__restrict__ wm_out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y;
  if (i >= g.nx) return;
  const long c = g.idx(i, j, g.nz);
  const double w = pup ? rk3coef * wp[c] : wm[c] + rk3coef * wp[c];
  const long wr = !wrap ? 0 : ((j == 0) ? (long)g.sy * g.ny : ((j == g.ny - 1) ? -(long)g.sy * g.ny : 0));
  w0[c] = w;
  if (wr) w0[c + wr] = w;
  if (zero) wp[c] = 0.;
  if (last) { wm_out[c] = w; if (wr) wm_out[c + wr] = w; }
}

// ---- reductions ---------------------------------------------------------------------
__device__ __forceinline__ double wave_max(double v) {
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ double wave_sum(double v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// Two-stage, atomic-free reductions (deterministic order): every workgroup writes its two partial
// results to part[2*block .. +1]; reduce_partials_kernel folds them.  OP 0 = max, 1 = sum.
template <int OP> __device__ __forceinline__ double red_op(double a, double b) { return OP == 0 ? fmax(a, b) : a + b; }
template <int OP> __device__ __forceinline__ double wave_red(double v) {
  for (int o = 32; o > 0; o >>= 1) v = red_op<OP>(v, __shfl_xor(v, o, 64));
  return v;
}
template <int OP0, int OP1>
__device__ __forceinline__ void block_reduce2(double a, double b, double *__restrict__ part) {
  __shared__ double sa[16], sb[16];
  const int tid = threadIdx.y * blockDim.x + threadIdx.x, wv = tid >> 6, nw = (blockDim.x * blockDim.y + 63) >> 6;
  a = wave_red<OP0>(a); b = wave_red<OP1>(b);
  if ((tid & 63) == 0) { sa[wv] = a; sb[wv] = b; }
  __syncthreads();
  if (tid == 0) {
    for (int q = 1; q < nw; ++q) { a = red_op<OP0>(a, sa[q]); b = red_op<OP1>(b, sb[q]); }
    part[2 * (size_t)blockIdx.x] = a;
    part[2 * (size_t)blockIdx.x + 1] = b;
  }
}
template <int OP0, int OP1>
__global__ __launch_bounds__(1024) void reduce_partials_kernel(const double *__restrict__ part, long n, double init0,
                                                                double init1, double *__restrict__ out) {
  double a = init0, b = init1;
  for (long q = threadIdx.x; q < n; q += blockDim.x) { a = red_op<OP0>(a, part[2 * q]); b = red_op<OP1>(b, part[2 * q + 1]); }
  block_reduce2<OP0, OP1>(a, b, out);
}

// tstep_update, src/modtstep.f90:113-128; CHECKSIM: the diagnostics of checksim instead (calccourant, src/modchecksim.f90:102-127: the
// SIGNED sum um dxhi + vm dyi + wm dzhi; calcdiffnr, :130-157: 1 / dzh(k)**2 formed on the spot)
template <bool CHECKSIM>
__global__ __launch_bounds__(256) void maxima_kernel(Geo g, TileGrid tg, Metrics m, double dt, const double *__restrict__ um,
    const double *__restrict__ vm, const double *__restrict__ wm, const double *__restrict__ ekm,
    const double *__restrict__ ekh, double *__restrict__ out) {
  int i, j, k;
  const bool inside_ = tile_decode(g, tg, i, j, k) && i >= g.xg && i < g.nx - g.xg;      // (open x boundaries: not the ghost columns)
  double cour = 0., dif = 0.;
  if (inside_) {
    const long c = g.idx(i, j, k);
    if (CHECKSIM) {
      cour = (um[c] * m.dxi + vm[c] * m.dyi + wm[c] * m.dzhi[k + 1]) * dt;
      const double dzh = m.dzh[k + 1];
      const double f = (1 / (dzh * dzh) + m.dx2i + m.dy2i);
      dif = fmax(ekm[c] * f * dt, ekh[c] * f * dt);
    } else {
      cour = (fabs(um[c]) * m.dxi + fabs(vm[c]) * m.dyi + fabs(wm[c]) / m.dzh[k + 1]) * dt;
      const double f = (m.dzh2i[k + 1] + m.dx2i + m.dy2i);
      dif = fmax(ekm[c] * f * dt, ekh[c] * f * dt);
    }
  }
  block_reduce2<0, 0>(cour, dif, out);
}

// chkdiv, src/modchecksim.f90:179-191
__global__ __launch_bounds__(256) void divcheck_kernel(Geo g, TileGrid tg, Metrics m, const double *__restrict__ u,
    const double *__restrict__ v, const double *__restrict__ w, double *__restrict__ out) {
  int i, j, k;
  const bool inside_ = tile_decode(g, tg, i, j, k) && i >= g.xg && i < g.nx - g.xg;
  double dmax = 0., dsum = 0.;
  if (inside_) {
    const long r0 = g.idx(0, j, k);
    const long c = r0 + i, xp = r0 + wrapp(i, g.nx);
    const double div = (u[xp] - u[c]) * m.dxi + (v[c + g.sy] - v[c]) * m.dyi + (w[c + g.sz] - w[c]) * m.dzfi[k + 1];
    dmax = fabs(div);
    dsum = div * m.dx * m.dy * m.dzf[k + 1];
  }
  block_reduce2<0, 1>(dmax, dsum, out);
}


// checksim's three routines in one sweep (the state is read once: 40 B per cell instead of 64) for the usual case that um is u0
// (after RK stage 3 of an aliased fused substep): calccourant + calcdiffnr as maxima_kernel<true>, chkdiv as divcheck_kernel.
// out: [2 nblocks] (courant, diffusion) maxima, then [2 nblocks] (divmax, divtot).
__global__ __launch_bounds__(256) void checksim_kernel(Geo g, TileGrid tg, Metrics m, double dt, const double *__restrict__ u,
    const double *__restrict__ v, const double *__restrict__ w, const double *__restrict__ ekm, const double *__restrict__ ekh,
    double *__restrict__ out) {
  int i, j, k;
  const bool inside_ = tile_decode(g, tg, i, j, k) && i >= g.xg && i < g.nx - g.xg;
  double cour = 0., dif = 0., dmax = 0., dsum = 0.;
  if (inside_) {
    const long r0 = g.idx(0, j, k);
    const long c = r0 + i, xp = r0 + wrapp(i, g.nx);
    const double uc = u[c], vc = v[c], wc = w[c];
    cour = (uc * m.dxi + vc * m.dyi + wc * m.dzhi[k + 1]) * dt;
    const double dzh = m.dzh[k + 1];
    const double f = (1 / (dzh * dzh) + m.dx2i + m.dy2i);
    dif = fmax(ekm[c] * f * dt, ekh[c] * f * dt);
    const double div = (u[xp] - uc) * m.dxi + (v[c + g.sy] - vc) * m.dyi + (w[c + g.sz] - wc) * m.dzfi[k + 1];
    dmax = fabs(div);
    dsum = div * m.dx * m.dy * m.dzf[k + 1];
  }
  block_reduce2<0, 0>(cour, dif, out);
  __syncthreads();      // (block_reduce2's staging arrays are reused)
  block_reduce2<0, 1>(dmax, dsum, out + 2 * (size_t)gridDim.x);
}

// masscorr, src/modforces.f90:328-497: volume-flow branches (luvolflowr :389-417, lvvolflowr :467-494) and the u outflow-rate
// branch (luoutflowr :352-387).
// flowsum: S_a = sum(a w(k)), S_b = sum(b w(k)) over the slab interior (b may be null); w = dzf, or with an immersed
// boundary dzf(k) / (fluid cells of level k) / zh(ke+1) so that the sum is the volume average over the fluid.  only_i >= 0: the
// plane i = only_i alone (the outlet plane i = ie; w = dy dzf(k) / outlet area).
__global__ __launch_bounds__(256) void flowsum_kernel(Geo g, TileGrid tg, const double *__restrict__ wlev, const double *__restrict__ a,
                                                      const double *__restrict__ b, double *__restrict__ out, int only_i) {
  int i, j, k;
  const bool inside_ = tile_decode(g, tg, i, j, k) && i >= g.xg && i < g.nx - g.xg;      // (open x boundaries: ib .. ie)
  double sa = 0., sb = 0.;
  if (inside_ && (only_i < 0 || i == only_i)) {
    const long c = g.idx(i, j, k);
    const double w = wlev[k + 1];
    sa = a[c] * w;
    if (b) sb = b[c] * w;
  }
  block_reduce2<1, 1>(sa, sb, out);
}
// field += (target - (ca S[0] + cb S[1])) rk3coefi for up to two components (null field = skip)
struct FlowShift { double *f; const double *S; double target, ca, cb; };
__global__ __launch_bounds__(256) void flowshift_kernel(Geo g, TileGrid tg, FlowShift u, FlowShift v, double rk3coefi, int wrap_vp) {
  int i, j, k;
  const bool inside_ = tile_decode(g, tg, i, j, k);
  if (!inside_) return;
  const long c = g.idx(i, j, k);
  if (u.f) {
    const double def = u.target - (u.ca * u.S[0] + u.cb * u.S[1]);
    u.f[c] = u.f[c] + def * rk3coefi;
  }
  if (v.f) {
    const double def = v.target - (v.ca * v.S[0] + v.cb * v.S[1]);
    const double t = v.f[c] + def * rk3coefi;
    v.f[c] = t;
    if (wrap_vp && j == 0) v.f[c + (long)g.sy * g.ny] = t;
  }
}

}  // namespace

// --------------------------------------------------------------------------------------
// rocfft_setup exactly once per process, whichever handle (and host thread: the virtual ranks of the tests create theirs
// concurrently) gets there first
static int rocfft_setup_once() {
  static std::once_flag once;
  static rocfft_status st = rocfft_status_success;
  std::call_once(once, [] { st = rocfft_setup(); });
  if (st != rocfft_status_success) { udc_set_error("rocfft_setup failed: rocfft status %d", (int)st); return 1; }
  return 0;
}

int pois_init(udc_handle *h) {
  const Geo &g = h->g;
  const int nx = g.nx, ny = g.ny, nz = g.nz;
  const int nkx = nx / 2 + 1;
  h->nkx = nkx;
  // row pitch of the spectral array (complex elements): nkx = nx/2+1 is odd; padding it lets
  // rocFFT's strided y pass and the Thomas sweep run on aligned rows.  Padding modes hold zeros.
  const int pad = (8 - nkx % 8) % 8;
  const int nkxp = nkx + pad;
  h->nkxp = nkxp;
  const long nmodes = (long)nkxp * ny;
  const double pi = 3.141592653589793116;   // src/modglobal.f90:270
  const double dxi = h->m.dxi, dyi = h->m.dyi;
  // eigenvalues, src/modpois.f90:100-107,124-131 (value of complex mode kx = slot 2kx)
  std::vector<double> xrt(nkx), yrt(ny), ev(nmodes, -1.0);
  {
    const double fac = 1. / (2. * nx);
    for (int kx = 1; kx < nx / 2; ++kx) { double s = sin((double)(2 * kx) * pi * fac); xrt[kx] = -4. * dxi * dxi * (s * s); }
    xrt[0] = 0.; xrt[nx / 2] = -4. * dxi * dxi;
  }
  {
    const double fac = 1. / (2. * ny);
    for (int ky = 0; ky < ny; ++ky) {
      const int mm = ky <= ny / 2 ? ky : ny - ky;
      if (mm == 0) yrt[ky] = 0.;
      else if (mm == ny / 2) yrt[ky] = -4. * dyi * dyi;
      else { double s = sin((double)(2 * mm) * pi * fac); yrt[ky] = -4. * dyi * dyi * (s * s); }
    }
  }
  for (int ky = 0; ky < ny; ++ky)
    for (int kx = 0; kx < nkx; ++kx) ev[(long)ky * nkxp + kx] = 1. * (xrt[kx] + yrt[ky] + 0.);
  // tridiagonal coefficients, :154-176 (rhobf = rhobh = 1)
  std::vector<double> tri(3 * (nz + 2), 0.0);
  double *a = &tri[0], *b = &tri[nz + 2], *c = &tri[2 * (nz + 2)];
  for (int k = 1; k <= nz; ++k) {
    a[k] = 1. / (h->cfg.dzf[k] * h->cfg.dzh[k]);
    c[k] = 1. / (h->cfg.dzf[k] * h->cfg.dzh[k + 1]);
    b[k] = -(a[k] + c[k]);
  }
  b[1] = b[1] + a[1];
  const double b_top_N = b[nz] + c[nz];
  const double b_top_D = b[nz] - c[nz];
  b[nz] = b_top_N;
  a[1] = 0.; c[nz] = 0.;
  h->btopD = b_top_D;

  HIP_OK(hipMalloc(&h->spec, sizeof(double) * 2 * nmodes * nz));
  HIP_OK(hipMemsetAsync(h->spec, 0, sizeof(double) * 2 * nmodes * nz, h->stream));
  h->thomas_lds = thomas_wants_lds(h, (long)nmodes, nz);
  HIP_OK(hipMalloc(&h->ztab, sizeof(double) * ztab_doubles((long)nmodes, nz)));
  HIP_OK(hipMalloc(&h->ev, sizeof(double) * nmodes));
  HIP_OK(hipMalloc(&h->tri, sizeof(double) * tri.size()));
  HIP_OK(hipMemcpy(h->ev, ev.data(), sizeof(double) * nmodes, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(h->tri, tri.data(), sizeof(double) * tri.size(), hipMemcpyHostToDevice));
  hipLaunchKernelGGL(thomas_table_kernel, dim3((unsigned)((nmodes + 63) / 64)), dim3(64), 0, h->stream,
                     (int)nmodes, nz, h->ev, h->tri, b_top_D, h->ztab, h->thomas_lds ? 1 : 0);
  HIP_OK(hipGetLastError());

  // rocFFT: batched 2-D real <-> Hermitian-interleaved, reading/writing the padded p field
  if (rocfft_setup_once()) return 1;
  size_t lengths[2] = {(size_t)nx, (size_t)ny};
  size_t cstr[2] = {1, (size_t)nkxp};
  size_t off[1] = {0};
  // rocFFT (ROCm 7.2) refuses some padded real layouts (e.g. 8x8, 16x16 inverse): fall back to a
  // compact real staging buffer + one strided copy for those sizes only.
  for (int dir = 0; dir < 2; ++dir) {
    rocfft_plan *plan = dir == 0 ? &h->plan_fwd : &h->plan_bwd;
    bool *compact = dir == 0 ? &h->fwd_compact : &h->bwd_compact;
    for (int attempt = 0; attempt < 2 && !*plan; ++attempt) {
      const bool cmp = attempt == 1;
      size_t rstr[2] = {1, cmp ? (size_t)nx : (size_t)g.sy};
      const size_t rdist = cmp ? (size_t)nx * ny : (size_t)g.sz;
      rocfft_plan_description d = nullptr;
      FFT_OK(rocfft_plan_description_create(&d));
      if (dir == 0)
        FFT_OK(rocfft_plan_description_set_data_layout(d, rocfft_array_type_real, rocfft_array_type_hermitian_interleaved,
                                                       off, off, 2, rstr, rdist, 2, cstr, (size_t)nmodes));
      else
        FFT_OK(rocfft_plan_description_set_data_layout(d, rocfft_array_type_hermitian_interleaved, rocfft_array_type_real,
                                                       off, off, 2, cstr, (size_t)nmodes, 2, rstr, rdist));
      rocfft_status st = rocfft_plan_create(plan, rocfft_placement_notinplace,
                                            dir == 0 ? rocfft_transform_type_real_forward : rocfft_transform_type_real_inverse,
                                            rocfft_precision_double, 2, lengths, (size_t)nz, d);
      rocfft_plan_description_destroy(d);
      if (st != rocfft_status_success) *plan = nullptr;
      else *compact = cmp;
    }
    if (!*plan) { udc_set_error("rocfft_plan_create failed for %dx%d (batch %d)", nx, ny, nz); return 1; }
  }
  if (h->fwd_compact || h->bwd_compact) HIP_OK(hipMalloc(&h->rbuf, sizeof(double) * (size_t)nx * ny * nz));
  size_t wf = 0, wb = 0;
  FFT_OK(rocfft_plan_get_work_buffer_size(h->plan_fwd, &wf));
  FFT_OK(rocfft_plan_get_work_buffer_size(h->plan_bwd, &wb));
  const size_t wmax = wf > wb ? wf : wb;
  if (wmax) HIP_OK(hipMalloc(&h->fft_work, wmax));
  FFT_OK(rocfft_execution_info_create(&h->info_fwd));
  FFT_OK(rocfft_execution_info_create(&h->info_bwd));
  FFT_OK(rocfft_execution_info_set_stream(h->info_fwd, h->stream));
  FFT_OK(rocfft_execution_info_set_stream(h->info_bwd, h->stream));
  if (wf) FFT_OK(rocfft_execution_info_set_work_buffer(h->info_fwd, h->fft_work, wf));
  if (wb) FFT_OK(rocfft_execution_info_set_work_buffer(h->info_bwd, h->fft_work, wb));
  if (fft_nat_init(h)) return 1;
  return 0;
}


static void poisson_coefficients(udc_handle *h, std::vector<double> &xrt, std::vector<double> &yrt,
                                 std::vector<double> &tri, double &b_top_D) {
  const int nx = h->g.nx, ny = h->jtot, nz = h->g.nz, nkx = nx / 2 + 1;
  const double pi = 3.141592653589793116;   // src/modglobal.f90:270
  const double dxi = h->m.dxi, dyi = h->m.dyi;
  xrt.assign(nkx, 0.); yrt.assign(ny, 0.);
  {
    const double fac = 1. / (2. * nx);
    for (int kx = 1; kx < nx / 2; ++kx) { double s = sin((double)(2 * kx) * pi * fac); xrt[kx] = -4. * dxi * dxi * (s * s); }
    xrt[0] = 0.; xrt[nx / 2] = -4. * dxi * dxi;
  }
  {
    const double fac = 1. / (2. * ny);
    for (int ky = 0; ky < ny; ++ky) {
      const int mm = ky <= ny / 2 ? ky : ny - ky;
      if (mm == 0) yrt[ky] = 0.;
      else if (mm == ny / 2) yrt[ky] = -4. * dyi * dyi;
      else { double s = sin((double)(2 * mm) * pi * fac); yrt[ky] = -4. * dyi * dyi * (s * s); }
    }
  }
  tri.assign(3 * (nz + 2), 0.0);
  double *a = &tri[0], *b = &tri[nz + 2], *c = &tri[2 * (nz + 2)];
  for (int k = 1; k <= nz; ++k) {
    a[k] = 1. / (h->cfg.dzf[k] * h->cfg.dzh[k]);
    c[k] = 1. / (h->cfg.dzf[k] * h->cfg.dzh[k + 1]);
    b[k] = -(a[k] + c[k]);
  }
  b[1] = b[1] + a[1];
  const double b_top_N = b[nz] + c[nz];
  b_top_D = b[nz] - c[nz];
  b[nz] = b_top_N;
  a[1] = 0.; c[nz] = 0.;
}

int pois_slab_init(udc_handle *h) {
  const Geo &g = h->g;
  const int nx = g.nx, nyl = g.ny, ny = h->jtot, nz = g.nz, P = h->cfg.nranks, r = h->cfg.rank;
  const int nkx = nx / 2 + 1, cx = (nkx + P - 1) / P;
  const int nkxp = nkx + (8 - nkx % 8) % 8;      // aligned row pitch of specA
  h->nkx = nkx; h->cx = cx; h->nkxp = nkxp;
  std::vector<double> xrt, yrt, tri;
  double btopD;
  poisson_coefficients(h, xrt, yrt, tri, btopD);
  h->btopD = btopD;
  const size_t nmodes = (size_t)cx * ny;
  std::vector<double> ev(nmodes);
  for (int kxl = 0; kxl < cx; ++kxl) {
    const int kx = r * cx + kxl;
    for (int y = 0; y < ny; ++y)
      ev[(size_t)kxl * ny + y] = kx < nkx ? 1. * (xrt[kx] + yrt[y] + 0.) : -1.0;   // padding modes carry zeros
  }
  // k-chunks of the all-to-all pipeline (UDC_A2A_CHUNKS overrides; chunks must divide nz)
  // four where there is something to pipeline (more than one rank, or a forced slab standing in for one) and a chunk keeps four levels
  // (measured with a one-rank RCCL communicator at 1024 x 512 x 512: 21.1 / 19.7 / 19.1 / 19.8 / 20.0 ms for 1 / 2 / 4 / 8 / 16)
  int nch = ((P > 1 || h->sw.force_slab) && nz % 4 == 0 && nz >= 16) ? 4 : 1;
  if (h->sw.a2a_chunks > 0) nch = h->sw.a2a_chunks;
  if (nch < 1 || nch > 16 || nz % nch) nch = 1;
  h->nch = nch;
  const int nzc = nz / nch;
  const size_t rows = (size_t)g.py * nz;
  HIP_OK(hipMalloc(&h->specA, sizeof(double) * 2 * nkxp * rows));
  HIP_OK(hipMemsetAsync(h->specA, 0, sizeof(double) * 2 * nkxp * rows, h->stream));
  HIP_OK(hipMalloc(&h->specB, sizeof(double) * 2 * nmodes * nz));
  // (+ 2 rows per block: the backward blocks may carry p's ghost rows, fft_y_bwd_pack)
  const size_t a2a_doubles = (size_t)2 * cx * ((size_t)ny + 2 * (size_t)P) * nz;
  HIP_OK(hipMalloc(&h->a2a_send, sizeof(double) * a2a_doubles));
  HIP_OK(hipMalloc(&h->a2a_recv, sizeof(double) * a2a_doubles));
  HIP_OK(hipMalloc(&h->ev_slab, sizeof(double) * nmodes));
  h->thomas_lds_slab = thomas_wants_lds(h, (long)nmodes, nz);
  HIP_OK(hipMalloc(&h->ztab_slab, sizeof(double) * ztab_doubles((long)nmodes, nz)));
  HIP_OK(hipMalloc(&h->tri, sizeof(double) * tri.size()));
  HIP_OK(hipMemcpy(h->ev_slab, ev.data(), sizeof(double) * nmodes, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(h->tri, tri.data(), sizeof(double) * tri.size(), hipMemcpyHostToDevice));
  hipLaunchKernelGGL(thomas_table_kernel, dim3((unsigned)((nmodes + 63) / 64)), dim3(64), 0, h->stream,
                     (int)nmodes, nz, h->ev_slab, h->tri, btopD, h->ztab_slab, h->thomas_lds_slab ? 1 : 0);
  HIP_OK(hipGetLastError());

  if (rocfft_setup_once()) return 1;
  size_t off[1] = {0}, one[1] = {1};
  size_t lx[1] = {(size_t)nx}, ly[1] = {(size_t)ny};
  rocfft_plan_description d = nullptr;
  FFT_OK(rocfft_plan_description_create(&d));
  FFT_OK(rocfft_plan_description_set_data_layout(d, rocfft_array_type_real, rocfft_array_type_hermitian_interleaved,
                                                 off, off, 1, one, (size_t)g.sy, 1, one, (size_t)nkxp));
  FFT_OK(rocfft_plan_create(&h->plan_xf, rocfft_placement_notinplace, rocfft_transform_type_real_forward,
                            rocfft_precision_double, 1, lx, (size_t)g.py * nzc, d));
  rocfft_plan_description_destroy(d);
  FFT_OK(rocfft_plan_description_create(&d));
  FFT_OK(rocfft_plan_description_set_data_layout(d, rocfft_array_type_hermitian_interleaved, rocfft_array_type_real,
                                                 off, off, 1, one, (size_t)nkxp, 1, one, (size_t)g.sy));
  FFT_OK(rocfft_plan_create(&h->plan_xb, rocfft_placement_notinplace, rocfft_transform_type_real_inverse,
                            rocfft_precision_double, 1, lx, (size_t)g.py * nzc, d));
  rocfft_plan_description_destroy(d);
  FFT_OK(rocfft_plan_create(&h->plan_yf, rocfft_placement_inplace, rocfft_transform_type_complex_forward,
                            rocfft_precision_double, 1, ly, (size_t)cx * nzc, nullptr));
  FFT_OK(rocfft_plan_create(&h->plan_yb, rocfft_placement_inplace, rocfft_transform_type_complex_inverse,
                            rocfft_precision_double, 1, ly, (size_t)cx * nzc, nullptr));
  h->fft_fused = fft_fused_possible(h) && h->sw.fft_fused;
  if (h->fft_fused && fft_fused_init(h)) return 1;
  {
    // exchanges run beside the transforms / the interior launches of the split kernels: highest priority, so that their few
    // workgroups are dispatched ahead of the compute stream's queue instead of behind it
    int lo = 0, hi = 0;
    HIP_OK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    HIP_OK(hipStreamCreateWithPriority(&h->comm_stream, hipStreamNonBlocking, hi));
    HIP_OK(hipEventCreateWithFlags(&h->ev_halo_ready, hipEventDisableTiming));
    HIP_OK(hipEventCreateWithFlags(&h->ev_halo_done, hipEventDisableTiming));
    h->no_halo_overlap = !h->sw.halo_overlap;
    h->no_mom_pipe = !h->sw.mom_pipe;
  }
  for (int c = 0; c < nch; ++c) {
    HIP_OK(hipEventCreateWithFlags(&h->ev_ready[c], hipEventDisableTiming));
    HIP_OK(hipEventCreateWithFlags(&h->ev_done[c], hipEventDisableTiming));
  }
  size_t w = 0, wmax = 0;
  rocfft_plan plans[4] = {h->plan_xf, h->plan_xb, h->plan_yf, h->plan_yb};
  for (auto pl : plans) { FFT_OK(rocfft_plan_get_work_buffer_size(pl, &w)); if (w > wmax) wmax = w; }
  if (wmax) HIP_OK(hipMalloc(&h->fft_work_slab, wmax));
  FFT_OK(rocfft_execution_info_create(&h->info_x));
  FFT_OK(rocfft_execution_info_set_stream(h->info_x, h->stream));
  if (wmax) FFT_OK(rocfft_execution_info_set_work_buffer(h->info_x, h->fft_work_slab, wmax));
  (void)nyl;
  return 0;
}

int k_poisson_solve_slab(udc_handle *h) {
  const Geo &g = h->g;
  const int P = h->cfg.nranks, cx = h->cx, nkx = h->nkx, ny = h->jtot, nch = h->nch, nzc = g.nz / nch;
  const size_t nmodes = (size_t)cx * ny;
  const size_t block = (size_t)2 * nzc * cx * g.ny;            // doubles per all-to-all block of one chunk
  const size_t chunk = block * P;                              // doubles per chunk in the send/recv buffers
  double *prow0 = h->fields[UDC_P] + g.idx(0, -HY, 0);         // first padded row of plane k = 0
  const dim3 tb(16, 16), tg((cx * P + 15) / 16, (g.ny + 15) / 16, nzc);
  const unsigned lb = g.ny >= 256 ? 256 : (g.ny >= 128 ? 128 : 64);      // threads along a run of ny_local
  const dim3 lin3((unsigned)((g.ny + lb - 1) / lb), (unsigned)cx, (unsigned)(P * nzc));
  const int pitch = h->nkxp;
  auto specA_at = [&](int k0) { return h->specA + (size_t)2 * pitch * g.py * k0; };
  auto specB_at = [&](int k0) { return h->specB + (size_t)2 * nmodes * k0; };
  // the exchange of chunk c runs on the communication stream while the compute stream transforms and
  // packs chunk c+1 (forward) / unpacks and transforms chunk c-1: xGMI transfers hide behind rocFFT
  // one rank without a communicator (UDC_FORCE_SLAB on a single GPU): the exchange is the identity, the "received"
  // blocks are the packed ones
  const bool self = P == 1 && !h->nccl && !h->local_group;
  double *const rbuf = self ? h->a2a_send : h->a2a_recv;
  // backward blocks with p's two ghost rows behind the slab's own (substep_fused asked for them; own line transforms only)
  const bool pg = h->p_ghost_in_transpose && h->fft_fused;
  const size_t blockB = pg ? (size_t)2 * nzc * cx * (g.ny + 2) : block, chunkB = blockB * P;
  auto exchange = [&](int c, bool back = false) -> int {
    if (self) return 0;
    HIP_OK(hipEventRecord(h->ev_ready[c], h->stream));
    HIP_OK(hipStreamWaitEvent(h->comm_stream, h->ev_ready[c], 0));
    const size_t ch = back ? chunkB : chunk;
    if (comm_alltoall(h, h->a2a_send + ch * c, rbuf + ch * c, back ? blockB : block, h->comm_stream)) return 1;
    HIP_OK(hipEventRecord(h->ev_done[c], h->comm_stream));
    return 0;
  };
  // The pipelined momentum sweep, every piece over all tile rows (UDC_MOM_PIPE=2): piece c + 1 is launched ahead of the x transform
  // of chunk c, so that vp's ghost row of chunk c -- sent when piece c was done -- has a whole piece's time to arrive and the transform
  // is one launch that waits for nothing:  p0 p1 x0 p2 x1 p3 x2 x3.  (Two other orders were measured at 1024 x 64 x 512 and dropped,
  // profiles/r06/mom_pipe_orders_ab.txt: the transform split at its last row group with a join in between, 8 launches, +0.09 ms on the
  // transforms; that last row group on the communication stream behind the row's unpack, which slows the sweep running beside it.)
  const bool ahead = h->mom_pipe.active && h->mom_pipe.rows_all;
  {
    for (int c = 0; c < nch; ++c) {
      const int k0 = c * nzc;
      // (the momentum sweep's levels for this chunk, when it is pipelined with the solve)
      if (ahead) {
        // (late: the row of piece c + 1 starts travelling behind x transform c, i.e. beside piece c + 2 and not beside the transform,
        //  which its pack / copy / unpack slowed down: 0.304 -> 0.286 ms at 1024 x 64 x 512, profiles/r06/mom_pipe_orders_ab.txt; the last
        //  piece's row leaves at once: its transform follows the one before without a piece in between)
        const bool late = c + 2 < nch;
        if (c == 0 && k_momentum_pipe_stage(h, 0)) return 1;
        if (c + 1 < nch && k_momentum_pipe_stage(h, c + 1, !late)) return 1;
        if (h->ev_vp[c]) HIP_OK(hipStreamWaitEvent(h->stream, h->ev_vp[c], 0));      // vp's ghost row of this chunk's levels is in
        if (c == nch - 1) { h->vp_halo_pending = false; h->halo_async_pending = false; }
      } else if (k_momentum_pipe_stage(h, c)) return 1;
      PROF(h, c == nch - 1 ? "fftx_pack_fwd" : "fftx_pack_fwd_edge");      // (one table row per substep: bench.py folds *_edge)
      if (h->fft_fused) {
        const int G = fft_x_row_groups(h);
        if (ahead) {
          if (fft_x_fwd_pack(h, k0, nzc, h->a2a_send + chunk * c)) return 1;
          if (exchange(c)) return 1;
          if (c + 2 < nch && k_momentum_pipe_row(h, c + 1)) return 1;
          continue;
        }
        if (h->vp_halo_pending && G >= 2) {
          // vp's ghost row (the divergence of the slab's last row reads it) is still travelling: every row group but the last first
          if (fft_x_fwd_pack(h, k0, nzc, h->a2a_send + chunk * c, 0, G - 1)) return 1;
          if (k_halo_y_join(h)) return 1;
          h->vp_halo_pending = false;
          if (fft_x_fwd_pack(h, k0, nzc, h->a2a_send + chunk * c, G - 1, G)) return 1;
        } else {
          if (h->vp_halo_pending) { if (k_halo_y_join(h)) return 1; h->vp_halo_pending = false; }
          if (fft_x_fwd_pack(h, k0, nzc, h->a2a_send + chunk * c)) return 1;
        }
      } else {
        if (h->vp_halo_pending) { if (k_halo_y_join(h)) return 1; h->vp_halo_pending = false; }
        void *in[1] = {prow0 + g.sz * k0}, *out[1] = {specA_at(k0)};
        FFT_OK(rocfft_execute(h->plan_xf, in, out, h->info_x));
        hipLaunchKernelGGL(slab_pack_fwd_kernel, tg, tb, 0, h->stream, g, nkx, pitch, cx, P, k0, nzc,
                           reinterpret_cast<const double2 *>(h->specA), reinterpret_cast<double2 *>(h->a2a_send + chunk * c));
      }
      if (exchange(c)) return 1;
    }
    HIP_OK(hipGetLastError());
  }
  {
    PROF(h, "unpack_ffty_fwd");
    for (int c = 0; c < nch; ++c) {
      const int k0 = c * nzc;
      if (!self) HIP_OK(hipStreamWaitEvent(h->stream, h->ev_done[c], 0));
      if (h->fft_fused) {
        if (fft_y_fwd_unpack(h, k0, nzc, rbuf + chunk * c)) return 1;
        continue;
      }
      hipLaunchKernelGGL(slab_unpack_fwd_kernel, lin3, dim3(lb), 0, h->stream, g, cx, P, ny, k0, nzc,
                         reinterpret_cast<const double2 *>(rbuf + chunk * c), reinterpret_cast<double2 *>(h->specB));
      void *io[1] = {specB_at(k0)};
      FFT_OK(rocfft_execute(h->plan_yf, io, nullptr, h->info_x));
    }
    HIP_OK(hipGetLastError());
  }
  {
    PROF(h, "thomas");
    if (launch_thomas(h, h->thomas_lds_slab, nmodes, g.nz, 1. / ((double)g.nx * (double)ny), h->ev_slab, h->ztab_slab,
                      reinterpret_cast<double2 *>(h->specB), 0, 0, ny)) return 1;
    if (h->bczp == 2 && h->cfg.rank == 0)      // (rank 0 holds kx = 0: specB[k][kx_l = 0][y = 0])
      hipLaunchKernelGGL(zero_mode_mean_kernel, dim3(1), dim3(256), 0, h->stream, g.nz, nmodes, reinterpret_cast<double2 *>(h->specB));
    HIP_OK(hipGetLastError());
  }
  {
    PROF(h, "ffty_pack_bwd");
    for (int c = 0; c < nch; ++c) {
      const int k0 = c * nzc;
      if (h->fft_fused) {
        if (fft_y_bwd_pack(h, k0, nzc, h->a2a_send + chunkB * c, pg)) return 1;
      } else {
        void *io[1] = {specB_at(k0)};
        FFT_OK(rocfft_execute(h->plan_yb, io, nullptr, h->info_x));
        hipLaunchKernelGGL(slab_pack_bwd_kernel, lin3, dim3(lb), 0, h->stream, g, cx, P, ny, k0, nzc,
                           reinterpret_cast<const double2 *>(h->specB), reinterpret_cast<double2 *>(h->a2a_send + chunk * c));
      }
      if (exchange(c, true)) return 1;
    }
    HIP_OK(hipGetLastError());
  }
  {
    PROF(h, "unpack_fftx_bwd");
    for (int c = 0; c < nch; ++c) {
      const int k0 = c * nzc;
      if (!self) HIP_OK(hipStreamWaitEvent(h->stream, h->ev_done[c], 0));
      if (h->fft_fused) {
        if (fft_x_bwd_unpack(h, k0, nzc, rbuf + chunkB * c, pg)) return 1;
        continue;
      }
      hipLaunchKernelGGL(slab_unpack_bwd_kernel, tg, tb, 0, h->stream, g, nkx, pitch, cx, P, k0, nzc,
                         reinterpret_cast<const double2 *>(rbuf + chunk * c), reinterpret_cast<double2 *>(h->specA));
      void *in[1] = {specA_at(k0)}, *out[1] = {prow0 + g.sz * k0};
      FFT_OK(rocfft_execute(h->plan_xb, in, out, h->info_x));
    }
    HIP_OK(hipGetLastError());
  }
  return 0;
}

void pois_destroy(udc_handle *h) {
  if (h->plan_fwd) rocfft_plan_destroy(h->plan_fwd);
  if (h->plan_bwd) rocfft_plan_destroy(h->plan_bwd);
  if (h->info_fwd) rocfft_execution_info_destroy(h->info_fwd);
  if (h->info_bwd) rocfft_execution_info_destroy(h->info_bwd);
  if (h->fft_work) hipFree(h->fft_work);
  if (h->spec) hipFree(h->spec);
  if (h->rbuf) hipFree(h->rbuf);
  if (h->ztab) hipFree(h->ztab);
  if (h->outlet_w) hipFree(h->outlet_w);
  if (h->ev) hipFree(h->ev);
  if (h->tri) hipFree(h->tri);
  if (h->partials) hipFree(h->partials);
  rocfft_plan sp[4] = {h->plan_xf, h->plan_xb, h->plan_yf, h->plan_yb};
  for (auto pl : sp) if (pl) rocfft_plan_destroy(pl);
  if (h->info_x) rocfft_execution_info_destroy(h->info_x);
  if (h->comm_stream) hipStreamDestroy(h->comm_stream);
  if (h->ev_halo_ready) hipEventDestroy(h->ev_halo_ready);
  if (h->ev_halo_done) hipEventDestroy(h->ev_halo_done);
  for (int c = 0; c < 16; ++c) { if (h->ev_ready[c]) hipEventDestroy(h->ev_ready[c]); if (h->ev_done[c]) hipEventDestroy(h->ev_done[c]); if (h->ev_vp[c]) hipEventDestroy(h->ev_vp[c]); }
  if (h->fft_tw) hipFree(h->fft_tw);
  double *bufs[7] = {h->specA, h->specB, h->a2a_send, h->a2a_recv, h->ev_slab, h->ztab_slab, (double *)h->fft_work_slab};
  for (auto b : bufs) if (b) hipFree(b);
}

int k_divergence_rhs(udc_handle *h, double rk3coef, bool pup) {
  const Geo &g = h->g;
  dim3 b(64, 4, 1), gr = cell_grid(g, b);
  const int lid = h->p.bctopm == UDC_TOP_PRESSURE ? 1 : 0;
  // open x boundaries: straight into the solver's doubled row (k_xo_poisson then skips its own copy)
  double *p2 = (h->xg && h->xpois) ? h->xpois->fields[UDC_P] : nullptr;
  const int sy2 = p2 ? h->xpois->g.sy : 0;
  const long sz2 = p2 ? h->xpois->g.sz : 0;
  PROF(h, "div_rhs");
  if (pup)
    hipLaunchKernelGGL((div_rhs_kernel<true>), gr, b, 0, h->stream, g, tile_grid(g), h->m, 1. / rk3coef,
                       h->fields[UDC_UP], h->fields[UDC_VP], h->fields[UDC_WP], h->fields[UDC_UM],
                       h->fields[UDC_VM], h->fields[UDC_WM], h->fields[UDC_P], lid, p2, sy2, sz2);
  else
    hipLaunchKernelGGL((div_rhs_kernel<false>), gr, b, 0, h->stream, g, tile_grid(g), h->m, 1. / rk3coef,
                       h->fields[UDC_UP], h->fields[UDC_VP], h->fields[UDC_WP], h->fields[UDC_UM],
                       h->fields[UDC_VM], h->fields[UDC_WM], h->fields[UDC_P], lid, p2, sy2, sz2);
  HIP_OK(hipGetLastError());
  h->xo_rhs_mirrored = p2 != nullptr;
  return 0;
}

int k_poisson_solve(udc_handle *h) {
  if (h->xg) return k_xo_poisson(h);      // open x boundaries: on the mirrored row (udc_xopen.hip)
  if (h->slab) return k_poisson_solve_slab(h);
  const Geo &g = h->g;
  const long nmodes = (long)h->nkxp * g.ny;
  double *pin = h->fields[UDC_P] + g.idx(0, 0, 0);
  dim3 b(64, 4, 1), gr = cell_grid(g, b);
  if (h->fwd_compact) {
    PROF(h, "fft_pack");
    hipLaunchKernelGGL((real_copy_kernel<false>), gr, b, 0, h->stream, g, tile_grid(g), h->fields[UDC_P], h->rbuf);
  }
  if (h->div_in_fft) {
    if (fft_nat_forward(h)) return 1;
  } else {
    PROF(h, "fft_fwd");
    void *in[1] = {h->fwd_compact ? h->rbuf : pin}, *out[1] = {h->spec};
    FFT_OK(rocfft_execute(h->plan_fwd, in, out, h->info_fwd));
  }
  {
    PROF(h, "thomas");
    // (rows of nkxp complex with nkxp a multiple of eight: the mirrored rows ky, ny - ky are solved together)
    const bool pairable = h->nkxp % ZB == 0 && g.ny % 2 == 0;
    if (launch_thomas(h, h->thomas_lds, nmodes, g.nz, 1. / ((double)g.nx * (double)g.ny), h->ev, h->ztab,
                      reinterpret_cast<double2 *>(h->spec), pairable ? h->nkxp / ZB : 0, pairable ? g.ny : 0)) return 1;
    if (h->bczp == 2)      // spec[k][ky = 0][kx = 0]
      hipLaunchKernelGGL(zero_mode_mean_kernel, dim3(1), dim3(256), 0, h->stream, g.nz, (size_t)nmodes, reinterpret_cast<double2 *>(h->spec));
    HIP_OK(hipGetLastError());
  }
  {
    PROF(h, "fft_bwd");
    void *in[1] = {h->spec}, *out[1] = {h->bwd_compact ? h->rbuf : pin};
    FFT_OK(rocfft_execute(h->plan_bwd, in, out, h->info_bwd));
  }
  if (h->bwd_compact) {
    PROF(h, "fft_unpack");
    hipLaunchKernelGGL((real_copy_kernel<true>), gr, b, 0, h->stream, g, tile_grid(g), h->fields[UDC_P], h->rbuf);
  }
  HIP_OK(hipGetLastError());
  return 0;
}

int k_project(udc_handle *h) {
  const Geo &g = h->g;
  dim3 b(64, 4, 1), gr = cell_grid(g, b);
  PROF(h, "project");
  hipLaunchKernelGGL(project_kernel, gr, b, 0, h->stream, g, tile_grid(g), h->m, h->fields[UDC_P], h->fields[UDC_UP],
                     h->fields[UDC_VP], h->fields[UDC_WP], h->fields[UDC_PRES0]);
  HIP_OK(hipGetLastError());
  return 0;
}

// the open lid's three steps (kernels above).  cnt: the fluid c cells of level ke in the whole domain
static double lid_count(const udc_handle *h) {
  // (IIc is all ones where the c grid's lists were not read -- no scalar field at all, src/modibm.f90:181)
  const bool masked = h->ibm_on && h->ibm[3].given && (int)h->ibm[3].fluid_cnt.size() > h->g.nz;
  return masked ? h->ibm[3].fluid_cnt[h->g.nz] : (double)(h->g.nx - 2 * h->g.xg) * (double)h->cfg.jtot;
}
// the slab mean the lid's rows take is a masked one (solid c cells in level ke: avexy_ibm's fluid-cell mean is then not the zero mode's)
bool k_lid_masked(const udc_handle *h) {
  return lid_count(h) != (double)(h->g.nx - 2 * h->g.xg) * (double)h->cfg.jtot;
}
int k_lid_bcpup(udc_handle *h, double rk3coef, bool pup, bool ptotal) {
  const Geo &g = h->g;
  if (ptotal && !pup) { udc_set_error("k_lid_bcpup: the pressure-total form needs the predicted-velocity form of the tendencies"); return 1; }
  if (!ptotal && k_level_sums_dev(h, UDC_PRES0, 1, g.nz - 1)) return 1;
  PROF(h, "lid");
  hipLaunchKernelGGL(lid_bcpup_kernel, dim3((g.nx + 63) / 64, g.ny), dim3(64), 0, h->stream, g, 1. / rk3coef, h->dzhi_top, lid_count(h),
                     (const double *)(ptotal ? nullptr : h->lev_sum16), (const double *)h->fields[UDC_WM], h->fields[UDC_WP], pup ? 1 : 0);
  HIP_OK(hipGetLastError());
  return 0;
}
int k_lid_tderive(udc_handle *h) {
  const Geo &g = h->g;
  if (k_level_sums_dev(h, UDC_P, 1, g.nz - 1)) return 1;
  PROF(h, "lid");
  hipLaunchKernelGGL(lid_tderive_kernel, dim3((g.nx + 63) / 64, g.ny), dim3(64), 0, h->stream, g, h->dzhi_top, lid_count(h),
                     (const double *)h->lev_sum16, h->fields[UDC_WP]);
  HIP_OK(hipGetLastError());
  return 0;
}
int k_lid_integrate(udc_handle *h, int rk3step, double dt, bool pup, bool zero, bool wrap) {
  const Geo &g = h->g;
  PROF(h, "lid");
  hipLaunchKernelGGL(lid_integrate_kernel, dim3((g.nx + 63) / 64, g.ny), dim3(64), 0, h->stream, g, dt / (4. - (double)rk3step), pup ? 1 : 0,
                     zero ? 1 : 0, rk3step == 3 ? 1 : 0, wrap ? 1 : 0, (const double *)h->fields[UDC_WM], h->fields[UDC_WP], h->fields[UDC_W0],
                     h->fields[UDC_WM]);
  HIP_OK(hipGetLastError());
  return 0;
}

static IntArgs int_args(udc_handle *h) {
  IntArgs a;
  a.u0 = h->fields[UDC_U0]; a.v0 = h->fields[UDC_V0]; a.w0 = h->fields[UDC_W0];
  a.um = h->fields[UDC_UM]; a.vm = h->fields[UDC_VM]; a.wm = h->fields[UDC_WM];
  a.up = h->fields[UDC_UP]; a.vp = h->fields[UDC_VP]; a.wp = h->fields[UDC_WP];
  a.nsv = 0;
  for (int n : h->slots) {
    if (h->sv_inline[n]) continue;      // (updated inside its own sweep, udc_scalar_lds.hip)
    a.sv0[a.nsv] = h->fields[UDC_SV0 + 3 * n];
    a.svm[a.nsv] = h->fields[UDC_SVM + 3 * n];
    a.svp[a.nsv] = h->fields[UDC_SVP + 3 * n];
    a.clip[a.nsv] = h->slot[n].tke ? h->tke.e12min : -1.;
    ++a.nsv;
  }
  return a;
}

// RK stage 3: the reference's `um = u0`, `thlm = thl0`, `svm = sv0` ... are whole-array assignments (src/modtstep.f90:331-339),
// ghost planes included; the kernels above update the cells (and, folded, the lateral / top ghosts `halos` and `boundary` own).
// The planes below the floor belong to no boundary routine -- they keep what the start-up put there -- so the m-fields take
// them from the 0-fields here (statsdump reads thlm(kb-1), src/modstatsdump.f90:861).
static int copy_floor_planes(udc_handle *h, bool vel) {
  const size_t bytes = sizeof(double) * (size_t)HZ * (size_t)h->g.sz;
  if (vel)
    for (int q = 0; q < 3; ++q)
      HIP_OK(hipMemcpyAsync(h->fields[UDC_UM + q], h->fields[UDC_U0 + q], bytes, hipMemcpyDeviceToDevice, h->stream));
  for (int n : h->slots)
    HIP_OK(hipMemcpyAsync(h->fields[UDC_SVM + 3 * n], h->fields[UDC_SV0 + 3 * n], bytes, hipMemcpyDeviceToDevice, h->stream));
  return 0;
}

int k_integrate(udc_handle *h, int rk3step, double dt) {
  const Geo &g = h->g;
  dim3 b(64, 4, 1), gr = cell_grid(g, b);
  const double rk3coef = dt / (4. - (double)rk3step);
  PROF(h, "integrate");
  hipLaunchKernelGGL((integrate_kernel<false, true, false>), gr, b, 0, h->stream, g, tile_grid(g), h->m, int_args(h),
                     (const double *)nullptr, (double *)nullptr, rk3coef, rk3step == 3 ? 3 : 0, 0, h->p);
  HIP_OK(hipGetLastError());
  if (rk3step == 3 && copy_floor_planes(h, true)) return 1;
  return 0;
}

int k_project_integrate(udc_handle *h, int rk3step, double dt, bool zero_tend, bool pup, bool ghosts,
                        bool write_um, bool out_to_um, int rows, int r0, int r1, bool ptotal) {
  const Geo &g = h->g;
  const dim3 b(64, 4, 1);
  // rows 1 / 2: the tile rows next to the neighbouring ranks / the rows in between (the caller exchanges ghost rows in between);
  // 3 / 4: the tile rows [r0, r1)
  const TileGrid tg = rows == 0 ? tile_grid(g) : (rows >= 3 ? tile_range(tile_grid(g), r0, r1) : tile_rows(tile_grid(g), 1, rows == 1));
  const dim3 gr((unsigned)tg.tiles * (unsigned)g.nz, 1, 1);
  const double rk3coef = dt / (4. - (double)rk3step);
  PROF(h, (rows == 1 || rows == 3) ? "project_integrate_edge" : "project_integrate");
  const int lastf = rk3step == 3 ? (write_um ? 3 : 2) : 0;
  IntArgs ia = int_args(h);
  if (out_to_um) { ia.u0 = ia.um; ia.v0 = ia.vm; ia.w0 = ia.wm; }   // pointer rotation at RK stage 1 (um_alias)
  if (ptotal && !pup) { udc_set_error("k_project_integrate: the pressure-total form needs the predicted-velocity form of the tendencies"); return 1; }
  if (ptotal)      // (the caller swaps p and pres0 once every row is integrated)
    hipLaunchKernelGGL((integrate_kernel<true, false, true, true>), gr, b, 0, h->stream, g, tg, h->m, ia,
                       (const double *)h->fields[UDC_P], h->fields[UDC_P], rk3coef, lastf, ghosts ? 1 : 0, h->p);
  else if (pup)
    hipLaunchKernelGGL((integrate_kernel<true, false, true>), gr, b, 0, h->stream, g, tg, h->m, ia,
                       (const double *)h->fields[UDC_P], h->fields[UDC_PRES0], rk3coef, lastf, ghosts ? 1 : 0, h->p);
  else if (zero_tend)
    hipLaunchKernelGGL((integrate_kernel<true, true, false>), gr, b, 0, h->stream, g, tg, h->m, ia,
                       (const double *)h->fields[UDC_P], h->fields[UDC_PRES0], rk3coef, lastf, ghosts ? 1 : 0, h->p);
  else
    hipLaunchKernelGGL((integrate_kernel<true, false, false>), gr, b, 0, h->stream, g, tg, h->m, ia,
                       (const double *)h->fields[UDC_P], h->fields[UDC_PRES0], rk3coef, lastf, ghosts ? 1 : 0, h->p);
  HIP_OK(hipGetLastError());
  if (rk3step == 3 && (rows == 0 || rows == 1) && copy_floor_planes(h, write_um)) return 1;      // (whole planes below the floor: once)
  return 0;
}

static int ensure_partials(udc_handle *h, size_t nblocks) {
  if (h->partials_cap >= nblocks) return 0;
  if (h->partials) HIP_OK(hipFree(h->partials));
  HIP_OK(hipMalloc(&h->partials, sizeof(double) * 2 * nblocks));
  h->partials_cap = nblocks;
  return 0;
}

// masscorr: pup_mode = the UP/VP arrays hold pup = up + um/rk3coef (fused substep), so rk3coef <pup> is the
// predicted flow rate and um is not read.  Sums are per-slab, then all-reduced (avexy_ibm's MPI_ALLREDUCE).
int k_masscorr(udc_handle *h, double rk3coef, bool pup_mode, bool wrap_vp) {
  if (!h->luvolflowr && !h->lvvolflowr) return 0;
  // inflow / outflow in x: every branch of the reference's masscorr sits under `.not. linoutflow` (src/modforces.f90:352, 393, 424, 467);
  // a prescribed volume flow then only names the outlet's speed (uouttot = ubulk, src/modboundary.f90:158-160: udc_set_open_x_outflow)
  if (h->xg) return 0;
  const Geo &g = h->g;
  // luoutflowr (h->luvolflowr == 2): the flow through the outlet plane i = ie over its fluid u points, per area of its fluid c cells
  // (uoutletarea, src/modforces.f90:499-522: sum of IIc(ie, j, k) dy dzf(k); all cells count when no c-grid lists were read)
  const bool outlet = h->luvolflowr == 2;
  if (outlet && !h->outlet_w) {
    std::vector<double> cnt(g.nz + 2, (double)h->cfg.jtot), w(g.nz + 2, 0.);
    if (h->ibm_on && h->ibm[3].given)
      for (size_t q = 0; q < h->ibm[3].solid_g.size() / 3; ++q)
        if (h->ibm[3].solid_g[3 * q] == g.nx) cnt[h->ibm[3].solid_g[3 * q + 2]] -= 1.;
    double area = 0.;
    for (int k = 1; k <= g.nz; ++k) area += cnt[k] * h->cfg.dy * h->cfg.dzf[k];
    if (!(area > 0.)) { udc_set_error("masscorr (luoutflowr): the outlet plane i = itot has no fluid cell"); return 1; }
    for (int k = 1; k <= g.nz; ++k) w[k] = h->cfg.dy * h->cfg.dzf[k] / area;
    HIP_OK(hipMalloc(&h->outlet_w, sizeof(double) * w.size()));
    HIP_OK(hipMemcpy(h->outlet_w, w.data(), sizeof(double) * w.size(), hipMemcpyHostToDevice));
  }
  dim3 b(64, 4, 1), gr = cell_grid(g, b);
  if (ensure_partials(h, gr.x)) return 1;
  PROF(h, "masscorr");
  const double vol = (double)(g.nx - 2 * g.xg) * (double)h->cfg.jtot * h->zsize;     // IIus(k) = itot*jtot cells per level, zh(ke+1)
  double *S = h->red + 16;
  FlowShift fu{nullptr, S, 0., 0., 0.}, fv{nullptr, S + 2, 0., 0., 0.};
  const int mo = h->um_alias ? UDC_U0 : UDC_UM;
  // immersed boundary: avexy_ibm's level averages run over the fluid cells (IIu / IIv, src/modforces.f90:404-405), so the
  // weight of a cell becomes dzf(k) / IIus(k) / zh(ke+1) and the listed solid points are taken back out of the sum
  if (h->ibm_on && !h->ibm_wlev) {
    std::vector<double> w(2 * (g.nz + 2), 0.);
    for (int q = 0; q < 2; ++q)
      for (int k = 1; k <= g.nz; ++k) {
        const double c = h->ibm[q].fluid_cnt[k];
        w[q * (g.nz + 2) + k] = c > 0. ? h->cfg.dzf[k] / c / h->zsize : 0.;
      }
    HIP_OK(hipMalloc(&h->ibm_wlev, sizeof(double) * w.size()));
    HIP_OK(hipMemcpy(h->ibm_wlev, w.data(), sizeof(double) * w.size(), hipMemcpyHostToDevice));
  }
  for (int c = 0; c < 2; ++c) {
    if (!(c == 0 ? h->luvolflowr : h->lvvolflowr)) continue;
    const double *a = h->fields[UDC_UP + c], *bm = pup_mode ? nullptr : h->fields[mo + c];
    const bool out_c = outlet && c == 0;
    const double *wlev = out_c ? h->outlet_w : (h->ibm_on ? h->ibm_wlev + (size_t)c * (g.nz + 2) : h->m.dzf);
    const int only_i = out_c ? g.nx - 1 : -1;
    hipLaunchKernelGGL(flowsum_kernel, gr, b, 0, h->stream, g, tile_grid(g), wlev, a, bm, h->partials, only_i);
    hipLaunchKernelGGL((reduce_partials_kernel<1, 1>), dim3(1), dim3(1024), 0, h->stream, h->partials, (long)gr.x, 0., 0.,
                       S + 2 * c);
    if (h->ibm_on && k_ibm_flowsum_correct(h, c, a, bm, wlev, S + 2 * c, only_i)) return 1;
    FlowShift &f = c == 0 ? fu : fv;
    f.f = h->fields[UDC_UP + c];
    f.target = c == 0 ? h->uflowrate : h->vflowrate;
    const bool weighted = h->ibm_on || out_c;      // the weights already carry the normalisation
    f.ca = weighted ? rk3coef : rk3coef / vol;
    f.cb = pup_mode ? 0. : (weighted ? 1. : 1. / vol);
  }
  HIP_OK(hipGetLastError());
  if (comm_allreduce(h, S, 4, 1)) return 1;
  hipLaunchKernelGGL(flowshift_kernel, gr, b, 0, h->stream, g, tile_grid(g), fu, fv, 1. / rk3coef, wrap_vp ? 1 : 0);
  HIP_OK(hipGetLastError());
  return 0;
}

int k_maxima(udc_handle *h, double dt, double *cour, double *diffn, bool checksim) {
  const Geo &g = h->g;
  dim3 b(64, 4, 1), gr = cell_grid(g, b);
  if (ensure_partials(h, gr.x)) return 1;
  const int mo = h->um_alias ? UDC_U0 : UDC_UM;      // aliased: um == u0
  if (checksim)
    hipLaunchKernelGGL(maxima_kernel<true>, gr, b, 0, h->stream, g, tile_grid(g), h->m, dt, h->fields[mo], h->fields[mo + 1],
                       h->fields[mo + 2], h->fields[UDC_EKM], h->fields[UDC_EKH], h->partials);
  else
    hipLaunchKernelGGL(maxima_kernel<false>, gr, b, 0, h->stream, g, tile_grid(g), h->m, dt, h->fields[mo], h->fields[mo + 1],
                       h->fields[mo + 2], h->fields[UDC_EKM], h->fields[UDC_EKH], h->partials);
  // diffnrtotl starts at 1e-5, src/modtstep.f90:115 (checksim's at 0, src/modchecksim.f90:141)
  hipLaunchKernelGGL((reduce_partials_kernel<0, 0>), dim3(1), dim3(1024), 0, h->stream, h->partials, (long)gr.x, 0.,
                     checksim ? 0. : 1e-5, h->red);
  HIP_OK(hipGetLastError());
  if (comm_allreduce(h, h->red, 2, 0)) return 1;     // MPI_ALLREDUCE(MAX), src/modtstep.f90:131-132
  HIP_OK(hipMemcpyAsync(h->red_host, h->red, 2 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIP_OK(hipStreamSynchronize(h->stream));
  *cour = h->red_host[0];
  *diffn = h->red_host[1];
  return 0;
}

// checksim, asynchronous: the four numbers are computed and copied into pinned memory behind whatever the stream holds; the
// caller picks them up later (k_checksim_end) -- so the time loop's only per-step wait disappears from runs that report every step
int k_checksim_begin(udc_handle *h, double dtmn) {
  const Geo &g = h->g;
  dim3 b(64, 4, 1), gr = cell_grid(g, b);
  if (ensure_partials(h, 2 * (size_t)gr.x)) return 1;
  if (!h->chk_host) {
    HIP_OK(hipHostMalloc(&h->chk_host, 4 * sizeof(double)));
    HIP_OK(hipMalloc(&h->chk_dev, 4 * sizeof(double)));
    HIP_OK(hipEventCreateWithFlags(&h->ev_chk, hipEventDisableTiming));
  }
  if (h->um_alias) {
    hipLaunchKernelGGL(checksim_kernel, gr, b, 0, h->stream, g, tile_grid(g), h->m, dtmn, h->fields[UDC_U0], h->fields[UDC_V0],
                       h->fields[UDC_W0], h->fields[UDC_EKM], h->fields[UDC_EKH], h->partials);
  } else {
    hipLaunchKernelGGL(maxima_kernel<true>, gr, b, 0, h->stream, g, tile_grid(g), h->m, dtmn, h->fields[UDC_UM], h->fields[UDC_VM],
                       h->fields[UDC_WM], h->fields[UDC_EKM], h->fields[UDC_EKH], h->partials);
    hipLaunchKernelGGL(divcheck_kernel, gr, b, 0, h->stream, g, tile_grid(g), h->m, h->fields[UDC_U0], h->fields[UDC_V0],
                       h->fields[UDC_W0], h->partials + 2 * (size_t)gr.x);
  }
  hipLaunchKernelGGL((reduce_partials_kernel<0, 0>), dim3(1), dim3(1024), 0, h->stream, h->partials, (long)gr.x, 0., 0., h->chk_dev);
  hipLaunchKernelGGL((reduce_partials_kernel<0, 1>), dim3(1), dim3(1024), 0, h->stream, h->partials + 2 * (size_t)gr.x, (long)gr.x, 0., 0.,
                     h->chk_dev + 2);
  HIP_OK(hipGetLastError());
  if (comm_allreduce(h, h->chk_dev, 3, 0)) return 1;      // courant, diffusion number, divmax: MPI_MAX
  if (comm_allreduce(h, h->chk_dev + 3, 1, 1)) return 1;  // divtot: MPI_SUM (src/modchecksim.f90:193-196)
  HIP_OK(hipMemcpyAsync(h->chk_host, h->chk_dev, 4 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIP_OK(hipEventRecord(h->ev_chk, h->stream));
  h->chk_pending = true;
  return 0;
}
int k_checksim_end(udc_handle *h, double out[4]) {
  if (!h->chk_pending) { udc_set_error("udc_checksim_end: no report pending"); return 1; }
  HIP_OK(hipEventSynchronize(h->ev_chk));
  for (int q = 0; q < 4; ++q) out[q] = h->chk_host[q];
  h->chk_pending = false;
  return 0;
}

int k_divergence_check(udc_handle *h, double *divmax, double *divtot) {
  const Geo &g = h->g;
  dim3 b(64, 4, 1), gr = cell_grid(g, b);
  if (ensure_partials(h, gr.x)) return 1;
  hipLaunchKernelGGL(divcheck_kernel, gr, b, 0, h->stream, g, tile_grid(g), h->m, h->fields[UDC_U0], h->fields[UDC_V0],
                     h->fields[UDC_W0], h->partials);
  hipLaunchKernelGGL((reduce_partials_kernel<0, 1>), dim3(1), dim3(1024), 0, h->stream, h->partials, (long)gr.x, 0.,
                     0., h->red);
  HIP_OK(hipGetLastError());
  if (comm_allreduce(h, h->red, 1, 0)) return 1;         // divmax: MPI_MAX
  if (comm_allreduce(h, h->red + 1, 1, 1)) return 1;     // divtot: MPI_SUM (src/modchecksim.f90:193-196)
  HIP_OK(hipMemcpyAsync(h->red_host, h->red, 2 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIP_OK(hipStreamSynchronize(h->stream));
  *divmax = h->red_host[0];
  *divtot = h->red_host[1];
  return 0;
}
