// Line FFTs of the slab (multi-GPU) Poisson solve with the all-to-all packing fused in.
//
// The slab path (udc_pois.hip, k_poisson_solve_slab) transforms x on the y-slabs, exchanges blocks [d][k][kx_l][j]
// with every rank, transforms y on the kx-slabs, and back.  With rocFFT each of the four transforms is followed /
// preceded by a separate transpose kernel (4 extra passes over the spectral data).  Here a workgroup holds whole
// lines in LDS -- L rows of x for the x transforms, C columns of y for the y transforms -- runs a Stockham autosort
// FFT on them (radix 4, one radix-2 stage when log2 is odd) and reads / writes the exchange buffers directly:
//
//   fftx_fwd_pack    p rows (real, coalesced)          -> R2C  -> send[d][k][kx_l][j]   (runs of L in j)
//   ffty_fwd_unpack  recv[s][k][kx_l][j] (runs of ny_l) -> C2C  -> specB[k][kx_l][y]     (what the Thomas kernel reads)
//   ffty_bwd_pack    specB[k][kx_l][y]                  -> C2C^-1 -> send[d][k][kx_l][j]
//   fftx_bwd_r8      recv[s][k][kx_l][j] (runs of L)    -> C2R  -> p rows      (radix-8 butterflies in registers, persistent)
//
// R2C / C2R of length N run as a complex transform of length M = N/2 on z[n] = x[2n] + i x[2n+1] with the usual
// split / merge step.  All transforms are unnormalised, like rocFFT's and FFTW's (the Thomas kernel carries 1/(nx ny)).
// Power-of-two lengths only (16 <= nx <= 2048, 8 <= ny <= 1024, power-of-two local rows >= 4); anything else keeps the
// rocFFT path.
//
// One GPU (no exchange): the forward half of the solve also runs here, in rocFFT's own spectral layout spec[k][j][kx] --
//   fftx_fwd_nat_kernel   divergence of (pup, pvp, pwp) -> R2C -> rows of spec            (div_rhs + rocFFT's x pass in one sweep)
//   ffty_natreg_kernel    columns of spec, in place, 16 x N2 in registers with one trip through LDS   (ny = 128, 256, 512)
// and the slab path's y transforms use the same register scheme on their y-contiguous lines (ffty_slabreg_kernel).
#include "udc_internal.h"
#include <algorithm>
#include <cmath>
#include <cstdlib>

namespace {

constexpr int FT = 256;      // threads per workgroup of the y kernels; the x kernels take theirs from the line length (xthreads)
// Measured at 1024 x 512 x 512 (x lines of 512 complex, four per workgroup; profiles/r03/fft_threads_scan.txt): 256 / 512 / 1024
// threads: x forward 2.39 / 2.08 / 1.79 ms, x backward 2.07 / 1.41 / 1.23 ms -- the five stages of a line are separated by
// barriers and only a fat workgroup keeps enough of them in flight; the y kernels (lines of 512, eight per workgroup) go the
// other way, 1.00 / 1.10 / 1.65 ms.
#ifndef UDC_XT_LM8
#define UDC_XT_LM8 512
#endif
__host__ __device__ constexpr int xthreads(int LM) { return LM >= 9 ? 1024 : (LM == 8 ? UDC_XT_LM8 : 256); }

__device__ __forceinline__ double2 cmul(double2 a, double2 b) {
  return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ double2 cadd(double2 a, double2 b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ double2 csub(double2 a, double2 b) { return make_double2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ double2 cconj(double2 a) { return make_double2(a.x, -a.y); }

// LDS index of element n of a line: one pad element per 16 so that the strided accesses of the radix-4 stages (stride 4
// and 16 elements between neighbouring lanes) spread over the banks
__device__ __forceinline__ int pad(int n) { return n + (n >> 4); }
__host__ __device__ constexpr int padded(int n) { return n + (n >> 4) + 1; }

// Stockham autosort FFT of 1 << LNL lines of M = 1 << LM complex each, held in LDS at a[line * MP + pad(n)]; b is the
// other half of the ping-pong.  tw[n] = exp(-2 pi i n / M), n < M, in LDS; INV conjugates it.  Returns the buffer
// holding the result.  All NTH threads of the workgroup must call it.  Everything is a shift or a mask.
template <bool INV, int LM, int NTH = FT>
__device__ __forceinline__ double2 *fft_lines(double2 *a, double2 *b, const double2 *tw, int MP, int nl) {
  constexpr int M = 1 << LM;
  const int tid = threadIdx.x;
#pragma unroll
  for (int lp = 0; lp < LM;) {
    const bool r4 = LM - lp >= 2;
    const int lR = r4 ? 2 : 1, lT = LM - lR, T = 1 << lT, p = 1 << lp, lstep = LM - lp - lR;
    const int work = nl << lT;
    for (int wi = tid; wi < work; wi += NTH) {
      const int line = wi >> lT, j = wi & (T - 1);
      const int k = j & (p - 1);
      const double2 *src = a + line * MP;
      double2 *dst = b + line * MP;
      const int o = ((j - k) << lR) + k;
      if (r4) {
        double2 u0 = src[pad(j)], u1 = src[pad(j + T)], u2 = src[pad(j + 2 * T)], u3 = src[pad(j + 3 * T)];
        if (lp > 0) {
          const int t1 = k << lstep;
          double2 w1 = tw[t1], w2 = tw[2 * t1], w3 = tw[3 * t1];
          if (INV) { w1.y = -w1.y; w2.y = -w2.y; w3.y = -w3.y; }
          u1 = cmul(u1, w1); u2 = cmul(u2, w2); u3 = cmul(u3, w3);
        }
        const double2 t0 = cadd(u0, u2), t1_ = csub(u0, u2), t2 = cadd(u1, u3);
        const double2 d = csub(u1, u3);
        const double2 t3 = INV ? make_double2(-d.y, d.x) : make_double2(d.y, -d.x);      // (u1 - u3) * (-/+ i)
        dst[pad(o)] = cadd(t0, t2);
        dst[pad(o + p)] = cadd(t1_, t3);
        dst[pad(o + 2 * p)] = csub(t0, t2);
        dst[pad(o + 3 * p)] = csub(t1_, t3);
      } else {
        double2 u0 = src[pad(j)], u1 = src[pad(j + T)];
        if (lp > 0) {
          double2 w1 = tw[k << lstep];
          if (INV) w1.y = -w1.y;
          u1 = cmul(u1, w1);
        }
        dst[pad(o)] = cadd(u0, u1);
        dst[pad(o + p)] = csub(u0, u1);
      }
    }
    __syncthreads();
    double2 *t = a; a = b; b = t;
    lp += lR;
  }
  (void)M;
  return a;
}

struct XArgs {
  int nx, M, MP;            // real length, complex length nx/2, LDS pitch of a line
  int nyl, py;              // local rows, padded rows per plane of the real field
  int sy;                   // row stride of the real field (doubles)
  long sz;                  // plane stride of the real field (doubles)
  int nkx, cx, P;           // r2c modes nx/2+1, modes per rank, ranks
  int k0, nzc;              // chunk
  int lL;                   // log2 of the rows per workgroup
  int jg0;                  // first row group of this launch (fftx_fwd_pack: the last group may follow vp's ghost row)
  unsigned cxmul;           // floor(2^32 / cx) + 1: kx / cx = __umulhi(kx, cxmul) for every mode index (r8 kernels)
  // backward blocks (fftx_bwd_r8): elements per source block; ghost != 0: behind the nzc cx nyl elements of a block's own rows sit
  // [kc][kxl][2] -- the rows just outside this slab (j = -1 from the previous rank's last row, j = nyl from the next rank's first),
  // which the y pass put there (p's ghost rows ride in the transpose: no exchange of their own)
  long bstride;
  int ghost;
};

// LDS: two line buffers [L][MP], then the twiddles of the length-M transform [M], then the rank of every mode [cx*P]
template <int LM>
__device__ __forceinline__ void x_lds(const XArgs &q, double2 *lds, double2 *&a, double2 *&b, double2 *&tw, int *&dmap,
                                      const double2 *__restrict__ twM) {
  const int L = 1 << q.lL;
  a = lds; b = lds + L * q.MP; tw = b + L * q.MP;
  dmap = reinterpret_cast<int *>(tw + (1 << LM));
  constexpr int NTH = xthreads(LM);
  for (int n = threadIdx.x; n < (1 << LM); n += NTH) tw[n] = twM[n];
  for (int kx = threadIdx.x; kx < q.cx * q.P; kx += NTH) dmap[kx] = kx / q.cx;
}

// fillps folded into the x forward transform (fused substep, PUP mode): the row is not read from p but evaluated as
// the divergence of (pup, pvp, pwp) (src/modpois.f90:968-970; pwp(ke+1) = 0 of bcpup), which then never exists in memory
// nz: the level from which pwp(k+1) is bcpup's zero -- ktot under a closed lid, ktot + 1 (never) under the open one, whose row bcpup fills
// xo_n > 0 (fftx_fwd_nat_kernel<LM, true>): the arrays are those of a handle with open x boundaries (strides sy, sz; udc_xopen.hip) and the row
// transformed is the divergence of its xo_n interior columns, from column xo_g on, followed by its mirror image -- what div_rhs_kernel wrote
// into the solver's doubled row, 40 B per cell that now stay in LDS
struct DivArgs { const double *pu, *pv, *pw; const double *dzfi; double dxi, dyi; int nz; int xo_n = 0, xo_g = 0, sy = 0; long sz = 0; };

// x forward: rows j0..j0+L-1 of plane k0+kc -> send blocks
template <int LM, bool DIV>
__global__ __launch_bounds__(xthreads(LM)) void fftx_fwd_pack_kernel(XArgs q, const double *__restrict__ p, DivArgs dv, const double2 *__restrict__ twM,
                                                           const double2 *__restrict__ twN, double2 *__restrict__ send) {
  extern __shared__ double2 lds[];
  constexpr int M = 1 << LM, NTH = xthreads(LM);
  double2 *a, *b, *tw; int *dmap;
  x_lds<LM>(q, lds, a, b, tw, dmap, twM);
  const int tid = threadIdx.x, L = 1 << q.lL;
  // (the launch order as it is: dealing the XCDs contiguous runs, as the backward kernel does, halves this kernel's partial-line
  // writes in the counters but costs it 5 % -- profiles/r05/same_box_ab_r04_r05.txt)
  const int j0 = (blockIdx.x + q.jg0) << q.lL, kc = blockIdx.y, k = q.k0 + kc;
  // load: row l holds M complex = nx reals, read as double2 (16-B aligned: nx even, rows nx*8 B apart, base 16-B aligned)
  for (int wi = tid; wi < (M << q.lL); wi += NTH) {
    const int l = wi >> LM, n = wi & (M - 1);
    const long ro = q.sz * (long)(k + HZ) + (long)q.sy * (j0 + l + HY);
    if (DIV) {
      const double2 *ru = reinterpret_cast<const double2 *>(dv.pu + ro), *rv = reinterpret_cast<const double2 *>(dv.pv + ro);
      const double2 *rv1 = reinterpret_cast<const double2 *>(dv.pv + ro + q.sy), *rw = reinterpret_cast<const double2 *>(dv.pw + ro);
      const double2 u = ru[n], un = ru[(n + 1) & (M - 1)], v = rv[n], v1 = rv1[n], w = rw[n];
      double2 w1 = make_double2(0., 0.);
      if (k < dv.nz - 1) w1 = reinterpret_cast<const double2 *>(dv.pw + ro + q.sz)[n];
      const double dz = dv.dzfi[k + 1];
      a[l * q.MP + pad(n)] = make_double2((u.y - u.x) * dv.dxi + (v1.x - v.x) * dv.dyi + (w1.x - w.x) * dz,
                                          (un.x - u.y) * dv.dxi + (v1.y - v.y) * dv.dyi + (w1.y - w.y) * dz);
    } else {
      a[l * q.MP + pad(n)] = reinterpret_cast<const double2 *>(p + ro)[n];
    }
  }
  __syncthreads();
  double2 *z = fft_lines<false, LM, NTH>(a, b, tw, q.MP, L);
  // split: X[kx] = (Z[kx] + conj(Z[M-kx]))/2 - (i/2) e^{-2 pi i kx/N} (Z[kx] - conj(Z[M-kx])), kx = 0..M (Z[M] = Z[0]);
  // written j-fastest: send[((d*nzc + kc)*cx + kxl)*nyl + j]; the padding modes kx >= nkx of the last rank are zero
  const int nk = q.cx * q.P;
  for (int wi = tid; wi < (nk << q.lL); wi += NTH) {
    const int kx = wi >> q.lL, l = wi & (L - 1);
    double2 X = make_double2(0., 0.);
    if (kx < q.nkx) {
      const double2 *zl = z + l * q.MP;
      const double2 zk = zl[pad(kx == M ? 0 : kx)], zc = cconj(zl[pad(kx == 0 ? 0 : M - kx)]);
      const double2 s = cadd(zk, zc), d = csub(zk, zc);
      const double2 w = twN[kx];                              // e^{-2 pi i kx / N}
      const double2 wd = cmul(w, d);                          // -(i/2) w d = (wd.y, -wd.x)/2
      X = make_double2(0.5 * (s.x + wd.y), 0.5 * (s.y - wd.x));
    }
    const int d_ = dmap[kx], kxl = kx - d_ * q.cx;
    send[(((size_t)d_ * q.nzc + kc) * q.cx + kxl) * q.nyl + j0 + l] = X;
  }
}

// ------------------------------------------------------------------------------------------------ radix-8 line transforms (round 5)
// The Stockham kernels above pass a line through LDS once per radix-4 stage (five round trips and seven barriers for 512 complex,
// half of the 1024 threads idle in four of them): measured on the x backward transform of a rank's slab of 1024 x 64 x 512
// (profiles/r05/xbwd_phase_ablation.txt) the stages, the gather and the store each cost about a third and nothing overlaps.  Here a
// thread holds a radix-8 butterfly in registers: M = 2^c 8^a points are a pre-stage of radix 2^c (c = 1, 2; eight neighbouring
// elements per thread) and `a` decimation-in-time stages of radix 8, in place in ONE line buffer (a stage reads and writes the same
// eight slots of the same thread: one barrier between stages), input placed digit-reversed, so that the last stage holds
// X[t + (M/8) q] in thread t: the stores to memory are coalesced straight from registers.  M/8 threads per line.
// LDS slot of element p: p + p/8 + p/64 (the stride-1 stage reads eight neighbours per thread: one pad per eight keeps eight lanes
// on eight different bank groups; the second term does the same for the digit-reversed placement, whose neighbouring inputs stand
// M/8 slots apart), lines LP = M + M/8 + M/64 + 1 apart (an odd number of 16-B slots: the lines of a workgroup start on different banks).
template <int LM> struct R8 {
  static constexpr int M = 1 << LM, A = LM / 3, C = LM % 3, TPL = M / 8, LP = M + M / 8 + M / 64 + 1;
  // slot (before padding) at which input element n has to stand: stage radices R_1 .. R_S = [2^C,] 8 x A, n = r_S + R_S (r_{S-1} + ...),
  // slot = r_1 + R_1 (r_2 + R_2 (...))
  __host__ __device__ static constexpr int place(int n) {
    int p = 0, w = M;
    for (int i = 0; i < A; ++i) { w >>= 3; p += (n & 7) * w; n >>= 3; }
    return p + n;
  }
};
__device__ __forceinline__ int pad8(int p) { return p + (p >> 3) + (p >> 6); }

// eight-point transform in registers, natural order in and out; INV: e^{+2 pi i / 8}
template <bool INV>
__device__ __forceinline__ void dft8(double2 (&x)[8]) {
  constexpr double h = 0.70710678118654752440;
  auto rot = [](double2 v) { return INV ? make_double2(-v.y, v.x) : make_double2(v.y, -v.x); };      // v * (-/+ i)
  const double2 e0 = cadd(x[0], x[4]), e1 = csub(x[0], x[4]), e2 = cadd(x[2], x[6]), e3 = rot(csub(x[2], x[6]));
  const double2 o0 = cadd(x[1], x[5]), o1 = csub(x[1], x[5]), o2 = cadd(x[3], x[7]), o3 = rot(csub(x[3], x[7]));
  const double2 E0 = cadd(e0, e2), E2 = csub(e0, e2), E1 = cadd(e1, e3), E3 = csub(e1, e3);
  const double2 O0 = cadd(o0, o2), O2 = rot(csub(o0, o2));
  const double2 q1 = cadd(o1, o3), q3 = csub(o1, o3);
  // w O1, w^3 O3 with w = (1 -/+ i) / sqrt 2
  const double2 O1 = INV ? make_double2((q1.x - q1.y) * h, (q1.x + q1.y) * h) : make_double2((q1.x + q1.y) * h, (q1.y - q1.x) * h);
  const double2 O3 = INV ? make_double2(-(q3.x + q3.y) * h, (q3.x - q3.y) * h) : make_double2((q3.y - q3.x) * h, -(q3.x + q3.y) * h);
  x[0] = cadd(E0, O0); x[4] = csub(E0, O0);
  x[1] = cadd(E1, O1); x[5] = csub(E1, O1);
  x[2] = cadd(E2, O2); x[6] = csub(E2, O2);
  x[3] = cadd(E3, O3); x[7] = csub(E3, O3);
}

// the stages of one line held at `ln` (slots pad8(p)), thread t of the line's M/8; tw[n] = exp(-2 pi i n / M) in memory.  On return
// x[q] = X[t + (M/8) q].  Every thread of the workgroup must call it (barriers).
template <bool INV, int LM>
__device__ __forceinline__ void r8_stages(double2 *ln, int t, const double2 *tw, double2 (&x)[8]) {
  using G = R8<LM>;
  if (G::C > 0) {                     // pre-stage: 8 / 2^C transforms of radix 2^C on this thread's eight neighbouring slots
#pragma unroll
    for (int r = 0; r < 8; ++r) x[r] = ln[pad8(8 * t) + r];
    if (G::C == 1) {
#pragma unroll
      for (int m = 0; m < 8; m += 2) { const double2 a = x[m], b = x[m + 1]; x[m] = cadd(a, b); x[m + 1] = csub(a, b); }
    } else {
#pragma unroll
      for (int m = 0; m < 8; m += 4) {
        const double2 t0 = cadd(x[m], x[m + 2]), t1 = csub(x[m], x[m + 2]), t2 = cadd(x[m + 1], x[m + 3]), d = csub(x[m + 1], x[m + 3]);
        const double2 t3 = INV ? make_double2(-d.y, d.x) : make_double2(d.y, -d.x);
        x[m] = cadd(t0, t2); x[m + 1] = cadd(t1, t3); x[m + 2] = csub(t0, t2); x[m + 3] = csub(t1, t3);
      }
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) ln[pad8(8 * t) + r] = x[r];
    __syncthreads();
  }
#pragma unroll
  for (int st = 0; st < G::A; ++st) {
    const int lB = G::C + 3 * st, B = 1 << lB;            // slots between the butterfly's elements
    const int j = t & (B - 1), base = ((t >> lB) << (lB + 3)) + j;
#pragma unroll
    for (int r = 0; r < 8; ++r) x[r] = ln[pad8(base + (r << lB))];
    if (lB > 0) {
      const int step = G::M >> (lB + 3);                  // W_{8B}^m = tw[m step]
#pragma unroll
      for (int r = 1; r < 8; ++r) {
        double2 w = tw[r * j * step];
        if (INV) w.y = -w.y;
        x[r] = cmul(x[r], w);
      }
    }
    dft8<INV>(x);
    if (st + 1 < G::A) {
#pragma unroll
      for (int r = 0; r < 8; ++r) ln[pad8(base + (r << lB))] = x[r];
      __syncthreads();
    }
  }
}

// x backward: recv blocks -> rows j0 .. j0+L-1 of plane k0+kc (unnormalised C2R); L M/8 threads.  The merge step of the C2R runs on
// the way in: the thread that fetches X[kx] and X[M - kx] leaves both Z[kx] and Z[M - kx] -- every element of the blocks is read once.
// Workgroups are persistent: the launch holds as many as are resident at once (LDS: four of 256 threads per CU) and each walks
// items (row group, level) vb = blockIdx.x, + gridDim.x, ...; the loads of the next item are issued before the stages of the
// current one.  (All workgroups of a launch start together, and with one item each they also gathered, transformed and stored
// together: the three phases added up -- 0.056 + 0.037 + 0.026 ms of 0.152 at 1024 x 64 x 512, profiles/r05/xbwd_phase_ablation.txt.)
template <int LM>
__global__ __launch_bounds__(512) void fftx_bwd_r8_kernel(XArgs q, const double2 *__restrict__ recv, const double2 *__restrict__ twM,
                                                          const double2 *__restrict__ twN, double *__restrict__ p) {
  using G = R8<LM>;
  constexpr int M = G::M, TPL = G::TPL, LP = G::LP;
  extern __shared__ double2 lds[];
  const int tid = threadIdx.x, L = 1 << q.lL;
  const unsigned gx = (unsigned)(q.nyl >> q.lL), nmain = gx * (unsigned)q.nzc, nitems = nmain + (q.ghost ? (unsigned)q.nzc : 0u), nwg = gridDim.x;
  // item -> (row group, level of the chunk): XCD c (workgroups b % 8 == c) walks a contiguous run of (level, row group) pairs, so
  // that row groups sharing 128-B lines of the blocks (L = 4: runs of 64 B) meet in one L2.  Behind them (q.ghost) one item per level
  // for the two rows just outside the slab: j0 = -1 marks it, lines 0 and 1 are the rows j = -1 and j = nyl (the other lines of the
  // workgroup repeat them and store nothing).
  auto decode = [&](unsigned vb, int &j0, int &kc) {
    if (vb >= nmain) { kc = (int)(vb - nmain); j0 = -1; return; }
    unsigned v = vb;
    if ((nmain & 7u) == 0 && (nwg & 7u) == 0) v = (vb & 7u) * (nmain >> 3) + (vb >> 3);
    kc = (int)(v / gx);
    j0 = (int)(v - (unsigned)kc * gx) << q.lL;
  };
  auto src = [&](int kx, int l, int j0, int kc) {
    const int s_ = (int)__umulhi((unsigned)kx, q.cxmul), kxl = kx - s_ * q.cx;
    const size_t b = (size_t)s_ * q.bstride, col = (size_t)kc * q.cx + kxl;
    return j0 < 0 ? recv[b + (size_t)q.nzc * q.cx * q.nyl + col * 2 + (l & 1)] : recv[b + col * q.nyl + j0 + l];
  };
  // pairs (kx, M - kx), kx = 0 .. M/2 (kx = 0 pairs X[0] with X[M]; kx = M/2 with itself), l fastest: runs of L complex in the blocks
  constexpr int IT = (M / 2) / TPL;      // = 4 passes of all threads, then the L pairs kx = M/2
  const int lg = tid & (L - 1), pi0 = tid >> q.lL;      // gather: line and first pair of this thread
  const int l = tid / TPL, t = tid - l * TPL;           // stages: line and thread of the line
  // the stages' twiddles from LDS and the merge's in registers: a load from memory behind the prefetch below would wait for it
  double2 *tws = lds + (LP << q.lL);
  for (int n = tid; n < M; n += TPL << q.lL) tws[n] = twM[n];
  double2 wn[IT];
#pragma unroll
  for (int it = 0; it < IT; ++it) wn[it] = cconj(twN[pi0 + it * TPL]);
  double2 xa[IT], xb[IT], xh = make_double2(0., 0.);
  auto fetch = [&](unsigned vb) {
    int j0, kc;
    decode(vb, j0, kc);
#pragma unroll
    for (int it = 0; it < IT; ++it) { const int kx = pi0 + it * TPL; xa[it] = src(kx, lg, j0, kc); xb[it] = src(M - kx, lg, j0, kc); }
    if (tid < L) xh = src(M / 2, tid, j0, kc);
  };
  unsigned vb = blockIdx.x;
  if (vb < nitems) fetch(vb);
  for (; vb < nitems; vb += nwg) {
    {
      double2 *ln = lds + lg * LP;
#pragma unroll
      for (int it = 0; it < IT; ++it) {
        const int kx = pi0 + it * TPL;
        // Z[kx] = (X[kx] + conj X[M-kx]) + i e^{+2 pi i kx/N} (X[kx] - conj X[M-kx]);  Z[M-kx]: s -> conj s, i w d -> -conj(i w d)
        const double2 xc = cconj(xb[it]);
        const double2 s_ = cadd(xa[it], xc), d = csub(xa[it], xc);
        const double2 wd = cmul(wn[it], d);                                      // i w d = (-wd.y, wd.x)
        ln[pad8(G::place(kx))] = make_double2(s_.x - wd.y, s_.y + wd.x);
        if (kx > 0) ln[pad8(G::place(M - kx))] = make_double2(s_.x + wd.y, wd.x - s_.y);
      }
      if (tid < L) lds[tid * LP + pad8(G::place(M / 2))] = make_double2(2. * xh.x, -2. * xh.y);      // w = i: Z = 2 conj X
    }
    __syncthreads();
    if (vb + nwg < nitems) fetch(vb + nwg);      // in flight while this item is transformed
    int j0, kc;
    decode(vb, j0, kc);
    double2 x[8];
    r8_stages<true, LM>(lds + l * LP, t, tws, x);
    const int jrow = j0 < 0 ? ((l & 1) ? q.nyl : -1) : j0 + l;
    double2 *row = reinterpret_cast<double2 *>(p + q.sz * (long)(q.k0 + kc + HZ) + (long)q.sy * (jrow + HY));
    if (j0 >= 0 || l < 2) {
#pragma unroll
      for (int r = 0; r < 8; ++r) row[t + r * TPL] = x[r];
    }
    __syncthreads();                             // the last stage's reads are done before the next item is placed
  }
}

struct YArgs {
  int ny, MP;               // line length (global rows), LDS pitch
  int nyl, lnyl, cx, P;     // local rows and their log2
  int k0, nzc;
  int C;                    // columns (kx_l) per workgroup
  long bstride;             // backward pack: elements per destination block (nzc cx nyl, + nzc cx 2 with the ghost rows)
  int ghost;                // backward pack: also leave every slab's edge rows in the neighbouring blocks' ghost slots
};
// element of the ghost slots of block e: row g (0: j = -1, 1: j = nyl) of column kxl at chunk level kc
__device__ __forceinline__ size_t ghost_off(const YArgs &q, int e, int kc, int kxl, int g_) {
  return (size_t)e * q.bstride + (size_t)q.nzc * q.cx * q.nyl + ((size_t)kc * q.cx + kxl) * 2 + g_;
}
// backward pack: where line element y of column kxl goes -- its own block, and (edge rows) a neighbour's ghost slot
template <class F>
__device__ __forceinline__ void bwd_store(const YArgs &q, double2 *__restrict__ send, int kc, int kxl, int y, double2 v, F) {
  const int d_ = y >> q.lnyl, j = y & (q.nyl - 1);
  send[(size_t)d_ * q.bstride + ((size_t)kc * q.cx + kxl) * q.nyl + j] = v;
  if (q.ghost) {
    if (j == q.nyl - 1) send[ghost_off(q, d_ + 1 == q.P ? 0 : d_ + 1, kc, kxl, 0)] = v;
    if (j == 0) send[ghost_off(q, d_ == 0 ? q.P - 1 : d_ - 1, kc, kxl, 1)] = v;
  }
}

// y forward: recv[s][kc][kxl][j] -> C2C -> specB[k][kxl][y]
template <int LM>
__global__ __launch_bounds__(FT) void ffty_fwd_unpack_kernel(YArgs q, const double2 *__restrict__ recv, const double2 *__restrict__ twg,
                                                             double2 *__restrict__ specB) {
  extern __shared__ double2 lds[];
  constexpr int M = 1 << LM;
  double2 *a = lds, *b = lds + q.C * q.MP, *tw = b + q.C * q.MP;
  const int tid = threadIdx.x;
  for (int n = tid; n < M; n += FT) tw[n] = twg[n];
  const int c0 = blockIdx.x * q.C, kc = blockIdx.y, k = q.k0 + kc;
  const int ncol = min(q.C, q.cx - c0);
  for (int wi = tid; wi < (ncol << LM); wi += FT) {
    const int col = wi >> LM, y = wi & (M - 1);
    const int s_ = y >> q.lnyl, j = y & (q.nyl - 1);
    a[col * q.MP + pad(y)] = recv[(((size_t)s_ * q.nzc + kc) * q.cx + c0 + col) * q.nyl + j];
  }
  __syncthreads();
  double2 *z = fft_lines<false, LM>(a, b, tw, q.MP, ncol);
  for (int wi = tid; wi < (ncol << LM); wi += FT) {
    const int col = wi >> LM, y = wi & (M - 1);
    specB[((size_t)k * q.cx + c0 + col) * M + y] = z[col * q.MP + pad(y)];
  }
}

// y backward: specB[k][kxl][y] -> C2C^-1 -> send[d][kc][kxl][j]
template <int LM>
__global__ __launch_bounds__(FT) void ffty_bwd_pack_kernel(YArgs q, const double2 *__restrict__ specB, const double2 *__restrict__ twg,
                                                           double2 *__restrict__ send) {
  extern __shared__ double2 lds[];
  constexpr int M = 1 << LM;
  double2 *a = lds, *b = lds + q.C * q.MP, *tw = b + q.C * q.MP;
  const int tid = threadIdx.x;
  for (int n = tid; n < M; n += FT) tw[n] = twg[n];
  const int c0 = blockIdx.x * q.C, kc = blockIdx.y, k = q.k0 + kc;
  const int ncol = min(q.C, q.cx - c0);
  for (int wi = tid; wi < (ncol << LM); wi += FT) {
    const int col = wi >> LM, y = wi & (M - 1);
    a[col * q.MP + pad(y)] = specB[((size_t)k * q.cx + c0 + col) * M + y];
  }
  __syncthreads();
  double2 *z = fft_lines<true, LM>(a, b, tw, q.MP, ncol);
  for (int wi = tid; wi < (ncol << LM); wi += FT) {
    const int col = wi >> LM, y = wi & (M - 1);
    bwd_store(q, send, kc, c0 + col, y, z[col * q.MP + pad(y)], 0);
  }
}

// ------------------------------------------------------------------------------------------------ one GPU, natural layout
// The single-slab solve keeps rocFFT's spectral layout spec[k][j][kx] (row pitch nkxp) for the Thomas sweep and the backward
// 2-D transform; its forward half can run here instead: the x transform with fillps' divergence folded in (div_rhs and rocFFT's
// x pass in one sweep: 32 B per cell instead of 32 + 16) followed by a y pass over columns of that layout, which rocFFT has no
// single plan for (a strided transform inside two batch dimensions, kx and k).  UDC_OWN_FWD=0 restores div_rhs + rocFFT (A/B switch).
struct NatArgs {
  int nx, M, MP;            // real length, complex length nx/2, LDS pitch of an x line
  int ny, nkx, nkxp;        // rows, r2c modes, row pitch of spec (complex elements)
  int sy; long sz;          // strides of the real fields (doubles)
  int lL;                   // log2 of the rows per workgroup (x kernel)
  int C, YP;                // columns per workgroup and LDS pitch of a column (y kernel)
};

template <int LM, bool XO = false>
__global__ __launch_bounds__(xthreads(LM)) void fftx_fwd_nat_kernel(NatArgs q, DivArgs dv, const double2 *__restrict__ twM,
                                                                    const double2 *__restrict__ twN, double2 *__restrict__ spec) {
  extern __shared__ double2 lds[];
  constexpr int M = 1 << LM, NTH = xthreads(LM);
  const int tid = threadIdx.x, L = 1 << q.lL;
  double2 *a = lds, *b = lds + L * q.MP, *tw = b + L * q.MP;
  for (int n = tid; n < M; n += NTH) tw[n] = twM[n];
  const int j0 = blockIdx.x << q.lL, k = blockIdx.y;
  if (XO) {
    // M pairs per row = 2 xo_n reals: pair nn < M/2 holds the cells 2 nn, 2 nn + 1 of the interior, pair M - 1 - nn their mirror image
    for (int wi = tid; wi < ((M / 2) << q.lL); wi += NTH) {
      const int l = wi >> (LM - 1), nn = wi & (M / 2 - 1);
      const long ro = dv.sz * (long)(k + HZ) + (long)dv.sy * (j0 + l + HY) + dv.xo_g + 2 * nn;
      const double u0 = dv.pu[ro], u1 = dv.pu[ro + 1], u2 = dv.pu[ro + 2];
      const double va = dv.pv[ro], vb = dv.pv[ro + 1], va1 = dv.pv[ro + dv.sy], vb1 = dv.pv[ro + dv.sy + 1];
      const double wa = dv.pw[ro], wb = dv.pw[ro + 1];
      double wa1 = 0., wb1 = 0.;
      if (k < dv.nz - 1) { wa1 = dv.pw[ro + dv.sz]; wb1 = dv.pw[ro + dv.sz + 1]; }
      const double dz = dv.dzfi[k + 1];
      const double d0 = (u1 - u0) * dv.dxi + (va1 - va) * dv.dyi + (wa1 - wa) * dz;
      const double d1 = (u2 - u1) * dv.dxi + (vb1 - vb) * dv.dyi + (wb1 - wb) * dz;
      a[l * q.MP + pad(nn)] = make_double2(d0, d1);
      a[l * q.MP + pad(M - 1 - nn)] = make_double2(d1, d0);
    }
  } else
  for (int wi = tid; wi < (M << q.lL); wi += NTH) {
    const int l = wi >> LM, n = wi & (M - 1);
    const long ro = q.sz * (long)(k + HZ) + (long)q.sy * (j0 + l + HY);
    const double2 *ru = reinterpret_cast<const double2 *>(dv.pu + ro), *rv = reinterpret_cast<const double2 *>(dv.pv + ro);
    const double2 *rv1 = reinterpret_cast<const double2 *>(dv.pv + ro + q.sy), *rw = reinterpret_cast<const double2 *>(dv.pw + ro);
    const double2 u = ru[n], un = ru[(n + 1) & (M - 1)], v = rv[n], v1 = rv1[n], w = rw[n];
    double2 w1 = make_double2(0., 0.);
    if (k < dv.nz - 1) w1 = reinterpret_cast<const double2 *>(dv.pw + ro + q.sz)[n];
    const double dz = dv.dzfi[k + 1];
    a[l * q.MP + pad(n)] = make_double2((u.y - u.x) * dv.dxi + (v1.x - v.x) * dv.dyi + (w1.x - w.x) * dz,
                                        (un.x - u.y) * dv.dxi + (v1.y - v.y) * dv.dyi + (w1.y - w.y) * dz);
  }
  __syncthreads();
  double2 *z = fft_lines<false, LM, NTH>(a, b, tw, q.MP, L);
  // split (as in fftx_fwd_pack_kernel), stored kx-fastest: one row of spec per line; kx = M by the first L threads
  for (int wi = tid; wi < (M << q.lL) + L; wi += NTH) {
    const bool last = wi >= (M << q.lL);
    const int l = last ? wi - (M << q.lL) : wi >> LM, kx = last ? M : (wi & (M - 1));
    const double2 *zl = z + l * q.MP;
    const double2 zk = zl[pad(kx == M ? 0 : kx)], zc = cconj(zl[pad(kx == 0 ? 0 : M - kx)]);
    const double2 s_ = cadd(zk, zc), d = csub(zk, zc);
    const double2 wd = cmul(twN[kx], d);
    spec[((size_t)k * q.ny + j0 + l) * q.nkxp + kx] = make_double2(0.5 * (s_.x + wd.y), 0.5 * (s_.y - wd.x));
  }
}

// y pass, in place: columns c0 .. c0+C-1 of plane k (C consecutive complex per row)
template <int LM>
__global__ __launch_bounds__(FT) void ffty_nat_kernel(NatArgs q, const double2 *__restrict__ twg, double2 *__restrict__ spec) {
  extern __shared__ double2 lds[];
  constexpr int M = 1 << LM;
  double2 *a = lds, *b = lds + q.C * q.YP, *tw = b + q.C * q.YP;
  const int tid = threadIdx.x;
  for (int n = tid; n < M; n += FT) tw[n] = twg[n];
  const int c0 = blockIdx.x * q.C, k = blockIdx.y;
  const int ncol = min(q.C, q.nkxp - c0);
  double2 *pl = spec + (size_t)k * q.ny * q.nkxp + c0;
  for (int wi = tid; wi < M * q.C; wi += FT) {
    const int y = wi / q.C, col = wi - y * q.C;
    if (col < ncol) a[col * q.YP + pad(y)] = pl[(size_t)y * q.nkxp + col];
  }
  __syncthreads();
  double2 *z = fft_lines<false, LM>(a, b, tw, q.YP, ncol);
  for (int wi = tid; wi < M * q.C; wi += FT) {
    const int y = wi / q.C, col = wi - y * q.C;
    if (col < ncol) pl[(size_t)y * q.nkxp + col] = z[col * q.YP + pad(y)];
  }
}

// y pass of 256 points with one trip through LDS: 256 = 16 x 16 (Cooley-Tukey, n = 16 n1 + n2, k = k1 + 16 k2).  Thread (col, n2)
// loads its 16 rows n = 16 n1 + n2 straight from memory, transforms them in registers (two layers of radix 4), applies
// W_256^(n2 k1) and leaves Y[k1] in LDS; after the barrier thread (col, k1) picks up its 16 values over n2, transforms them and
// stores rows k = k1 + 16 k2.  C x 16 threads, C x 16 x 17 complex of LDS (rows of 17: the second step reads with stride 17).
__device__ __forceinline__ double2 mulw16(double2 v, int e) {        // v * exp(-2 pi i e / 16); e is a constant after unrolling
  constexpr double c1 = 0.92387953251128673848, s1 = 0.38268343236508978178, h = 0.70710678118654752440;
  switch (e & 15) {
    case 0: return v;
    case 1: return make_double2(v.x * c1 + v.y * s1, v.y * c1 - v.x * s1);
    case 2: return make_double2((v.x + v.y) * h, (v.y - v.x) * h);
    case 3: return make_double2(v.x * s1 + v.y * c1, v.y * s1 - v.x * c1);
    case 4: return make_double2(v.y, -v.x);
    case 6: return make_double2((v.y - v.x) * h, -(v.x + v.y) * h);
    case 9: return make_double2(-v.x * c1 - v.y * s1, v.x * s1 - v.y * c1);
    default: return v;      // (not reached: the exponents are j q, j, q < 4)
  }
}
__device__ __forceinline__ void r4fwd(double2 &a, double2 &b, double2 &c, double2 &d) {
  const double2 t0 = cadd(a, c), t1 = csub(a, c), t2 = cadd(b, d), e = csub(b, d);
  const double2 t3 = make_double2(e.y, -e.x);
  a = cadd(t0, t2); b = cadd(t1, t3); c = csub(t0, t2); d = csub(t1, t3);
}
// forward transform of 16 values in registers; X[q + 4 r] ends up in x[4 q + r]
__device__ __forceinline__ void fft16(double2 (&x)[16]) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    r4fwd(x[j], x[j + 4], x[j + 8], x[j + 12]);
#pragma unroll
    for (int q = 1; q < 4; ++q) x[j + 4 * q] = mulw16(x[j + 4 * q], j * q);
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) r4fwd(x[4 * q], x[4 * q + 1], x[4 * q + 2], x[4 * q + 3]);
}

// v * exp(-2 pi i e / 32), e a constant after unrolling
__device__ __forceinline__ double2 mulw32(double2 v, int e) {
  constexpr double C32[16] = {1.00000000000000000000, 0.98078528040323043058, 0.92387953251128673848, 0.83146961230254523567, 0.70710678118654757274, 0.55557023301960228867, 0.38268343236508983729, 0.19509032201612833135, 0.00000000000000006123, -0.19509032201612819257, -0.38268343236508972627, -0.55557023301960195560, -0.70710678118654746172, -0.83146961230254534669, -0.92387953251128673848, -0.98078528040323043058};
  constexpr double S32[16] = {0.00000000000000000000, 0.19509032201612824808, 0.38268343236508978178, 0.55557023301960217765, 0.70710678118654746172, 0.83146961230254523567, 0.92387953251128673848, 0.98078528040323043058, 1.00000000000000000000, 0.98078528040323043058, 0.92387953251128673848, 0.83146961230254545772, 0.70710678118654757274, 0.55557023301960217765, 0.38268343236508989280, 0.19509032201612860891};
  if (e == 0) return v;
  if (e == 8) return make_double2(v.y, -v.x);
  return make_double2(v.x * C32[e & 15] + v.y * S32[e & 15], v.y * C32[e & 15] - v.x * S32[e & 15]);
}
// forward transforms of 8 / 32 values in registers: one radix-2 layer (decimation in frequency) over two halves;
// X[k] ends up in x[regpos<N>(k)]
__device__ __forceinline__ void fft8(double2 (&x)[8]) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const double2 a = cadd(x[j], x[j + 4]), b = csub(x[j], x[j + 4]);
    x[j] = a; x[j + 4] = mulw32(b, 4 * j);
  }
  r4fwd(x[0], x[1], x[2], x[3]);
  r4fwd(x[4], x[5], x[6], x[7]);
}
__device__ __forceinline__ void fft32(double2 (&x)[32]) {
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const double2 a = cadd(x[j], x[j + 16]), b = csub(x[j], x[j + 16]);
    x[j] = a; x[j + 16] = mulw32(b, j);
  }
  double2 (&lo)[16] = *reinterpret_cast<double2 (*)[16]>(&x[0]);
  double2 (&hi)[16] = *reinterpret_cast<double2 (*)[16]>(&x[16]);
  fft16(lo);
  fft16(hi);
}
__device__ __forceinline__ constexpr int regpos16(int k) { return 4 * (k & 3) + (k >> 2); }
template <int N> __device__ __forceinline__ constexpr int regpos(int k) {
  return N == 16 ? regpos16(k) : (N == 8 ? ((k & 1) ? 4 + (k >> 1) : (k >> 1)) : ((k & 1) ? 16 + regpos16(k >> 1) : regpos16(k >> 1)));
}
template <int N> __device__ __forceinline__ void fft_reg(double2 (&x)[N]);
template <> __device__ __forceinline__ void fft_reg<8>(double2 (&x)[8]) { fft8(x); }
template <> __device__ __forceinline__ void fft_reg<16>(double2 (&x)[16]) { fft16(x); }
template <> __device__ __forceinline__ void fft_reg<32>(double2 (&x)[32]) { fft32(x); }

#ifndef NATREG_CPAD
#define NATREG_CPAD 2
#endif
// N = 16 x N2 (N2 = 8, 16, 32: ny = 128, 256, 512); max(16, N2) threads per column
template <int LN2>
__global__ __launch_bounds__(256) void ffty_natreg_kernel(NatArgs q, const double2 *__restrict__ twg, double2 *__restrict__ spec) {
  extern __shared__ double2 lds[];
  constexpr int N2 = 1 << LN2, LP = N2 + 1;
  const int tid = threadIdx.x;
  const int col = tid % q.C, t = tid / q.C;            // t = n2 in the first step, k1 in the second
  const int c0 = blockIdx.x * q.C, k = blockIdx.y;
  const bool on = c0 + col < q.nkxp;
  double2 *pl = spec + (size_t)k * q.ny * q.nkxp + c0 + col;
  double2 *mine = lds + col * (16 * LP + NATREG_CPAD);      // lanes run along the columns here: +2 keeps 8 columns x 2 rows on 16 different 16-B slots
  if (on && t < N2) {
    double2 x[16];
#pragma unroll
    for (int n1 = 0; n1 < 16; ++n1) x[n1] = pl[(size_t)(N2 * n1 + t) * q.nkxp];
    double2 w[16];                                    // all fifteen twiddles in flight behind the rows, not one wait each
#pragma unroll
    for (int k1 = 1; k1 < 16; ++k1) w[k1] = twg[t * k1];
    fft16(x);
#pragma unroll
    for (int k1 = 0; k1 < 16; ++k1) mine[k1 * LP + t] = k1 == 0 ? x[regpos16(k1)] : cmul(x[regpos16(k1)], w[k1]);
  }
  __syncthreads();
  if (on && t < 16) {
    double2 x[N2];
#pragma unroll
    for (int n2 = 0; n2 < N2; ++n2) x[n2] = mine[t * LP + n2];
    fft_reg<N2>(x);
#pragma unroll
    for (int k2 = 0; k2 < N2; ++k2) pl[(size_t)(t + 16 * k2) * q.nkxp] = x[regpos<N2>(k2)];
  }
}

// The slab path's y transforms the same way (16 x N2 in registers, one trip through LDS), on its y-contiguous lines: lanes run
// along y, so the loads are runs of N2 and the stores runs of 16 complex.  INV: conj(FFT(conj(x))), unnormalised.
//   forward:  recv[s][kc][kxl][j]  -> specB[k][kxl][y]        backward:  specB[k][kxl][y] -> send[d][kc][kxl][j]
template <int LN2, bool INV>
__global__ __launch_bounds__(256) void ffty_slabreg_kernel(YArgs q, const double2 *__restrict__ in, const double2 *__restrict__ twg,
                                                           double2 *__restrict__ out) {
  extern __shared__ double2 lds[];
  constexpr int N2 = 1 << LN2, LP = N2 + 1, T = N2 > 16 ? N2 : 16, M = 16 * N2;
  const int tid = threadIdx.x;
  const int t = tid % T, col = tid / T;               // t = n2 in the first step, k1 in the second
  const int c0 = blockIdx.x * q.C, kc = blockIdx.y, k = q.k0 + kc;
  const bool on = c0 + col < q.cx;
  double2 *mine = lds + col * (16 * LP);
  const size_t lineB = ((size_t)k * q.cx + c0 + col) * M;                       // specB line
  auto xoff = [&](int y) {                                                      // exchange-buffer element of (this column, y)
    const int s_ = y >> q.lnyl, j = y & (q.nyl - 1);
    return (((size_t)s_ * q.nzc + kc) * q.cx + c0 + col) * q.nyl + j;
  };
  if (on && t < N2) {
    double2 x[16];
#pragma unroll
    for (int n1 = 0; n1 < 16; ++n1) {
      const int y = N2 * n1 + t;
      x[n1] = INV ? cconj(in[lineB + y]) : in[xoff(y)];
    }
    double2 w[16];
#pragma unroll
    for (int k1 = 1; k1 < 16; ++k1) w[k1] = twg[t * k1];
    fft16(x);
#pragma unroll
    for (int k1 = 0; k1 < 16; ++k1) mine[k1 * LP + t] = k1 == 0 ? x[regpos16(k1)] : cmul(x[regpos16(k1)], w[k1]);
  }
  __syncthreads();
  if (on && t < 16) {
    double2 x[N2];
#pragma unroll
    for (int n2 = 0; n2 < N2; ++n2) x[n2] = mine[t * LP + n2];
    fft_reg<N2>(x);
#pragma unroll
    for (int k2 = 0; k2 < N2; ++k2) {
      const int y = t + 16 * k2;
      if (INV) bwd_store(q, out, kc, c0 + col, y, cconj(x[regpos<N2>(k2)]), 0);
      else out[lineB + y] = x[regpos<N2>(k2)];
    }
  }
}

inline int ilog2(int n) { int l = 0; while ((1 << l) < n) ++l; return l; }
inline bool pow2(int n) { return n > 0 && (n & (n - 1)) == 0; }

}  // namespace

// ------------------------------------------------------------------------------------------------ host side
bool fft_fused_possible(const udc_handle *h) {
  const int nx = h->g.nx, ny = h->jtot, nyl = h->g.ny;
  return pow2(nx) && nx >= 16 && nx <= 2048 && pow2(ny) && ny >= 8 && ny <= 1024 && pow2(nyl) && nyl >= 4;
}

static size_t x_lds_bytes(const udc_handle *h, int L) {
  const int M = h->g.nx / 2;
  return (size_t)2 * L * padded(M + 1) * 16 + (size_t)M * 16 + (size_t)h->cx * h->cfg.nranks * 4;
}
static size_t r8_lds_bytes(int M, int L) { return ((size_t)L * (M + M / 8 + M / 64 + 1) + M) * 16; }      // lines + the stages' twiddles
static size_t y_lds_bytes(const udc_handle *h, int C) { return (size_t)2 * C * padded(h->jtot) * 16 + (size_t)h->jtot * 16; }

#define FFT_DISPATCH(LMV, CALL)                                                        \
  switch (LMV) {                                                                       \
    case 3: { constexpr int LM = 3; CALL; } break;   case 4: { constexpr int LM = 4; CALL; } break;   \
    case 5: { constexpr int LM = 5; CALL; } break;   case 6: { constexpr int LM = 6; CALL; } break;   \
    case 7: { constexpr int LM = 7; CALL; } break;   case 8: { constexpr int LM = 8; CALL; } break;   \
    case 9: { constexpr int LM = 9; CALL; } break;   case 10: { constexpr int LM = 10; CALL; } break; \
    default: udc_set_error("fused FFT: unsupported length 2^%d", LMV); return 1;       \
  }

// twiddle tables of the line transforms: exp(-2 pi i n / M), n < M = nx/2 | exp(-2 pi i n / nx), n <= M | exp(-2 pi i n / ny), n < ny
static int fft_twiddles(udc_handle *h, int nx, int ny) {
  if (h->fft_tw) return 0;
  const int M = nx / 2;
  const double pi = 3.141592653589793238462643383279502884;
  std::vector<double> t;
  t.reserve((size_t)2 * (M + (M + 1) + ny));
  for (int n = 0; n < M; ++n) { t.push_back(cos(2. * pi * n / M)); t.push_back(-sin(2. * pi * n / M)); }          // twM
  for (int n = 0; n <= M; ++n) { t.push_back(cos(2. * pi * n / nx)); t.push_back(-sin(2. * pi * n / nx)); }       // twN
  for (int n = 0; n < ny; ++n) { t.push_back(cos(2. * pi * n / ny)); t.push_back(-sin(2. * pi * n / ny)); }       // twY
  HIP_OK(hipMalloc(&h->fft_tw, sizeof(double) * t.size()));
  HIP_OK(hipMemcpy(h->fft_tw, t.data(), sizeof(double) * t.size(), hipMemcpyHostToDevice));
  return 0;
}

static int slab_yreg_cols(int ny);
int fft_fused_init(udc_handle *h) {
  const int nx = h->g.nx, ny = h->jtot, M = nx / 2;
  if (fft_twiddles(h, nx, ny)) return 1;
  // rows per workgroup of the x kernels (= the run length of the packed writes): as long as four workgroups still fit
  // a CU's 160 KB of LDS, at least 4; columns per workgroup of the y kernels likewise (UDC_FFT_L / UDC_FFT_C override)
  int L = 16;
  while (L > tune::FFT_X_MIN_ROWS && x_lds_bytes(h, L) > (size_t)tune::FFT_X_LDS_BUDGET) L >>= 1;
  while (h->g.ny % L) L >>= 1;
  int C = tune::FFT_Y_COLS;
  while (C > 1 && y_lds_bytes(h, C) > (size_t)tune::FFT_X_LDS_BUDGET) C >>= 1;
  h->fft_L = L; h->fft_C = C;
  h->slab_yreg = (ny == 128 || ny == 256 || ny == 512) && h->sw.slab_yreg;
  if (h->slab_yreg && ny == 512) {
    const int ldsb = slab_yreg_cols(ny) * 16 * 33 * 16;
    HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void *>(ffty_slabreg_kernel<5, false>), hipFuncAttributeMaxDynamicSharedMemorySize, ldsb));
    HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void *>(ffty_slabreg_kernel<5, true>), hipFuncAttributeMaxDynamicSharedMemorySize, ldsb));
  }
  const int ldsx = (int)x_lds_bytes(h, L), ldsy = (int)y_lds_bytes(h, C);
  if (ldsx > 160 * 1024 || ldsy > 160 * 1024) { h->fft_fused = false; return 0; }
  if (M == 1024) HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void *>(fftx_bwd_r8_kernel<10>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)r8_lds_bytes(M, 4)));
  const int lmx = ilog2(M), lmy = ilog2(ny);
  if (ldsx > 65536) {
    FFT_DISPATCH(lmx, HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void *>(fftx_fwd_pack_kernel<LM, false>), hipFuncAttributeMaxDynamicSharedMemorySize, ldsx));
                      HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void *>(fftx_fwd_pack_kernel<LM, true>), hipFuncAttributeMaxDynamicSharedMemorySize, ldsx)))
  }
  if (ldsy > 65536) {
    FFT_DISPATCH(lmy, HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void *>(ffty_fwd_unpack_kernel<LM>), hipFuncAttributeMaxDynamicSharedMemorySize, ldsy));
                      HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void *>(ffty_bwd_pack_kernel<LM>), hipFuncAttributeMaxDynamicSharedMemorySize, ldsy)))
  }
  return 0;
}

static XArgs xargs(const udc_handle *h, int k0, int nzc) {
  const Geo &g = h->g;
  const int M = g.nx / 2;
  return XArgs{g.nx, M, padded(M + 1), g.ny, g.py, g.sy, g.sz, h->nkx, h->cx, h->cfg.nranks, k0, nzc, ilog2(h->fft_L), 0,
               (unsigned)((1ULL << 32) / (unsigned long long)h->cx + 1ULL), (long)nzc * h->cx * g.ny, 0};
}
static YArgs yargs(const udc_handle *h, int k0, int nzc) {
  return YArgs{h->jtot, padded(h->jtot), h->g.ny, ilog2(h->g.ny), h->cx, h->cfg.nranks, k0, nzc, h->fft_C, (long)nzc * h->cx * h->g.ny, 0};
}

int fft_x_row_groups(const udc_handle *h) { return h->g.ny / h->fft_L; }

// g0, g1: row groups [g0, g1) of the slab (fft_x_row_groups; g1 <= 0: all)
int fft_x_fwd_pack(udc_handle *h, int k0, int nzc, double *send, int g0, int g1, hipStream_t st) {
  if (!st) st = h->stream;
  XArgs q = xargs(h, k0, nzc);
  const double2 *tw = reinterpret_cast<const double2 *>(h->fft_tw);
  if (g1 <= 0) { g0 = 0; g1 = q.nyl >> q.lL; }
  q.jg0 = g0;
  const dim3 gr((unsigned)(g1 - g0), (unsigned)nzc);
  const size_t lds = x_lds_bytes(h, h->fft_L);
  // (open lid, BCtopm = 3: lid_bcpup_kernel has filled plane ke+1 of pwp -- "nz + 1 levels" makes the top cell read it instead of bcpup's zero)
  const DivArgs dv{h->fields[UDC_UP], h->fields[UDC_VP], h->fields[UDC_WP], h->m.dzfi, h->m.dxi, h->m.dyi,
                   h->g.nz + (h->p.bctopm == UDC_TOP_PRESSURE ? 1 : 0)};
  if (h->div_in_fft) {
    FFT_DISPATCH(ilog2(q.M), hipLaunchKernelGGL((fftx_fwd_pack_kernel<LM, true>), gr, dim3(xthreads(LM)), lds, st, q, (const double *)h->fields[UDC_P], dv,
                                                tw, tw + q.M, reinterpret_cast<double2 *>(send)))
  } else {
    FFT_DISPATCH(ilog2(q.M), hipLaunchKernelGGL((fftx_fwd_pack_kernel<LM, false>), gr, dim3(xthreads(LM)), lds, st, q, (const double *)h->fields[UDC_P], dv,
                                                tw, tw + q.M, reinterpret_cast<double2 *>(send)))
  }
  HIP_OK(hipGetLastError());
  return 0;
}
// rows per workgroup of the radix-8 x kernel: as many as 256 threads hold (512 for lines of 1024 complex)
static int r8_rows(const udc_handle *h) {
  const int tpl = h->g.nx / 16, ntmax = tpl * 4 > 256 ? tpl * 4 : 256;
  int L = ntmax / tpl;
  if (L > 16) L = 16;
  while (L > 1 && h->g.ny % L) L >>= 1;
  return L;
}
// ghost: the blocks carry the two rows just outside the slab behind their own (fft_y_bwd_pack put them there): p's ghost rows are
// transformed and stored with the others
int fft_x_bwd_unpack(udc_handle *h, int k0, int nzc, const double *recv, bool ghost) {
  XArgs q = xargs(h, k0, nzc);
  if (ghost) { q.ghost = 1; q.bstride = (long)nzc * h->cx * (h->g.ny + 2); }
  const double2 *tw = reinterpret_cast<const double2 *>(h->fft_tw);
  const int L = r8_rows(h), tpl = q.M / 8;
  q.lL = ilog2(L);
  const size_t lds = r8_lds_bytes(q.M, L);
  const unsigned items = (unsigned)(q.nyl >> q.lL) * (unsigned)nzc + (ghost ? (unsigned)nzc : 0u);
  // persistent workgroups: as many as are resident at once (LDS, 2048 threads per CU)
  const unsigned per_cu = (unsigned)std::max<size_t>(1, std::min<size_t>(160 * 1024 / lds, 2048 / (size_t)(L * tpl)));
  const dim3 gr(std::min(items, 256u * per_cu));
  FFT_DISPATCH(ilog2(q.M), hipLaunchKernelGGL(fftx_bwd_r8_kernel<LM>, gr, dim3(L * tpl), lds, h->stream, q, reinterpret_cast<const double2 *>(recv),
                                              tw, tw + q.M, h->fields[UDC_P]))
  HIP_OK(hipGetLastError());
  return 0;
}
// register y pass of the slab path: ny = 128, 256, 512 (UDC_SLAB_YREG=0: the Stockham kernels)
static int slab_yreg_cols(int ny) { const int n2 = ny / 16, tpc = n2 > 16 ? n2 : 16; int C = 8; while (C > 1 && C * tpc > 256) C >>= 1; return C; }
template <bool INV>
static int launch_slab_yreg(udc_handle *h, YArgs q, const double2 *in, const double2 *tw, double2 *out) {
  const int n2 = q.ny / 16, tpc = n2 > 16 ? n2 : 16;
  q.C = slab_yreg_cols(q.ny);
  const size_t ldsb = (size_t)q.C * 16 * (n2 + 1) * 16;
  const dim3 gr((unsigned)((q.cx + q.C - 1) / q.C), (unsigned)q.nzc), b((unsigned)(tpc * q.C));
  if (n2 == 8) hipLaunchKernelGGL((ffty_slabreg_kernel<3, INV>), gr, b, ldsb, h->stream, q, in, tw, out);
  else if (n2 == 16) hipLaunchKernelGGL((ffty_slabreg_kernel<4, INV>), gr, b, ldsb, h->stream, q, in, tw, out);
  else hipLaunchKernelGGL((ffty_slabreg_kernel<5, INV>), gr, b, ldsb, h->stream, q, in, tw, out);
  HIP_OK(hipGetLastError());
  return 0;
}

int fft_y_fwd_unpack(udc_handle *h, int k0, int nzc, const double *recv) {
  const YArgs q = yargs(h, k0, nzc);
  const double2 *tw = reinterpret_cast<const double2 *>(h->fft_tw) + (h->g.nx / 2) + (h->g.nx / 2 + 1);
  if (h->slab_yreg) return launch_slab_yreg<false>(h, q, reinterpret_cast<const double2 *>(recv), tw, reinterpret_cast<double2 *>(h->specB));
  const dim3 gr((unsigned)((q.cx + q.C - 1) / q.C), (unsigned)nzc);
  const size_t lds = y_lds_bytes(h, q.C);
  FFT_DISPATCH(ilog2(q.ny), hipLaunchKernelGGL(ffty_fwd_unpack_kernel<LM>, gr, dim3(FT), lds, h->stream, q, reinterpret_cast<const double2 *>(recv),
                                               tw, reinterpret_cast<double2 *>(h->specB)))
  HIP_OK(hipGetLastError());
  return 0;
}
int fft_y_bwd_pack(udc_handle *h, int k0, int nzc, double *send, bool ghost) {
  YArgs q = yargs(h, k0, nzc);
  if (ghost) { q.ghost = 1; q.bstride = (long)nzc * h->cx * (h->g.ny + 2); }
  const double2 *tw = reinterpret_cast<const double2 *>(h->fft_tw) + (h->g.nx / 2) + (h->g.nx / 2 + 1);
  if (h->slab_yreg) return launch_slab_yreg<true>(h, q, reinterpret_cast<const double2 *>(h->specB), tw, reinterpret_cast<double2 *>(send));
  const dim3 gr((unsigned)((q.cx + q.C - 1) / q.C), (unsigned)nzc);
  const size_t lds = y_lds_bytes(h, q.C);
  FFT_DISPATCH(ilog2(q.ny), hipLaunchKernelGGL(ffty_bwd_pack_kernel<LM>, gr, dim3(FT), lds, h->stream, q,
                                               reinterpret_cast<const double2 *>(h->specB), tw, reinterpret_cast<double2 *>(send)))
  HIP_OK(hipGetLastError());
  return 0;
}

// ---- one GPU: forward half in rocFFT's layout (default where the register y pass exists; UDC_OWN_FWD=0 / 1)
static size_t nat_x_lds(int M, int L) { return (size_t)2 * L * padded(M + 1) * 16 + (size_t)M * 16; }
static size_t nat_y_lds(int ny, int C) { return (size_t)2 * C * padded(ny) * 16 + (size_t)ny * 16; }

int fft_nat_init(udc_handle *h) {
  // default: on where the register y pass exists (ny = 128, 256, 512); UDC_OWN_FWD=0 keeps div_rhs + rocFFT's forward plan,
  // UDC_OWN_FWD=1 also takes the other power-of-two sizes (Stockham y pass: slower than rocFFT's, for tests)
  h->own_fwd = false;
  const int want = h->sw.own_fwd;
  if (want == 0) return 0;
  const int nx = h->g.nx, ny = h->g.ny, M = nx / 2;
  if (h->slab || h->fwd_compact || !pow2(nx) || nx < 16 || nx > 2048 || !pow2(ny) || ny < 8 || ny > 1024) return 0;
  if (want < 0 && !(ny == 128 || ny == 256 || ny == 512)) return 0;
  if (fft_twiddles(h, nx, ny)) return 1;
  // rows per workgroup of the x kernel: 4 (256^3: 0.085 ms against 0.088 with 2 and 0.097 with 8, profiles/r03/own_fwd_ab.json;
  // nx = 512: 0.387 against 0.417 / 0.424, nx = 1024: 1.64 against 1.67 / 2.24, profiles/r03/nat_l_scan.json)
  int L = tune::NAT_X_ROWS;
  while (ny % L) L >>= 1;
  int C = tune::NAT_Y_COLS;
  h->nat_reg16 = (ny == 128 || ny == 256 || ny == 512) && h->sw.nat_reg;
  if (h->nat_reg16) {
    const int n2 = ny / 16, tpc = n2 > 16 ? n2 : 16;
    while (C > 1 && C * tpc > 256) C >>= 1;
    const int ldsb = C * (16 * (n2 + 1) + NATREG_CPAD) * 16;
    if (ldsb > 65536) {
      if (n2 == 32) HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void *>(ffty_natreg_kernel<5>), hipFuncAttributeMaxDynamicSharedMemorySize, ldsb));
      else if (n2 == 16) HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void *>(ffty_natreg_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, ldsb));
    }
  }
  h->nat_L = L; h->nat_C = C;
  const int ldsx = (int)nat_x_lds(M, L), ldsy = (int)nat_y_lds(ny, C);
  if (ldsx > 160 * 1024 || ldsy > 160 * 1024) return 0;
  const int lmx = ilog2(M), lmy = ilog2(ny);
  if (ldsx > 65536) {
    FFT_DISPATCH(lmx, HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void *>(fftx_fwd_nat_kernel<LM>), hipFuncAttributeMaxDynamicSharedMemorySize, ldsx)))
    FFT_DISPATCH(lmx, HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void *>(fftx_fwd_nat_kernel<LM, true>), hipFuncAttributeMaxDynamicSharedMemorySize, ldsx)))
  }
  if (ldsy > 65536) {
    FFT_DISPATCH(lmy, HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void *>(ffty_nat_kernel<LM>), hipFuncAttributeMaxDynamicSharedMemorySize, ldsy)))
  }
  h->own_fwd = true;
  return 0;
}

// divergence of (pup, pvp, pwp) -> x R2C -> y C2C, into h->spec
int fft_nat_forward(udc_handle *h) {
  const Geo &g = h->g;
  const int M = g.nx / 2;
  NatArgs q{g.nx, M, padded(M + 1), g.ny, h->nkx, h->nkxp, g.sy, g.sz, ilog2(h->nat_L), h->nat_C, padded(g.ny)};
  const double2 *tw = reinterpret_cast<const double2 *>(h->fft_tw);
  // (this handle is the solver's doubled row of a handle with open x boundaries: the tendencies are that handle's, k_xo_poisson)
  const udc_handle *src = h->xo_src ? h->xo_src : h;
  DivArgs dv{src->fields[UDC_UP], src->fields[UDC_VP], src->fields[UDC_WP], src->m.dzfi, src->m.dxi, src->m.dyi,
             g.nz + (src->p.bctopm == UDC_TOP_PRESSURE ? 1 : 0)};      // (open lid: as in fft_x_fwd_pack)
  if (src != h) { dv.xo_n = g.nx / 2; dv.xo_g = src->g.xg; dv.sy = src->g.sy; dv.sz = src->g.sz; }
  double2 *spec = reinterpret_cast<double2 *>(h->spec);
  {
    PROF(h, "fftx_pack_fwd");
    const dim3 gr((unsigned)(g.ny >> q.lL), (unsigned)g.nz);
    if (src != h) {
      FFT_DISPATCH(ilog2(M), hipLaunchKernelGGL((fftx_fwd_nat_kernel<LM, true>), gr, dim3(xthreads(LM)), nat_x_lds(M, h->nat_L), h->stream, q, dv, tw, tw + M, spec))
    } else {
      FFT_DISPATCH(ilog2(M), hipLaunchKernelGGL(fftx_fwd_nat_kernel<LM>, gr, dim3(xthreads(LM)), nat_x_lds(M, h->nat_L), h->stream, q, dv, tw, tw + M, spec))
    }
    HIP_OK(hipGetLastError());
  }
  {
    PROF(h, "unpack_ffty_fwd");
    const dim3 gr((unsigned)((h->nkxp + q.C - 1) / q.C), (unsigned)g.nz);
    if (h->nat_reg16) {
      const int n2 = g.ny / 16, tpc = n2 > 16 ? n2 : 16;
      const size_t ldsb = (size_t)q.C * (16 * (n2 + 1) + NATREG_CPAD) * 16;
      if (n2 == 8) hipLaunchKernelGGL(ffty_natreg_kernel<3>, gr, dim3(tpc * q.C), ldsb, h->stream, q, tw + M + (M + 1), spec);
      else if (n2 == 16) hipLaunchKernelGGL(ffty_natreg_kernel<4>, gr, dim3(tpc * q.C), ldsb, h->stream, q, tw + M + (M + 1), spec);
      else hipLaunchKernelGGL(ffty_natreg_kernel<5>, gr, dim3(tpc * q.C), ldsb, h->stream, q, tw + M + (M + 1), spec);
    } else {
      FFT_DISPATCH(ilog2(g.ny), hipLaunchKernelGGL(ffty_nat_kernel<LM>, gr, dim3(FT), nat_y_lds(g.ny, q.C), h->stream, q, tw + M + (M + 1), spec))
    }
    HIP_OK(hipGetLastError());
  }
  return 0;
}
