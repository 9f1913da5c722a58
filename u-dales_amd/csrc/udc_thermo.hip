// Slab (horizontal-plane) sums and what is built on them: the dry buoyancy term of `forces`
// (src/modforces.f90:73-84 with calc_halflev / the slab average of src/modthermodynamics.f90:76,208,518-524), diagfld's
// slab averages (:262-279) and the per-level forcings tend += A(k) + B(k) field of lstend / nudge / grwdamp
// (src/modforces.f90:719-860, src/modboundary.f90:1447-1488; tables built on the host, udcore/forcings.py).
#include "udc_internal.h"

namespace {

inline dim3 cell_grid(const Geo &g, dim3 b) {
  (void)b;
  return dim3((unsigned)tile_grid(g).tiles * (unsigned)g.nz, 1, 1);
}

// Buoyancy (forces with lbuoyancy, src/modforces.f90:73-84; dry air: thv0h = thl0h, src/modthermodynamics.f90:208):
//   wp(k) += grav (thl0h(i,j,k) - thvh(k)) / thvh(k),  k = kb+1..ke,
//   thl0h(k) = (thl0(k) dzf(k-1) + thl0(k-1) dzf(k)) / (2 dzh(k))          (calc_halflev, :518-524)
//   thvh(k)  = slab average of thl0h(k)                                     (thermodynamics :76, avexy_ibm)
__device__ __forceinline__ double thl_half(const Geo &g, const Metrics &m, const double *__restrict__ t, long c, int k) {
  const int kf = k + 1;
  return (t[c] * m.dzf[kf - 1] + t[c - g.sz] * m.dzf[kf]) / (2 * m.dzh[kf]);
}
// stage 1: one workgroup per (xy tile, level) -> part[level * tiles + tile]; stage 2: one workgroup per level
__global__ __launch_bounds__(256) void levelsum_kernel(Geo g, Metrics m, int gx, const double *__restrict__ thl, double *__restrict__ part) {
  __shared__ double sw[4];
  const int tile = blockIdx.x, k = blockIdx.y;
  const int by = tile / gx, bx = tile - by * gx;
  const int i = bx * 64 + threadIdx.x, j = by * 4 + threadIdx.y;
  double v = 0.;
  if (i < g.nx && j < g.ny && k >= 1) v = thl_half(g, m, thl, g.idx(i, j, k), k);
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  if (threadIdx.x == 0) sw[threadIdx.y] = v;
  __syncthreads();
  if (threadIdx.x == 0 && threadIdx.y == 0) part[(size_t)k * gridDim.x + tile] = (sw[0] + sw[1]) + (sw[2] + sw[3]);
}
__global__ __launch_bounds__(256) void levelsum_final_kernel(int tiles, const double *__restrict__ part, double *__restrict__ S) {
  __shared__ double sw[4];
  const int k = blockIdx.x;
  double v = 0.;
  for (int q = threadIdx.x; q < tiles; q += 256) v += part[(size_t)k * tiles + q];
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  if ((threadIdx.x & 63) == 0) sw[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) S[k] = (sw[0] + sw[1]) + (sw[2] + sw[3]);
}
__global__ __launch_bounds__(256) void buoyancy_kernel(Geo g, TileGrid tg, Metrics m, const double *__restrict__ thl,
                                                        const double *__restrict__ S, double cnt, double grav, double *__restrict__ wp) {
  int i, j, k;
  if (!tile_decode(g, tg, i, j, k) || k < 1) return;
  const long c = g.idx(i, j, k);
  const double thvh = S[k] / cnt;
  wp[c] = wp[c] + grav * (thl_half(g, m, thl, c, k) - thvh) / thvh;
}

// plain per-level slab sums (stage 1; levelsum_final_kernel is stage 2): levels k = 0..nlev-1 (device), i.e. 1..nlev
__global__ __launch_bounds__(256) void levelsum_plain_kernel(Geo g, int gx, const double *__restrict__ f, double *__restrict__ part) {
  __shared__ double sw[4];
  const int tile = blockIdx.x, k = blockIdx.y;
  const int by = tile / gx, bx = tile - by * gx;
  const int i = bx * 64 + threadIdx.x, j = by * 4 + threadIdx.y;
  double v = 0.;
  if (i < g.nx && j < g.ny) v = f[g.idx(i, j, k)];
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  if (threadIdx.x == 0) sw[threadIdx.y] = v;
  __syncthreads();
  if (threadIdx.x == 0 && threadIdx.y == 0) part[(size_t)k * gridDim.x + tile] = (sw[0] + sw[1]) + (sw[2] + sw[3]);
}
// tend += A(k) + B(k) src   (A, B indexed by the reference's k)
__global__ __launch_bounds__(256) void level_affine_kernel(Geo g, TileGrid tg, const double *__restrict__ A, const double *__restrict__ B,
                                                            const double *__restrict__ src, double *__restrict__ tend, int wrap) {
  int i, j, k;
  if (!tile_decode(g, tg, i, j, k)) return;
  const long c = g.idx(i, j, k);
  double t = tend[c] + A[k + 1];
  if (src) t = t + B[k + 1] * src[c];
  tend[c] = t;
  if (wrap && j == 0) tend[c + (long)g.sy * g.ny] = t;
}

}  // namespace

int k_slab_average(udc_handle *h, int field, double *avg_host, int n) {
  const Geo &g = h->g;
  if (n < 1 || n > g.nz + 1) { udc_set_error("udc_slab_average: 1 <= n <= ktot+1"); return 1; }
  if (field < 0 || field >= (int)h->fields.size() || !h->fields[field]) { udc_set_error("udc_slab_average: unknown field %d", field); return 1; }
  const TileGrid tg = tile_grid(g);
  const size_t need = (size_t)tg.tiles * (g.nz + 1);
  if (h->lev_cap < need) {
    if (h->lev_part) HIP_OK(hipFree(h->lev_part));
    HIP_OK(hipMalloc(&h->lev_part, sizeof(double) * need));
    h->lev_cap = need;
  }
  if (!h->lev_sum) HIP_OK(hipMalloc(&h->lev_sum, sizeof(double) * (g.nz + 2)));
  hipLaunchKernelGGL(levelsum_plain_kernel, dim3((unsigned)tg.tiles, (unsigned)n), dim3(64, 4), 0, h->stream, g, tg.gx,
                     (const double *)h->fields[field], h->lev_part);
  hipLaunchKernelGGL(levelsum_final_kernel, dim3((unsigned)n), dim3(256), 0, h->stream, tg.tiles, h->lev_part, h->lev_sum);
  HIP_OK(hipGetLastError());
  if (comm_allreduce(h, h->lev_sum, n, 1)) return 1;
  HIP_OK(hipMemcpyAsync(h->red_host, h->lev_sum, sizeof(double) * n, hipMemcpyDeviceToHost, h->stream));
  HIP_OK(hipStreamSynchronize(h->stream));
  const double cnt = (double)g.nx * (double)h->cfg.jtot;
  for (int k = 0; k < n; ++k) avg_host[k] = h->red_host[k] / cnt;
  return 0;
}

int k_level_forcings(udc_handle *h, int when, bool wrap_vp) {
  const Geo &g = h->g;
  dim3 b(64, 4, 1), gr = cell_grid(g, b);
  for (const auto &f : h->level_forcings) {
    if (f.when != when) continue;
    PROF(h, "level_forcing");
    hipLaunchKernelGGL(level_affine_kernel, gr, b, 0, h->stream, g, tile_grid(g), (const double *)f.A, (const double *)f.B,
                       f.src >= 0 ? (const double *)h->fields[f.src] : nullptr, h->fields[f.tend], (wrap_vp && f.tend == UDC_VP) ? 1 : 0);
    HIP_OK(hipGetLastError());
  }
  return 0;
}

int k_buoyancy(udc_handle *h) {
  const Geo &g = h->g;
  if (!h->lbuoyancy) return 0;
  if ((int)h->fields.size() <= UDC_THL0 || !h->fields[UDC_THL0]) { udc_set_error("buoyancy needs the temperature equation (udc_set_tempeq)"); return 1; }
  const TileGrid tg = tile_grid(g);
  const size_t need = (size_t)tg.tiles * g.nz;
  if (h->lev_cap < need) {
    if (h->lev_part) HIP_OK(hipFree(h->lev_part));
    HIP_OK(hipMalloc(&h->lev_part, sizeof(double) * need));
    h->lev_cap = need;
  }
  if (!h->lev_sum) HIP_OK(hipMalloc(&h->lev_sum, sizeof(double) * (g.nz + 2)));
  PROF(h, "buoyancy");
  const double *thl = h->fields[UDC_THL0];
  hipLaunchKernelGGL(levelsum_kernel, dim3((unsigned)tg.tiles, (unsigned)g.nz), dim3(64, 4), 0, h->stream, g, h->m, tg.gx, thl, h->lev_part);
  hipLaunchKernelGGL(levelsum_final_kernel, dim3((unsigned)g.nz), dim3(256), 0, h->stream, tg.tiles, h->lev_part, h->lev_sum);
  HIP_OK(hipGetLastError());
  if (comm_allreduce(h, h->lev_sum, g.nz, 1)) return 1;       // avexy_ibm's MPI_ALLREDUCE over the slabs
  dim3 b(64, 4, 1), gr = cell_grid(g, b);
  hipLaunchKernelGGL(buoyancy_kernel, gr, b, 0, h->stream, g, tg, h->m, thl, h->lev_sum, (double)g.nx * (double)h->cfg.jtot,
                     h->grav, h->fields[UDC_WP]);
  HIP_OK(hipGetLastError());
  return 0;
}

