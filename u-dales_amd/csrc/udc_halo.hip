// Ghost rows / planes: y-periodic wrap (single slab) and the top/bottom rows of `boundary`.
// x has no ghosts on the device (index wrap in the kernels).
#include "udc_internal.h"

namespace {

struct FieldList {
  double *f[16];
};

// rows j = -w..-1 <- ny-w..ny-1 and rows ny..ny+w-1 <- 0..w-1, for every k plane (ghost
// planes included, as the reference's ym_periodic does: src/modboundary.f90:596-627)
__global__ void halo_y_wrap_kernel(Geo g, FieldList fl, int width) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= g.nx) return;
  const int r = blockIdx.y % (2 * width);
  const int fi = blockIdx.y / (2 * width);
  const int k = (int)blockIdx.z - HZ;
  int jdst, jsrc;
  if (r < width) { jdst = -1 - r; jsrc = g.ny - 1 - r; }
  else { jdst = g.ny + (r - width); jsrc = r - width; }
  double *a = fl.f[fi];
  a[g.idx(i, jdst, k)] = a[g.idx(i, jsrc, k)];
}

// multi-slab: pack `width` boundary rows of every field into a contiguous buffer
// buf[((f*width + r)*pz + kk)*nx + i]; low = rows 0..w-1 (to previous rank), high = rows ny-w..ny-1.
// dirs: bit 0 = the rows that go to the previous rank (and arrive from the next one), bit 1 = to the next (from the previous)
// (kk0, gridDim.z: the planes kk0 .. kk0 + gridDim.z - 1 of the padded array -- all of them, or one k-chunk's)
__global__ void halo_pack_kernel(Geo g, FieldList fl, int width, double *__restrict__ to_prev, double *__restrict__ to_next, int dirs, int kk0) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= g.nx) return;
  const int r = blockIdx.y % width, fi = blockIdx.y / width;
  const int kk = kk0 + blockIdx.z, k = kk - HZ;
  const double *a = fl.f[fi];
  const size_t o = (((size_t)fi * width + r) * gridDim.z + blockIdx.z) * g.nx + i;
  if (dirs & 1) to_prev[o] = a[g.idx(i, r, k)];
  if (dirs & 2) to_next[o] = a[g.idx(i, g.ny - width + r, k)];
}
__global__ void halo_unpack_kernel(Geo g, FieldList fl, int width, const double *__restrict__ from_prev,
                                   const double *__restrict__ from_next, int dirs, int kk0) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= g.nx) return;
  const int r = blockIdx.y % width, fi = blockIdx.y / width;
  const int kk = kk0 + blockIdx.z, k = kk - HZ;
  double *a = fl.f[fi];
  const size_t o = (((size_t)fi * width + r) * gridDim.z + blockIdx.z) * g.nx + i;
  if (dirs & 2) a[g.idx(i, -width + r, k)] = from_prev[o];
  if (dirs & 1) a[g.idx(i, g.ny + r, k)] = from_next[o];
}

struct BoundaryArgs {
  double *u0, *v0, *w0, *um, *vm, *wm;
  double *sv[32];
  int kind[32];      // top condition per array: 0 zero-flux copy, 1 fluxtop (top_flux_kernel), 2 valuetop, 3 fixed ghost value
  double val[32];
  int nscal;         // number of scalar arrays in sv (sv0 and svm of every scalar)
};

// boundary: src/modboundary.f90:163-247 (periodic lateral subset): w(kb) = 0; top ghost rows
// by fluxtop with zero flux (:1494-1507) or valuetop (:1509-1519); w(ke+1) = 0; scalars
// zero-flux top (:1521-1537).  Runs over the whole padded y extent after the y ghosts.
__global__ void top_bottom_kernel(Geo g, Params pr, BoundaryArgs a, int uv_only) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int j = (int)blockIdx.y - HY;
  if (i >= g.nx) return;
  const long top = g.idx(i, j, g.nz - 1), ghost = top + g.sz;
  if (pr.bctopm == UDC_TOP_NOSLIP) {
    if (!uv_only) {
      a.um[ghost] = 2 * pr.uinf - a.um[top]; a.u0[ghost] = 2 * pr.uinf - a.u0[top];
      a.vm[ghost] = 2 * pr.vinf - a.vm[top]; a.v0[ghost] = 2 * pr.vinf - a.v0[top];
    }
  } else {
    a.um[ghost] = a.um[top]; a.u0[ghost] = a.u0[top];
    a.vm[ghost] = a.vm[top]; a.v0[ghost] = a.v0[top];
  }
  if (uv_only) return;
  const long bot = g.idx(i, j, 0);
  a.wm[bot] = 0.; a.w0[bot] = 0.;
  if (pr.bctopm != UDC_TOP_PRESSURE) { a.w0[ghost] = 0.; a.wm[ghost] = 0.; }      // (open lid: "w considered in modpois", :191-200)
  for (int s = 0; s < a.nscal; ++s) {
    double *c = a.sv[s];
    if (a.kind[s] == 1) continue;                     // non-zero flux: top_flux_kernel (needs ekh)
    const double t = a.kind[s] == 2 ? 2 * a.val[s] - c[top]                // valuetop, src/modboundary.f90:1516
                   : (a.kind[s] == 3 ? a.val[s] : c[top]);                 // e120(ke+1) = e12min, :180-181
    c[ghost] = t;
    c[ghost + g.sz] = t;
  }
}

}  // namespace

int k_halo_y(udc_handle *h, const int *fields, int nf, int width, int dirs) {
  const Geo &g = h->g;
  if (width > HY) { udc_set_error("k_halo_y: bad arguments"); return 1; }
  if (nf > 16) {      // one launch (and one exchange buffer) carries 16 fields; more go in rounds
    for (int q = 0; q < nf; q += 16)
      if (k_halo_y(h, fields + q, nf - q < 16 ? nf - q : 16, width, dirs)) return 1;
    return 0;
  }
  FieldList fl;
  for (int q = 0; q < nf; ++q) fl.f[q] = h->fields[fields[q]];
  // an exchange started beside the compute stream (k_halo_y_begin) uses the same pack buffers and events: it is joined first
  if (h->halo_async_pending && k_halo_y_join(h)) return 1;
  if (!h->slab) {
    PROF(h, "halo_y");
    hipLaunchKernelGGL(halo_y_wrap_kernel, dim3((g.nx + 63) / 64, 2 * width * nf, g.pz), dim3(64), 0, h->stream,
                       g, fl, width);
    HIP_OK(hipGetLastError());
    return 0;
  }
  // y-slabs: rows travel to the neighbouring ranks (periodic), as 2decomp's exchange_halo_z did
  const size_t count = (size_t)nf * width * g.pz * g.nx;
  if (count > h->halo_cap) { udc_set_error("k_halo_y: pack buffer too small"); return 1; }
  {
    PROF(h, "halo_pack");
    hipLaunchKernelGGL(halo_pack_kernel, dim3((g.nx + 63) / 64, width * nf, g.pz), dim3(64), 0, h->stream, g, fl,
                       width, h->halo_buf[0], h->halo_buf[1], dirs, 0);
    HIP_OK(hipGetLastError());
  }
  {
    PROF(h, "halo_xchg");
    if (comm_neighbours(h, h->halo_buf[0], h->halo_buf[1], h->halo_buf[2], h->halo_buf[3], count, nullptr, dirs)) return 1;
  }
  {
    PROF(h, "halo_unpack");
    hipLaunchKernelGGL(halo_unpack_kernel, dim3((g.nx + 63) / 64, width * nf, g.pz), dim3(64), 0, h->stream, g, fl,
                       width, h->halo_buf[2], h->halo_buf[3], dirs, 0);
    HIP_OK(hipGetLastError());
  }
  return 0;
}

// ---- ghost-row exchange beside the compute stream (y-slabs)
// The producer of the rows is launched twice: first over the tile rows next to the neighbouring ranks, then over the rows in
// between; the exchange of what the first launch wrote -- pack, send / receive, unpack -- is queued in between on the communication
// stream (highest priority: its few workgroups go ahead of the second launch's queue) and the compute stream waits for it after the
// second launch.  The second launch touches no row the exchange reads or writes (its tiles lie >= one tile row inside).
bool halo_overlap(const udc_handle *h, int tile_rows_y) {
  return h->slab && h->comm_stream && !h->no_halo_overlap && tile_rows_y >= 3;
}

// klev0 >= 0: only the levels klev0 .. klev0 + nlev - 1 (device levels; no ghost planes) -- a k-chunk's rows, as the pipelined momentum
// sweep hands them on.  Exchanges begun one after the other share the pack buffers: they queue on the communication stream.
int k_halo_y_begin(udc_handle *h, const int *fields, int nf, int width, double *const *ptrs, int dirs, int klev0, int nlev) {
  const Geo &g = h->g;
  if (width > HY || nf > 16 || !h->slab || !h->comm_stream) { udc_set_error("k_halo_y_begin: bad arguments"); return 1; }
  if (klev0 >= 0 && (nlev < 1 || klev0 + nlev > g.nz)) { udc_set_error("k_halo_y_begin: bad level range"); return 1; }
  FieldList fl;
  for (int q = 0; q < nf; ++q) fl.f[q] = ptrs ? ptrs[q] : h->fields[fields[q]];
  const int kk0 = klev0 >= 0 ? klev0 + HZ : 0, nkk = klev0 >= 0 ? nlev : g.pz;
  const size_t count = (size_t)nf * width * nkk * g.nx;
  if (count > h->halo_cap) { udc_set_error("k_halo_y_begin: pack buffer too small"); return 1; }
  hipStream_t cs = h->comm_stream;
  HIP_OK(hipEventRecord(h->ev_halo_ready, h->stream));
  HIP_OK(hipStreamWaitEvent(cs, h->ev_halo_ready, 0));
  hipLaunchKernelGGL(halo_pack_kernel, dim3((g.nx + 63) / 64, width * nf, nkk), dim3(64), 0, cs, g, fl, width, h->halo_buf[0], h->halo_buf[1], dirs, kk0);
  HIP_OK(hipGetLastError());
  if (comm_neighbours(h, h->halo_buf[0], h->halo_buf[1], h->halo_buf[2], h->halo_buf[3], count, cs, dirs)) return 1;
  hipLaunchKernelGGL(halo_unpack_kernel, dim3((g.nx + 63) / 64, width * nf, nkk), dim3(64), 0, cs, g, fl, width, h->halo_buf[2], h->halo_buf[3], dirs, kk0);
  HIP_OK(hipGetLastError());
  HIP_OK(hipEventRecord(h->ev_halo_done, cs));
  h->halo_async_pending = true;      // (several begins in a row queue behind each other on the communication stream; one join covers them)
  return 0;
}

int k_halo_y_join(udc_handle *h) {
  HIP_OK(hipStreamWaitEvent(h->stream, h->ev_halo_done, 0));
  h->halo_async_pending = false;
  return 0;
}

static BoundaryArgs boundary_args(udc_handle *h) {
  BoundaryArgs a;
  a.u0 = h->fields[UDC_U0]; a.v0 = h->fields[UDC_V0]; a.w0 = h->fields[UDC_W0];
  a.um = h->fields[UDC_UM]; a.vm = h->fields[UDC_VM]; a.wm = h->fields[UDC_WM];
  a.nscal = 0;
  for (int n : h->slots) {
    for (int q = 0; q < 2; ++q) {
      a.kind[a.nscal] = h->slot[n].top; a.val[a.nscal] = h->slot[n].topval;
      a.sv[a.nscal++] = h->fields[(q == 0 ? UDC_SV0 : UDC_SVM) + 3 * n];
    }
  }
  return a;
}

int k_top_bottom(udc_handle *h) {
  const Geo &g = h->g;
  {
    PROF(h, "top_bottom");
    hipLaunchKernelGGL(top_bottom_kernel, dim3((g.nx + 63) / 64, g.py), dim3(64), 0, h->stream, g, h->p,
                       boundary_args(h), 0);
    HIP_OK(hipGetLastError());
  }
  return k_scalar_top_flux(h);
}

// reassure_fluxtop_boundary, src/modboundary.f90:392-431 (free-slip: top rows of um,u0,vm,v0)
int k_top_rows_after_closure(udc_handle *h) {
  const Geo &g = h->g;
  if (h->p.bctopm == UDC_TOP_NOSLIP) return 0;
  PROF(h, "top_rows");
  hipLaunchKernelGGL(top_bottom_kernel, dim3((g.nx + 63) / 64, g.py), dim3(64), 0, h->stream, g, h->p,
                     boundary_args(h), 1);
  HIP_OK(hipGetLastError());
  return 0;
}
