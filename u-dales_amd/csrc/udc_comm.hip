// Inter-slab communication.  One process per GPU; the y-slab neighbours exchange ghost rows and
// the Poisson solver does two all-to-alls (kx <-> y).  Two transports implement the same three
// primitives (neighbour rows, all-to-all blocks, tiny all-reduce):
//
//   * RCCL over xGMI (ncclSend/ncclRecv groups, ncclAllReduce) -- the production path; replaces
//     2decomp-fft's MPI halo exchange and alltoall transposes.
//   * "local group" (ONLY in libudcore_test.so, built with -DUDC_TEST_TRANSPORT; the product library does not
//     contain it): P handles inside ONE process on ONE device, one host thread per virtual
//     rank, buffers exchanged with device-to-device copies behind a pthread barrier.  It exists so
//     that the multi-rank code path (packing, index maps, transposed spectral layout) can be
//     parity-tested on a single-GPU box; everything except the byte mover is shared.
#include "udc_internal.h"
#include <rccl/rccl.h>
#include <pthread.h>
#include <mutex>
#include <cstring>
#include <vector>
#include <cstdlib>

#ifdef UDC_TEST_TRANSPORT
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
// Second test transport, between PROCESSES on one device (mpiexec -n P of the Fortran drop-in build on a one-GPU box, where RCCL
// refuses two ranks per device): every rank owns an outbox in a POSIX shared-memory segment; an exchange is "copy what I send into
// my outbox, barrier, copy what is meant for me out of the others', barrier".  Slow and only for small grids -- the point is that
// the multi-rank route through MPI start-up, the Fortran modules' slab logic, per-rank files and the library's packing runs at all.
struct ShmGroup {
  pthread_barrier_t bar;
  int P, ready;
  size_t cap;                 // bytes per outbox
  // followed by P outboxes
  unsigned char *box(int r) { return reinterpret_cast<unsigned char *>(this) + 4096 + (size_t)r * cap; }
};
constexpr size_t SHM_CAP = (size_t)48 << 20;

struct LocalGroup {
  int P;
  pthread_barrier_t bar;
  // published per rank: send pointers by tag (0 = to-prev rows, 1 = to-next rows, 2 = all-to-all base)
  const double *send[64][3];
  double red[64][4096];
};

static std::mutex g_groups_mu;
static std::vector<LocalGroup *> g_groups;
#endif

#define NCCL_OK(expr)                                                                     \
  do {                                                                                    \
    ncclResult_t r_ = (expr);                                                             \
    if (r_ != ncclSuccess) {                                                              \
      udc_set_error("%s failed: %s (%s:%d)", #expr, ncclGetErrorString(r_), __FILE__,     \
                    __LINE__);                                                            \
      return 1;                                                                           \
    }                                                                                     \
  } while (0)

extern "C" int udc_comm_unique_id(unsigned char id[128]) {
  ncclUniqueId u;
  NCCL_OK(ncclGetUniqueId(&u));
  static_assert(sizeof(u) == 128, "ncclUniqueId size");
  memcpy(id, &u, 128);
  return 0;
}

// UDC_FORCE_COMM=1: build the RCCL communicator even for a single rank and send the "exchanges" of the forced
// slab path (UDC_FORCE_SLAB=1) through ncclSend/ncclRecv to self -- a hardware test of the RCCL plumbing
// (communicator, grouped point-to-point on the library's streams, all-reduce) on a one-GPU box.

extern "C" int udc_comm_init(udc_handle *h, const unsigned char id[128]) {
  NO_OPEN_X(h, "udc_comm_init");
  if (h->cfg.nranks == 1 && !h->sw.force_comm) return 0;
  HIP_OK(hipSetDevice(h->device));
  ncclUniqueId u;
  memcpy(&u, id, 128);
  ncclComm_t c;
  NCCL_OK(ncclCommInitRank(&c, h->cfg.nranks, u, h->cfg.rank));
  h->nccl = (void *)c;
  return 0;
}

#ifdef UDC_TEST_TRANSPORT
extern "C" int udc_local_group_create(int nranks) {
  if (nranks < 1 || nranks > 64) { udc_set_error("udc_local_group_create: 1..64 ranks"); return -1; }
  LocalGroup *g = new LocalGroup();
  g->P = nranks;
  pthread_barrier_init(&g->bar, nullptr, (unsigned)nranks);
  std::lock_guard<std::mutex> lk(g_groups_mu);
  g_groups.push_back(g);
  return (int)g_groups.size();   // ids start at 1
}

extern "C" int udc_comm_init_local(udc_handle *h, int group) {
  NO_OPEN_X(h, "udc_comm_init_local");
  std::lock_guard<std::mutex> lk(g_groups_mu);
  if (group < 1 || group > (int)g_groups.size() || g_groups[group - 1]->P != h->cfg.nranks) {
    udc_set_error("udc_comm_init_local: bad group");
    return 1;
  }
  h->local_group = g_groups[group - 1];
  return 0;
}
#endif

void comm_destroy(udc_handle *h) {
  for (auto &t : h->ctimed) { hipEventDestroy(t.a); hipEventDestroy(t.b); }
  h->ctimed.clear();
  if (h->nccl) { ncclCommDestroy((ncclComm_t)h->nccl); h->nccl = nullptr; }
#ifdef UDC_TEST_TRANSPORT
  if (h->shm_group) { munmap(h->shm_group, 4096 + (size_t)h->cfg.nranks * SHM_CAP); h->shm_group = nullptr; }
#endif
}

#ifdef UDC_TEST_TRANSPORT
extern "C" int udc_comm_init_shm(udc_handle *h, const char *name) {
  if (!h || !name || name[0] != '/') { udc_set_error("udc_comm_init_shm: name must start with '/'"); return 1; }
  const int P = h->cfg.nranks, r = h->cfg.rank;
  const size_t bytes = 4096 + (size_t)P * SHM_CAP;
  ShmGroup *g = nullptr;
  if (r == 0) {
    shm_unlink(name);
    const int fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, (off_t)bytes) != 0) { udc_set_error("udc_comm_init_shm: cannot create %s", name); return 1; }
    g = (ShmGroup *)mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (g == MAP_FAILED) { udc_set_error("udc_comm_init_shm: mmap failed"); return 1; }
    pthread_barrierattr_t at;
    pthread_barrierattr_init(&at);
    pthread_barrierattr_setpshared(&at, PTHREAD_PROCESS_SHARED);
    pthread_barrier_init(&g->bar, &at, (unsigned)P);
    g->P = P; g->cap = SHM_CAP;
    __atomic_store_n(&g->ready, 1, __ATOMIC_RELEASE);
  } else {
    int fd = -1;
    for (int t = 0; t < 60000 && fd < 0; ++t) { fd = shm_open(name, O_RDWR, 0600); if (fd < 0) usleep(1000); }
    struct stat sb;
    for (int t = 0; t < 60000 && fd >= 0; ++t) { if (fstat(fd, &sb) == 0 && (size_t)sb.st_size >= bytes) break; usleep(1000); }
    if (fd < 0) { udc_set_error("udc_comm_init_shm: %s never appeared", name); return 1; }
    g = (ShmGroup *)mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (g == MAP_FAILED) { udc_set_error("udc_comm_init_shm: mmap failed"); return 1; }
    for (int t = 0; t < 60000 && !__atomic_load_n(&g->ready, __ATOMIC_ACQUIRE); ++t) usleep(1000);
    if (!g->ready || g->P != P) { udc_set_error("udc_comm_init_shm: segment of another run"); return 1; }
  }
  pthread_barrier_wait(&g->bar);
  if (r == 0) shm_unlink(name);      // everyone is attached: the name can go
  h->shm_group = g;
  return 0;
}
// one exchange step: `put` bytes from device memory into this rank's outbox; after the barrier `get(g)` copies out of the others'
template <class Get>
static int shm_step(udc_handle *h, hipStream_t st, const void *src, size_t put, Get get) {
  ShmGroup *g = (ShmGroup *)h->shm_group;
  if (put > g->cap) { udc_set_error("test transport (shared memory): %zu bytes exceed the outbox", put); return 1; }
  HIP_OK(hipStreamSynchronize(st));
  if (put) HIP_OK(hipMemcpy(g->box(h->cfg.rank), src, put, hipMemcpyDeviceToHost));
  pthread_barrier_wait(&g->bar);
  if (get(g)) return 1;
  pthread_barrier_wait(&g->bar);
  return 0;
}
#endif

// bookkeeping around every exchange: counters always, a pair of timing events on the exchange's stream when asked for
struct CommScope {
  udc_handle *h; hipStream_t st; hipEvent_t b = nullptr;
  CommScope(udc_handle *h_, hipStream_t st_, int kind) : h(h_), st(st_) {
    if (!h->comm_timing) return;
    // (bounded: a caller that switches the timing on and never collects -- udc_comm_stats mode 1 / 2 -- stops adding pairs here)
    if (h->ctimed.size() >= 65536) return;
    hipEvent_t a = nullptr;
    if (hipEventCreate(&a) != hipSuccess) return;
    if (hipEventCreate(&b) != hipSuccess) { hipEventDestroy(a); b = nullptr; return; }
    hipEventRecord(a, st);
    h->ctimed.push_back({a, b, kind});
  }
  ~CommScope() { if (b) hipEventRecord(b, st); }
};

extern "C" int udc_comm_info(udc_handle *h, int info[8]) {
  if (!h || !info) { udc_set_error("udc_comm_info: null argument"); return 1; }
  for (int q = 0; q < 8; ++q) info[q] = 0;
  info[0] = h->nccl ? 1 : (h->local_group ? 2 : (h->shm_group ? 3 : 0));
  info[1] = info[2] = info[3] = -1;
  if (h->nccl) {
    ncclComm_t c = (ncclComm_t)h->nccl;
    NCCL_OK(ncclCommCount(c, &info[1]));
    NCCL_OK(ncclCommUserRank(c, &info[2]));
    NCCL_OK(ncclCommCuDevice(c, &info[3]));
  }
  NCCL_OK(ncclGetVersion(&info[4]));
  info[5] = h->nch; info[6] = h->cfg.nranks; info[7] = h->cfg.rank;
  return 0;
}

extern "C" int udc_comm_stats(udc_handle *h, int mode, double out[16]) {
  if (!h) { udc_set_error("udc_comm_stats: null handle"); return 1; }
  HIP_OK(hipSetDevice(h->device));
  if (udc_flush_pending(h)) return 1;      // (deferred execution: the counters include the substeps recorded so far)
  if (mode == 1) {
    HIP_OK(hipDeviceSynchronize());
    for (auto &t : h->ctimed) { hipEventDestroy(t.a); hipEventDestroy(t.b); }
    h->ctimed.clear();
    h->cstat = udc_handle::CommStats();
    h->comm_timing = true;
    return 0;
  }
  if (!out) { udc_set_error("udc_comm_stats: null output"); return 1; }
  HIP_OK(hipDeviceSynchronize());
  double ms[2] = {0., 0.};
  for (auto &t : h->ctimed) {
    float m = 0.f;
    if (hipEventElapsedTime(&m, t.a, t.b) == hipSuccess) ms[t.kind] += m;
  }
  for (int q = 0; q < 16; ++q) out[q] = 0.;
  const auto &s = h->cstat;
  out[0] = s.a2a_ops; out[1] = s.a2a_block_bytes; out[2] = s.a2a_bytes; out[3] = ms[0];
  out[4] = s.halo_ops; out[5] = s.halo_prev; out[6] = s.halo_next; out[7] = ms[1];
  out[8] = s.red_ops; out[9] = s.red_doubles;
  if (mode == 2) {
    for (auto &t : h->ctimed) { hipEventDestroy(t.a); hipEventDestroy(t.b); }
    h->ctimed.clear();
    h->comm_timing = false;
  }
  return 0;
}

extern "C" int udc_comm_dry_run(udc_handle *h, int on) {
  if (!h) { udc_set_error("udc_comm_dry_run: null handle"); return 1; }
  h->comm_dry = on != 0;
  return 0;
}

static int need_comm(udc_handle *h) {
  if (h->nccl || h->local_group || h->shm_group) return 0;      // (the last two are only ever set by the test build)
  udc_set_error("multi-rank handle used before udc_comm_init");
  return 1;
}

// neighbour rows: to_prev/to_next are packed send buffers of `count` doubles each;
// from_next receives the next rank's to_prev, from_prev the previous rank's to_next.
int comm_neighbours(udc_handle *h, const double *to_prev, const double *to_next, double *from_prev,
                    double *from_next, size_t count, hipStream_t st, int dirs) {
  if (!st) st = h->stream;
  const int P = h->cfg.nranks, r = h->cfg.rank;
  const int prev = (r + P - 1) % P, next = (r + 1) % P;
  const bool dp = (dirs & 1) != 0, dn = (dirs & 2) != 0;      // rows to the previous rank (and from the next) / to the next (from the previous)
  h->cstat.halo_ops += 1;
  if (dp) h->cstat.halo_prev += (double)(count * sizeof(double));
  if (dn) h->cstat.halo_next += (double)(count * sizeof(double));
  if (h->comm_dry) return 0;
  CommScope scope(h, st, 1);
  if (P == 1 && !h->nccl) {   // single slab driven through the slab code path (UDC_FORCE_SLAB): periodic wrap onto itself
    if (dp) HIP_OK(hipMemcpyAsync(from_next, to_prev, count * sizeof(double), hipMemcpyDeviceToDevice, st));
    if (dn) HIP_OK(hipMemcpyAsync(from_prev, to_next, count * sizeof(double), hipMemcpyDeviceToDevice, st));
    return 0;
  }
  if (need_comm(h)) return 1;
  if (h->nccl) {
    ncclComm_t c = (ncclComm_t)h->nccl;
    // with P == 2 both neighbours are the same peer: sends and receives pair up in posting order,
    // so post "to prev" first on the send side and "from next" first on the receive side.
    const size_t piece = (size_t)1 << 26;      // at most 512 MiB per operation (see comm_alltoall)
    NCCL_OK(ncclGroupStart());
    for (size_t o = 0; o < count; o += piece) {
      const size_t n = count - o < piece ? count - o : piece;
      if (dp) NCCL_OK(ncclSend(to_prev + o, n, ncclDouble, prev, c, st));
      if (dn) NCCL_OK(ncclSend(to_next + o, n, ncclDouble, next, c, st));
      if (dp) NCCL_OK(ncclRecv(from_next + o, n, ncclDouble, next, c, st));
      if (dn) NCCL_OK(ncclRecv(from_prev + o, n, ncclDouble, prev, c, st));
    }
    NCCL_OK(ncclGroupEnd());
    return 0;
  }
#ifdef UDC_TEST_TRANSPORT
  if (h->shm_group) {      // outbox = [to_prev | to_next]
    const size_t nb = count * sizeof(double);
    ShmGroup *sg = (ShmGroup *)h->shm_group;
    if (2 * nb > sg->cap) { udc_set_error("test transport (shared memory): ghost rows exceed the outbox"); return 1; }
    HIP_OK(hipStreamSynchronize(st));
    if (dp) HIP_OK(hipMemcpy(sg->box(r), to_prev, nb, hipMemcpyDeviceToHost));
    if (dn) HIP_OK(hipMemcpy(sg->box(r) + nb, to_next, nb, hipMemcpyDeviceToHost));
    pthread_barrier_wait(&sg->bar);
    if (dp) HIP_OK(hipMemcpy(from_next, sg->box(next), nb, hipMemcpyHostToDevice));
    if (dn) HIP_OK(hipMemcpy(from_prev, sg->box(prev) + nb, nb, hipMemcpyHostToDevice));
    pthread_barrier_wait(&sg->bar);
    return 0;
  }
  LocalGroup *g = (LocalGroup *)h->local_group;
  g->send[r][0] = to_prev; g->send[r][1] = to_next;
  HIP_OK(hipStreamSynchronize(st));
  pthread_barrier_wait(&g->bar);
  if (dp) HIP_OK(hipMemcpyAsync(from_next, g->send[next][0], count * sizeof(double), hipMemcpyDeviceToDevice, st));
  if (dn) HIP_OK(hipMemcpyAsync(from_prev, g->send[prev][1], count * sizeof(double), hipMemcpyDeviceToDevice, st));
  HIP_OK(hipStreamSynchronize(st));
  pthread_barrier_wait(&g->bar);
  return 0;
#else
  (void)prev; (void)next;
  return need_comm(h) ? 1 : 1;
#endif
}

// all-to-all of equal blocks: block d of `send` goes to rank d, arriving as block r of its `recv`
int comm_alltoall(udc_handle *h, const double *send, double *recv, size_t block, hipStream_t st) {
  const int P = h->cfg.nranks, r = h->cfg.rank;
  h->cstat.a2a_ops += 1;
  h->cstat.a2a_block_bytes = (double)(block * sizeof(double));
  h->cstat.a2a_bytes += (double)(block * sizeof(double)) * (P > 1 ? P - 1 : 1);
  if (h->comm_dry) return 0;
  CommScope scope(h, st, 0);
  if (P == 1 && !h->nccl) {
    HIP_OK(hipMemcpyAsync(recv, send, block * sizeof(double), hipMemcpyDeviceToDevice, st));
    return 0;
  }
  if (need_comm(h)) return 1;
  if (h->nccl) {
    ncclComm_t c = (ncclComm_t)h->nccl;
    // own block: plain device copy; the other P-1 blocks use every xGMI link at once
    const int q0 = (P == 1) ? 0 : 1;    // P == 1 only under UDC_FORCE_COMM: the own block goes through RCCL too
    if (q0) HIP_OK(hipMemcpyAsync(recv + (size_t)r * block, send + (size_t)r * block, block * sizeof(double),
                                  hipMemcpyDeviceToDevice, st));
    // one operation carries at most 2^26 doubles (512 MiB): measured on this RCCL, a single send / receive of more than 1 GiB (the
    // 1024 x 512 x 512 transposes in one or two k-chunks on one rank) delivers garbage without an error; pieces go in the same group
    const size_t piece = (size_t)1 << 26;
    NCCL_OK(ncclGroupStart());
    for (int q = q0; q < P; ++q) {
      const int to = (r + q) % P, from = (r + P - q) % P;     // staggered so that pairs differ per step
      for (size_t o = 0; o < block; o += piece) {
        const size_t n = block - o < piece ? block - o : piece;
        NCCL_OK(ncclSend(send + (size_t)to * block + o, n, ncclDouble, to, c, st));
        NCCL_OK(ncclRecv(recv + (size_t)from * block + o, n, ncclDouble, from, c, st));
      }
    }
    NCCL_OK(ncclGroupEnd());
    return 0;
  }
#ifdef UDC_TEST_TRANSPORT
  if (h->shm_group)
    return shm_step(h, st, send, (size_t)P * block * sizeof(double), [&](ShmGroup *sg) {
      for (int s = 0; s < P; ++s)
        HIP_OK(hipMemcpy(recv + (size_t)s * block, sg->box(s) + (size_t)r * block * sizeof(double), block * sizeof(double), hipMemcpyHostToDevice));
      return 0;
    });
  LocalGroup *g = (LocalGroup *)h->local_group;
  g->send[r][2] = send;
  HIP_OK(hipStreamSynchronize(st));
  pthread_barrier_wait(&g->bar);
  for (int s = 0; s < P; ++s)
    HIP_OK(hipMemcpyAsync(recv + (size_t)s * block, g->send[s][2] + (size_t)r * block, block * sizeof(double),
                          hipMemcpyDeviceToDevice, st));
  HIP_OK(hipStreamSynchronize(st));
  pthread_barrier_wait(&g->bar);
  return 0;
#else
  return 1;
#endif
}

// in-place all-reduce of n doubles held in device memory `buf`; op 0 = max, 1 = sum
int comm_allreduce(udc_handle *h, double *buf, int n, int op) {
  if (h->cfg.nranks == 1 && !h->nccl) return 0;
  h->cstat.red_ops += 1; h->cstat.red_doubles += n;
  if (need_comm(h)) return 1;
  if (h->nccl) {
    NCCL_OK(ncclAllReduce(buf, buf, (size_t)n, ncclDouble, op == 0 ? ncclMax : ncclSum, (ncclComm_t)h->nccl,
                          h->stream));
    return 0;
  }
#ifdef UDC_TEST_TRANSPORT
  if (h->shm_group)
    return shm_step(h, h->stream, buf, (size_t)n * sizeof(double), [&](ShmGroup *sg) {
      std::vector<double> acc(n);
      for (int q = 0; q < n; ++q) {
        double v = reinterpret_cast<const double *>(sg->box(0))[q];
        for (int s = 1; s < sg->P; ++s) {
          const double w = reinterpret_cast<const double *>(sg->box(s))[q];
          v = op == 0 ? (w > v ? w : v) : v + w;
        }
        acc[q] = v;
      }
      HIP_OK(hipMemcpy(buf, acc.data(), (size_t)n * sizeof(double), hipMemcpyHostToDevice));
      return 0;
    });
  LocalGroup *g = (LocalGroup *)h->local_group;
  const int P = h->cfg.nranks, r = h->cfg.rank;
  std::vector<double> out(4096);
  for (int o = 0; o < n; o += 4096) {          // the test transport's staging rows hold 4096 values: go in chunks
    const int m = n - o < 4096 ? n - o : 4096;
    HIP_OK(hipMemcpyAsync(g->red[r], buf + o, m * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_OK(hipStreamSynchronize(h->stream));
    pthread_barrier_wait(&g->bar);
    for (int q = 0; q < m; ++q) {
      double v = g->red[0][q];
      for (int s = 1; s < P; ++s) v = op == 0 ? (g->red[s][q] > v ? g->red[s][q] : v) : v + g->red[s][q];
      out[q] = v;
    }
    pthread_barrier_wait(&g->bar);
    HIP_OK(hipMemcpyAsync(buf + o, out.data(), m * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIP_OK(hipStreamSynchronize(h->stream));
  }
  return 0;
#else
  return 1;
#endif
}
