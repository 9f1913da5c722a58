// libudcore: handle, residency, orchestration of the reference's call surface.
#include "udc_internal.h"
#include <cstdarg>
#include <cstring>
#include <cmath>
#include <cstdlib>
#include <utility>
#include <algorithm>

static thread_local char g_err[512] = "";

void udc_set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int udc_flush_pending(udc_handle *h);
#define ENTRY(h)                                        \
  do {                                                  \
    if (!(h)) { udc_set_error("null handle"); return 1; } \
    HIP_OK(hipSetDevice((h)->device));                  \
  } while (0)
#define ENTRY_FLUSH(h)                                  \
  do {                                                  \
    ENTRY(h);                                           \
    if (udc_flush_pending(h)) return 1;                 \
  } while (0)

extern "C" const char *udc_last_error(void) { return g_err; }
extern "C" int udc_version(void) { return 100; }

// ------------------------------------------------------------------------------ profiling
ProfScope::ProfScope(udc_handle *h_, const char *name) : h(h_), id(-1) {
  if (!h->prof) return;
  if (h->prof_focus_on && (strncmp(name, h->prof_focus.c_str(), h->prof_focus.size()) != 0 || (h->substep_seq - h->prof_phase - 1) % h->prof_every != 0)) return;
  auto it = h->prof_ids.find(name);
  if (it == h->prof_ids.end()) {
    id = (int)h->prof_names.size();
    h->prof_names.push_back(name);
    h->prof_ids[name] = id;
  } else {
    id = it->second;
  }
  // One marker between consecutive launches: the end event of the previous scope doubles as the start of this one
  // (event records are not free on the GPU timeline: two per launch cost ~4 % of a 256^3 substep, one costs ~2 %).
  // Events come from a pool.
  if (h->prof_chain) {
    a = h->prof_chain;
    own_a = false;
  } else {
    a = prof_take(h);
    own_a = true;
    hipEventRecord(a, h->stream);
  }
}
ProfScope::~ProfScope() {
  if (id < 0) return;
  b = prof_take(h);
  hipEventRecord(b, h->stream);
  h->prof_events.push_back({a, b, id, own_a});
  h->prof_chain = h->prof_focus_on ? nullptr : b;      // focus mode: the next timed launch is not adjacent
}

hipEvent_t prof_take(udc_handle *h) {
  hipEvent_t e;
  if (h->prof_pool.empty()) hipEventCreate(&e);
  else { e = h->prof_pool.back(); h->prof_pool.pop_back(); }
  return e;
}

static void prof_drain(udc_handle *h) {
  for (auto &e : h->prof_events) {
    hipEventSynchronize(e.b);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e.a, e.b);
    auto &acc = h->prof_acc[e.name];
    acc.first += ms;
    acc.second += 1;
  }
  for (auto &e : h->prof_events) {          // every event is the `b` of exactly one entry, or an owned `a`
    if (e.own_a) h->prof_pool.push_back(e.a);
    h->prof_pool.push_back(e.b);
  }
  h->prof_events.clear();
  h->prof_chain = nullptr;                  // host-side pause: the next scope records its own start
}

extern "C" int udc_profile_enable(udc_handle *h, int on) {
  if (!h) return 1;
  prof_drain(h);
  if (on == 2 && h->prof_focus.empty()) { udc_set_error("udc_profile_enable(2): call udc_profile_focus first"); return 1; }
  h->prof = on != 0;
  h->prof_focus_on = on == 2;
  return 0;
}
extern "C" int udc_profile_focus(udc_handle *h, const char *name_prefix) {
  if (!h || !name_prefix) { udc_set_error("udc_profile_focus: null argument"); return 1; }
  h->prof_focus = name_prefix;
  return 0;
}
extern "C" int udc_profile_every(udc_handle *h, int n) {
  if (!h || n < 1) { udc_set_error("udc_profile_every: null handle or n < 1"); return 1; }
  h->prof_every = n;
  h->prof_phase = h->substep_seq;       // the next fused substep is a timed one, then every n-th after it
  return 0;
}
extern "C" int udc_profile_reset(udc_handle *h) {
  prof_drain(h);
  h->prof_acc.clear();
  return 0;
}
extern "C" int udc_profile_get(udc_handle *h, int cap, char names[][64], double *total_ms, int *launches) {
  prof_drain(h);
  int n = 0;
  for (auto &kv : h->prof_acc) {
    if (n < cap) {
      snprintf(names[n], 64, "%s", h->prof_names[kv.first].c_str());
      total_ms[n] = kv.second.first;
      launches[n] = kv.second.second;
    }
    ++n;
  }
  return n;
}

// ------------------------------------------------------------------------------ lifetime
static int alloc_field(udc_handle *h, int id) {
  if ((int)h->fields.size() <= id) h->fields.resize(id + 1, nullptr);
  if (h->fields[id]) return 0;
  double *p = nullptr;
  HIP_OK(hipMalloc(&p, sizeof(double) * h->g.n));
  HIP_OK(hipMemsetAsync(p, 0, sizeof(double) * h->g.n, h->stream));
  h->fields[id] = p;
  return 0;
}

static int env_int(const char *name, int dflt) {
  const char *e = getenv(name);
  return (e && *e) ? atoi(e) : dflt;
}
// The one place the library reads its environment (include/udcore.h, "Environment").
void udc_read_switches(Switches &sw) {
  sw.force_slab = env_int("UDC_FORCE_SLAB", 0) != 0;
  sw.force_comm = env_int("UDC_FORCE_COMM", 0) != 0;
  sw.halo_overlap = env_int("UDC_HALO_OVERLAP", 1) != 0;
  sw.mom_pipe = env_int("UDC_MOM_PIPE", 2);
  sw.a2a_chunks = env_int("UDC_A2A_CHUNKS", 0);
  sw.fft_fused = env_int("UDC_FFT_FUSED", 1) != 0;
  sw.own_fwd = env_int("UDC_OWN_FWD", -1);
  sw.div_in_fft = env_int("UDC_DIV_IN_FFT", 1) != 0;
  sw.ptotal = env_int("UDC_PTOTAL", 1) != 0;
  sw.p_transpose = env_int("UDC_P_TRANSPOSE", 1) != 0;
  sw.sv_inline = env_int("UDC_SV_INLINE", 1) != 0;
  sw.no_fold = env_int("UDC_NO_FOLD", 0) != 0;
  sw.no_alias = env_int("UDC_NO_ALIAS", 0) != 0;
  sw.ek_always = env_int("UDC_EK_ALWAYS", 0) != 0;
  sw.scalar_pair = env_int("UDC_SCALAR_PAIR", 1) != 0;
  sw.thomas = env_int("UDC_THOMAS", -1);
  sw.thomas_pair = env_int("UDC_THOMAS_PAIR", 1) != 0;
  sw.thomas_mirror_min = env_int("UDC_THOMAS_MIRROR_MIN", 256);
  sw.nat_reg = env_int("UDC_NAT_REG", 1) != 0;
  sw.slab_yreg = env_int("UDC_SLAB_YREG", 1) != 0;
}

static int create_on_device(const udc_config *cfg, udc_handle *h);
extern "C" int udc_create(const udc_config *cfg, udc_handle **out) {
  if (!cfg || !out) { udc_set_error("udc_create: null argument"); return 1; }
  if (cfg->itot < 4 || cfg->jtot < 4 || cfg->ktot < 3) { udc_set_error("udc_create: grid too small"); return 1; }
  if (cfg->itot % 2 || cfg->jtot % 2) { udc_set_error("udc_create: itot and jtot must be even (half-complex FFT, src/modpois.f90:482-487)"); return 1; }
  if (cfg->nranks < 1 || cfg->rank < 0 || cfg->rank >= cfg->nranks) { udc_set_error("udc_create: bad rank/nranks"); return 1; }
  if (cfg->jtot % cfg->nranks) { udc_set_error("udc_create: jtot must be divisible by nranks (src/modstartup.f90:730-760)"); return 1; }
  if (cfg->nranks > 64) { udc_set_error("udc_create: at most 64 slabs"); return 1; }
  if ((cfg->jtot / cfg->nranks) < 2 * HY) { udc_set_error("udc_create: slab thinner than the ghost width"); return 1; }
  if (cfg->nsv < 0 || cfg->nsv > 16) { udc_set_error("udc_create: nsv out of range"); return 1; }
  if (cfg->lbottom && !(cfg->z0 > 0.)) { udc_set_error("udc_create: lbottom needs a roughness length z0 > 0 (src/modwallfunctions.f90:314)"); return 1; }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
    udc_set_error("udc_create: no HIP device visible -- libudcore has no CPU fallback");
    return 1;
  }
  // device < 0: the library deals the ranks round over the node's devices (one rank per GPU)
  const int device = cfg->device >= 0 ? cfg->device : cfg->rank % ndev;
  if (device >= ndev) { udc_set_error("udc_create: device index beyond the visible HIP devices"); return 1; }
  udc_handle *h = new udc_handle();
  h->device = device;
  // whatever fails from here on: the handle, its stream and what it has allocated so far go back (udc_destroy takes a half-built handle)
  if (create_on_device(cfg, h)) { udc_destroy(h); return 1; }
  *out = h;
  return 0;
}

// &BC BCxm = 2 (udc_xopen.hip): the handle's rows carry one ghost column at either end
extern "C" int udc_create_open_x(const udc_config *cfg, const double *uprof, const double *vprof, udc_handle **out) {
  if (!cfg || !out || !uprof || !vprof) { udc_set_error("udc_create_open_x: null argument"); return 1; }
  if (cfg->nranks != 1) { udc_set_error("udc_create_open_x: one rank (the y-slab exchanges know nothing of the outlet's planes)"); return 1; }
  if (cfg->bctopm != UDC_TOP_PRESSURE) {
    udc_set_error("udc_create_open_x: BCtopm must be 3 (the reference opens the lid with BCxm = 2 itself, src/modstartup.f90:845-848)");
    return 1;
  }
  if (cfg->nsv < 0 || cfg->nsv > 12) { udc_set_error("udc_create_open_x: nsv out of range (at most 12 passive scalars)"); return 1; }
  if (cfg->sgs == UDC_SGS_ONEEQN) { udc_set_error("udc_create_open_x: the one-equation closure is not offered with open x boundaries"); return 1; }
  if (cfg->itot < 8 || cfg->itot % 2) { udc_set_error("udc_create_open_x: itot even and >= 8"); return 1; }
  // one ghost column at either end of a row; two with passive scalars (advecc_kappa reads i-2 .. i+1: ib-2, ib-1 / ie+1, ie+2)
  const int xg = cfg->nsv > 0 ? 2 : 1;
  udc_config c2 = *cfg;
  c2.itot = cfg->itot + 2 * xg;
  // (udc_create's own checks, then the same construction with the ghost columns switched on)
  if (cfg->jtot < 4 || cfg->ktot < 3 || cfg->jtot % 2) { udc_set_error("udc_create_open_x: grid too small, or jtot odd"); return 1; }
  if (cfg->jtot < 2 * HY) { udc_set_error("udc_create_open_x: slab thinner than the ghost width"); return 1; }
  if (cfg->lbottom && !(cfg->z0 > 0.)) { udc_set_error("udc_create_open_x: lbottom needs a roughness length z0 > 0"); return 1; }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { udc_set_error("udc_create_open_x: no HIP device visible -- libudcore has no CPU fallback"); return 1; }
  const int device = cfg->device >= 0 ? cfg->device : 0;
  if (device >= ndev) { udc_set_error("udc_create_open_x: device index beyond the visible HIP devices"); return 1; }
  udc_handle *h = new udc_handle();
  h->device = device;
  h->xg = xg;
  if (create_on_device(&c2, h) || xo_init(h, uprof, vprof)) { udc_destroy(h); return 1; }
  if (h->slab) { udc_set_error("udc_create_open_x: not with UDC_FORCE_SLAB"); udc_destroy(h); return 1; }
  *out = h;
  return 0;
}
// the inflow profiles again (a driver that makes its handle before prof.inp is in)
extern "C" int udc_set_open_x_profile(udc_handle *h, const double *uprof, const double *vprof) {
  ENTRY_FLUSH(h);
  if (!h->xg || !uprof || !vprof) { udc_set_error("udc_set_open_x_profile: a handle of udc_create_open_x and both profiles"); return 1; }
  const size_t nk = (size_t)h->g.nz + 2;
  HIP_OK(hipStreamSynchronize(h->stream));
  HIP_OK(hipMemcpy(h->xo_prof, uprof, sizeof(double) * nk, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(h->xo_prof + nk, vprof, sizeof(double) * nk, hipMemcpyHostToDevice));
  return 0;
}
// the outlet's convection speed uouttot (src/modboundary.f90:141-160): a constant (a prescribed volume flow's ubulk), or -- wlev
// [ktot] given -- sum_k wlev(k) u0av(k) of the state every substep starts from, wlev(k) = dzf(k) / (zh(ke+1) - zh(kb+1)).
// hold_first: the first substep after this call still convects with `uouttot` -- the reference's start-up forms u0av BEFORE its
// first `boundary` (src/modstartup.f90:1601, src/program.f90:118) and the first substep's `boundary` still reads that one
extern "C" int udc_set_open_x_outflow(udc_handle *h, const double *wlev, double uouttot, int hold_first) {
  ENTRY_FLUSH(h);
  if (!h->xg) { udc_set_error("udc_set_open_x_outflow: not a handle of udc_create_open_x"); return 1; }
  HIP_OK(hipStreamSynchronize(h->stream));
  HIP_OK(hipMemcpy(h->bcx_uout_dev, &uouttot, sizeof(double), hipMemcpyHostToDevice));
  h->bcx_uout = uouttot;
  h->bcx_uout_avg = wlev != nullptr;
  h->xo_hold = wlev != nullptr && hold_first != 0;
  if (wlev) {
    if (!h->bcx_wlev) HIP_OK(hipMalloc(&h->bcx_wlev, sizeof(double) * h->g.nz));
    HIP_OK(hipMemcpy(h->bcx_wlev, wlev, sizeof(double) * h->g.nz, hipMemcpyHostToDevice));
  }
  return 0;
}

static int create_on_device(const udc_config *cfg, udc_handle *h) {
  udc_read_switches(h->sw);
  h->cfg = *cfg;
  HIP_OK(hipSetDevice(h->device));
  if (!h->stream) HIP_OK(hipStreamCreate(&h->stream));      // (a solver-only handle works on its owner's stream: udc_create_open_x)
  Geo &g = h->g;
  g.nx = cfg->itot; g.ny = cfg->jtot / cfg->nranks; g.nz = cfg->ktot;
  g.xg = h->xg;
  h->jtot = cfg->jtot;
  h->no_fold = h->sw.no_fold;
  h->no_alias = h->sw.no_alias;
  h->ek_always = h->sw.ek_always;
  h->no_div_in_fft = !h->sw.div_in_fft;
  h->slab = cfg->nranks > 1 || h->sw.force_slab;
  g.py = g.ny + 2 * HY; g.pz = g.nz + 2 * HZ;
  // row stride: rows of a power-of-two nx put every row of a tile column on the same few L2 channels and sets, and the
  // stencil kernels' halo lines evict each other (closure at 1024x512x512: 2.2 x its algorithmic reads, 3.2 -> 2.55 ms with
  // the padding; no effect at nx = 256).  16 doubles (one 128-B line) of padding after the nx cells of a row for
  // nx >= 512 and a multiple of 256 (udc_tuning.h); never read (x is periodic by index wrap).
  // (open x boundaries: itot + 2 columns, padded to whole 128-byte lines so that rows start on one)
  const int xpad = h->xg ? (16 - g.nx % 16) % 16 : tune::row_padding(g.nx);
  g.sy = g.nx + xpad; g.sz = (long)g.sy * g.py; g.n = g.sz * g.pz;
  h->p = Params{cfg->numol, cfg->prandtlmoli, cfg->prandtli, cfg->c_vreman, cfg->csz,
                cfg->uinf, cfg->vinf, cfg->sgs, cfg->bctopm, cfg->lbottom ? 1 : 0, cfg->z0};

  // metrics exactly as src/modglobal.f90:812-838
  const int nk = g.nz + 2;
  std::vector<double> hm(13 * nk, 0.0);
  double *dzf = &hm[0], *dzfi = dzf + nk, *dzfi5 = dzfi + nk, *dzfiq = dzfi5 + nk, *dzf2 = dzfiq + nk;
  double *dzhi = dzf2 + nk, *dzhiq = dzhi + nk, *dzh2i = dzhiq + nk, *dzh = dzh2i + nk;
  for (int k = 0; k < nk; ++k) {
    dzf[k] = cfg->dzf[k];
    dzfi[k] = 1. / dzf[k]; dzfi5[k] = 0.5 * dzfi[k]; dzfiq[k] = 0.25 * dzfi[k]; dzf2[k] = dzf[k] * dzf[k];
  }
  for (int k = 1; k < nk; ++k) {
    dzh[k] = cfg->dzh[k];
    dzhi[k] = 1. / dzh[k]; dzhiq[k] = 0.25 * dzhi[k]; dzh2i[k] = dzhi[k] * dzhi[k];
  }
  h->zsize = 0.;
  for (int k = 1; k <= g.nz; ++k) h->zsize += dzf[k];      // zh(ke+1), src/modglobal.f90:747-750
  h->dzhi_top = dzhi[g.nz + 1];                             // dzhi(ke+1): the open lid's rows (k_lid_*)
  double *mlen = &hm[11 * nk];
  // delta(i,k) = (dxf(i)*dy*dzf(k))**(1/3), src/modglobal.f90:793-797 (uniform x)
  for (int k = 0; k < nk; ++k) mlen[k] = cfg->csz * pow(cfg->dx * cfg->dy * dzf[k], 1. / 3.);
  for (int k = 0; k < nk; ++k) hm[12 * nk + k] = pow(cfg->dx * cfg->dy * dzf[k], 1. / 3.);
  dzh[0] = dzh[1]; dzhi[0] = dzhi[1]; dzhiq[0] = dzhiq[1]; dzh2i[0] = dzh2i[1];
  HIP_OK(hipMalloc(&h->metrics_dev, sizeof(double) * hm.size()));
  HIP_OK(hipMemcpy(h->metrics_dev, hm.data(), sizeof(double) * hm.size(), hipMemcpyHostToDevice));
  Metrics &m = h->m;
  double *b = h->metrics_dev;
  m.dzf = b; m.dzfi = b + nk; m.dzfi5 = b + 2 * nk; m.dzfiq = b + 3 * nk; m.dzf2 = b + 4 * nk;
  m.dzhi = b + 5 * nk; m.dzhiq = b + 6 * nk; m.dzh2i = b + 7 * nk; m.dzh = b + 8 * nk;
  m.dpdxl = b + 9 * nk; m.dpdyl = b + 10 * nk; m.mlen = b + 11 * nk; m.delta = b + 12 * nk;
  m.dx = cfg->dx; m.dy = cfg->dy;
  m.dxi = 1. / cfg->dx; m.dyi = 1. / cfg->dy;
  m.dx2 = cfg->dx * cfg->dx; m.dy2 = cfg->dy * cfg->dy;
  m.dxiq = 0.25 * m.dxi; m.dyiq = 0.25 * m.dyi;
  m.dx2i = m.dxi * m.dxi; m.dy2i = m.dyi * m.dyi;
  m.dxi5 = 0.5 * m.dxi; m.dyi5 = 0.5 * m.dyi;

  for (int f = UDC_U0; f <= UDC_EKH; ++f) {
    if (h->poisson_only && f != UDC_P) { if ((int)h->fields.size() <= f) h->fields.resize(f + 1, nullptr); continue; }
    if (alloc_field(h, f)) return 1;
  }
  for (int n = 0; n < cfg->nsv; ++n) {
    for (int q = 0; q < 3; ++q)
      if (alloc_field(h, UDC_SV0 + 3 * n + q)) return 1;
    h->slots.push_back(n);
  }
  h->red_cap = std::max((size_t)4096, (size_t)16 * (g.nz + 2));      // udc_slab_averages: 16 fields x (ktot + 1) levels
  HIP_OK(hipMalloc(&h->red, sizeof(double) * h->red_cap));
  HIP_OK(hipHostMalloc(&h->red_host, sizeof(double) * h->red_cap));
  if (h->xg) {      // open x boundaries: the solve runs on a handle of its own, twice as wide (udc_xopen.hip)
    udc_handle *hp = new udc_handle();
    h->xpois = hp;
    hp->device = h->device; hp->stream = h->stream; hp->poisson_only = true;
    udc_config c2 = *cfg;
    c2.itot = 2 * (cfg->itot - 2 * h->xg); c2.nsv = 0;
    if (create_on_device(&c2, hp)) return 1;
  } else if (h->slab) { if (pois_slab_init(h)) return 1; }
  else if (pois_init(h)) return 1;
  if (h->slab) {
    h->halo_cap = (size_t)16 * HY * g.nx * g.pz;
    for (int q = 0; q < 4; ++q) HIP_OK(hipMalloc(&h->halo_buf[q], sizeof(double) * h->halo_cap));
  }
  HIP_OK(hipStreamSynchronize(h->stream));
  return 0;
}

extern "C" int udc_destroy(udc_handle *h) {
  if (!h) return 0;
  hipSetDevice(h->device);
  h->pend.clear();
  if (h->stream) hipStreamSynchronize(h->stream);
  prof_drain(h);
  for (hipEvent_t e : h->prof_pool) hipEventDestroy(e);
  xo_destroy(h);
  pois_destroy(h);
  comm_destroy(h);
  ibm_destroy(h);
  stats_destroy(h);
  ibm_wf_destroy(h);
  for (double *&p : h->bcx_east) if (p) { hipFree(p); p = nullptr; }
  if (h->bcx_prof) { hipFree(h->bcx_prof); h->bcx_prof = nullptr; }
  if (h->bcx_uout_dev) { hipFree(h->bcx_uout_dev); h->bcx_uout_dev = nullptr; }
  if (h->bcx_wlev) { hipFree(h->bcx_wlev); h->bcx_wlev = nullptr; }
  for (int q = 0; q < 4; ++q) if (h->halo_buf[q]) hipFree(h->halo_buf[q]);
  for (auto &f : h->level_forcings) { if (f.A) hipFree(f.A); if (f.stage) hipHostFree(f.stage); if (f.copied) hipEventDestroy(f.copied); }
  for (double *p : h->fields) if (p) hipFree(p);
  if (h->metrics_dev) hipFree(h->metrics_dev);
  for (int q = 0; q < 3; ++q) if (h->bottom_diag[q]) hipFree(h->bottom_diag[q]);
  if (h->red) hipFree(h->red);
  if (h->red_host) hipHostFree(h->red_host);
  if (h->chk_host) hipHostFree(h->chk_host);
  if (h->chk_dev) hipFree(h->chk_dev);
  if (h->ev_chk) hipEventDestroy(h->ev_chk);
  if (h->thlpcar) hipFree(h->thlpcar);
  if (h->dthv_top) hipFree(h->dthv_top);
  if (h->mt) hipFree(h->mt);
  if (h->shift_tab) hipFree(h->shift_tab);
  for (auto &s : h->svsrc) if (s.d) hipFree(s.d);
  if (h->ug) hipFree(h->ug);
  if (h->lev_part) hipFree(h->lev_part);
  if (h->lev_sum) hipFree(h->lev_sum);
  if (h->lev_sum16) hipFree(h->lev_sum16);
  if (h->stream && !h->poisson_only) hipStreamDestroy(h->stream);
  delete h;
  return 0;
}

extern "C" int udc_sync(udc_handle *h) {
  ENTRY_FLUSH(h);
  HIP_OK(hipStreamSynchronize(h->stream));
  h->prof_chain = nullptr;
  return 0;
}

// ------------------------------------------------------------------------------ residency
static int field_ptr(udc_handle *h, int field, double **p) {
  if (field < 0 || field >= (int)h->fields.size() || !h->fields[field]) {
    udc_set_error("unknown or unallocated field id %d", field);
    return 1;
  }
  *p = h->fields[field];
  return 0;
}

static int tend_clean(udc_handle *h);
static int um_materialise(udc_handle *h);

static int copy3d_ptr(udc_handle *h, int field, double *dev, double *host, const int lb[3], const int ub[3], bool up);
static int copy3d(udc_handle *h, int field, double *host, const int lb[3], const int ub[3], bool up) {
  double *dev;
  if (field_ptr(h, field, &dev)) return 1;
  if (((field >= UDC_UP && field <= UDC_WP) || (field >= UDC_SV0 && (field - UDC_SV0) % 3 == 2)) && tend_clean(h)) return 1;
  if (field >= UDC_UM && field <= UDC_WM && um_materialise(h)) return 1;
  if (field >= UDC_UM && field <= UDC_WM && field_ptr(h, field, &dev)) return 1;
  if (up) h->halos_fresh = h->boundary_fresh = h->thermo_fresh = false;
  return copy3d_ptr(h, field, dev, host, lb, ub, up);
}

static int copy3d_ptr(udc_handle *h, int field, double *dev, double *host, const int lb[3], const int ub[3], bool up) {
  const Geo &g = h->g;
  const int hnx = ub[0] - lb[0] + 1, hny = ub[1] - lb[1] + 1;
  // (open x boundaries: the device row starts xg columns west of the reference's ib, udc_xopen.hip)
  const int xg = g.xg, itot = g.nx - 2 * xg;
  int i0 = lb[0] > 1 - xg ? lb[0] : 1 - xg, i1 = ub[0] < itot + xg ? ub[0] : itot + xg;
  int j0 = lb[1] > 1 - HY ? lb[1] : 1 - HY, j1 = ub[1] < g.ny + HY ? ub[1] : g.ny + HY;
  int k0 = lb[2] > 1 - HZ ? lb[2] : 1 - HZ, k1 = ub[2] < g.nz + HZ ? ub[2] : g.nz + HZ;
  if (i1 < i0 || j1 < j0 || k1 < k0) { udc_set_error("field %d: empty overlap with host bounds", field); return 1; }
  hipMemcpy3DParms p;
  memset(&p, 0, sizeof(p));
  hipPitchedPtr hp = make_hipPitchedPtr((void *)host, (size_t)hnx * 8, (size_t)hnx * 8, (size_t)hny);
  hipPitchedPtr dp = make_hipPitchedPtr((void *)dev, (size_t)g.sy * 8, (size_t)g.sy * 8, (size_t)g.py);
  hipPos hpos = make_hipPos((size_t)(i0 - lb[0]) * 8, (size_t)(j0 - lb[1]), (size_t)(k0 - lb[2]));
  hipPos dpos = make_hipPos((size_t)(i0 - 1 + xg) * 8, (size_t)(j0 - 1 + HY), (size_t)(k0 - 1 + HZ));
  p.extent = make_hipExtent((size_t)(i1 - i0 + 1) * 8, (size_t)(j1 - j0 + 1), (size_t)(k1 - k0 + 1));
  if (up) { p.srcPtr = hp; p.srcPos = hpos; p.dstPtr = dp; p.dstPos = dpos; p.kind = hipMemcpyHostToDevice; }
  else    { p.srcPtr = dp; p.srcPos = dpos; p.dstPtr = hp; p.dstPos = hpos; p.kind = hipMemcpyDeviceToHost; }
  HIP_OK(hipStreamSynchronize(h->stream));
  HIP_OK(hipMemcpy3D(&p));
  if (!up && !xg) {
    // the reference's x ghost columns are periodic images (src/modboundary.f90:516-529):
    // rebuild them on the host so that untouched host routines see what they expect
    const long hsy = hnx, hsz = (long)hnx * hny;
    for (int k = k0; k <= k1; ++k)
      for (int j = j0; j <= j1; ++j) {
        double *row = host + (long)(j - lb[1]) * hsy + (long)(k - lb[2]) * hsz - lb[0];
        for (int i = lb[0]; i < 1; ++i) row[i] = row[i + g.nx];
        for (int i = g.nx + 1; i <= ub[0]; ++i) row[i] = row[i - g.nx];
      }
  }
  return 0;
}

extern "C" int udc_field_upload(udc_handle *h, int field, const double *host, const int lb[3], const int ub[3]) {
  ENTRY_FLUSH(h);
  if (copy3d(h, field, const_cast<double *>(host), lb, ub, true)) return 1;
  if (field == UDC_EKM || field == UDC_EKH) h->ek_stale = false;
  if (field == UDC_EKH) h->ekh_stale = false;
  if (field == UDC_P) h->p_scratch = false;
  if (h->xg && xo_capture_east(h, field, host, lb, ub)) return 1;      // v, w at ie+1: the convective outlet's own state
  if (h->scal_bcx == 2 && field >= UDC_SV0 && (field - UDC_SV0) % 3 == 0)      // sv0 with its east ghost columns (BCxs = 2)
    return k_scalar_bcx_capture(h, (field - UDC_SV0) / 3, host, lb, ub);
  return 0;
}

extern "C" int udc_set_scalar_bcx_outflow(udc_handle *h, const double *wlev) {
  ENTRY_FLUSH(h);
  if (h->scal_bcx != 2 || !h->bcx_uout_dev) { udc_set_error("udc_set_scalar_bcx_outflow: call udc_set_scalar_bcx(h, 2, ...) first"); return 1; }
  if (!wlev) { h->bcx_uout_avg = false; return 0; }
  HIP_OK(hipStreamSynchronize(h->stream));
  if (!h->bcx_wlev) HIP_OK(hipMalloc(&h->bcx_wlev, sizeof(double) * h->g.nz));
  HIP_OK(hipMemcpy(h->bcx_wlev, wlev, sizeof(double) * h->g.nz, hipMemcpyHostToDevice));
  h->bcx_uout_avg = true;
  return 0;
}

extern "C" int udc_set_scalar_bcx(udc_handle *h, int bcxs, const double *svprof, double uouttot) {
  NO_OPEN_X(h, "udc_set_scalar_bcx");
  ENTRY_FLUSH(h);
  if (bcxs != 1 && bcxs != 2) { udc_set_error("udc_set_scalar_bcx: BCxs must be 1 (periodic) or 2 (inflow profile, convective outflow)"); return 1; }
  const Geo &g = h->g;
  h->scal_bcx = bcxs;
  if (bcxs == 1) return 0;
  if (!svprof) { udc_set_error("udc_set_scalar_bcx: the inflow profiles svprof[nsv][ktot+2] are needed"); return 1; }
  if (h->cfg.nsv < 1) { udc_set_error("udc_set_scalar_bcx: no passive scalars"); return 1; }
  if (g.nx < 8) { udc_set_error("udc_set_scalar_bcx: itot >= 8"); return 1; }
  h->bcx_uout = uouttot;
  const size_t np = (size_t)h->cfg.nsv * (g.nz + 2);
  HIP_OK(hipStreamSynchronize(h->stream));
  if (!h->bcx_prof) HIP_OK(hipMalloc(&h->bcx_prof, sizeof(double) * np));
  HIP_OK(hipMemcpy(h->bcx_prof, svprof, sizeof(double) * np, hipMemcpyHostToDevice));
  h->bcx_prof_host.assign(svprof, svprof + np);
  if (!h->bcx_uout_dev) HIP_OK(hipMalloc(&h->bcx_uout_dev, sizeof(double)));
  HIP_OK(hipMemcpy(h->bcx_uout_dev, &uouttot, sizeof(double), hipMemcpyHostToDevice));
  h->bcx_uout_avg = false;
  for (int n = 0; n < h->cfg.nsv && n < 13; ++n)
    if (!h->bcx_east[n]) {
      HIP_OK(hipMalloc(&h->bcx_east[n], sizeof(double) * 2 * g.pz * g.py));
      HIP_OK(hipMemset(h->bcx_east[n], 0, sizeof(double) * 2 * g.pz * g.py));
    }
  return 0;
}
// ekm / ekh in memory are those of the last substep that wrote them: RK stages 1 and 2 of a substep whose closure ran inside
// the momentum sweep keep them in LDS only (udc_mom_fused.hip).  Nothing of the reference's loop looks at them there; a caller
// that does is told so instead of being handed an older substep's values.
static int ek_current(udc_handle *h, const char *who, bool ekh_only = false) {
  if (!h->ek_stale && !(h->ekh_stale && ekh_only)) return 0;
  udc_set_error("%s: ekm / ekh were not (both) written by the last substep -- an RK stage 1 or 2 in which nothing of the reference's "
                "loop reads them; set UDC_EK_ALWAYS=1 before udc_create to have every substep write them", who);
  return 1;
}

extern "C" int udc_field_download(udc_handle *h, int field, double *host, const int lb[3], const int ub[3]) {
  ENTRY_FLUSH(h);
  if ((field == UDC_EKM || field == UDC_EKH) && ek_current(h, "udc_field_download", field == UDC_EKH)) return 1;
  if (field == UDC_P && h->p_scratch) {
    udc_set_error("udc_field_download: p is scratch after a fused substep in the pressure-total form (it holds the previous pres0, not the "
                  "increment; udc_poisson leaves the reference's p, UDC_PTOTAL=0 keeps it in the fused substep too)");
    return 1;
  }
  if (copy3d(h, field, host, lb, ub, false)) return 1;
  if (h->scal_bcx == 2 && field >= UDC_SV0 && (field - UDC_SV0) % 3 == 0)      // sv0: its x ghost columns under BCxs = 2
    return k_scalar_bcx_fill_host(h, (field - UDC_SV0) / 3, host, lb, ub);
  return 0;
}

double *stats_ptr(udc_handle *h, int id);
extern "C" int udc_stats_get(udc_handle *h, int id, double *host, const int lb[3], const int ub[3]) {
  ENTRY_FLUSH(h);
  double *dev = stats_ptr(h, id);
  if (!dev) { udc_set_error("udc_stats_get: statistic %d is not being accumulated", id); return 1; }
  return copy3d_ptr(h, -1, dev, host, lb, ub, false);
}

extern "C" int udc_set_forcing(udc_handle *h, const double *dpdxl, const double *dpdyl, int n) {
  ENTRY_FLUSH(h);
  if (n != h->g.nz) { udc_set_error("udc_set_forcing: expected %d levels", h->g.nz); return 1; }
  const int nk = h->g.nz + 2;
  std::vector<double> t(2 * nk, 0.0);
  for (int k = 1; k <= n; ++k) { t[k] = dpdxl[k - 1]; t[nk + k] = dpdyl[k - 1]; }
  HIP_OK(hipMemcpyAsync(const_cast<double *>(h->m.dpdxl), t.data(), sizeof(double) * 2 * nk,
                        hipMemcpyHostToDevice, h->stream));
  HIP_OK(hipStreamSynchronize(h->stream));
  return 0;
}

// The fused substep leaves up,vp,wp unwritten (its next sweep does not read them).  Anything that
// observes the tendencies through the reference's semantics (zero after tstep_integrate) goes
// through here first.
static int tend_clean(udc_handle *h) {
  if (!h->tend_scratch) return 0;
  for (int f = UDC_UP; f <= UDC_WP; ++f)
    HIP_OK(hipMemsetAsync(h->fields[f], 0, sizeof(double) * h->g.n, h->stream));
  for (int n : h->slots)
    HIP_OK(hipMemsetAsync(h->fields[UDC_SVP + 3 * n], 0, sizeof(double) * h->g.n, h->stream));
  h->tend_scratch = false;
  return 0;
}

// After RK stage 3 the fused substep does not copy u0,v0,w0 into um,vm,wm: it marks them aliased and
// rotates the buffers at the next stage 1.  Anything that needs real um,vm,wm arrays comes through here.
static int um_materialise(udc_handle *h) {
  if (!h->um_alias) return 0;
  for (int q = 0; q < 3; ++q)
    HIP_OK(hipMemcpyAsync(h->fields[UDC_UM + q], h->fields[UDC_U0 + q], sizeof(double) * h->g.n,
                          hipMemcpyDeviceToDevice, h->stream));
  h->um_alias = false;
  return 0;
}

// ------------------------------------------------------------------------------ call surface
static int vel_fields(udc_handle *h, int rk3step, int *f) {
  int n = 0;
  f[n++] = UDC_U0; f[n++] = UDC_V0; f[n++] = UDC_W0;
  if (rk3step == 3) { f[n++] = UDC_UM; f[n++] = UDC_VM; f[n++] = UDC_WM; }
  (void)h;
  return n;
}

static int now_advection(udc_handle *h) {
  if (tend_clean(h) || um_materialise(h)) return 1;
  if (k_momentum_lds(h, true, false, false, false, 0.)) return 1;
  for (int n : h->slots)
    if (k_scalar_adv(h, n)) return 1;
  return 0;
}

static int now_subgrid(udc_handle *h) {
  if (tend_clean(h) || um_materialise(h)) return 1;
  if (k_closure(h)) return 1;
  if (h->lbuoycorr && k_vreman_buoycorr(h)) return 1;
  if (k_ek_ghosts(h) || k_xo_ek_ghosts(h)) return 1;
  h->ek_stale = h->ekh_stale = false;
  if (k_top_rows_after_closure(h)) return 1;
  if (k_momentum_lds(h, false, true, false, false, 0.)) return 1;
  if (k_scalar_top_flux(h)) return 1;      // reassure_fluxtop_boundary for a non-zero thl top flux (uses the new ekh)
  if (k_xo_thl_top(h, false)) return 1;    // ... and for a zero one above the inlet's first column (BCxT = 2)
  for (int n : h->slots)
    if (k_scalar_diff(h, n)) return 1;
  if (h->p.sgs == UDC_SGS_ONEEQN && k_tke_sources(h)) return 1;
  return 0;
}

extern "C" int udc_set_tempeq(udc_handle *h, int iadv_thl, int bctopt, double wttop, double thl_top, int bcbott, double wtsurf) {
  ENTRY_FLUSH(h);
  if (h->xg && iadv_thl != 2) {
    udc_set_error("udc_set_tempeq: with open x boundaries the temperature takes the central scheme (iadv_thl = 2; kappa reads two ghost columns)");
    return 1;
  }
  if (h->cfg.nsv > 15) { udc_set_error("udc_set_tempeq: thl uses scalar slot 15, nsv must be <= 15"); return 1; }
  if (iadv_thl != 2 && iadv_thl != 7) { udc_set_error("udc_set_tempeq: iadv_thl must be 2 (cd2, advecc_2nd) or 7 (kappa, advecc_kappa)"); return 1; }
  if (bctopt != 1 && bctopt != 2) { udc_set_error("udc_set_tempeq: BCtopT must be 1 (flux) or 2 (value)"); return 1; }
  if (h->p.lbottom && bcbott != 1 && bcbott != 2) { udc_set_error("udc_set_tempeq: BCbotT must be 1 (flux) or 2 (wall function)"); return 1; }
  const bool have = (int)h->fields.size() > UDC_THL0 && h->fields[UDC_THL0];
  if (!have) {
    for (int q = 0; q < 3; ++q)
      if (alloc_field(h, UDC_SV0 + 3 * 15 + q)) return 1;
    h->slots.push_back(15);
    if (h->slab) {      // more rows travel per exchange
      const size_t need = (size_t)16 * HY * h->g.nx * h->g.pz;
      if (need > h->halo_cap) { udc_set_error("udc_set_tempeq: halo buffers too small"); return 1; }
    }
  }
  udc_handle::Slot &sl = h->slot[15];
  sl.adv = iadv_thl == 7 ? 1 : 2;
  sl.kappa_ghosts = iadv_thl == 7 ? (bctopt == 2 ? 2 : 1) : 0;
  sl.top = bctopt == 2 ? 2 : (wttop != 0. ? 1 : 0);
  sl.topval = bctopt == 2 ? thl_top : wttop;
  sl.floorflux = wtsurf;
  return 0;
}

extern "C" int udc_set_chem(udc_handle *h, int lchem, double k1, double jno2) {
  ENTRY_FLUSH(h);
  if (lchem && h->cfg.nsv < 3) { udc_set_error("udc_set_chem: the NO, NO2, O3 chemistry needs nsv >= 3 (src/modchem.f90:50-52)"); return 1; }
  h->lchem = lchem ? 1 : 0; h->chem_k1 = k1; h->chem_jno2 = jno2;
  return 0;
}

extern "C" int udc_set_shifted_pbc(udc_handle *h, double a, const double *sinx, int nx, const double *u0av, int nz) {
  NO_OPEN_X(h, "udc_set_shifted_pbc");
  HIP_OK(hipSetDevice(h->device));
  h->shift_a = a;
  if (a == 0.) return 0;
  if (nx != h->g.nx || nz != h->g.nz) { udc_set_error("udc_set_shifted_pbc: expected sinx[%d], u0av[%d]", h->g.nx, h->g.nz); return 1; }
  if (!h->shift_tab) HIP_OK(hipMalloc(&h->shift_tab, sizeof(double) * (size_t)(nx + nz)));
  HIP_OK(hipMemcpyAsync(h->shift_tab, sinx, sizeof(double) * nx, hipMemcpyHostToDevice, h->stream));
  HIP_OK(hipMemcpyAsync(h->shift_tab + nx, u0av, sizeof(double) * nz, hipMemcpyHostToDevice, h->stream));
  HIP_OK(hipStreamSynchronize(h->stream));
  return 0;
}

static int now_shifted_pbcs(udc_handle *h) {
  if (tend_clean(h) || um_materialise(h)) return 1;
  return k_shifted_pbcs(h, false);
}

extern "C" int udc_set_scalar_top(udc_handle *h, int n, int bctops, double value) {
  ENTRY_FLUSH(h);
  if (n < 0 || n >= h->cfg.nsv) { udc_set_error("udc_set_scalar_top: scalar %d of %d", n, h->cfg.nsv); return 1; }
  if (bctops != 1 && bctops != 2) { udc_set_error("udc_set_scalar_top: BCtops must be 1 (flux) or 2 (value)"); return 1; }
  h->slot[n].top = bctops == 2 ? 2 : (value != 0. ? 1 : 0);
  h->slot[n].topval = value;
  return 0;
}

extern "C" int udc_set_scalar_source(udc_handle *h, int n, const double *src, const int lb[3], const int ub[3]) {
  ENTRY_FLUSH(h);
  if (n < 0 || n >= h->cfg.nsv) { udc_set_error("udc_set_scalar_source: scalar %d of %d", n, h->cfg.nsv); return 1; }
  udc_handle::ScalarSource &s = h->svsrc[n];
  if (s.d) { HIP_OK(hipFree(s.d)); s.d = nullptr; }
  if (!src) return 0;
  const int ext[3] = {h->g.nx - 2 * h->g.xg, h->g.ny, h->g.nz};
  size_t cnt = 1;
  for (int q = 0; q < 3; ++q) {
    if (lb[q] < 1 || ub[q] > ext[q] || ub[q] < lb[q]) { udc_set_error("udc_set_scalar_source: box outside the interior (dimension %d: %d..%d of 1..%d)", q, lb[q], ub[q], ext[q]); return 1; }
    s.lo[q] = lb[q] - 1 + (q == 0 ? h->g.xg : 0); s.hi[q] = ub[q] - 1 + (q == 0 ? h->g.xg : 0);      // (open x boundaries: the device row starts xg columns west of ib)
    cnt *= (size_t)(ub[q] - lb[q] + 1);
  }
  HIP_OK(hipMalloc(&s.d, sizeof(double) * cnt));
  HIP_OK(hipMemcpy(s.d, src, sizeof(double) * cnt, hipMemcpyHostToDevice));
  return 0;
}

static int now_scalsource(udc_handle *h) {
  if (tend_clean(h)) return 1;
  return k_scalsource(h);
}

extern "C" int udc_set_floor_air_temperature(udc_handle *h, double thl_kb) {
  ENTRY_FLUSH(h);
  h->floor_thl_air_on = true;
  h->floor_thl_air = thl_kb;
  return 0;
}

extern "C" int udc_set_floor_wf(udc_handle *h, int bcbotm, int bcbott, double thls, double z0h, double prandtlturb) {
  ENTRY_FLUSH(h);
  if (!h->p.lbottom) { udc_set_error("udc_set_floor_wf: the floor is off (udc_config.lbottom)"); return 1; }
  if (bcbotm != 2 && bcbotm != 3) { udc_set_error("udc_set_floor_wf: BCbotm must be 2 (wfuno) or 3 (wfmneutral)"); return 1; }
  if (bcbott != 1 && bcbott != 2) { udc_set_error("udc_set_floor_wf: BCbotT must be 1 (flux) or 2 (wfuno)"); return 1; }
  if (bcbotm == 2 || bcbott == 2) {
    const bool have_thl = (int)h->fields.size() > UDC_THL0 && h->fields[UDC_THL0];
    if (bcbott == 2 && !have_thl) { udc_set_error("udc_set_floor_wf: BCbotT = 2 needs the temperature equation (call udc_set_tempeq first)"); return 1; }
    // (BCbotm = 2 without the temperature equation: udc_set_floor_air_temperature, checked when `bottom` runs)
    // (thls: the reference's default is -1 and it runs with it -- only a wall temperature of exactly zero has no meaning, :99)
    if (thls == 0. || !(z0h > 0.) || !(prandtlturb > 0.)) { udc_set_error("udc_set_floor_wf: thls must not be zero, z0h and prandtlturb must be positive"); return 1; }
  }
  h->floor_bcbotm = bcbotm; h->floor_bcbott = bcbott;
  h->floor_thls = thls; h->floor_z0h = z0h; h->floor_prt = prandtlturb;
  return 0;
}

extern "C" int udc_set_moisture(udc_handle *h, int iadv_qt, int bctopq, double wqtop, double qt_top, int bcbotq, double wqsurf) {
  ENTRY_FLUSH(h);
  if (h->cfg.nsv > 13) { udc_set_error("udc_set_moisture: qt uses scalar slot 13, nsv must be <= 13"); return 1; }
  if (iadv_qt != 2) { udc_set_error("udc_set_moisture: only iadv_qt = 2 (cd2, advecc_2nd) exists (src/modadvection.f90:79-85)"); return 1; }
  if (bctopq != 1 && bctopq != 2) { udc_set_error("udc_set_moisture: BCtopq must be 1 (flux) or 2 (value)"); return 1; }
  if (h->p.lbottom && bcbotq != 1) { udc_set_error("udc_set_moisture: BCbotq must be 1 (flux) (src/modibm.f90:2051,2062-2064)"); return 1; }
  if (h->lbuoyancy) { udc_set_error("udc_set_moisture: call it before udc_set_buoyancy (and udc_set_moist_thermo in between)"); return 1; }
  const bool have = (int)h->fields.size() > UDC_QT0 && h->fields[UDC_QT0];
  if (!have) {
    for (int q = 0; q < 3; ++q)
      if (alloc_field(h, UDC_SV0 + 3 * 13 + q)) return 1;
    // keep the slot list ordered: passive scalars, qt (13), e12 (14), thl (15)
    auto it = h->slots.begin();
    while (it != h->slots.end() && *it < 13) ++it;
    h->slots.insert(it, 13);
  }
  udc_handle::Slot &sl = h->slot[13];
  sl.adv = 2;
  sl.top = bctopq == 2 ? 2 : (wqtop != 0. ? 1 : 0);
  sl.topval = bctopq == 2 ? qt_top : wqtop;
  sl.floorflux = -wqsurf;      // the reference adds wqsurf where it subtracts wtsurf (src/modibm.f90:2058 vs :2043)
  h->lmoist = true;
  return 0;
}

extern "C" int udc_set_moist_thermo(udc_handle *h, double thls, double qts, double ps, const double *zf, const double *zh, int n, int lqlnr) {
  ENTRY_FLUSH(h);
  const int nz = h->g.nz, n2 = nz + 2;
  if (!h->lmoist) { udc_set_error("udc_set_moist_thermo: call udc_set_moisture first"); return 1; }
  if ((int)h->fields.size() <= UDC_THL0 || !h->fields[UDC_THL0]) { udc_set_error("udc_set_moist_thermo: call udc_set_tempeq first"); return 1; }
  if (n != nz + 1) { udc_set_error("udc_set_moist_thermo: expected %d levels (zf, zh of kb..ke+kh)", nz + 1); return 1; }
  if (nz + 2 > 2000) { udc_set_error("udc_set_moist_thermo: ktot <= 1998 (diagfld keeps four level tables in LDS)"); return 1; }
  // (the reference checks nothing here and its own test deck tests/cases/526 runs with the defaults thls = qts = -1, src/modsurfdata.f90:61-64:
  //  diagfld's exner functions and calc_halflev's floor values then follow from those numbers as they are -- so do the kernels;
  //  refused is only what the formulas cannot take: 1 / thls, log(ps))
  if (thls == 0. || !(ps > 0.)) { udc_set_error("udc_set_moist_thermo: thls must not be zero, ps must be positive"); return 1; }
  std::vector<double> t((size_t)udc_handle::MT_N * n2, 0.0);
  for (int k = 1; k <= nz + 1; ++k) { t[udc_handle::MT_ZF * n2 + k] = zf[k - 1]; t[udc_handle::MT_ZH * n2 + k] = zh[k - 1]; }
  if (!h->mt) HIP_OK(hipMalloc(&h->mt, sizeof(double) * t.size()));
  HIP_OK(hipMemcpy(h->mt, t.data(), sizeof(double) * t.size(), hipMemcpyHostToDevice));
  h->thls = thls; h->qts = qts; h->ps = ps; h->lqlnr = lqlnr ? 1 : 0;
  if (h->p.sgs == UDC_SGS_ONEEQN && alloc_field(h, UDC_QL0)) return 1;
  h->mt_valid = false;
  return 0;
}

extern "C" int udc_set_poisson_bczp(udc_handle *h, int bczp) {
  ENTRY_FLUSH(h);
  if (bczp != 1 && bczp != 2) { udc_set_error("udc_set_poisson_bczp: BCzp is 1 (tridiagonal solve in z) or 2 (cosine transform in z)"); return 1; }
  if (bczp == 2) {      // (src/modpois.f90:180: "Assumes equidistant in z")
    std::vector<double> dzf(h->g.nz + 2);
    HIP_OK(hipMemcpy(dzf.data(), h->m.dzf, sizeof(double) * dzf.size(), hipMemcpyDeviceToHost));
    for (int k = 2; k <= h->g.nz; ++k)
      if (fabs(dzf[k] - dzf[1]) > 1e-12 * fabs(dzf[1])) { udc_set_error("udc_set_poisson_bczp: BCzp = 2 (cosine transform in z) needs equidistant levels"); return 1; }
  }
  h->bczp = bczp;
  return 0;
}

extern "C" int udc_set_fkar(udc_handle *h, double fkar) {
  ENTRY_FLUSH(h);
  if (!(fkar > 0.)) { udc_set_error("udc_set_fkar: the von Karman constant must be positive"); return 1; }
  h->fkar = fkar;
  return 0;
}

extern "C" int udc_calthv(udc_handle *h) {
  ENTRY_FLUSH(h);
  return k_calthv_capture(h);
}

extern "C" int udc_thermodynamics(udc_handle *h) {
  ENTRY_FLUSH(h);
  if (h->thermo_fresh) return 0;      // the fused substep ended with it (src/program.f90:214) and nothing changed since
  if (k_calthv_capture(h)) return 1;
  if (k_thermodynamics(h)) return 1;
  h->thermo_fresh = true;
  return 0;
}

extern "C" int udc_thermo_state(udc_handle *h, double *tables, int n, int set) {
  ENTRY_FLUSH(h);
  const int nz = h->g.nz, n2 = nz + 2;
  if (!h->mt) { udc_set_error("udc_thermo_state: call udc_set_moist_thermo first"); return 1; }
  if (n != nz + 1) { udc_set_error("udc_thermo_state: expected %d levels", nz + 1); return 1; }
  std::vector<double> t((size_t)udc_handle::MT_STATE_N * n2, 0.0);
  if (set) {
    for (int q = 0; q < udc_handle::MT_STATE_N; ++q)
      for (int k = 1; k <= nz + 1; ++k) t[(size_t)q * n2 + k] = tables[(size_t)q * n + k - 1];
    HIP_OK(hipMemcpyAsync(h->mt, t.data(), sizeof(double) * t.size(), hipMemcpyHostToDevice, h->stream));
    HIP_OK(hipStreamSynchronize(h->stream));
    h->mt_valid = true;
    h->thermo_fresh = false;
  } else {
    if (!h->mt_valid) { udc_set_error("udc_thermo_state: no thermodynamics call has been made yet"); return 1; }
    HIP_OK(hipMemcpyAsync(t.data(), h->mt, sizeof(double) * t.size(), hipMemcpyDeviceToHost, h->stream));
    HIP_OK(hipStreamSynchronize(h->stream));
    for (int q = 0; q < udc_handle::MT_STATE_N; ++q)
      for (int k = 1; k <= nz + 1; ++k) tables[(size_t)q * n + k - 1] = t[(size_t)q * n2 + k];
  }
  return 0;
}

extern "C" int udc_set_tke(udc_handle *h, double cm, double cn, double ch1, double ch2, double ce1, double ce2, double e12min,
                           double grav, double thvs, int ldelta) {
  NO_OPEN_X(h, "udc_set_tke");
  ENTRY_FLUSH(h);
  if (h->cfg.nsv > 14) { udc_set_error("udc_set_tke: e12 uses scalar slot 14, nsv must be <= 14"); return 1; }
  if (!(thvs > 0.) || !(e12min > 0.)) { udc_set_error("udc_set_tke: thvs and e12min must be positive"); return 1; }
  const bool have = (int)h->fields.size() > UDC_E120 && h->fields[UDC_E120];
  if (!have) {
    for (int q = 0; q < 3; ++q)
      if (alloc_field(h, UDC_SV0 + 3 * 14 + q)) return 1;
    // keep the slot list ordered: passive scalars, e12 (14), thl (15)
    auto it = h->slots.begin();
    while (it != h->slots.end() && *it < 14) ++it;
    h->slots.insert(it, 14);
  }
  udc_handle::Slot &sl = h->slot[14];
  sl.adv = 2; sl.top = 3; sl.topval = e12min; sl.floorflux = 0.; sl.tke = true;
  h->tke = udc_handle::Tke{cm, cn, ch1, ch2, ce1, ce2, e12min, grav, thvs, ldelta ? 1 : 0};
  h->p.sgs = UDC_SGS_ONEEQN;
  if (h->mt && alloc_field(h, UDC_QL0)) return 1;      // calthv's moist dthvdz reads ql0
  return 0;
}

extern "C" int udc_set_buoyancy(udc_handle *h, int lbuoyancy, double grav) {
  ENTRY_FLUSH(h);
  if (lbuoyancy && h->lmoist && !h->mt) {
    udc_set_error("udc_set_buoyancy: with moisture, call udc_set_moist_thermo first (surface values, pressure, level heights)");
    return 1;
  }
  if (lbuoyancy && ((int)h->fields.size() <= UDC_THL0 || !h->fields[UDC_THL0])) {
    udc_set_error("udc_set_buoyancy: call udc_set_tempeq first (thv0h comes from thl0)");
    return 1;
  }
  h->lbuoyancy = lbuoyancy ? 1 : 0;
  h->grav = grav;
  return 0;
}

extern "C" int udc_set_buoycorr(udc_handle *h, int lbuoycorr, double rigc) {
  ENTRY_FLUSH(h);
  if (!lbuoycorr) { h->lbuoycorr = 0; h->p.bare = 0; return 0; }
  if (h->p.sgs != UDC_SGS_VREMAN) { udc_set_error("udc_set_buoycorr: the correction belongs to the Vreman closure (src/modsubgrid.f90:269-353)"); return 1; }
  if (!h->lbuoyancy) { udc_set_error("udc_set_buoycorr: call udc_set_buoyancy first (the reference applies it only with lbuoyancy)"); return 1; }
  if (!(rigc > 0.)) { udc_set_error("udc_set_buoycorr: Rigc must be positive"); return 1; }
  h->lbuoycorr = 1; h->rigc = rigc; h->p.bare = 1;
  return 0;
}

extern "C" int udc_set_thl_source(udc_handle *h, const double *thlpcar, int n) {
  ENTRY_FLUSH(h);
  if (n != h->g.nz) { udc_set_error("udc_set_thl_source: expected %d levels", h->g.nz); return 1; }
  if ((int)h->fields.size() <= UDC_THL0 || !h->fields[UDC_THL0]) { udc_set_error("udc_set_thl_source: call udc_set_tempeq first"); return 1; }
  std::vector<double> t(h->g.nz + 2, 0.0);
  for (int k = 1; k <= n; ++k) t[k] = thlpcar[k - 1];
  if (!h->thlpcar) HIP_OK(hipMalloc(&h->thlpcar, sizeof(double) * t.size()));
  HIP_OK(hipMemcpy(h->thlpcar, t.data(), sizeof(double) * t.size(), hipMemcpyHostToDevice));
  return 0;
}

static int now_bottom(udc_handle *h) {
  if (h->p.sgs == UDC_SGS_ONEEQN && k_tke_floor(h)) return 1;      // unconditional part of `bottom`
  if (!h->p.lbottom) return 0;
  if (tend_clean(h) || um_materialise(h)) return 1;
  return k_bottom(h, false);
}

extern "C" int udc_set_coriolis(udc_handle *h, int mode, double om22, double om23, const double *ug, int n) {
  ENTRY_FLUSH(h);
  if (mode < 0 || mode > 2) { udc_set_error("udc_set_coriolis: mode 0 (off), 1 (lcoriol) or 2 (lprofforc)"); return 1; }
  if (mode == 2 && (!ug || n != h->g.nz)) { udc_set_error("udc_set_coriolis: lprofforc needs ug(kb:ke)"); return 1; }
  h->coriolis_mode = mode; h->om22 = om22; h->om23 = om23;
  std::vector<double> t(h->g.nz + 2, 0.0);
  if (ug) for (int k = 1; k <= n && k <= h->g.nz; ++k) t[k] = ug[k - 1];
  if (!h->ug) HIP_OK(hipMalloc(&h->ug, sizeof(double) * t.size()));
  HIP_OK(hipMemcpy(h->ug, t.data(), sizeof(double) * t.size(), hipMemcpyHostToDevice));
  return 0;
}

static int now_coriolis(udc_handle *h) {
  if (!h->coriolis_mode) return 0;
  if (tend_clean(h) || um_materialise(h)) return 1;
  return k_coriolis(h, false);
}

extern "C" int udc_slab_average(udc_handle *h, int field, double *avg, int n) {
  ENTRY_FLUSH(h);
  if (field >= UDC_UM && field <= UDC_WM && um_materialise(h)) return 1;
  return k_slab_average(h, field, avg, n);
}

extern "C" int udc_slab_averages(udc_handle *h, const int *fields, int nf, double *avg, int n) {
  ENTRY_FLUSH(h);
  if (um_materialise(h)) return 1;
  return k_slab_averages(h, fields, nf, avg, n);
}

extern "C" int udc_set_level_forcing(udc_handle *h, int tend, int src, const double *A, const double *B, int n, int when) {
  HIP_OK(hipSetDevice(h->device));
  if (tend < 0 || tend >= (int)h->fields.size() || !h->fields[tend]) { udc_set_error("udc_set_level_forcing: unknown tendency field %d", tend); return 1; }
  if (src >= 0 && (src >= (int)h->fields.size() || !h->fields[src])) { udc_set_error("udc_set_level_forcing: unknown source field %d", src); return 1; }
  auto it = h->level_forcings.begin();
  when = when ? 1 : 0;
  while (it != h->level_forcings.end() && !(it->tend == tend && it->when == when)) ++it;
  if (!A) {                       // remove
    if (it != h->level_forcings.end()) {
      HIP_OK(hipStreamSynchronize(h->stream));
      hipFree(it->A); hipHostFree(it->stage); hipEventDestroy(it->copied);
      h->level_forcings.erase(it);
    }
    return 0;
  }
  if (n != h->g.nz) { udc_set_error("udc_set_level_forcing: expected %d levels", h->g.nz); return 1; }
  if (it == h->level_forcings.end()) {
    udc_handle::LevelForcing f;
    f.tend = tend; f.when = when;
    HIP_OK(hipMalloc(&f.A, sizeof(double) * 2 * (n + 2)));
    f.B = f.A + (n + 2);
    HIP_OK(hipHostMalloc(&f.stage, sizeof(double) * 2 * (n + 2)));
    HIP_OK(hipEventCreateWithFlags(&f.copied, hipEventDisableTiming));
    HIP_OK(hipEventRecord(f.copied, h->stream));
    h->level_forcings.push_back(f);
    it = h->level_forcings.end() - 1;
  }
  it->src = (src >= 0 && B) ? src : -1;
  // one asynchronous upload from the entry's pinned block (waiting first for the previous upload of the same entry)
  HIP_OK(hipEventSynchronize(it->copied));
  double *t = it->stage;
  for (int k = 0; k < 2 * (n + 2); ++k) t[k] = 0.0;
  for (int k = 1; k <= n; ++k) { t[k] = A[k - 1]; if (B) t[n + 2 + k] = B[k - 1]; }
  HIP_OK(hipMemcpyAsync(it->A, t, sizeof(double) * 2 * (n + 2), hipMemcpyHostToDevice, h->stream));
  HIP_OK(hipEventRecord(it->copied, h->stream));
  return 0;
}

static int now_level_forcings(udc_handle *h, int when) {
  if (h->level_forcings.empty()) return 0;
  if (tend_clean(h) || um_materialise(h)) return 1;
  return k_level_forcings(h, when ? 1 : 0, false);
}

static void masscorr_effective(udc_handle *h) {
  if (h->uout_req) { h->luvolflowr = 2; h->uflowrate = h->uout_rate; }
  else { h->luvolflowr = h->uvol_req; h->uflowrate = h->uvol_rate; }
}

extern "C" int udc_set_masscorr(udc_handle *h, int luvolflowr, double uflowrate, int lvvolflowr, double vflowrate) {
  ENTRY_FLUSH(h);
  // the two requests for u are kept side by side; an outflow-rate correction (udc_set_masscorr_outflow) takes precedence while it
  // is on (src/modforces.f90:352,389), whatever order the two setters are called in and however often
  h->uvol_req = luvolflowr ? 1 : 0; h->uvol_rate = uflowrate;
  masscorr_effective(h);
  h->lvvolflowr = lvvolflowr ? 1 : 0; h->vflowrate = vflowrate;
  return 0;
}

extern "C" int udc_set_masscorr_outflow(udc_handle *h, int luoutflowr, double uflowrate) {
  NO_OPEN_X(h, "udc_set_masscorr_outflow");
  ENTRY_FLUSH(h);
  h->uout_req = luoutflowr ? 1 : 0; h->uout_rate = uflowrate;
  masscorr_effective(h);
  return 0;
}

static int now_masscorr(udc_handle *h, int rk3step, double dt) {
  if (!h->luvolflowr && !h->lvvolflowr) return 0;
  if (tend_clean(h) || um_materialise(h)) return 1;
  return k_masscorr(h, dt / (4. - (double)rk3step), false, false);
}

static int now_ibmwallfun(udc_handle *h) {
  if (!h->ibm_on) return 0;
  if (tend_clean(h) || um_materialise(h)) return 1;
  return k_ibm_wallfun(h);
}
static int now_ibmnorm(udc_handle *h) {
  if (!h->ibm_on) return 0;
  if (tend_clean(h) || um_materialise(h)) return 1;
  h->ptotal_now = false;      // (routine by routine: the reference's form)
  return k_ibm_norm(h);
}

static int now_forces(udc_handle *h) {
  if (tend_clean(h) || um_materialise(h)) return 1;
  if (h->thlpcar && k_level_source(h, 15, h->thlpcar)) return 1;      // thlp += thlpcar(k), src/modforces.f90:104-110
  if (k_forces(h)) return 1;
  return k_buoyancy(h);
}

static int now_poisson(udc_handle *h, int rk3step, double dt) {
  if (tend_clean(h) || um_materialise(h)) return 1;
  const double rk3coef = rk3step == 0 ? 1. : dt / (4. - (double)rk3step);
  // pvp(je+1) = pvp(jb) of bcpup: pvp = vp + vm / rk3coef is formed on the fly from the two, so both ghost rows count
  // (ibmnorm edits vm at the listed points after `halos` has run)
  const int fvp[2] = {UDC_VP, UDC_VM};
  if (k_halo_y(h, fvp, h->ibm_on ? 2 : 1, 1)) return 1;
  const bool lid = h->p.bctopm == UDC_TOP_PRESSURE;
  if (lid && k_lid_bcpup(h, rk3coef, false)) return 1;     // bcpup's open-lid rows, src/modboundary.f90:1234-1243
  if (k_xo_bcpup(h, rk3coef, false)) return 1;             // ... and its inflow / outflow columns, :1257-1280
  if (k_divergence_rhs(h, rk3coef, false)) return 1;       // fillps
  if (k_poisson_solve(h)) return 1;
  const int fp[1] = {UDC_P};
  if (k_halo_y(h, fp, 1, 1)) return 1;              // bcp
  h->p_scratch = false;
  if (k_project(h)) return 1;                       // tderive
  if (lid && k_lid_tderive(h)) return 1;            // src/modpois.f90:1058-1069
  const int fpr[1] = {UDC_PRES0};
  if (k_halo_y(h, fpr, 1, 1)) return 1;
  return 0;
}

static int now_tstep_integrate(udc_handle *h, int rk3step, double dt) {
  if (tend_clean(h) || um_materialise(h)) return 1;
  h->bcx_rk3coef = dt / (4. - (double)rk3step);
  if (k_scalar_bcx_uout(h)) return 1;
  h->halos_fresh = h->boundary_fresh = h->thermo_fresh = false;
  h->dthv_top_on = false;      // new fields: the next dthvdz is the one of the thermodynamics call that follows `boundary`
  if (h->p.bctopm == UDC_TOP_PRESSURE && k_lid_integrate(h, rk3step, dt, false, true, false)) return 1;   // src/modtstep.f90:270-286
  if (k_integrate(h, rk3step, dt)) return 1;
  if (k_xo_after_integrate(h, rk3step)) return 1;
  if (rk3step == 3 && k_chem(h, dt)) return 1;      // src/modtstep.f90:236-238
  return 0;
}

static int scalar_halo_list(udc_handle *h, int rk3step, std::vector<int> &f) {
  for (int n : h->slots) {
    f.push_back(UDC_SV0 + 3 * n);
    // (the m-fields change on RK stage 3 only -- except under an immersed boundary, whose `solid` rewrites them at the solid cells on
    //  every substep and reads, for a solid cell in the first / last row, the ghost row the last `halos` left: src/modibm.f90:748-826)
    if (rk3step == 3 || rk3step < 0 || (h->ibm_on && h->ibm[3].given)) f.push_back(UDC_SVM + 3 * n);
  }
  return 0;
}

// tau_x, tau_y, thl_flux of `bottom` (src/modibm.f90:2015-2018, 2094-2097)
extern "C" int udc_bottom_diagnostics(udc_handle *h, int on) {
  if (!h) { udc_set_error("null handle"); return 1; }
  HIP_OK(hipSetDevice(h->device));
  if (udc_flush_pending(h)) return 1;
  const size_t n = (size_t)h->g.nx * h->g.ny;
  for (int q = 0; q < 3; ++q) {
    if (on && !h->bottom_diag[q]) {
      HIP_OK(hipMalloc(&h->bottom_diag[q], sizeof(double) * n));
      HIP_OK(hipMemsetAsync(h->bottom_diag[q], 0, sizeof(double) * n, h->stream));
    } else if (!on && h->bottom_diag[q]) {
      HIP_OK(hipStreamSynchronize(h->stream));
      HIP_OK(hipFree(h->bottom_diag[q]));
      h->bottom_diag[q] = nullptr;
    }
  }
  return 0;
}
extern "C" int udc_bottom_diag_get(udc_handle *h, int which, double *out) {
  if (!h || !out) { udc_set_error("udc_bottom_diag_get: null argument"); return 1; }
  if (which < 0 || which > 2 || !h->bottom_diag[which]) { udc_set_error("udc_bottom_diag_get: enable with udc_bottom_diagnostics first"); return 1; }
  HIP_OK(hipSetDevice(h->device));
  if (udc_flush_pending(h)) return 1;
  // (open x boundaries: the deck's itot columns, not the two ghost columns of the device row)
  const int xg = h->g.xg, itot = h->g.nx - 2 * xg;
  HIP_OK(hipMemcpy2DAsync(out, sizeof(double) * itot, h->bottom_diag[which] + xg, sizeof(double) * h->g.nx, sizeof(double) * itot, (size_t)h->g.ny,
                          hipMemcpyDeviceToHost, h->stream));
  HIP_OK(hipStreamSynchronize(h->stream));
  return 0;
}

extern "C" int udc_halos(udc_handle *h) {
  ENTRY_FLUSH(h);
  if (h->halos_fresh) return 0;       // the fused substep already exchanged them and nothing changed since
  if (um_materialise(h)) return 1;
  if (k_xo_halos(h)) return 1;
  const int f[6] = {UDC_U0, UDC_V0, UDC_W0, UDC_UM, UDC_VM, UDC_WM};
  if (k_halo_y(h, f, 6, 1)) return 1;
  std::vector<int> s;
  scalar_halo_list(h, -1, s);
  if (!s.empty() && k_halo_y(h, s.data(), (int)s.size(), 2)) return 1;
  h->halos_fresh = true;
  h->boundary_fresh = false;
  return 0;
}

// the rk3coef the NEXT udc_boundary convects the outlets with (xmo_convective / xso_convective form it from dt and rk3step themselves,
// src/modboundary.f90:914): udc_tstep_integrate and udc_substep leave their own; this is for a `boundary` no integration precedes --
// the start-up's (src/program.f90:118: rk3step = 0 and, on a cold start, dt = dtmax / 100, src/modstartup.f90:1099)
extern "C" int udc_set_boundary_rk3coef(udc_handle *h, double rk3coef) {
  ENTRY_FLUSH(h);
  h->bcx_rk3coef = rk3coef;
  h->boundary_fresh = false;
  return 0;
}

extern "C" int udc_boundary(udc_handle *h) {
  ENTRY_FLUSH(h);
  if (h->boundary_fresh) return 0;
  if (um_materialise(h)) return 1;
  if (k_top_bottom(h)) return 1;
  if (k_scalar_bcx_outlet(h)) return 1;    // BCxs = 2: xso_convective with the rk3coef of the substep just integrated
  if (k_xo_boundary(h)) return 1;          // BCxm = 2: xmi_profile, xmo_convective
  h->boundary_fresh = h->halos_fresh;      // (boundary before halos leaves the ghost rows of the top planes stale)
  h->xo_boundary_owed = false;
  return 0;
}

extern "C" int udc_tstep_maxima(udc_handle *h, double dt, double *courtot, double *diffnrtot) {
  ENTRY_FLUSH(h);
  if (ek_current(h, "udc_tstep_maxima", true)) return 1;
  return k_maxima(h, dt, courtot, diffnrtot);
}

extern "C" int udc_checksim_begin(udc_handle *h, double dtmn) {
  ENTRY_FLUSH(h);
  if (ek_current(h, "udc_checksim", true)) return 1;
  return k_checksim_begin(h, dtmn);
}
extern "C" int udc_checksim_end(udc_handle *h, double out[4]) { return k_checksim_end(h, out); }
extern "C" int udc_checksim(udc_handle *h, double dtmn, double out[4]) {
  if (udc_checksim_begin(h, dtmn)) return 1;
  return k_checksim_end(h, out);
}

extern "C" int udc_divergence(udc_handle *h, double *divmax, double *divtot) {
  ENTRY_FLUSH(h);
  return k_divergence_check(h, divmax, divtot);
}


// ------------------------------------------------------------------------------ whole substep
// The tendency routines of one RK3 substep, as bits (in the reference's call order, src/program.f90:142-193)
enum : unsigned {
  OP_ADV = 1u << 0, OP_SHIFT = 1u << 1, OP_SUBGRID = 1u << 2, OP_BOTTOM = 1u << 3, OP_CORIOLIS = 1u << 4,
  OP_FORCES = 1u << 5, OP_LEV0 = 1u << 6, OP_IBMWALL = 1u << 7, OP_MASSCORR = 1u << 8, OP_IBMNORM = 1u << 9,
  OP_SCALSRC = 1u << 10, OP_LEV1 = 1u << 11, OP_POISSON = 1u << 12
};

// advection, subgrid, poisson, tstep_integrate, halos, boundary, thermodynamics and the routines of `ops`, with kernels
// fused across routine boundaries.  The additive terms are applied in a fixed order of their own (the momentum sweep
// first); masscorr sees every momentum term the reference's masscorr sees.
static int substep_fused(udc_handle *h, int rk3step, double dt, unsigned ops) {
  const double rk3coef = dt / (4. - (double)rk3step);
  if (h->xo_boundary_owed) {      // the stage-3 substep before this one ended ahead of `boundary` (statistics sample there) and no udc_boundary came
    if (k_top_bottom(h) || k_scalar_bcx_outlet(h) || k_xo_boundary(h)) return 1;      // (with that substep's rk3coef, still in bcx_rk3coef)
    h->xo_boundary_owed = false;
    h->boundary_fresh = true;
    if (h->lmoist && h->mt && !h->thermo_fresh) { if (k_thermodynamics(h)) return 1; h->thermo_fresh = true; }
  }
  h->bcx_rk3coef = rk3coef;
  ++h->substep_seq;
  // BCxs = 2 without a prescribed volume flow: the outlet's speed from the state the substep starts from.  (BCxm = 2: bcpup still reads
  // the speed the PREVIOUS `boundary` used -- uouttot is refreshed by `boundary` only, src/modboundary.f90:141-160 -- so there the refresh
  // follows k_xo_bcpup below; u0 is the substep's starting state until the integration)
  if (!h->xg && k_scalar_bcx_uout(h)) return 1;
  // what runs, in which order, is decided in one place: plan_substep (udc_plan.h; DESIGN.md section 7 has the table, the CPU test
  // tests/test_substep_plan.py enumerates it)
  PlanIn pin{};
  pin.no_fold = h->no_fold; pin.no_alias = h->no_alias;
  pin.ek_always = h->ek_always; pin.halo_overlap = !h->no_halo_overlap; pin.mom_pipe = !h->no_mom_pipe; pin.div_in_fft = !h->no_div_in_fft; pin.ptotal = h->sw.ptotal; pin.p_transpose = h->sw.p_transpose;
  pin.slab = h->slab; pin.comm_stream = h->comm_stream != nullptr; pin.sgs = h->p.sgs; pin.lbuoycorr = h->lbuoycorr;
  pin.nslots = (int)h->slots.size(); pin.ibm_on = h->ibm_on; pin.stats_any = h->stats_on || h->xyt_on || h->yt_on;
  pin.fft_fused = h->fft_fused; pin.own_fwd = h->xg ? (h->xpois && h->xpois->own_fwd) : h->own_fwd;      // (open x boundaries: the solve's transforms run on the doubled row of h->xpois)
  // (inflow / outflow in x: masscorr is off altogether, k_masscorr -- a prescribed flow rate only names the outlet's speed)
  pin.tend_plane = !h->xg && (h->luvolflowr == 2 || (h->ibm_on && (h->luvolflowr || h->lvvolflowr)));
  pin.between = h->coriolis_mode || !h->level_forcings.empty() || h->luvolflowr || h->lvvolflowr || h->ibm_on || h->shift_a != 0. ||
                h->thlpcar || h->lbuoyancy;
  pin.closure_tile_rows = closure_lds_tile_rows(h->g); pin.mom_tile_rows = momentum_lds_tile_rows(h->g); pin.int_tile_rows = tile_grid(h->g).gy;
  pin.x_row_groups = (h->slab && h->fft_fused) ? fft_x_row_groups(h) : 0;
  pin.levels_per_chunk = h->g.nz / (h->nch > 0 ? h->nch : 1);
  pin.rk3step = rk3step; pin.um_alias = h->um_alias; pin.ibm_edits_now = (ops & (OP_IBMWALL | OP_IBMNORM)) != 0;
  pin.open_lid = h->p.bctopm == UDC_TOP_PRESSURE;
  pin.lid_masked = pin.open_lid && k_lid_masked(h);
  const Plan plan = plan_substep(pin);
  h->last_plan = plan; h->have_plan = true;
  h->ptotal_now = plan.ptotal != 0;
  // (whatever way this substep ends, the routine-by-routine entry points must find the reference's form again)
  struct FormGuard { udc_handle *h; ~FormGuard() { h->ptotal_now = false; for (bool &b : h->sv_inline) b = false; } } form_guard{h};
  const bool lds = true, pup = true, fold = plan.fold;      // (LDS-staged sweeps, tendencies as predicted velocity: always)
  const bool forces = (ops & OP_FORCES) != 0;
  if (plan.materialise_um) { if (um_materialise(h)) return 1; }
  const bool rotate = plan.rotate;
  h->halos_fresh = h->boundary_fresh = h->thermo_fresh = false;
  // closure + momentum.  (One sweep evaluating ekm in LDS was built and measured slower than the two kernels, profiles/HISTORY.md section 5:
  // not kept.)
  {
    // closure first: it only needs u0,v0,w0, and the momentum sweep below needs ekm
    h->ekh_stale = false;
    if (plan.closure == CLOSURE_FOLDED) {
      if (k_closure_lds(h, true, plan.need_ekh)) return 1;
      h->ekh_stale = !plan.need_ekh;
    } else if (plan.closure == CLOSURE_OVERLAPPED) {
      // y-slabs: the tile rows next to the neighbouring ranks first; their ekm / ekh rows travel while the rows in between are swept
      // (ekh -- its array and its ghost row -- only where something reads it: a transported scalar, or between time steps)
      const int fek[2] = {UDC_EKM, UDC_EKH};
      if (k_closure_lds(h, false, plan.need_ekh, 1)) return 1;
      if (k_halo_y_begin(h, fek, plan.need_ekh ? 2 : 1, 1)) return 1;
      if (k_closure_lds(h, false, plan.need_ekh, 2)) return 1;
      if (k_halo_y_join(h)) return 1;
      if (k_ek_ghosts(h, false)) return 1;
      h->ekh_stale = !plan.need_ekh;
    } else {
      if (k_closure(h)) return 1;
      if (h->lbuoycorr && k_vreman_buoycorr(h)) return 1;      // before closurebc, as in the reference
      if (k_ek_ghosts(h)) return 1;
    }
    h->ek_stale = false;
    if (k_xo_ek_ghosts(h)) return 1;
    // y-slabs, nothing between the sweep and the solve but the floor: the sweep is pipelined with the solve's k-chunks.  Tile row 0
    // first over all levels (+ the floor on its rows): it holds the row of vp that the previous rank's divergence reads, which then
    // travels; the other rows follow level range by level range from inside k_poisson_solve_slab, each ahead of the x forward
    // transform of the same k-chunk -- so the forward all-to-all of chunk c runs under the sweep of the levels above it
    const bool floor_on = (ops & OP_BOTTOM) && h->p.lbottom;
    const bool pipe = plan.mom_pipe;
    if (pipe) {
      // UDC_MOM_PIPE=1: tile row 0 first over all levels (its own piece: 1/8 of the work in short chunks, and the other pieces then
      // fill 7/8 of the chip's workgroup slots at 1024 x 64 x 512).  2 (default): no such piece -- every k-chunk's piece covers all
      // tile rows and hands vp's first row of its own levels on when it is done (k_momentum_pipe_stage)
      h->mom_pipe.rows_all = h->sw.mom_pipe >= 2;
      if (!h->mom_pipe.rows_all) {
        const MomPart row0{0, 1, 0, 0, true};
        if (k_momentum_lds(h, true, true, forces, true, 1. / rk3coef, rotate, &row0, !plan.ptotal)) return 1;
        if (floor_on && k_bottom(h, false, 0, momentum_lds_tile_height())) return 1;
        const int fvp[1] = {UDC_VP};
        if (k_halo_y_begin(h, fvp, 1, 1, nullptr, HALO_TO_PREV)) return 1;      // (only the divergence of the slab's last row reads a ghost row of vp)
        h->vp_halo_pending = true;
      }
      h->mom_pipe.active = true; h->mom_pipe.forces = forces; h->mom_pipe.um_is_u0 = rotate; h->mom_pipe.bottom = floor_on;
      h->mom_pipe.rk3coefi = 1. / rk3coef; h->mom_pipe.pgrad = !plan.ptotal;
    } else if (k_momentum_lds(h, true, true, forces, true, 1. / rk3coef, rotate, nullptr, !plan.ptotal)) return 1;
  }
  const bool piped = h->mom_pipe.active;      // (then nothing below up to the solve has anything to do: see `pipe`)
  // a passive kappa-advected scalar whose tendency nothing else touches (no source, chemistry, obstacle, inflow / outflow, top flux) takes
  // its RK3 update inside its sweep: the sweep writes the new value where the tendency would go, the two arrays swap after the
  // integration, which skips it (48 + 24 -> 56 B per cell and scalar).  Single slab (the slabs' scalar rows may leave before the swap).
  h->sv_inline_last = rk3step == 3; h->sv_inline_rk3coef = rk3coef;
  for (int n : h->slots) {
    bool forced = false;      // a level forcing (subsidence, nudging, sponge) acts on this scalar's tendency
    for (const auto &f : h->level_forcings) forced = forced || f.tend == UDC_SVP + 3 * n;
    h->sv_inline[n] = !forced && !h->xg && h->sw.sv_inline && lds && fold && n < h->cfg.nsv && n < 13 && !h->slot[n].tke && h->slot[n].adv == 1 && h->slot[n].top == 0 &&
                      h->slot[n].kappa_ghosts == 0 && h->scal_bcx == 1 && !h->ibm_on && !h->svsrc[n].d && !h->lchem && h->g.nx >= 32 && h->g.ny >= 4;
  }
  h->last_inline_scalars = 0;
  for (int n : h->slots) h->last_inline_scalars += h->sv_inline[n] ? 1 : 0;
  if (k_scalar_top_flux(h)) return 1;
  if (k_xo_thl_top(h, false)) return 1;
  // thl (slot 15) and qt (13) share velocities and diffusivity: one sweep for both where their schemes agree
  bool paired = false;
  if (lds && std::find(h->slots.begin(), h->slots.end(), 15) != h->slots.end() && std::find(h->slots.begin(), h->slots.end(), 13) != h->slots.end()) {
    const int r = k_scalar_fused_pair(h, 15, 13, lds);
    if (r > 0) return 1;
    paired = r == 0;
  }
  for (int n : h->slots) {
    if (paired && (n == 15 || n == 13)) continue;
    if (k_scalar_fused(h, n, lds)) return 1;       // lds: the tendencies are scratch between fused substeps (tend_scratch)
  }
  if (k_xo_thl_top(h, true)) return 1;              // (the sweep advected over the reassured row as well)
  if (h->p.sgs == UDC_SGS_ONEEQN) {
    if (k_tke_sources(h)) return 1;                // subgrid's `sources`, after the diffusion terms
    if ((ops & OP_BOTTOM) && k_tke_floor(h)) return 1;   // first lines of `bottom` (src/program.f90:152)
  }
  if (forces && h->thlpcar && k_level_source(h, 15, h->thlpcar)) return 1;
  if (forces && k_buoyancy(h)) return 1;        // additive on wp(kb+1..ke), hence on pwp
  // `bottom` (src/program.f90:152): additive on the k = kb tendencies, hence equally on pup = up + um/rk3coef
  if (!piped && (ops & OP_BOTTOM) && h->p.lbottom && k_bottom(h, fold)) return 1;
  if ((ops & OP_CORIOLIS) && k_coriolis(h, fold)) return 1;        // src/program.f90:158; wrap of vp's ghost row follows below
  if ((ops & OP_SHIFT) && k_shifted_pbcs(h, fold)) return 1;       // src/program.f90:144 (additive on the momentum tendencies)
  if ((ops & OP_LEV0) && !h->level_forcings.empty() && k_level_forcings(h, 0, fold)) return 1;   // lstend, nudge tables
  if ((ops & OP_IBMWALL) && k_ibm_wallfun(h)) return 1;                                      // src/program.f90:166
  // masscorr (src/program.f90:169); without pup the tendencies and um are summed separately
  if ((ops & OP_MASSCORR) && k_masscorr(h, rk3coef, pup, fold)) return 1;
  if ((ops & OP_IBMNORM) && k_ibm_norm(h)) return 1;                                         // src/program.f90:171
  if ((ops & OP_SCALSRC) && k_scalsource(h)) return 1;                                       // src/program.f90:181
  if ((ops & OP_LEV1) && !h->level_forcings.empty() && k_level_forcings(h, 1, fold)) return 1;   // fixuinf1, grwdamp tables
  // the ghost row of vp that fillps' divergence reads: folded into the kernels above when the slab is the whole domain --
  // except for the immersed-boundary routines, which edit listed points only (a solid v point in the first row of the
  // domain has its periodic image in the ghost row)
  // slab path with the own line FFTs: fillps' divergence is evaluated inside the x forward transform (udc_fft.hip)
  h->div_in_fft = plan.div_in_fft;
  if (plan.vp_row == ROW_BESIDE) {
    // y-slabs with the divergence inside the x transform: the row travels while all but the last row group of the first k-chunk
    // are transformed (k_poisson_solve_slab joins)
    const int fvp[1] = {UDC_VP};
    if (k_halo_y_begin(h, fvp, 1, 1, nullptr, HALO_TO_PREV)) return 1;
    h->vp_halo_pending = true;
  } else if (plan.vp_row == ROW_INLINE) {
    const int fvp[1] = {UDC_VP};
    if (k_halo_y(h, fvp, 1, 1, HALO_TO_PREV)) return 1;
  }      // (ROW_PIPED: already travelling; ROW_FOLDED: written by the kernels above)
  // open lid (BCtopm = 3): bcpup's row pwp(ke+1) from the slab mean of pres0(ke), before the divergence that reads it
  const bool lid = pin.open_lid != 0;
  if (lid && k_lid_bcpup(h, rk3coef, pup, plan.ptotal != 0)) return 1;
  if (k_xo_bcpup(h, rk3coef, pup, plan.ptotal != 0)) return 1;
  if (h->xg && k_scalar_bcx_uout(h)) return 1;
  if (!h->div_in_fft && k_divergence_rhs(h, rk3coef, pup)) return 1;
  h->p_ghost_in_transpose = plan.p_row == ROW_TRANSPOSED;
  if (k_poisson_solve(h)) return 1;
  h->p_ghost_in_transpose = false;
  h->div_in_fft = false;
  h->mom_pipe.active = false;
  if (h->vp_halo_pending) { if (k_halo_y_join(h)) return 1; h->vp_halo_pending = false; }      // (no path leaves it pending)
  const int gyI = tile_grid(h->g).gy;
  // y-slabs: p's ghost row (the projection of the slab's first row reads it) travels while a first part of the interior rows is
  // integrated; those rows need neither it nor anything the edge launch writes
  const bool ov_p = plan.p_row == ROW_BESIDE;
  if (plan.p_row != ROW_FOLDED && plan.p_row != ROW_TRANSPOSED) {      // (ROW_TRANSPOSED: the backward transpose brought both rows along)
    const int fp[1] = {UDC_P};
    // (only the projection of the slab's first row reads a ghost row of p: the previous rank's last row; in the pressure-total form
    // p is the new pres0 and its other ghost row travels with it -- pres0 then leaves the exchange of the velocities' rows below)
    const int pdirs = plan.ptotal ? HALO_BOTH : HALO_TO_NEXT;
    if (ov_p) { if (k_halo_y_begin(h, fp, 1, 1, nullptr, pdirs)) return 1; }
    else if (k_halo_y(h, fp, 1, 1, pdirs)) return 1;
  }
  const bool skip_um = plan.skip_um;
  // open lid: tderive's and tstep_integrate's row w(ke+1), ahead of the sweep (it reads the solve's output and the row bcpup left;
  // the exchange of the new velocities' rows below then carries its plane like any other)
  if (lid && (k_lid_tderive(h) || k_lid_integrate(h, rk3step, dt, pup, false, fold))) return 1;
  // y-slabs: the rows next to the neighbouring ranks first; the ghost rows of the new velocities and of pres0 travel while the rows
  // in between are integrated (the exchange names the arrays as they will be known after the pointer rotation below)
  const bool ov_int = plan.integrate == INT_EDGES_FIRST;
  bool ov_scal = false;
  const bool ptot = plan.ptotal != 0;
  if (ov_int) {
    const int rb = ov_p ? 1 + std::max(1, (gyI - 2) / 4) : 1;      // interior tile rows [1, rb) first, [rb, gyI - 1) last
    if (ov_p) {
      if (k_project_integrate(h, rk3step, dt, !lds, pup, fold, !skip_um, rotate, 3, 1, rb, ptot)) return 1;
      if (k_halo_y_join(h)) return 1;
    }
    if (k_project_integrate(h, rk3step, dt, !lds, pup, fold, !skip_um, rotate, 1, 0, 0, ptot)) return 1;
    int f[8];
    int nf = vel_fields(h, skip_um ? 0 : rk3step, f);
    if (!ptot) f[nf++] = UDC_PRES0;
    double *ptr[8];
    for (int q = 0; q < nf; ++q) {
      int id = f[q];
      if (rotate && id >= UDC_U0 && id < UDC_U0 + 3) id = UDC_UM + (id - UDC_U0);
      else if (rotate && id >= UDC_UM && id < UDC_UM + 3) id = UDC_U0 + (id - UDC_UM);
      ptr[q] = h->fields[id];
    }
    if (k_halo_y_begin(h, f, nf, 1, ptr)) return 1;
    // the transported scalars' two ghost rows behind them on the same stream (not on a chemistry step: k_chem below edits them first)
    std::vector<int> sc;
    scalar_halo_list(h, rk3step, sc);
    ov_scal = !sc.empty() && sc.size() <= 16 && !(h->lchem && rk3step == 3);
    if (ov_scal && k_halo_y_begin(h, sc.data(), (int)sc.size(), 2)) return 1;
    if (rb < gyI - 1 && k_project_integrate(h, rk3step, dt, !lds, pup, fold, !skip_um, rotate, 4, rb, gyI - 1, ptot)) return 1;
    if (k_halo_y_join(h)) return 1;
  } else if (k_project_integrate(h, rk3step, dt, !lds, pup, fold, !skip_um, rotate, 0, 0, 0, ptot)) return 1;
  h->ptotal_now = false;
  if (ptot) { std::swap(h->fields[UDC_P], h->fields[UDC_PRES0]); h->p_scratch = true; }
  else h->p_scratch = false;
  for (int n : h->slots) {
    if (!h->sv_inline[n]) continue;
    // the array the sweep wrote is the scalar now; the planes below the floor (never written by a kernel: what the start-up put there) go along
    HIP_OK(hipMemcpyAsync(h->fields[UDC_SVP + 3 * n], h->fields[UDC_SV0 + 3 * n], sizeof(double) * (size_t)HZ * (size_t)h->g.sz, hipMemcpyDeviceToDevice, h->stream));
    std::swap(h->fields[UDC_SV0 + 3 * n], h->fields[UDC_SVP + 3 * n]);
    h->sv_inline[n] = false;      // (the integration has run: from here on the scalar is like any other)
  }      // what the solve returned is pres0 now; the old pres0 array takes the next solve's output
  h->dthv_top_on = false;
  if (rk3step == 3 && k_chem(h, dt)) return 1;      // src/modtstep.f90:236-238 (before the ghosts are refreshed)
  if (rotate) {
    for (int q = 0; q < 3; ++q) std::swap(h->fields[UDC_U0 + q], h->fields[UDC_UM + q]);
    h->um_alias = false;
  }
  if (skip_um) h->um_alias = true;
  h->tend_scratch = lds;
  if (!fold && !ov_int) {
    int f[8];
    int nf = vel_fields(h, skip_um ? 0 : rk3step, f);
    if (!ptot) f[nf++] = UDC_PRES0;
    if (k_halo_y(h, f, nf, 1)) return 1;
  }
  std::vector<int> s;
  scalar_halo_list(h, rk3step, s);
  if (!s.empty() && !ov_scal && k_halo_y(h, s.data(), (int)s.size(), 2)) return 1;
  if (h->xg && rk3step == 3 && (h->stats_on || h->xyt_on || h->xo_sample_gap)) {
    h->xo_sample_gap = false;
    // Open x boundaries with the device's statistics on: the reference samples them between tstep_integrate / halos and `boundary`
    // (src/program.f90:199-207), i.e. with the x ghost columns as the PREVIOUS `boundary` left them -- so stage 3 ends here, the ghost
    // columns put back; udc_boundary (the host's `boundary`, or the next substep's first act) and udc_thermodynamics follow the sample
    if (k_xo_after_integrate(h, rk3step, false)) return 1;
    h->halos_fresh = true; h->boundary_fresh = false; h->thermo_fresh = false;
    h->xo_boundary_owed = true;
    return 0;
  }
  if (!fold || !h->slots.empty()) { if (k_top_bottom(h)) return 1; }
  if (k_scalar_bcx_outlet(h)) return 1;
  if (k_xo_after_integrate(h, rk3step, true) || k_xo_boundary(h, rk3step == 3 ? 1 : 0)) return 1;
  h->halos_fresh = h->boundary_fresh = true;
  if (h->lmoist && h->mt) {                                             // src/program.f90:214
    if (k_thermodynamics(h)) return 1;
    h->thermo_fresh = true;
  }
  return 0;
}

extern "C" int udc_last_plan(udc_handle *h, int out[16]) {
  if (!h || !out) { udc_set_error("udc_last_plan: null argument"); return 1; }
  ENTRY_FLUSH(h);      // (deferred execution: the plan asked for is the one of the substep recorded last, not of an older one)
  if (!h->have_plan) { udc_set_error("udc_last_plan: no fused substep has run on this handle"); return 1; }
  const Plan &p = h->last_plan;
  const int v[16] = {p.fold, p.closure, p.need_ekh, p.mom_pipe, p.div_in_fft, p.vp_row, p.p_row, p.integrate, p.rotate, p.skip_um,
                     p.materialise_um, h->slab ? 1 : 0, h->fft_fused ? 1 : 0, h->nch, h->own_fwd ? 1 : 0, p.ptotal | (h->last_inline_scalars << 1)};
  for (int q = 0; q < 16; ++q) out[q] = v[q];
  return 0;
}

// the momentum sweep's piece that feeds k-chunk c of the slab solve: tile rows 1 .. of levels [c nzc + 1, (c + 1) nzc + 1) (the x
// forward transform of chunk c reads pwp one level up), from level 0 for c = 0 and to the top for the last; after the first piece the
// floor on those rows (`bottom` acts on level kb only)
// vp's first row of the levels chunk c's divergence reads (complete once piece c is done: level c nzc is the previous piece's last)
// starts travelling behind whatever the compute stream holds at this call
int k_momentum_pipe_row(udc_handle *h, int c) {
  if (!h->mom_pipe.active || !h->mom_pipe.rows_all) return 0;
  const int nzc = h->g.nz / h->nch;
  const int fvp[1] = {UDC_VP};
  if (k_halo_y_begin(h, fvp, 1, 1, nullptr, HALO_TO_PREV, c * nzc, nzc)) return 1;
  if (c < 16) {
    if (!h->ev_vp[c]) HIP_OK(hipEventCreateWithFlags(&h->ev_vp[c], hipEventDisableTiming));
    HIP_OK(hipEventRecord(h->ev_vp[c], h->comm_stream));      // (the x transform of chunk c waits for this one, not for the last begun)
  }
  h->vp_halo_pending = true;
  return 0;
}
int k_momentum_pipe_stage(udc_handle *h, int c, bool row) {
  if (!h->mom_pipe.active) return 0;
  const int nch = h->nch, nzc = h->g.nz / nch, gy = momentum_lds_tile_rows(h->g);
  const int kbeg = c == 0 ? 0 : c * nzc + 1, kend = c == nch - 1 ? h->g.nz : (c + 1) * nzc + 1;
  const bool all = h->mom_pipe.rows_all;
  const MomPart part{all ? 0 : 1, gy, kbeg, kend, c != nch - 1};
  if (k_momentum_lds(h, true, true, h->mom_pipe.forces, true, h->mom_pipe.rk3coefi, h->mom_pipe.um_is_u0, &part, h->mom_pipe.pgrad)) return 1;
  if (c == 0 && h->mom_pipe.bottom && k_bottom(h, false, all ? 0 : momentum_lds_tile_height(), -1)) return 1;
  if (all && row) return k_momentum_pipe_row(h, c);
  return 0;
}

// ------------------------------------------------------------------------------ deferred execution
static int run_op(udc_handle *h, unsigned op) {
  switch (op) {
    case OP_ADV: return now_advection(h);
    case OP_SHIFT: return now_shifted_pbcs(h);
    case OP_SUBGRID: return now_subgrid(h);
    case OP_BOTTOM: return now_bottom(h);
    case OP_CORIOLIS: return now_coriolis(h);
    case OP_FORCES: return now_forces(h);
    case OP_LEV0: return now_level_forcings(h, 0);
    case OP_IBMWALL: return now_ibmwallfun(h);
    case OP_IBMNORM: return now_ibmnorm(h);
    case OP_MASSCORR: return now_masscorr(h, h->pend_rk, h->pend_dt);
    case OP_SCALSRC: return now_scalsource(h);
    case OP_LEV1: return now_level_forcings(h, 1);
    case OP_POISSON: return now_poisson(h, h->pend_rk, h->pend_dt);
  }
  udc_set_error("deferred execution: unknown routine bit %u", op);
  return 1;
}

// Run whatever is recorded, routine by routine, in call order.
int udc_flush_pending(udc_handle *h) {
  if (h->pend.empty()) return 0;
  std::vector<unsigned> ops;
  ops.swap(h->pend);
  for (unsigned op : ops)
    if (run_op(h, op)) return 1;
  h->pend_rk = 0;
  ++h->n_unfused;
  return 0;
}

static int defer(udc_handle *h, unsigned op, int rk3step = 0, double dt = 0.) {
  if (rk3step) {
    if (h->pend_rk && (h->pend_rk != rk3step || h->pend_dt != dt)) { if (udc_flush_pending(h)) return 1; }
    h->pend_rk = rk3step; h->pend_dt = dt;
  }
  h->pend.push_back(op);
  return 0;
}


extern "C" int udc_set_deferred(udc_handle *h, int on) {
  ENTRY_FLUSH(h);
  h->deferred = on != 0;
  return 0;
}
extern "C" int udc_flush(udc_handle *h) { ENTRY_FLUSH(h); return 0; }
extern "C" int udc_deferred_stats(udc_handle *h, long *fused, long *unfused) {
  if (!h) return 1;
  if (fused) *fused = h->n_fused;
  if (unfused) *unfused = h->n_unfused;
  return 0;
}

extern "C" int udc_advection(udc_handle *h) { ENTRY(h); return h->deferred ? defer(h, OP_ADV) : now_advection(h); }
extern "C" int udc_shifted_pbcs(udc_handle *h) { ENTRY(h); return h->deferred ? defer(h, OP_SHIFT) : now_shifted_pbcs(h); }
extern "C" int udc_subgrid(udc_handle *h) { ENTRY(h); return h->deferred ? defer(h, OP_SUBGRID) : now_subgrid(h); }
extern "C" int udc_bottom(udc_handle *h) { ENTRY(h); return h->deferred ? defer(h, OP_BOTTOM) : now_bottom(h); }
extern "C" int udc_coriolis(udc_handle *h) { ENTRY(h); return h->deferred ? defer(h, OP_CORIOLIS) : now_coriolis(h); }
extern "C" int udc_forces(udc_handle *h) { ENTRY(h); return h->deferred ? defer(h, OP_FORCES) : now_forces(h); }
extern "C" int udc_level_forcings(udc_handle *h, int when) {
  ENTRY(h);
  return h->deferred ? defer(h, when ? OP_LEV1 : OP_LEV0) : now_level_forcings(h, when);
}
extern "C" int udc_masscorr(udc_handle *h, int rk3step, double dt) {
  ENTRY(h);
  if (rk3step < 1 || rk3step > 3) { udc_set_error("udc_masscorr: rk3step %d", rk3step); return 1; }
  return h->deferred ? defer(h, OP_MASSCORR, rk3step, dt) : now_masscorr(h, rk3step, dt);
}
extern "C" int udc_ibmwallfun(udc_handle *h) { ENTRY(h); return h->deferred ? defer(h, OP_IBMWALL) : now_ibmwallfun(h); }
extern "C" int udc_ibmnorm(udc_handle *h) { ENTRY(h); return h->deferred ? defer(h, OP_IBMNORM) : now_ibmnorm(h); }
extern "C" int udc_scalsource(udc_handle *h) { ENTRY(h); return h->deferred ? defer(h, OP_SCALSRC) : now_scalsource(h); }
extern "C" int udc_poisson(udc_handle *h, int rk3step, double dt) {
  ENTRY(h);
  // (rk3step 0 = "rk3coef 1", a bare projection: never part of a substep, runs at once)
  if (h->deferred && rk3step >= 1 && rk3step <= 3) return defer(h, OP_POISSON, rk3step, dt);
  if (udc_flush_pending(h)) return 1;
  return now_poisson(h, rk3step, dt);
}

extern "C" int udc_tstep_integrate(udc_handle *h, int rk3step, double dt) {
  ENTRY(h);
  if (!h->pend.empty()) {
    // the reference's own sequence (each routine at most once, in program.f90's order, advection + subgrid + poisson
    // present, masscorr/poisson recorded with this substep's rk3step and dt) runs as the fused substep
    unsigned mask = 0, last = 0;
    bool canon = true;
    for (unsigned op : h->pend) {
      if (op <= last) canon = false;
      last = op; mask |= op;
    }
    const unsigned need = OP_ADV | OP_SUBGRID | OP_POISSON;
    if ((mask & need) != need || h->pend_rk != rk3step || h->pend_dt != dt) canon = false;
    if (canon) {
      h->pend.clear();
      h->pend_rk = 0;
      ++h->n_fused;
      return substep_fused(h, rk3step, dt, mask);
    }
    if (udc_flush_pending(h)) return 1;
  }
  return now_tstep_integrate(h, rk3step, dt);
}

extern "C" int udc_substep(udc_handle *h, int rk3step, double dt, int with_forces) {
  ENTRY_FLUSH(h);
  unsigned ops = OP_ADV | OP_SHIFT | OP_SUBGRID | OP_BOTTOM | OP_IBMWALL | OP_MASSCORR | OP_IBMNORM | OP_SCALSRC | OP_POISSON;
  if (with_forces) ops |= OP_CORIOLIS | OP_FORCES | OP_LEV0 | OP_LEV1;
  return substep_fused(h, rk3step, dt, ops);
}

extern "C" int udc_run(udc_handle *h, int nsubsteps, int rk3step0, double dt, int with_forces) {
  int rk = rk3step0;
  for (int s = 0; s < nsubsteps; ++s) {
    if (udc_substep(h, rk, dt, with_forces)) return 1;
    rk = rk % 3 + 1;
  }
  return 0;
}
