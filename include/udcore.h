/* udcore.h -- C ABI of libudcore.so, the MI355X-native dynamical core for uDALES.
 *
 * The reference has no FFI: its seam is four Fortran modules whose argument-less
 * procedures work on module-global arrays.  Each entry point
 * below replaces one of those procedures; the Fortran drop-in modules in
 * u-dales_amd/fortran/ (same module and procedure names as the reference) are thin
 * ISO_C_BINDING wrappers over these symbols -- see INTEGRATION.md.
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on failure; the message is
 *     available from udc_last_error().  Nothing here calls exit().
 *   - Environment: the library reads its UDC_* switches exactly once, inside udc_create (A/B switches of the substep order, tuning
 *     knobs; DESIGN.md section 7 lists them); no other entry point looks at the environment, and none changes results beyond
 *     round-off.
 *   - plain pointers and sizes only; host arrays are owned by the caller (Fortran's
 *     modfields), device arrays by the library; no host pointer is retained.
 *   - host 3-D arrays are Fortran-ordered (i fastest) real(8) with inclusive index
 *     bounds lb[3]..ub[3] given in the reference's own (1-based interior) indexing,
 *     e.g. u0(ib-ih:ie+ih, jb-jh:je+jh, kb-kh:ke+kh) -> lb={0,0,0}, ub={nx+1,ny+1,nz+1}.
 *   - one handle per GPU / per rank; calls on a handle are serialised by the caller
 *     (the reference is single-threaded per rank).  Entry points are asynchronous on the
 *     handle's HIP stream unless stated; udc_field_download() and udc_sync() synchronise.
 */
#ifndef UDCORE_H
#define UDCORE_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct udc_handle udc_handle;

/* Field identifiers (reference array it mirrors, src/modfields.f90:30-80,
 * src/modsubgriddata.f90:64-73, src/modpois.f90:44-46) */
enum {
  UDC_U0 = 0, UDC_V0, UDC_W0,        /* velocities, current RK stage            */
  UDC_UM, UDC_VM, UDC_WM,            /* velocities, start of time step          */
  UDC_UP, UDC_VP, UDC_WP,            /* tendencies                              */
  UDC_PRES0,                         /* modified pressure                       */
  UDC_P,                             /* pressure correction (modpois: p)        */
  UDC_EKM, UDC_EKH,                  /* eddy viscosity / diffusivity            */
  UDC_SV0,                           /* passive scalar n: UDC_SV0 + 3*n  (sv0)  */
  UDC_SVM,                           /*                   UDC_SVM + 3*n  (svm)  */
  UDC_SVP,                           /*                   UDC_SVP + 3*n  (svp)  */
  /* ql0 as the reference's `thermo` leaves it (one level low: profiles/HISTORY.md, section 8); kept only for the one-equation
   * closure with moisture (calthv's moist dthvdz reads it), written by udc_thermodynamics */
  UDC_QL0 = UDC_SV0 + 3 * 16,
  UDC_FIELD_MAX = UDC_QL0 + 1,
  /* temperature equation (ltempeq): thl0, thlm, thlp live in scalar slot 15 (so nsv <= 15 with ltempeq) */
  UDC_THL0 = UDC_SV0 + 3 * 15, UDC_THLM = UDC_SVM + 3 * 15, UDC_THLP = UDC_SVP + 3 * 15,
  /* one-equation closure (loneeqn): e120, e12m, e12p live in scalar slot 14 (so nsv <= 14 with loneeqn) */
  UDC_E120 = UDC_SV0 + 3 * 14, UDC_E12M = UDC_SVM + 3 * 14, UDC_E12P = UDC_SVP + 3 * 14,
  /* moisture (lmoist): qt0, qtm, qtp live in scalar slot 13 (so nsv <= 13 with lmoist) */
  UDC_QT0 = UDC_SV0 + 3 * 13, UDC_QTM = UDC_SVM + 3 * 13, UDC_QTP = UDC_SVP + 3 * 13
};

/* SGS closure selector: &NAMSUBGRID lsmagorinsky / lvreman (src/modsubgriddata.f90:39-42),
 * DNS = lles .false. (src/modglobal.f90:194) */
enum { UDC_SGS_DNS = 0, UDC_SGS_SMAGORINSKY = 1, UDC_SGS_VREMAN = 2, UDC_SGS_ONEEQN = 3 /* set by udc_set_tke */ };
/* BCtopm (src/modglobal.f90:150-153) */
enum { UDC_TOP_FREESLIP = 1, UDC_TOP_NOSLIP = 2, UDC_TOP_PRESSURE = 3 };      /* BCtopm_*, src/modglobal.f90:140-142 */

/* Everything the reference's initglobal / initsubgrid / initpois derive the kernels'
 * constants from (src/modglobal.f90:536-874, src/modsubgrid.f90:44-79, src/modpois.f90:66-220). */
typedef struct udc_config {
  int itot, jtot, ktot;     /* global grid (&DOMAIN)                                     */
  int nranks, rank;         /* y-slab decomposition: nprocx = 1, nprocy = nranks          */
  int device;               /* HIP device ordinal for this rank; < 0: rank % visible devices */
  double dx, dy;            /* xlen/itot, ylen/jtot                                       */
  const double *dzf;        /* [ktot+2] = dzf(kb-1:ke+1)   (copied)                       */
  const double *dzh;        /* [ktot+2], entry k = dzh(k), k = 1..ktot+1 (entry 0 unused) */
  double numol;             /* src/modglobal.f90:300                                      */
  double prandtlmoli;       /* src/modglobal.f90:303                                      */
  double prandtli;          /* 1/Prandtl, src/modsubgrid.f90:117                          */
  double c_vreman;          /* src/modsubgriddata.f90:61                                  */
  double csz;               /* Smagorinsky constant, src/modsubgrid.f90:73-77             */
  int sgs;                  /* UDC_SGS_*                                                  */
  int bctopm;               /* UDC_TOP_*                                                  */
  double uinf, vinf;        /* no-slip top wall velocity (valuetop)                       */
  int nsv;                  /* passive scalars (kappa scheme, src/modglobal.f90:557-559)  */
  int lbottom;              /* &WALLS lbottom (src/modibm.f90:49): floor wall function, BCbotm = 3, BCbots = 1 */
  double z0;                /* &BC z0 roughness length (src/modsurfdata.f90:72); > 0 when lbottom          */
} udc_config;

/* ---- lifetime ------------------------------------------------------------------- */
/* replaces initfields/initsubgrid/initpois allocations (src/program.f90:77-89) */
int udc_create(const udc_config *cfg, udc_handle **out);
/* replaces exitsubgrid/exitpois (src/modstartup.f90:2352-2353) */
int udc_destroy(udc_handle *h);
const char *udc_last_error(void);
int udc_version(void);

/* ---- multi-GPU (RCCL over xGMI replaces 2decomp-fft's MPI) --- */
/* id must point to 128 bytes; rank 0 fills it and the host broadcasts it (MPI_Bcast in the
 * Fortran driver, torch.distributed in bench.py), then every rank calls udc_comm_init. */
int udc_comm_unique_id(unsigned char id[128]);
int udc_comm_init(udc_handle *h, const unsigned char id[128]);
/* What the communicator says about itself (bench.py's N > 1 line echoes it): info[0] transport (0 none: one rank; 1 RCCL; 2, 3 the
 * test transports of libudcore_test.so), [1] ncclCommCount, [2] ncclCommUserRank, [3] ncclCommCuDevice, [4] ncclGetVersion,
 * [5] k-chunks of the two transposes, [6] ranks the handle was created for, [7] its rank.  Replaces nothing in the reference
 * (2decomp-fft's decomp_2d_init prints its own layout, src/modstartup.f90:676). */
int udc_comm_info(udc_handle *h, int info[8]);
/* Exchange bookkeeping.  mode 1: reset the counters and time every exchange from now on with a pair of events on the stream it runs
 * on; mode 0: read (waits for the device); mode 2: read and stop timing.  out[0] all-to-all operations, [1] bytes one of them (the last: a backward transpose, with p's ghost rows
 * when they ride along) sends to ONE peer, [2] bytes sent to other ranks by all of them, [3] their time on the communication stream (ms, sum); [4] ghost-row
 * exchanges, [5] bytes sent to the previous rank, [6] to the next, [7] their time (ms, sum); [8] all-reduces, [9] doubles reduced. */
int udc_comm_stats(udc_handle *h, int mode, double out[16]);
/* on != 0: every exchange returns at once without moving anything -- the substep then costs what its kernels cost (results are wrong
 * from there on: a timing device for bench.py's exposed-exchange figure, used on a state that is thrown away). */
int udc_comm_dry_run(udc_handle *h, int on);
/* The order the last fused substep ran in, as plan_substep decided it (udc_plan.h; DESIGN.md "order of a substep"): out[0] ghost rows
 * folded into the kernels (single slab), [1] closure 0 folded / 1 edge rows first, rows travel beside the interior / 2 plain,
 * [2] ekh written, [3] momentum sweep pipelined with the solve's k-chunks, [4] divergence inside the x transform, [5] vp's and
 * [6] p's ghost row: 0 folded / 1 beside a sweep / 2 in line / 3 with the pipelined sweep / 4 inside the backward transpose, [7] integration 0 one launch / 1 edge
 * rows first, [8] um rotated, [9] um left aliased, [10] um materialised, [11] slab layout, [12] own line transforms on the slab path,
 * [13] k-chunks of the transposes, [14] own forward half (one GPU), [15] bit 0: pressure-total form (see udc_substep), bits 1..: how
 * many scalars took their RK3 update inside their own sweep. */
int udc_last_plan(udc_handle *h, int out[16]);
#ifdef UDC_TEST_TRANSPORT
/* Test transport, NOT part of libudcore.so: libudcore_test.so (same sources + -DUDC_TEST_TRANSPORT) adds it for the virtual-rank
 * tests.  P handles (cfg.nranks = P, rank = 0..P-1) inside ONE process on ONE device, each driven by its own host thread; ghost
 * rows and all-to-all blocks move by device-to-device copies behind a barrier.  Lets the multi-slab code path be parity-tested
 * on a single-GPU box.  udc_local_group_create returns a group id > 0 (or -1). */
int udc_local_group_create(int nranks);
int udc_comm_init_local(udc_handle *h, int group);
/* ... and between P PROCESSES sharing one device (mpiexec -n P of the Fortran drop-in build on a one-GPU box, where RCCL refuses
 * two ranks per device): outboxes in the POSIX shared-memory segment `name` ("/something", the same on every rank; rank 0 creates
 * it), device <-> host copies around process-shared barriers.  For small grids only. */
int udc_comm_init_shm(udc_handle *h, const char *name);
#endif

/* ---- host <-> device residency --------------------------------------------------- */
int udc_field_upload(udc_handle *h, int field, const double *host, const int lb[3], const int ub[3]);
int udc_field_download(udc_handle *h, int field, double *host, const int lb[3], const int ub[3]);
/* dpdxl(kb:ke), dpdyl(kb:ke) of modfields (src/modstartup.f90:2071-2081); n = ktot */
int udc_set_forcing(udc_handle *h, const double *dpdxl, const double *dpdyl, int n);

/* Temperature equation, &PHYSICS ltempeq (src/modglobal.f90:176): thl is advected (advection: iadv_thl = 2 ->
 * advecc_2nd, src/modadvection.f90:103-155; 7 -> advecc_kappa on the thl0c copy, :66-72, whose vertical ghost planes
 * follow src/modboundary.f90:211-213), diffused (subgrid: diffc with ekh), integrated and given its top
 * (BCtopT 1 = flux wttop, 2 = value thl_top, src/modboundary.f90:207-220) and floor (lbottom, BCbotT 1 = flux wtsurf,
 * src/modibm.f90:2035-2047) conditions like the passive scalars.  udc_set_buoyancy switches on forces' buoyancy
 * term for dry air (lbuoyancy, src/modforces.f90:73-84): wp += grav (thv0h - thvh)/thvh with thv0h = thl0h of
 * calc_halflev and thvh its slab average (src/modthermodynamics.f90:76,208,518-524), applied by udc_forces and
 * inside udc_substep.
 * Call once after udc_create, before the first substep.  thlpcar (udc_set_thl_source, [ktot] = thlpcar(kb:ke), the
 * radiative tendency of src/modforces.f90:104-110) is optional. */
int udc_set_tempeq(udc_handle *h, int iadv_thl, int bctopt, double wttop, double thl_top, int bcbott, double wtsurf);
int udc_set_thl_source(udc_handle *h, const double *thlpcar, int n);
/* chem  src/modchem.f90:27-73 (&CHEMISTRY lchem, k1, JNO2; called by tstep_integrate on RK stage 3, src/modtstep.f90:236-238):
 * NO + O3 <-> NO2 on scalars 1-3 [ug/m3], fully implicit backward Euler over dt.  Needs nsv >= 3; applied inside
 * udc_tstep_integrate / udc_substep. */
int udc_set_chem(udc_handle *h, int lchem, double k1, double jno2);
/* shiftedPBCs  src/modforces.f90:953-980 (src/program.f90:144, &BC ds > 0): in the downstream half of the domain
 * (global i > itot/2) the momentum tendencies get -vs (phi(j) - phi(j-1))/dy with vs = a u0av(k) sinx(i),
 * a = 0.5 pi ds/(0.5 xlen), sinx(i) = sin(pi (xh(i) - xh(itot/2))/(0.5 xlen)) ([itot], zero where inactive).  u0av(kb:ke)
 * is diagfld's slab average, refreshed by the host before every substep (udc_slab_average); udc_shifted_pbcs applies
 * it, as does udc_substep.  a = 0 switches it off. */
int udc_set_shifted_pbc(udc_handle *h, double a, const double *sinx, int nx, const double *u0av, int nz);
int udc_shifted_pbcs(udc_handle *h);
/* Top condition of passive scalar n (0-based), src/modboundary.f90:236-247: BCtops 1 = flux wsvtop(n) (fluxtopscal,
 * :1521-1537; the default with zero flux), 2 = value sv_top(n) (valuetopscal, :1539-1553; the reference sets
 * sv_top = svprof(ke, n), src/modstartup.f90:1573-1574). */
int udc_set_scalar_top(udc_handle *h, int n, int bctops, double value);
/* scalsource  src/modscalsource.f90:379-483 (src/program.f90:181): Gaussian point and line sources of the scalars.  They
 * depend on neither time nor flow, so the host evaluates the reference's expressions once (udcore/sources.py; in Fortran
 * one call of the reference's own scalsource on a zeroed svp) and registers the result per scalar n (0-based) as a dense
 * box src[(k-lb[2]) nj ni + (j-lb[1]) ni + (i-lb[0])] over the local interior indices lb..ub (src = NULL removes it);
 * udc_scalsource adds it to svp, as does udc_substep after masscorr. */
int udc_set_scalar_source(udc_handle *h, int n, const double *src, const int lb[3], const int ub[3]);
int udc_scalsource(udc_handle *h);
/* Floor (lbottom) with the stability-dependent wall function wfuno (Louis 1979 / Uno et al. 1995 over a rough wall,
 * src/modwallfunctions.f90:24-261) as `bottom` selects it (src/modibm.f90:2021-2045): BCbotm = 2 -> momentum (case 91;
 * 3 = the neutral wfmneutral of udc_config), BCbotT = 2 -> temperature against a wall at thls with roughness length
 * z0h (case 92; 1 = the prescribed flux wtsurf of udc_set_tempeq).  prandtlturb: src/modglobal.f90:304 (= prandtlmol).
 * Needs the temperature equation; call after udc_set_tempeq (which must then be given the same BCbotT). */
/* Temperature equation off (ltempeq = .false.) with the wfuno floor (BCbotm = 2, the reference's default): the reference's
 * thl0 is allocated regardless, keeps the values of prof.inp for ever (src/modtstep.f90:240 integrates it under ltempeq
 * only; src/modstartup.f90 does not perturb it) and wfuno judges the stability on its first level (src/modibm.f90:2022,
 * src/modwallfunctions.f90:92-127).  thl_kb = thlprof(kb); call before the first `bottom`. */
int udc_set_floor_air_temperature(udc_handle *h, double thl_kb);
int udc_set_floor_wf(udc_handle *h, int bcbotm, int bcbott, double thls, double z0h, double prandtlturb);
/* The von Karman constant of every wall function -- the floor's (wfuno, wfmneutral: src/modwallfunctions.f90:72, 307) and the
 * facets' (src/modibm.f90:1878, 1915, 1942): &WALLS fkar (src/modstartup.f90:152-153), default 0.41 (src/modglobal.f90:317). */
int udc_set_fkar(udc_handle *h, double fkar);
/* Total water, &PHYSICS lmoist (src/modglobal.f90:402): qt is advected (iadv_qt = 2 -> advecc_2nd,
 * src/modadvection.f90:78-86), diffused (diffc with ekh, src/modsubgrid.f90:147), integrated
 * (src/modtstep.f90:256) and given its top (BCtopq 1 = flux wqtop, 2 = value qt_top, src/modboundary.f90:222-231)
 * and floor (lbottom, BCbotq 1 = flux, "+ wqsurf" as the reference has it, src/modibm.f90:2050-2066) conditions.
 * Without buoyancy qt is a passive field.  With lbuoyancy the moist thermodynamics below must be set up before
 * udc_set_buoyancy.  With the one-equation closure the moist thermodynamics must be set up too: calthv's dthvdz takes
 * its saturated branch where the (level-shifted) ql0 is positive (src/modthermodynamics.f90:154-205). */
int udc_set_moisture(udc_handle *h, int iadv_qt, int bctopq, double wqtop, double qt_top, int bcbotq, double wqsurf);
/* Moist thermodynamics (src/modthermodynamics.f90:57-124, lmoist): thls, qts, ps of &BC / modsurfdata
 * (src/modsurfdata.f90:41,58,64) and the level heights zf(kb:ke+kh), zh(kb:ke+kh) ([n = ktot+1] each,
 * src/modglobal.f90:747-751); lqlnr (&DYNAMICS, src/modthermodynamics.f90:37) picks `thermo`'s Newton-Raphson branch
 * (:448-473) instead of the one-step formula (:476-500).  udc_thermodynamics is the reference's `thermodynamics`: thermo (condensate, all-or-nothing,
 * Tetens), diagfld (slab averages, hydrostatic pressures by fromztop, exner functions), calc_halflev, thermo on the
 * half levels and calthv's thv0h with its slab average thvh -- what forces' buoyancy term then uses
 * (wp += grav (thv0h - thvh)/thvh, src/modforces.f90:73-84).  Call it once before the first substep
 * (src/program.f90:120); udc_substep calls it at its end (:214), a routine-by-routine caller does so itself after
 * udc_boundary.  The reference's off-by-one-level ql0 slab average is reproduced (profiles/HISTORY.md, section 8).
 * udc_thermo_state reads (set = 0) or writes (set = 1) what one call leaves for the next: nine tables of [n = ktot+1]
 * (k = kb..ke+kh) in the order presf, presh, exnf, exnh, thvh, thl0av, qt0av, ql0av, th0av. */
int udc_set_moist_thermo(udc_handle *h, double thls, double qts, double ps, const double *zf, const double *zh, int n, int lqlnr);
int udc_thermodynamics(udc_handle *h);
/* calthv's dthvdz (src/modthermodynamics.f90:154-232) is state in the reference: computed whenever `thermodynamics` runs and read by
 * the closure of the NEXT substep (Vreman's buoyancy correction src/modsubgrid.f90:330-353, the one-equation closure :363-400).  The
 * library evaluates it from thl0 / qt0 inside those kernels, which is the same thing because nothing changes the fields in between
 * -- except at a start-up, where the reference's thermodynamics runs before `boundary` has set the top ghost plane
 * (src/modstartup.f90:1601, src/program.f90:118-120).  udc_calthv records the top ghost planes of thl0 / qt0 as they are at the
 * call; the closures read level ke's dthvdz from them until the next time integration.  Call it where the reference calls
 * `thermodynamics` at start-up: after the fields are set, before udc_boundary (udc_thermodynamics does so itself; dry decks call
 * this).  Without the call the first closure after a cold start sees the top condition's ghost value one substep early. */
int udc_calthv(udc_handle *h);
int udc_thermo_state(udc_handle *h, double *tables, int n, int set);
int udc_set_buoyancy(udc_handle *h, int lbuoyancy, double grav);
/* Vreman buoyancy correction, &NAMSUBGRID lbuoycorr with lbuoyancy (src/modsubgrid.f90:330-353, Huusko et al. 2025):
 * ekm *= sqrt(1 - min(max(Rig, 0), Rigc)/Rigc) before ekh = ekm/Pr and the molecular parts are added, with the gradient
 * Richardson number Rig = (grav/thl0) dthvdz / (du0dz^2 + dv0dz^2 + 1e-10) and calthv's dthvdz (dry or moist).  Vreman
 * closure only; call after udc_set_buoyancy.  Part of udc_subgrid / udc_substep. */
int udc_set_buoycorr(udc_handle *h, int lbuoycorr, double rigc);

/* One-equation (TKE) closure, &NAMSUBGRID loneeqn (src/modsubgrid.f90:363-400): switches the closure to
 * ekm = cm zlt e120 + numol, ekh = (ch1 + ch2 zlt/delta) ekm + numol/Pr_mol with the stability-limited length zlt
 * (dthvdz from thl0 when the temperature equation is on, else neutral), and makes e120 a transported field:
 * advecc_2nd (iadv_tke = 2), diffe (src/modsubgrid.f90:627-669), sources (shear, buoyancy, dissipation, :415-538),
 * e120 = max(e12min, e12m + rk3coef e12p) (src/modtstep.f90:209-211), e120(kb-1) = e120(kb) in `bottom`
 * (src/modibm.f90:2012-2013), e120(ke+1) = e12min in `boundary` (src/modboundary.f90:180-181).
 * Constants as initsubgrid derives them (src/modsubgrid.f90:63-71).  Call once after udc_create (and after
 * udc_set_tempeq if used), before the first substep. */
int udc_set_tke(udc_handle *h, double cm, double cn, double ch1, double ch2, double ce1, double ce2, double e12min,
                double grav, double thvs, int ldelta);

/* ---- the reference's call surface (src/program.f90:134-207) ----------------------- */
/* advection   src/modadvection.f90:36   up,vp,wp (+svp) -= div(u phi) (+ grad pres0)     */
int udc_advection(udc_handle *h);
/* subgrid     src/modsubgrid.f90:128    closure+closurebc, then diffu/diffv/diffw/diffc  */
int udc_subgrid(udc_handle *h);
/* bottom      src/modibm.f90:1998       floor (lbottom): wfmneutral (src/modwallfunctions.f90:263-350) replaces the
 *             resolved viscous flux through the floor in up,vp(kb) by the neutral log-law stress; zero-flux floor
 *             for the scalars (:2073-2090).  Called between subgrid and forces (src/program.f90:146-160).
 *             No-op when cfg.lbottom == 0.  tau_x, tau_y, thl_flux (what bottom added to up, vp, thlp: :2015-2018, 2094-2097;
 *             nonzero on the k = kb plane only, the reference's fielddump variables) are kept on request:
 *             udc_bottom_diagnostics(h, 1), then udc_bottom_diag_get(h, 0 | 1 | 2, out[ny_l * nx]) after a substep. */
int udc_bottom(udc_handle *h);
int udc_bottom_diagnostics(udc_handle *h, int on);
int udc_bottom_diag_get(udc_handle *h, int which, double *out);      /* 0 tau_x, 1 tau_y, 2 thl_flux */
/* forces      src/modforces.f90:46      neutral branch: up -= dpdxl(k), vp -= dpdyl(k), wp(kb)=0 */
int udc_forces(udc_handle *h);
/* coriolis    src/modforces.f90:600     mode 1 = lcoriol: Coriolis terms with om22 = 2 omega cos(lat), om23 = 2 omega sin(lat)
 *             (src/modglobal.f90:666-673), wp(kb) = 0; mode 2 = lprofforc: up += om23 (ug(k) - u0); ug = ug(kb:ke).
 *             Called between bottom and forces (src/program.f90:158); inside udc_substep when with_forces != 0. */
int udc_set_coriolis(udc_handle *h, int mode, double om22, double om23, const double *ug, int n);
int udc_coriolis(udc_handle *h);
/* Per-level forcings (lstend src/modforces.f90:719-822, nudge :824-860, grwdamp src/modboundary.f90:1447-1488):
 * they all have the form  tendency(i,j,k) += A(k) + B(k) * field(i,j,k)  with A, B built from slab averages
 * (u0av, v0av, thl0av, sv0av of diagfld, src/modthermodynamics.f90:262-279) and input profiles.  The library provides
 * the two device halves; the per-level arithmetic stays with the host (udcore/forcings.py mirrors the reference lines).
 * udc_slab_average: avg[k-1] = mean over the whole horizontal plane (all slabs) of `field` at level k = 1..n
 *   (n <= ktot+1: level ktot+1 is the top ghost plane, as avexy_ibm averages it); synchronises.
 * udc_set_level_forcing: registers (or with A == NULL removes) the forcing of tendency field `tend`; src < 0 = no B
 *   term; A, B hold levels 1..ktot; when = 0: applied before masscorr (lstend, nudge: src/program.f90:162-164),
 *   when = 1: after it (grwdamp, :191).  Registered forcings are applied by udc_level_forcings(h, when) and, when
 *   with_forces != 0, inside udc_substep. */
int udc_slab_average(udc_handle *h, int field, double *avg, int n);
/* the same for nf <= 16 fields at once (avg[q n + k]): one reduction, one copy back, one synchronisation */
int udc_slab_averages(udc_handle *h, const int *fields, int nf, double *avg, int n);
int udc_set_level_forcing(udc_handle *h, int tend, int src, const double *A, const double *B, int n, int when);
int udc_level_forcings(udc_handle *h, int when);
/* Immersed boundary, the sparse corrections (src/modibm.f90).  The point lists are the reference's input files
 * solid_{u,v,w,c}.txt and fluid_boundary_{u,v,w,c}.txt: n rows of GLOBAL 1-based (i, j, k), here as int[n][3]; grid 0 = u,
 * 1 = v, 2 = w, 3 = c (needed when scalars are transported).  udc_ibm_commit builds what initibm derives from them (the masks
 * of :150-186, evaluated at the listed points) and keeps this slab's points.  Across the lateral boundaries of the domain a
 * neighbour is looked up in the periodic image by default; udc_set_ibm_mask_wrap(h, wrapx, wrapy) reproduces what the
 * reference's masks hold there for a given decomposition: initibm fills their ghost cells with exchange_halo_z only, which
 * wraps a direction only when it is split over more than one rank (periodic_bc, src/modstartup.f90:662-672) -- with
 * nprocx = 1 (nprocy = 1) the masks' ghost columns (rows) stay "fluid".  Call it before udc_ibm_commit.
 *   udc_ibmwallfun  ibmwallfun (:1167) without facet wall functions (iwallmom = 1): diffu_corr (:990), diffv_corr (:1033),
 *                   diffw_corr (:1075), diffc_corr (:1120) per scalar -- called after nudge (src/program.f90:166)
 *   udc_ibmnorm     ibmnorm (:697): solid (:748) -- um, vm, wm and their tendencies zeroed at the solid points, svm / svp set
 *                   to the mean of their fluid neighbours -- called after masscorr (src/program.f90:171)
 * Both are part of udc_substep once committed.  With an immersed boundary udc_masscorr and udc_slab_average(s) average over
 * the fluid cells only (avexy_ibm, src/modmpi.f90:623-664, with IIu / IIv / IIc), and so do the moist thermodynamics' slab averages
 * (udc_thermodynamics: diagfld's thl0av, qt0av, ql0av over IIc, thvh over IIw; src/modthermodynamics.f90:76,262-279).  Wall
 * fluxes of heat and moisture: udc_set_ibm_wallheat / udc_set_ibm_wallmoist below; without them adiabatic, impermeable walls.
 * A listed point whose (i, j) lies outside the domain is dropped without a word, as the reference's reader drops what no rank owns
 * (read_sparse_ijk, src/readinput.f90:90-100; the lists of its own tests/cases/526 reach beyond the domain); a level outside kb..ke is
 * an error. */
enum { UDC_IBM_U = 0, UDC_IBM_V = 1, UDC_IBM_W = 2, UDC_IBM_C = 3 };
int udc_set_ibm_points(udc_handle *h, int grid, const int *solid, int nsolid, const int *bound, int nbound);
int udc_set_ibm_mask_wrap(udc_handle *h, int wrapx, int wrapy);
/* temperature (udc_set_tempeq) and total water (udc_set_moisture) with an immersed boundary: ibmnorm's solid on thlm / qtm (thl:
 * the volume mean of the fluid-cell slab averages where an obstacle cell has no fluid neighbour, :715) and advecc2nd_corr --
 * _liberal (:936) or, with lconservativeibm (&PHYSICS), _conservative (:889) --, ibmwallfun's diffc_corr, and the buoyancy
 * term's thvh over the fluid w points.  The walls are adiabatic and impermeable unless udc_set_ibm_wallheat /
 * udc_set_ibm_wallmoist say otherwise. */
int udc_set_ibm_conservative(udc_handle *h, int lconservativeibm);
/* Facet wall functions for momentum (wallfunmom, src/modibm.f90:1286-1433; &WALLS iwallmom: 2 = with the Uno et al. stability
 * functions on the facet temperatures, the reference's default; 3 = neutral log law; 1 = none).  udc_set_ibm_wallfun: the
 * choice, prandtlturb and the level coordinates zf(1:ktot+1), zh(1:ktot+1) the reconstruction interpolates on.
 * udc_set_ibm_sections: the facet sections of one velocity grid as initibmwallfun (:273-644) leaves them, skipped sections
 * left out, in the order of facet_sections_<grid>.txt -- per section the fluid-boundary cell (global 1-based i, j, k), the
 * section's area, the wall distance of the cell, the facet's unit normal, roughness lengths z0 / z0h and surface temperature
 * (facT(:,1); read for iwallmom = 2 only), whether the velocity is taken at the cell itself (comprec = 1) or at the
 * reconstruction point recpt (x, y, z) inside the cells recids[4][3] (lower corners on the u, v, w, c grids, global 1-based),
 * and the c-grid fluid masks of the two cells interp_temperature_* averages (tmask[2]: the cell, its lower neighbour along the
 * grid's direction).  The host side that builds these from the reference's input files: udcore/facets.py.  Every slab
 * passes all sections; each keeps the ones of its rows.  udc_ibmwallfun then applies the wall stress before the diffusion
 * corrections, as the reference does. */
int udc_set_ibm_wallfun(udc_handle *h, int iwallmom, double prandtlturb, const double *zf, const double *zh);
/* Heat wall function (wallfunheat, src/modibm.f90:1436-1540, sensible part): iwalltemp = 2 takes the wall heat flux from the
 * facet temperatures with heat_transfer_coef_flux (:1920-1986) on the c-grid sections (udc_set_ibm_sections with grid 3) and
 * takes flux * area / (dx dy dzh(k)) out of thlp, before diffc_corr; iwalltemp = 1: the prescribed flux of the facet's direction
 * (bctfxm ... bctfz, :1508-1524) is handed over per section in the slot of the facet temperature (tsurf), the same sections act
 * (log-law and velocity checks included, as in the reference); 0 (the default): off, adiabatic walls.  After udc_set_tempeq and
 * udc_set_ibm_wallfun. */
int udc_set_ibm_wallheat(udc_handle *h, int iwalltemp);
/* Latent part of wallfunheat (src/modibm.f90:1556-1600; lmoist, vegetated facets only -- faclGR): per c-grid section, n and the
 * order as given to udc_set_ibm_sections(grid 3): lgr (1 = the section's facet is vegetated), and
 *   iwallmoist = 1: qwall = the prescribed moisture flux of the facet's direction (bcqfxm ... bcqfz, :1558-1568);
 *   iwallmoist = 2: qwall = facqsat (saturation humidity at the facet temperature), hurel = fachurel, resc / ress = the canopy /
 *                   soil resistances facf(:, 4) / facf(:, 5): flux = moist_flux (:1989) with the aerodynamic resistance
 *                   1 / (htc |utan|) of the sensible part's heat transfer coefficient -- needs iwalltemp = 2;
 *   0 (the default): off, impermeable walls.
 * flux * area / (dx dy dzh(k)) comes out of qtp in the same sweep as the sensible part.  After udc_set_moisture,
 * udc_set_ibm_sections(3, ...) and udc_set_ibm_wallheat. */
int udc_set_ibm_wallmoist(udc_handle *h, int iwallmoist, int n, const int *lgr, const double *qwall, const double *hurel,
                          const double *resc, const double *ress);
/* lwritefac (&WALLS; src/modibm.f90:196-243, 1246-1282): the facet output of ibmwallfun -- per facet the wall shear stresses of the three
 * directions (wallfunmom, :1413-1430), the pressure and its square at the boundary cells (wallfunheat, :1475-1476) and the heat transfer
 * coefficients (:1540-1541), area-weighted over the facet's sections, divided by the facet's area and integrated in time on every RK
 * stage 3.  udc_set_ibm_facet_output: nfcts facets with their areas (facetarea.inp), the facet of every section handed to
 * udc_set_ibm_sections (nsec[4] = the counts given there, fac_* 1-based), and ALL sections of the c grid (cell, area, facet: the pressure is
 * taken on the skipped ones too).  udc_ibm_facet_sample(dt): the next udc_ibmwallfun is stage 3 of a step of dt.  udc_ibm_facet_get: the
 * integrals [7][nfcts] (tau_x, tau_y, tau_z, pres, pres2, htc, cth), reset != 0: and back to zero (fac.NNN.nc's record: the caller
 * divides by the interval). */
int udc_set_ibm_facet_output(udc_handle *h, int nfcts, const double *faca, const int *nsec, const int *fac_u, const int *fac_v,
                             const int *fac_w, const int *fac_c, int npres, const int *pcell, const double *parea, const int *pfac);
int udc_ibm_facet_sample(udc_handle *h, double dt);
int udc_ibm_facet_get(udc_handle *h, double *out, int reset);
int udc_set_ibm_sections(udc_handle *h, int grid, int n, const int *cell, const double *area, const double *dist, const double *norm,
                         const double *z0, const double *z0h, const double *tsurf, const int *comprec, const double *recpt,
                         const int *recids, const double *tmask);
int udc_ibm_commit(udc_handle *h);
int udc_ibmwallfun(udc_handle *h);
int udc_ibmnorm(udc_handle *h);
/* masscorr    src/modforces.f90:328     volume-flow branches: up += (uflowrate - <um + rk3coef up>)/rk3coef (luvolflowr,
 *             :389-417) and the same for v (lvvolflowr, :467-494); <.> = volume average over the whole domain
 *             (all-reduced over the slabs).  Called after forces (src/program.f90:169).  No-op unless enabled with
 *             udc_set_masscorr (&PHYSICS luvolflowr/uflowrate, lvvolflowr/vflowrate). */
int udc_set_masscorr(udc_handle *h, int luvolflowr, double uflowrate, int lvvolflowr, double vflowrate);
/* masscorr's outflow-rate branch for u (&PHYSICS luoutflowr, src/modforces.f90:352-387; it takes precedence over luvolflowr): the
 * mean of um + rk3coef up over the fluid u points of the outlet plane i = itot -- integrated with dy dzf(k), divided by the area of
 * that plane's fluid c cells (uoutletarea :499-522) -- is brought to uflowrate by a uniform shift of up.  Call it after
 * udc_set_masscorr.  The v counterpart (lvoutflowr :424-465) is not offered: the reference hands a (kb:ke) array to a dummy of
 * shape (ib:ie, kb:ke) there and overruns it. */
int udc_set_masscorr_outflow(udc_handle *h, int luoutflowr, double uflowrate);
int udc_masscorr(udc_handle *h, int rk3step, double dt);
/* &BC BCzp (src/modglobal.f90:176): 1 (default) the tridiagonal solve in z (solmpj), 2 the cosine transform in z of
 * src/modpois.f90:179-191, 559-590 (equidistant levels).  The cosine modes diagonalise the matrix solmpj solves, so 2 is served by the same
 * solve with the singular mode's solution shifted to zero mean over the levels (what dropping its kz = 0 coefficient gives). */
int udc_set_poisson_bczp(udc_handle *h, int bczp);
/* poisson     src/modpois.f90:419       fillps+bcpup, FFT(x,y)+tridiagonal(z), tderive+bcp */
int udc_poisson(udc_handle *h, int rk3step, double dt);
/* tstep_integrate src/modtstep.f90:171  u0 = um + rk3coef*up ..., zero tendencies, m <- 0 on stage 3 */
int udc_tstep_integrate(udc_handle *h, int rk3step, double dt);
/* halos       src/modboundary.f90:67    periodic x (index wrap) and y (slab exchange / wrap) ghosts */
int udc_halos(udc_handle *h);
/* boundary    src/modboundary.f90:115   w(kb)=0 and top ghost rows (periodic lateral subset) */
int udc_boundary(udc_handle *h);
/* tstep_update src/modtstep.f90:49      adaptive dt: returns max Courant and diffusion numbers
 * (already max-reduced over ranks); the caller applies dt = min(dtmax, dt*courant/C, dt*diffnr/D). */
int udc_tstep_maxima(udc_handle *h, double dt, double *courtot, double *diffnrtot);

/* One whole RK3 substep = advection, subgrid, forces, poisson, tstep_integrate, halos,
 * boundary in the reference's order, with kernels fused across routine boundaries.
 * By default in the pressure-total form (UDC_PTOTAL=0: off): the momentum sweep leaves -grad pres0
 * (src/modadvection.f90:187,245,309) out, the solve returns pres0 + p, the projection applies that and it becomes pres0 -- the same
 * velocities and pres0 to round-off, with pres0 read nowhere.  After such a substep the array `p` (UDC_P) is scratch: it holds the
 * previous pres0, not the increment (udc_poisson called on its own leaves the increment there as the reference does), and
 * udc_field_download(UDC_P) says so instead of handing it out.  Not with an open lid (BCtopm = 3), a prescribed outflow rate, or a
 * volume flow over an immersed boundary: the reference's form there. */
int udc_substep(udc_handle *h, int rk3step, double dt, int with_forces);
/* n substeps with fixed dt, rk3step cycling 1,2,3 starting from rk3step0 */
int udc_run(udc_handle *h, int nsubsteps, int rk3step0, double dt, int with_forces);

/* Deferred execution -- how an UNTOUCHED driver (src/program.f90:132-222) gets the fused substep.  With it on, the
 * tendency routines (udc_advection, udc_shifted_pbcs, udc_subgrid, udc_bottom, udc_coriolis, udc_forces,
 * udc_level_forcings, udc_masscorr, udc_scalsource, udc_poisson) only record that they were called.  udc_tstep_integrate
 * then runs the record: as ONE fused substep (the kernels of udc_substep, which also does halos, boundary and
 * thermodynamics -- the driver's own calls of those right after find nothing left to do) when the record is the
 * reference's sequence -- each routine at most once, in program.f90's order, advection + subgrid + poisson present --
 * and routine by routine in call order otherwise.  Every other entry point (uploads, downloads, reductions, set-up
 * calls other than udc_set_level_forcing / udc_set_shifted_pbc, whose tables only the recorded routines read) first
 * runs what is recorded, so host-visible results never depend on the mode.  udc_deferred_stats counts the substeps
 * that ran fused / routine by routine. */
int udc_set_deferred(udc_handle *h, int on);
int udc_flush(udc_handle *h);
int udc_deferred_stats(udc_handle *h, long *fused, long *unfused);

/* Statistics: the time-averaged 3-D set of tdump (src/modstatsdump.f90:1137-1213 accumulation, :1557-1645 output).
 * udc_stats_sample(h, tsamplep, tstatsdumpp) takes one sample of the current um, vm, wm, pres0 (+ thlm, qtm, svm(1..4), ekh)
 * into the running averages X <- (X (tstatsdumpp - tsamplep) + sample tsamplep)/tstatsdumpp, levels kb..ke+kh as the
 * reference; the caller keeps the reference's two clocks and calls it on RK stage 3 when tsamplep >= tsample
 * (:802-811; udcore/stats.py).  udc_stats_get downloads an accumulator like udc_field_download.  The output variables
 * are means and <ab> - <a><b> of these (ut = UMT, upwpt = UWTIK - UTIK WTIK, tketc = ((UUTC - UTC^2) + ...)/2, ...).
 * On a handle of udc_create_open_x: the reference samples between tstep_integrate / halos and
 * `boundary` (src/program.f90:199-214), so with the statistics on a fused stage-3 udc_substep ends ahead of `boundary`; call
 * udc_stats_sample, then udc_boundary (+ udc_thermodynamics) -- the next udc_substep runs an owed boundary first if none came.  The masks of
 * udc_stats_set_masks are indexed by the deck's columns [nz][ny][itot] there as well, and so are the x-z tables of udc_stats_yt / udc_stats_y. */
enum {
  UDC_ST_UMT = 0, UDC_ST_VMT, UDC_ST_WMT, UDC_ST_PT,          /* um, vm, wm, pres0                               */
  UDC_ST_UTC, UDC_ST_VTC, UDC_ST_WTC,                         /* velocities at the cell centre                   */
  UDC_ST_UUTC, UDC_ST_VVTC, UDC_ST_WWTC,                      /* their squares                                   */
  UDC_ST_UWTIK, UDC_ST_VWTJK, UDC_ST_UVTIJ,                   /* products at the ik, jk, ij edges                */
  UDC_ST_UTIK, UDC_ST_WTIK, UDC_ST_VTJK, UDC_ST_WTJK, UDC_ST_UTIJ, UDC_ST_VTIJ,   /* the edge interpolations    */
  UDC_ST_MOM_N,
  UDC_ST_THL = UDC_ST_MOM_N,      /* + 0 thlt, + 1 thltk (half level), + 2 wthltk, + 3 thlthlt                    */
  UDC_ST_QT = UDC_ST_THL + 4,     /* + 0 qtt, + 1 qttk, + 2 wqttk, + 3 qtqtt                                      */
  UDC_ST_SV = UDC_ST_QT + 4,      /* scalar n < 4: UDC_ST_SV + 5 n + (0 svt, 1 svtk, 2 wsvtk, 3 svsvt, 4 svsgst)  */
  UDC_ST_SV_STRIDE = 5,
  UDC_ST_MAX = UDC_ST_SV + 4 * UDC_ST_SV_STRIDE
};
int udc_stats_enable(udc_handle *h, int on);      /* 0 off; 1 the 3-D accumulators; +2 xytdump's running profiles; +4 ytdump's running y-averages;
                                                   * +8 xydump's and +16 ydump's instantaneous tables of every sample (udc_stats_xy / udc_stats_y) */
int udc_stats_sample(udc_handle *h, double tsamplep, double tstatsdumpp);
int udc_stats_get(udc_handle *h, int id, double *host, const int lb[3], const int ub[3]);

/* xytdump (src/modstatsdump.f90:1037-1056 slab averages of a sample, :1086-1101 their running time averages, :1404-1460
 * the table written every tstatsdump): x-, y- and time-averaged profiles on levels kb..ke.  The slab averages are
 * avexy_ibm's (src/modmpi.f90:623-664, lnan = .false.): over the fluid points of createmasks' masks (src/modibm.f90:2141-2190)
 * IIu, IIv, IIw, IIc, IIuw, IIvw, IIuv.  udc_stats_set_masks hands those over as one byte per cell of this rank's slab
 * [ktot][jmax][itot] (bit 0 IIu ... bit 6 IIuv in that order; NULL = no obstacles) and the global fluid counts per level
 * [7][ktot]; the caller applies avexy_ibm's rule for a first level without fluid points (all bits of that level set, count
 * of level ke) before the call (udcore/stats.py: xyt_masks).  udc_stats_sample then also updates the nine running profiles;
 * udc_stats_xyt forms the remaining slab averages from the 3-D accumulators and returns the table [UDC_XYT_N][ktot], rows in
 * the order of the reference's output variables (-999 on levels without fluid points). */
enum {
  UDC_XYT_U = 0, UDC_XYT_V, UDC_XYT_W, UDC_XYT_THL, UDC_XYT_QT, UDC_XYT_P,          /* uxyt vxyt wxyt thlxyt qtxyt pxyt      */
  UDC_XYT_UPWP, UDC_XYT_WPTHLP, UDC_XYT_VPWP, UDC_XYT_UPVP,                         /* upwpxyt wpthlpxyt vpwpxyt upvpxyt     */
  UDC_XYT_UW, UDC_XYT_WTHL, UDC_XYT_UV, UDC_XYT_VW, UDC_XYT_WW,                     /* uwxyt wthlxyt uvxyt vwxyt wwxyt       */
  UDC_XYT_USGS, UDC_XYT_THLSGS, UDC_XYT_VSGS,                                       /* usgsxyt thlsgsxyt vsgsxyt             */
  UDC_XYT_THLPTHLP, UDC_XYT_UPUP, UDC_XYT_VPVP, UDC_XYT_WPWP, UDC_XYT_TKE,          /* thlpthlpt upuptxyc vpvptxyc wpwptxyc tketxyc */
  UDC_XYT_N
};
int udc_stats_set_masks(udc_handle *h, const unsigned char *bits, const int *counts);
int udc_stats_xyt(udc_handle *h, double *table);
/* ytdump (src/modstatsdump.f90:964-999 y-averages of a sample with avey_ibm, :1104-1132 their running time averages, :1466-1551
 * the table): y- and time-averaged x-z fields on levels kb..ke, udc_stats_enable with bit 4 (on = 5, or 7 with xytdump).  The masks
 * are those of udc_stats_set_masks; avey_ibm has no rule for a level without fluid points, so the caller names the masks whose
 * first level it filled for avexy_ibm's rule (udc_stats_set_forced, forced[7] in the masks' bit order) and the device empties
 * those again; the column counts IIut ... IIuwt are formed on the device.  udc_stats_yt returns the table
 * [UDC_YT_N][ktot][itot] in the order of the reference's output variables (uyt vyt wyt thlyt qtyt sca1-3yt, upwpyt wpthlpyt
 * wpqtpyt wpsca1-3pyt, uwyt wthlyt wqtyt wsca1-3yt, upupyt wpwpyt thlpthlpyt qtpqtpyt sca1-3psca1-3pyt, usgsyt wsgsyt thlsgsyt
 * qtsgsyt sca1-3sgsyt); -999 in columns without fluid points. */
enum { UDC_YT_N = 34 };
int udc_stats_set_forced(udc_handle *h, const int *forced);
int udc_stats_yt(udc_handle *h, double *table);
/* xydump and ydump (src/modstatsdump.f90:1002-1081 the averages of one sample, :1294-1349 the records written every tsample): the
 * instantaneous counterparts of the two tables above -- the x-y (x-z) averages of the LAST sample udc_stats_sample took, no time
 * averaging.  udc_stats_xy: [UDC_XY_N][ktot] in the order of the reference's record (uxy vxy wxy thlxy qtxy pxy upwpxyik wpthlpxyk
 * vpwpxyjk usgsxy thlsgsxy vsgsxy uwxyik wthlxyk vwxyjk; the products of the edge interpolations are averaged with avexy_ibm's lnan =
 * .true.: -999 on a level without fluid points, the first level included -- the flux rows then carry the reference's arithmetic on
 * those -999).  udc_stats_y: [UDC_Y_N][ktot][itot] (uy vy wy thly qty sca1-3y upwpyik wpthlpyk usgsy thlsgsy uwyik wthlyk), -999 in
 * columns without fluid points.  udc_stats_enable with bit 8 / 16. */
enum { UDC_XY_N = 15, UDC_Y_N = 14 };
int udc_stats_xy(udc_handle *h, double *table);
int udc_stats_y(udc_handle *h, double *table);

/* Passive scalars with an inflow and an outflow in x while the flow stays periodic (&BC BCxs = 2, the reference's dispersion
 * examples): inlet ghost cells mirrored about the inflow profile (xsi_profile, src/modboundary.f90:844-861), a convective outlet
 * ghost updated at every `boundary` (xso_convective, :983-996, with uouttot = ubulk of a prescribed volume flow, :159), no
 * periodic refresh of the scalars' x ghosts (:99).  svprof[nsv][ktot+2] is indexed by the reference's k; the east ghost columns
 * are taken from any upload of sv0 whose host array carries them (lb[0] <= itot+1, ub[0] >= itot+2).  advecc_kappa and diffc
 * then see those ghosts in the two cell columns next to either end; a download of sv0 into such an array fills them back in
 * (the inlet ones by xsi_profile's rule).  Not covered: obstacles touching the x ends of the domain; svm's ghost columns (no
 * routine reads them). */
int udc_set_scalar_bcx(udc_handle *h, int bcxs, const double *svprof, double uouttot);
/* Without a prescribed volume flow the reference takes the outlet's speed from diagfld's slab averages of the previous substep's end
 * (src/modboundary.f90:143-156): uouttot = sum_k wlev(k) u0av(k), wlev(k) = dzf(k) / (zh(ke+1) - zh(kb+1)), u0av over the fluid u
 * points.  wlev[ktot]; evaluated on the device at the start of every substep.  NULL: back to the constant of udc_set_scalar_bcx. */
int udc_set_scalar_bcx_outflow(udc_handle *h, const double *wlev);

/* Inflow and outflow in x for the flow itself (&BC BCxm = 2: the inlet's velocity from prof.inp's profile, a convective outlet;
 * the reference opens the lid with it, src/modstartup.f90:830-849, so cfg->bctopm must be 3).  What it replaces:
 *   xmi_profile      src/modboundary.f90:688-717     u(ib) = uprof(k); u, v, w at ib-1 mirrored about uprof, vprof, 0
 *   xmo_convective   :908-926                         v, w (the 0 and the m fields) at ie+1 carried out with uouttot at every `boundary`
 *   bcpup            :1257-1280                       pup(ib) = uprof / rk3coef, up(ib) = 0; pup(ie+1) = um(ie+1) / rk3coef
 *                                                     - (u0(ie+1) - u0(ie)) dxi uouttot, at kb a copy of pup(ie)
 *   bcp, closurebc   :1376-1394, :467-475             p, pres0, ekm, ekh at ib-1 / ie+1 = the column next to them
 *   initpois/poisson src/modpois.f90:113-121, 492-507 cosine transform in x (REDFT10 / REDFT01) instead of the real FFT
 *   tstep_integrate  src/modtstep.f90:262-264         u0(ie+1) = um(ie+1) + rk3coef up(ie+1)
 * udc_create_open_x makes the handle (cfg as for udc_create, itot the deck's); uprof, vprof: [ktot+2] indexed by the reference's k
 * (entry ktot+1 is what the reference's uprof(ke+1) holds: zero, src/modfields.f90:556).  Such a handle keeps the reference's x ghost
 * columns ib-1 and ie+1 of every field on the device: udc_field_upload takes them from a host array that carries them (lb[0] <= 0,
 * ub[0] >= itot+1: u0(ie+1), v0 / w0 / vm / wm(ie+1) are state), udc_field_download returns them; columns further out are left alone.
 * With it go: the immersed boundary (point lists, facet sections: wall functions for momentum and heat), the floor, Smagorinsky / Vreman, the
 * temperature (central scheme; udc_set_open_x_thl) and up to 12 passive scalars (cfg->nsv > 0 gives the rows a second ghost column either
 * side; udc_set_open_x_scalars).  A temperature without an inflow profile and the total water stay periodic in x (&BC BCxT = 1, BCxq = 1, the
 * reference's defaults -- its tests/cases/525 runs BCxm = 3 with them): halos' xT_periodic / xq_periodic (src/modboundary.f90:543-577) refresh
 * their ghost columns after every integration, the moist thermodynamics' slab averages run over ib .. ie.  One rank; no one-equation closure
 * yet (that entry point refuses the handle; udc_masscorr does nothing, as the reference's masscorr under linoutflow); the statistics
 * (udc_stats_*) are offered.  DESIGN.md sections 1 and 4 (udc_xopen.hip).
 * udc_set_open_x_outflow: the outlet's speed uouttot (src/modboundary.f90:141-160) -- wlev NULL: the constant given (ubulk of a prescribed
 * flow); wlev[ktot] = dzf(k) / (zh(ke+1) - zh(kb+1)): sum_k wlev(k) u0av(k) of the state each substep starts from, `uouttot` being
 * the value in force until the first refresh (bcpup reads the previous `boundary`'s speed) -- and, with hold_first, through the whole
 * first substep: the reference's start-up forms u0av before its first `boundary` puts uprof into u(ib), and the first substep's
 * `boundary` still reads that u0av (src/modstartup.f90:1601, src/program.f90:118, 214). */
/* The rk3coef the next udc_boundary convects the outlets with (BCxm = 2 and BCxs = 2; xmo_convective / xso_convective form it from dt
 * and rk3step, src/modboundary.f90:914, 989).  udc_tstep_integrate and udc_substep leave their own; this one is for a `boundary` that
 * no integration precedes: the start-up's (src/program.f90:118 -- rk3step = 0 and, on a cold start, dt = dtmax / 100,
 * src/modstartup.f90:1099, so dtmax / 400). */
int udc_set_boundary_rk3coef(udc_handle *h, double rk3coef);
int udc_create_open_x(const udc_config *cfg, const double *uprof, const double *vprof, udc_handle **out);
int udc_set_open_x_outflow(udc_handle *h, const double *wlev, double uouttot, int hold_first);
/* One-shot: the next fused RK stage-3 udc_substep ends ahead of `boundary` -- interior as integrated, x ghost columns as the previous
 * `boundary` left them -- which is the state the reference's checksim / fielddump / statsdump see (src/program.f90:199-214: they run between
 * halos and boundary; xTi_profile, src/modboundary.f90:785-791, overwrites an interior column there).  udc_boundary (+ udc_thermodynamics)
 * must follow; the next udc_substep runs an owed `boundary` first if none came.  With the device's own statistics on (udc_stats_enable)
 * every stage-3 substep ends there and this call is not needed; it is for a caller that downloads the fields for host-side statistics.
 * Does not run recorded substeps (udc_set_deferred): the one it is meant for may be among them. */
int udc_set_open_x_sample_gap(udc_handle *h, int on);
int udc_set_open_x_profile(udc_handle *h, const double *uprof, const double *vprof);      /* the inflow profiles again */
/* &BC BCxm = 3 (BCxm_driver): the inlet from the planes of a precursor run instead of the profile.  The reference's moddriver (host:
 * reads the driver files, interpolates in time -- drivergen, src/moddriver.f90:174) stays what it is; after every drivergen the six
 * planes it leaves in modinletdata go to the device: u0driver, umdriver, v0driver, vmdriver, w0driver, wmdriver, each
 * (lb[0]:ub[0], lb[1]:ub[1]) in the reference's j, k (j fastest), covering jb-1 .. je+1, kb .. ke+1.  They are what the NEXT `boundary`
 * applies (xmi_driver, src/modboundary.f90:720-749: u at ib and ib-1, v and w at ib-1); bcpup (:1282-1303: pup(ib) = u0driver / rk3coef,
 * the outlet convective on every level) keeps reading the planes the LAST `boundary` applied, as the reference's does between two
 * drivergen calls.  Everything else is the open-x handle's (udc_create_open_x; its profiles are then unused). */
/* The temperature on such a handle (&BC BCxT = 2 with BCxm = 2 / 3; after udc_set_tempeq with the central scheme, iadv_thl = 2): the inflow
 * profile thlprof [ktot+2] by the reference's k (entry ktot+1 as the reference's thlprof(ke+1): zero).  xTi_profile
 * (src/modboundary.f90:766-793: thl(ib-1) = thlprof on kb .. ke+1, thl(ib) = thlprof on kb .. ke) and xTo_convective (:947-957) run with every
 * `boundary`; thl0 / thlm at ie+1 are state like v and w there (udc_field_upload takes them from a host array that carries the column).
 * Not with the kappa scheme.  Without this call the temperature is periodic in x (BCxT = 1). */
int udc_set_open_x_thl(udc_handle *h, const double *thlprof);
/* The total water's inflow profile (&BC BCxq = 2; after udc_set_moisture): qtprof [ktot+2] by the reference's k.  xqi_profile
 * (src/modboundary.f90:811-823: qt(ib-1) = 2 qtprof - qt(ib) on jb-1 .. je+1, kb .. ke+1) and xqo_convective (:961-971 -- which starts from
 * qt(ie), not from qt(ie+1): kept) with every `boundary`.  Without this call the total water is periodic in x (BCxq = 1). */
int udc_set_open_x_qt(udc_handle *h, const double *qtprof);
/* &BC BCxT = 3 / BCxq = 3 / BCxs = 3 (next to BCxm = 3): a scalar's inlet ghost columns from the planes the reference's drivergen leaves
 * in modinletdata -- field = UDC_THL0 (thl0driver, thlmdriver -> xTi_driver, :795-808), UDC_QT0 (qt0driver, qtmdriver -> xqi_driver,
 * :826-839), the sv0 of passive scalar n (sv0driver(:, :, n), svmdriver(:, :, n) -> xsi_driver, :883-901: both ghost columns; after
 * udc_set_open_x_scalars) --, each (lb[0]:ub[0], lb[1]:ub[1]) in the reference's j, k (j fastest) covering jb-1 .. je+1, kb .. ke+1.  Handed
 * over after every drivergen like the flow's planes (udc_set_open_x_inlet); the next `boundary` applies them, the outlet is convective. */
int udc_set_open_x_inlet_scalar(udc_handle *h, int field, const double *f0driver, const double *fmdriver, const int lb[2], const int ub[2]);
/* Passive scalars on such a handle (&BC BCxs = 2 with BCxm = 2 / 3; cfg->nsv > 0 gives the rows two ghost columns either side, ib-2 .. ie+2:
 * advecc_kappa reads i-2 .. i+1): the inflow profiles svprof [nsv][ktot+2] by the reference's k.  xsi_profile (src/modboundary.f90:844-861:
 * sv(ib-1) = 2 svprof - sv(ib), sv(ib-2) = 2 svprof - sv(ib-1), rows jb .. je, levels kb .. ke+1) and xso_convective (:983-996: sv(ie+1)) run
 * with every `boundary`; sv(ie+2) keeps what an upload put there (no routine of the reference writes it).  udc_field_upload / _download of
 * sv0, svm carry the four ghost columns of a host c-array (lb[0] <= -1, ub[0] >= itot+2).  With obstacles: the c-grid lists as usual. */
int udc_set_open_x_scalars(udc_handle *h, const double *svprof);
int udc_set_open_x_inlet(udc_handle *h, const double *u0driver, const double *umdriver, const double *v0driver, const double *vmdriver,
                         const double *w0driver, const double *wmdriver, const int lb[2], const int ub[2]);

/* checksim's diagnostics (src/modchecksim.f90:76-203) of the state on the device: out[0] = calccourant's number -- the maximum of the
 * SIGNED sum (um dxhi + vm dyi + wm dzhi) dtmn, :111-117 --, out[1] = calcdiffnr's (:142-149), out[2], out[3] = chkdiv's divmax and
 * divtot of u0, v0, w0 (:179-196); dtmn: the mean time step since the last report (:83, :86). */
int udc_checksim(udc_handle *h, double dtmn, double out[4]);
/* The same in two halves: _begin queues the reductions and their copy to the host behind whatever the handle's stream holds and
 * returns at once; _end waits for them and hands the four numbers over.  A driver that reports every time step (tcheck <= dt,
 * src/modchecksim.f90:66) calls _end of the previous report just before the next _begin and so never waits on the device. */
int udc_checksim_begin(udc_handle *h, double dtmn);
int udc_checksim_end(udc_handle *h, double out[4]);
/* divergence of u0 as modchecksim's chkdiv (src/modchecksim.f90:161-203): max |div|, sum div */
int udc_divergence(udc_handle *h, double *divmax, double *divtot);

int udc_sync(udc_handle *h);

/* ---- measurement: HIP-event timing of kernel launches on the library's stream.
 * on = 1: every launch (one chained event per launch boundary: each costs ~6 us on the GPU timeline, ~4 % of a 256^3
 * substep); on = 2: only the launches whose name starts with the prefix given to udc_profile_focus (two events per
 * such launch, nothing between the others): what a timed run can afford. on = 0: off. --- */
int udc_profile_enable(udc_handle *h, int on);
int udc_profile_focus(udc_handle *h, const char *name_prefix);
/* on = 2 only: time the focused launches of the next fused substep and of every n-th after it (n >= 1; default 1 = every
 * one).  A timed run samples:
 * the two events around a launch cost the stream ~11 us, 1.2 % of a 256^3 substep when every launch carries them. */
int udc_profile_every(udc_handle *h, int n);
int udc_profile_reset(udc_handle *h);
/* returns the number of distinct kernels; fills up to cap entries */
int udc_profile_get(udc_handle *h, int cap, char names[][64], double *total_ms, int *launches);

#ifdef __cplusplus
}
#endif
#endif
