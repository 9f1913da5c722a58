/* TEST INFRASTRUCTURE -- CPU oracle, NOT the product.
 *
 * Plain-C restatement of the reference's per-substep dynamical core
 * (uDALES: src/modadvection.f90, src/modsubgrid.f90, src/modpois.f90,
 * src/modtstep.f90 and the modboundary.f90 routines they call), single rank,
 * periodic x/y (+ the inflow / outflow branch in x, orc_set_open_x), used only by tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg to CHECK the HIP library.  Every function cites the
 * reference lines it follows.  Pinned against per-routine and multi-substep
 * vectors produced by the reference's own unmodified Fortran (oracle/_ref,
 * fixtures in tests/golden/), see tests/test_oracle_vs_reference.py.
 *
 * Array conventions (identical to the reference's Fortran storage, i fastest):
 *   "m-arrays"  (u0,v0,w0,um,vm,wm,pres0,ekm,ekh,p,up,vp,wp,pup,pvp,pwp):
 *       extents (0:nx+1, 0:ny+1, 0:nz+1), halo 1, so Fortran index == C index:
 *       a[i + (nx+2)*(j + (ny+2)*k)].  Tendencies use the same box (the k=0
 *       plane, which the reference does not allocate, is simply unused).
 *   "c-arrays"  (sv0, svm, svp: kappa-advected scalars): extents
 *       (-1:nx+2, -1:ny+2, -1:nz+2), halo 2: a[(i+1) + (nx+4)*((j+1) + (ny+4)*(k+1))].
 *   dzf[0..nz+1]  = dzf(kb-1:ke+1),  dzh[0..nz+1] with dzh[k] = dzh(k), k=1..nz+1.
 */
#ifndef UDC_ORACLE_H
#define UDC_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
  int nx, ny, nz;           /* itot, jtot, ktot (single rank: imax=itot, ...) */
  double dx, dy;            /* xlen/itot, ylen/jtot (src/modglobal.f90:710-711) */
  const double *dzf;        /* [nz+2] */
  const double *dzh;        /* [nz+2], entry 0 unused */
  double numol;             /* 1.5e-5   src/modglobal.f90:300 */
  double prandtlmoli;       /* 1/0.71   src/modglobal.f90:303 */
  double prandtli;          /* 1/Prandtl src/modsubgrid.f90:117 */
  double c_vreman;          /* 0.07     src/modsubgriddata.f90:61 */
  double csz;               /* Smagorinsky constant src/modsubgrid.f90:73-77 */
  int sgs;                  /* 0 = DNS (lles false), 1 = Smagorinsky, 2 = Vreman, 3 = one-equation TKE */
  int bctopm;               /* 1 free-slip, 2 no-slip, 3 open to the pressure gradient (src/modglobal.f90:140-153) */
  double uinf, vinf;        /* only for no-slip top (valuetop) */
  int nsv;                  /* passive scalars, kappa scheme (src/modglobal.f90:557-559) */
  int lbottom;              /* floor wall function (src/modibm.f90:49,2021), BCbotm = 3, BCbots = 1 */
  double z0;                /* roughness length (src/modsurfdata.f90:72) */
  int luvolflowr, lvvolflowr;   /* masscorr volume-flow switches (src/modglobal.f90:231) */
  double uflowrate, vflowrate;  /* prescribed volume-mean velocities (src/modglobal.f90:331) */
  int ltempeq;              /* passive temperature equation, iadv_thl = 2 (src/modglobal.f90:176) */
  int bctopt;               /* BCtopT: 1 flux wttop, 2 value thl_top (src/modglobal.f90:144-154) */
  double wttop, thl_top, wtsurf;   /* src/modsurfdata.f90:62,80,81 */
  int lbuoyancy;            /* forces' buoyancy term, dry air (src/modforces.f90:73-84) */
  int coriolis_mode;        /* 0 off, 1 lcoriol, 2 lprofforc (src/modforces.f90:600-717) */
  double om22, om23;        /* src/modglobal.f90:666-673 */
  /* one-equation closure (sgs = 3, loneeqn): constants of initsubgrid src/modsubgrid.f90:63-71 */
  double cm, cn, ch1, ch2, ce1, ce2, thvs;
  int ldelta;
  int lmoist;               /* total water transported, iadv_qt = 2 (src/modglobal.f90:402) */
  int bctopq;               /* BCtopq: 1 flux wqtop, 2 value qt_top (src/modglobal.f90:147-155) */
  double wqtop, qt_top, wqsurf;    /* src/modsurfdata.f90:65,83,84 */
  /* moist thermodynamics (lmoist with lbuoyancy): surface values and pressure (src/modsurfdata.f90:41,58,64) and
   * the level heights zf(kb:ke+kh), zh(kb:ke+kh) as [nz+2] tables indexed by k (src/modglobal.f90:747-751) */
  double thls, qts, ps;
  const double *zf, *zh;
  /* floor with the stability-dependent wall function of Uno et al. 1995 (wfuno, src/modwallfunctions.f90:24-170):
   * BCbotm 2 (momentum, case 91) / 3 (neutral, wfmneutral), BCbotT 1 (flux wtsurf) / 2 (wall temperature thls, case 92);
   * roughness length for heat z0h (src/modsurfdata.f90:73), prandtlturb (src/modglobal.f90:304) */
  int bcbotm, bcbott;
  double z0h, prandtlturb;
  /* top condition of the scalars (src/modboundary.f90:236-247): BCtops 1 = flux wsvtop(n) (fluxtopscal, :1521-1537),
   * 2 = value sv_top(n) (valuetopscal, :1539-1553); up to 4 scalars here */
  int bctops;
  double wsvtop[4], sv_top[4];
  /* NO - NO2 - O3 chemistry on scalars 1-3 (lchem, src/modchem.f90:27-73): rate constant, photolysis rate */
  int lchem;
  double k1, JNO2;
  int lqlnr;                /* condensate by Newton-Raphson on T instead of the one-step formula (src/modthermodynamics.f90:37,448-473) */
  int iadv_thl;             /* 2 = cd2 (advecc_2nd), 7 = kappa (advecc_kappa on thl0c), src/modadvection.f90:64-76 */
  int lbuoycorr;            /* Vreman buoyancy correction (src/modsubgriddata.f90:41, src/modsubgrid.f90:330-353) */
  double Rigc;              /* critical Richardson number (src/modsubgriddata.f90:44) */
} orc_grid;

/* ---- advection: src/modadvection.f90 */
void orc_advecu_2nd(const orc_grid *g, const double *u0, const double *v0, const double *w0,
                    const double *pres0, double *up);
void orc_advecv_2nd(const orc_grid *g, const double *u0, const double *v0, const double *w0,
                    const double *pres0, double *vp);
void orc_advecw_2nd(const orc_grid *g, const double *u0, const double *v0, const double *w0,
                    const double *pres0, double *wp);
void orc_advecc_kappa(const orc_grid *g, const double *u0, const double *v0, const double *w0,
                      const double *c, double *cp);
/* ---- subgrid: src/modsubgrid.f90 + closurebc (src/modboundary.f90:434-505) */
void orc_closure(const orc_grid *g, const double *u0, const double *v0, const double *w0,
                 double *ekm, double *ekh);
/* immersed boundary (src/modibm.f90): per grid 0 u, 1 v, 2 w, 3 c the solid and fluid-boundary point lists ([n][3], 1-based
 * i j k as in solid_*.txt / fluid_boundary_*.txt) and the mask (m-array, 1 fluid / 0 solid, orc_ibm_mask) */
typedef struct {
  const int *sol[4]; int nsol[4];
  const int *bnd[4]; int nbnd[4];
  const double *mask[4];
  int conservative;          /* lconservativeibm (src/modglobal.f90:192): which advecc2nd_corr ibmnorm applies to thl / qt */
} orc_ibm;
void orc_ibm_mask(const orc_grid *g, int is_w, const int *solid, int nsolid, double *mask, int wrapx, int wrapy);
void orc_ibm_diffu_corr(const orc_grid *g, const int *bnd, int nbnd, const double *mask_u, const double *u0, const double *ekm, double *up);
void orc_ibm_diffv_corr(const orc_grid *g, const int *bnd, int nbnd, const double *mask_v, const double *v0, const double *ekm, double *vp);
void orc_ibm_diffw_corr(const orc_grid *g, const int *bnd, int nbnd, const double *mask_w, const double *w0, const double *ekm, double *wp);
void orc_ibm_diffc_corr(const orc_grid *g, const int *bnd, int nbnd, const double *mask_c, const double *var, const double *ekh, double *rhs);
void orc_ibm_solid_m(const orc_grid *g, const int *pts, int n, double *var, double *rhs, double val);
void orc_ibm_solid_c(const orc_grid *g, const int *pts, int n, const double *mask, double *var, double *rhs, double val);
/* thl / qt with an immersed boundary (m-arrays): diffc_corr :1120, solid with the c mask :748, advecc2nd_corr_conservative :889 /
 * _liberal :936; the slab average over the fluid cells of a level (avexy_ibm, src/modmpi.f90:623-664) */
void orc_ibm_diffc_corr_m(const orc_grid *g, const int *bnd, int nbnd, const double *mask_c, const double *var, const double *ekh, double *rhs);
void orc_ibm_solid_cm(const orc_grid *g, const int *pts, int n, const double *mask, double *var, double *rhs, double val);
void orc_ibm_advecc2nd_corr_m(const orc_grid *g, int conservative, const int *bnd, int nbnd, const double *mask_u, const double *mask_v,
                              const double *mask_w, const double *mask_c, const double *u0, const double *v0, const double *w0,
                              const double *var, double *rhs);
double orc_ibm_level_mean_m(const orc_grid *g, const double *f, const double *mask, int k);
void orc_set_ibm(const orc_ibm *b);      /* orc_substep then runs ibmwallfun / ibmnorm (src/program.f90:166, 171) */
void orc_set_closure_thl(const double *thl0);   /* thl0 for the Vreman buoyancy correction inside orc_closure */
void orc_closurebc(const orc_grid *g, double *ekm, double *ekh);
void orc_diffu(const orc_grid *g, const double *u0, const double *v0, const double *w0,
               const double *ekm, double *up);
void orc_diffv(const orc_grid *g, const double *u0, const double *v0, const double *w0,
               const double *ekm, double *vp);
void orc_diffw(const orc_grid *g, const double *u0, const double *v0, const double *w0,
               const double *ekm, double *wp);
void orc_diffc(const orc_grid *g, const double *c, const double *ekh, double *cp);
/* ---- forces (neutral branch): src/modforces.f90:46-133 */
void orc_forces(const orc_grid *g, const double *dpdxl, const double *dpdyl,
                double *up, double *vp, double *wp);
/* ---- floor: `bottom` src/modibm.f90:1998-2100 -> wfmneutral src/modwallfunctions.f90:263-350; momfluxb may be NULL */
void orc_bottom(const orc_grid *g, const double *u0, const double *v0, const double *ekm, const double *ekh,
                const double *sv0, double *up, double *vp, double *svp, double *momfluxb);
/* ---- temperature equation (passive): advecc_2nd src/modadvection.f90:103-155, diffc on an m-array, top / floor */
void orc_advecc_2nd(const orc_grid *g, const double *u0, const double *v0, const double *w0, const double *c, double *cp);
void orc_diffc_m(const orc_grid *g, const double *c, const double *ekh, double *cp);
void orc_thl_top(const orc_grid *g, const double *ekh, double *a);
/* `bottom` with wfuno: momentum (case 91) into up, vp; temperature (case 92, when g->bcbott == 2) into thlp */
void orc_bottom_uno(const orc_grid *g, const double *u0, const double *v0, const double *thl0, const double *ekm, const double *ekh,
                    double *up, double *vp, double *thlp);
/* scalsource, src/modscalsource.f90:379-483: Gaussian point sources (rows xS yS zS SS sigS) and line sources (rows xSb ySb
 * zSb xSe ySe zSe SS sigS) added to the tendency cp (c-array) of one scalar; zf as in orc_grid ([nz+2] by k) */
void orc_scalsource(const orc_grid *g, int npoint, const double *points, int nline, const double *lines, double *cp);
/* thl0c as the reference maintains it: interior = thl0 (src/modtstep.f90:249), periodic lateral ghosts two wide
 * (src/modboundary.f90:87,556-557,645-646), the two planes below the floor never written (zero), the two above the top
 * copies of level ke for a flux condition (:211-213) and never written (zero) for a value condition */
void orc_thl0c_from(const orc_grid *g, const double *thl0, double *thl0c);
/* advection of thl with iadv_thl = 7 (src/modadvection.f90:69-72): thlpc = thlp; advecc_kappa(thl0c, thlpc); thlp = thlpc */
void orc_advec_thl_kappa(const orc_grid *g, const double *u0, const double *v0, const double *w0, const double *thl0c, double *thlp);
/* chem, src/modchem.f90:27-73 (IBM-free: IIc = 1): fully implicit backward Euler of NO + O3 <-> NO2 on sv0(:,:,kb:ke+khc,1:3) */
void orc_chem(const orc_grid *g, double dt, double *sv0);
void orc_qt_top(const orc_grid *g, const double *ekh, double *a);
void orc_qt_floor(const orc_grid *g, const double *ekh, const double *qt0, double *qtp);
void orc_buoyancy(const orc_grid *g, const double *thl0, double *wp);
void orc_thl_floor(const orc_grid *g, const double *ekh, const double *thl0, double *thlp);
/* ---- one-equation closure: src/modsubgrid.f90:363-400 (closure), :627-669 (diffe), :415-538 (sources) */
void orc_closure_tke(const orc_grid *g, const double *e120, const double *thl0, double *ekm, double *ekh);
void orc_diffe(const orc_grid *g, const double *e120, const double *ekm, double *e12p);
void orc_sources(const orc_grid *g, const double *u0, const double *v0, const double *w0, const double *e120,
                 const double *thl0, const double *ekm, const double *ekh, double *e12p);
/* ---- coriolis: src/modforces.f90:600-717 */
void orc_coriolis(const orc_grid *g, const double *u0, const double *v0, const double *w0, const double *ug,
                  double *up, double *vp, double *wp);
/* ---- masscorr: src/modforces.f90:328-497 (volume-flow branches) */
void orc_masscorr(const orc_grid *g, int rk3step, double dt, double *up, const double *um, double *vp, const double *vm);
/* ---- pressure: src/modpois.f90 (ipoiss = POISS_FFT2D, BCzp = 1, periodic x,y) */
void orc_fillps(const orc_grid *g, double rk3coef, const double *up, const double *vp,
                const double *wp, const double *um, const double *vm, const double *wm,
                double *pup, double *pvp, double *pwp, double *p);
void orc_set_poisson_bczp(int bczp);      /* 1 (default): solmpj; 2: cosine transform in z, src/modpois.f90:559-590 */
void orc_set_poisson_bczp(int bczp);      /* 1 (default): solmpj; 2: cosine transform in z, src/modpois.f90:559-590 */
void orc_poisson_solve(const orc_grid *g, double *p);   /* in: rhs in p interior; out: p interior */
void orc_tderive(const orc_grid *g, double *p, double *up, double *vp, double *wp, double *pres0);
/* ---- time stepping: src/modtstep.f90:171-340 */
void orc_tstep_integrate(const orc_grid *g, int rk3step, double dt, double *u0, double *v0,
                         double *w0, double *um, double *vm, double *wm, double *up, double *vp,
                         double *wp, double *sv0, double *svm, double *svp);
/* ---- halos + boundary (periodic / top / bottom subset): src/modboundary.f90:67-109,115-247 */
void orc_halos_m(const orc_grid *g, double *a);          /* x then y periodic wrap, halo 1 */
void orc_halos_c(const orc_grid *g, double *a);          /* halo 2 */
/* fluxtopscal / valuetopscal on sv0 and svm (both ghost planes get the same value); ekh: the current eddy diffusivity */
void orc_scalar_tops(const orc_grid *g, const double *ekh, double *sv0, double *svm);
void orc_boundary(const orc_grid *g, double *u0, double *v0, double *w0, double *um, double *vm,
                  double *wm, double *sv0, double *svm);

/* ---- whole substep, src/program.f90:132-222 restricted to the dynamical core */
typedef struct {
  double *u0, *v0, *w0, *um, *vm, *wm, *up, *vp, *wp, *pres0, *ekm, *ekh, *p, *pup, *pvp, *pwp;
  double *sv0, *svm, *svp;                /* nsv consecutive c-arrays each */
  const double *dpdxl, *dpdyl;            /* [nz+2] indexed by Fortran k, or NULL (no forces) */
  double *thl0, *thlm, *thlp;             /* m-arrays, used when g->ltempeq */
  const double *thlpcar;                  /* [nz+2] or NULL */
  const double *ug;                       /* [nz+2] geostrophic wind (lprofforc) or NULL */
  double *e120, *e12m, *e12p;             /* m-arrays, used when g->sgs == 3 */
  double *qt0, *qtm, *qtp;                /* m-arrays, used when g->lmoist */
  /* moist thermodynamics state kept between calls: ORC_TH_N tables of [nz+2] indexed by k (presf, presh, exnf, exnh,
   * thvh, thl0av, qt0av, ql0av, th0av) followed by one flag (0 = diagfld has not run yet); and ql0 (m-array) */
  double *thermo, *ql0;
  double *thl0c;                          /* c-array: the wide copy of thl0 kappa advection runs on (g->iadv_thl == 7) */
  const double *svsrc;                    /* nsv consecutive c-arrays: constant scalar sources added to svp (scalsource), or NULL */
} orc_state;
enum { ORC_TH_PRESF, ORC_TH_PRESH, ORC_TH_EXNF, ORC_TH_EXNH, ORC_TH_THVH, ORC_TH_THL0AV, ORC_TH_QT0AV, ORC_TH_QL0AV,
       ORC_TH_TH0AV, ORC_TH_N };
/* thermodynamics, src/modthermodynamics.f90:57-124 (lmoist): thermo, diagfld (fromztop twice), calc_halflev, thermo on
 * the half levels, calthv's thv0h and its slab average thvh.  Called once before the first substep
 * (src/program.f90:120) and by orc_substep at its end (:214) when g->lmoist && g->lbuoyancy. */
void orc_thermodynamics(const orc_grid *g, orc_state *s);
/* forces' buoyancy term with the moist thv0h (src/modforces.f90:73-84, src/modthermodynamics.f90:142-152) */
/* the state calthv's moist dthvdz (one-equation closure with lmoist) reads: qt0, ql0, the thermodynamics tables; NULL = dry */
void orc_set_moist_context(const orc_state *s);
void orc_buoyancy_moist(const orc_grid *g, const orc_state *s, double *wp);
/* the open lid (BCtopm = 3): bcpup's and tderive's branches; orc_set_lid(pres0, wp) makes orc_fillps apply the first */
void orc_set_lid(const double *pres0, double *wp);
void orc_bcpup_lid(const orc_grid *g, double rk3coef, const double *pres0, const double *wm, double *wp, double *pwp);
void orc_tderive_lid(const orc_grid *g, const double *p, double *wp);
/* inflow / outflow in x (&BC BCxm = 2; the reference opens the lid with it: g->bctopm = 3).  orc_set_open_x(1, uprof, vprof) -- [nz+2]
 * by k, entry nz+1 zero like the reference's uprof(ke+1) -- switches closurebc, bcpup (inside orc_fillps; orc_substep hands it u0 and
 * up), bcp (inside orc_tderive), the x transform of orc_poisson_solve (REDFT10 / REDFT01), u0(ie+1) in orc_tstep_integrate and
 * orc_halos_m; orc_boundary_open_x is `boundary`'s xmi_profile + xmo_convective.  The outlet's speed uouttot is a variable of the run:
 * _uouttot sets it; _outflow(wlev[nz], u, hold_first) makes orc_substep refresh it like the reference (udcore_oracle.c). */
void orc_set_open_x(int on, const double *uprof, const double *vprof);
void orc_set_open_x_uouttot(double uouttot);
double orc_open_x_uouttot(void);
void orc_set_open_x_fields(const double *u0, double *up);
void orc_set_open_x_outflow(const double *wlev, double uouttot, int hold_first);
void orc_set_open_x_scalars(const double *svprof);      /* BCxs = 2: the scalars' inflow profiles [nsv][nz+2] by k (NULL: off) */
void orc_boundary_open_x_sv(const orc_grid *g, double rk3coef, double *sv0, double *svm);
void orc_set_open_x_thl(const double *thlprof);      /* BCxT = 2: the temperature's inflow profile [nz+2] by k (NULL: off) */
void orc_boundary_open_x_thl(const orc_grid *g, double rk3coef, double *thl0, double *thlm);
void orc_boundary_open_x(const orc_grid *g, double rk3coef, double *u0, double *v0, double *w0, double *um, double *vm, double *wm);
void orc_substep(const orc_grid *g, orc_state *s, int rk3step, double dt);

#ifdef __cplusplus
}
#endif
#endif
