!> ISO_C_BINDING interface to libudcore.so (include/udcore.h) plus the residency manager the
!! four drop-in modules (modadvection, modsubgrid, modpois, modtstep) share.
!!
!! The reference's seam is "argument-less module procedures working on the module-global
!! arrays of modfields" (SURVEY.md section 8b).  The device library owns device mirrors of
!! those arrays; this module decides what crosses PCIe around each call:
!!
!!   UDC_RESIDENCY=0 (default, "strict")   every entry point uploads the host arrays it reads and
!!                        downloads the ones it writes -- correct with ANY untouched host routine
!!                        in between (IBM, EB, trees, statistics ...), PCIe-bound.
!!   UDC_RESIDENCY=1 ("tendencies")        velocities/pressure stay resident; only up,vp,wp (the
!!                        arrays the reference's add-on physics modify, docs/udales-architecture.md:104)
!!                        are exchanged, and u0,v0,w0 are pulled after tstep_integrate.
!!   UDC_RESIDENCY=2 ("device")            inside the time loop nothing moves: the drop-in routines only record their call
!!                        (udc_set_deferred) and tstep_integrate launches the fused substep.  The host arrays are
!!                        refreshed (udc_pull_all) at the end of the run, when a restart file is due and every
!!                        UDC_PULL_EVERY time steps (default: never) -- see the drop-in `halos` / `thermodynamics`.
!!                        For decks whose remaining host routines in the loop are no-ops.
!! Before the first tstep_update (start-up: readinitfiles calls halos / boundary / thermodynamics on host arrays that
!! it is still filling) every mode behaves like 0.
module udc_iface
  use iso_c_binding
  implicit none
  public

  ! field ids, include/udcore.h
  integer(c_int), parameter :: UDC_U0 = 0, UDC_V0 = 1, UDC_W0 = 2, UDC_UM = 3, UDC_VM = 4, UDC_WM = 5, &
                               UDC_UP = 6, UDC_VP = 7, UDC_WP = 8, UDC_PRES0 = 9, UDC_P = 10, &
                               UDC_EKM = 11, UDC_EKH = 12, UDC_SV0 = 13, UDC_SVM = 14, UDC_SVP = 15, &
                               UDC_THL0 = 13 + 45, UDC_THLM = 14 + 45, UDC_THLP = 15 + 45, &   ! scalar slot 15
                               UDC_QT0 = 13 + 39, UDC_QTM = 14 + 39, UDC_QTP = 15 + 39, &      ! scalar slot 13
                               UDC_E120 = 13 + 42, UDC_E12M = 14 + 42, UDC_E12P = 15 + 42      ! scalar slot 14

  type, bind(C) :: udc_config
    integer(c_int) :: itot, jtot, ktot
    integer(c_int) :: nranks, rank, device
    real(c_double) :: dx, dy
    type(c_ptr)    :: dzf, dzh
    real(c_double) :: numol, prandtlmoli, prandtli, c_vreman, csz
    integer(c_int) :: sgs, bctopm
    real(c_double) :: uinf, vinf
    integer(c_int) :: nsv
    integer(c_int) :: lbottom
    real(c_double) :: z0
  end type udc_config

  type(c_ptr), save :: udc_h = c_null_ptr
  integer, save :: udc_residency = 0
  ! floor wall function on the device (device-resident mode only: in the other modes the host's own `bottom`,
  ! src/modibm.f90:1998, edits the pulled tendencies and must not be applied twice); set by udc_set_floor
  logical, save :: udc_floor_on = .false.
  real(c_double), save :: udc_floor_z0 = -1.
  logical, save :: udc_in_loop = .false.      !< set by the first tstep_update
  logical, save :: udc_lqlnr = .false.        !< &DYNAMICS lqlnr (owned by modthermodynamics; copied by initthermodynamics)
  logical, save :: udc_need_avg = .false.     !< some host routine reads diagfld's slab averages (lstend, nudge, grwdamp, fixuinf, shiftedPBCs)
  logical, save :: udc_host_fresh = .true.    !< the host arrays hold the current state
  logical, save :: udc_bottom_diag_on = .false.   !< the device keeps bottom's tau_x / tau_y / thl_flux planes
  !> per-level tables tend(i,j,k) += A(k) + B(k) field(i,j,k) under construction on the host (udc_tab_start/apply):
  !! (level, row, when) with when = 0: applied before masscorr (lstend, nudge), 1: after it (fixuinf1, grwdamp)
  integer, parameter :: ROW_UP = 1, ROW_VP = 2, ROW_WP = 3, ROW_THLP = 4, ROW_QTP = 5, ROW_SVP = 5
  real(c_double), allocatable, save :: udc_tabA(:, :, :), udc_tabB(:, :, :)
  logical, save :: udc_tab_open(0:1) = .false.
  logical, save :: udc_tab_registered(0:1) = .false.
  logical, save :: udc_row_registered(ROW_SVP + 16, 0:1) = .false.
  logical, save :: udc_scalsrc_on = .false.   !< constant scalar sources are registered with the device (device mode)
  logical, save :: udc_stats_on_device = .false.      !< the drop-in modstatsdump is linked and has something to do: the reference's
                                                      !! statsdump is not there to read the host arrays
  integer, save :: udc_pull_every = 0         !< UDC_PULL_EVERY: refresh the host arrays every n-th time step in device mode
  ! host-side phase clock (UDC_TIMERS=1): wall time the time loop spends in the transfers and in the drop-in routines, printed by
  ! udc_timers_report (exitstatsdump calls it): what of the loop's time is the one-time upload / download, what the per-step
  ! waits, what the host's own work
  integer, parameter :: UDC_T_UPLOAD = 1, UDC_T_DOWNLOAD = 2, UDC_T_INTEGRATE = 3, UDC_T_CHECKSIM = 4, UDC_T_STATS = 5, UDC_T_RECORD = 6, &
                        UDC_T_THERMO = 7, UDC_T_HALOS = 8, UDC_T_N = 8
  character(10), parameter :: udc_t_name(UDC_T_N) = [character(10) :: 'upload', 'download', 'integrate', 'checksim', 'statsdump', &
                                                     'record', 'thermo', 'halos']
  logical, save :: udc_timers_on = .false.
  real(8), save :: udc_t_sum(UDC_T_N) = 0.
  integer(8), save :: udc_t_cnt(UDC_T_N) = 0, udc_t_start(UDC_T_N) = 0

  interface
    integer(c_int) function udc_create(cfg, h) bind(C, name='udc_create')
      import :: c_int, c_ptr, udc_config
      type(udc_config), intent(in) :: cfg
      type(c_ptr), intent(out) :: h
    end function
    integer(c_int) function udc_destroy(h) bind(C, name='udc_destroy')
      import :: c_int, c_ptr
      type(c_ptr), value :: h
    end function
    type(c_ptr) function udc_last_error() bind(C, name='udc_last_error')
      import :: c_ptr
    end function
    integer(c_int) function udc_field_upload(h, field, host, lb, ub) bind(C, name='udc_field_upload')
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: h
      integer(c_int), value :: field
      real(c_double), intent(in) :: host(*)
      integer(c_int), intent(in) :: lb(3), ub(3)
    end function
    integer(c_int) function udc_field_download(h, field, host, lb, ub) bind(C, name='udc_field_download')
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: h
      integer(c_int), value :: field
      real(c_double), intent(inout) :: host(*)
      integer(c_int), intent(in) :: lb(3), ub(3)
    end function
    integer(c_int) function udc_set_forcing(h, dpdxl, dpdyl, n) bind(C, name='udc_set_forcing')
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: h
      real(c_double), intent(in) :: dpdxl(*), dpdyl(*)
      integer(c_int), value :: n
    end function
    integer(c_int) function udc_advection(h) bind(C, name='udc_advection')
      import :: c_int, c_ptr
      type(c_ptr), value :: h
    end function
    integer(c_int) function udc_subgrid(h) bind(C, name='udc_subgrid')
      import :: c_int, c_ptr
      type(c_ptr), value :: h
    end function
    integer(c_int) function udc_set_tempeq(h, iadv_thl, bctopt, wttop, thl_top, bcbott, wtsurf) bind(C, name='udc_set_tempeq')
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: h
      integer(c_int), value :: iadv_thl, bctopt, bcbott
      real(c_double), value :: wttop, thl_top, wtsurf
    end function
    integer(c_int) function udc_set_tke(h, cm, cn, ch1, ch2, ce1, ce2, e12min, grav, thvs, ldelta) bind(C, name='udc_set_tke')
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: h
      real(c_double), value :: cm, cn, ch1, ch2, ce1, ce2, e12min, grav, thvs
      integer(c_int), value :: ldelta
    end function
    integer(c_int) function udc_set_buoyancy(h, lbuoyancy, grav) bind(C, name='udc_set_buoyancy')
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: h
      integer(c_int), value :: lbuoyancy
      real(c_double), value :: grav
    end function
    integer(c_int) function udc_set_buoycorr(h, lbuoycorr, rigc) bind(C, name='udc_set_buoycorr')
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: h
      integer(c_int), value :: lbuoycorr
      real(c_double), value :: rigc
    end function
    integer(c_int) function udc_set_ibm_points(h, grid, solid, nsolid, bound, nbound) bind(C, name='udc_set_ibm_points')
      import :: c_int, c_ptr
      type(c_ptr), value :: h
      integer(c_int), value :: grid, nsolid, nbound
      integer(c_int), intent(in) :: solid(3, *), bound(3, *)
    end function
    integer(c_int) function udc_ibm_commit(h) bind(C, name='udc_ibm_commit')
      import :: c_int, c_ptr
      type(c_ptr), value :: h
    end function
    integer(c_int) function udc_bottom_diagnostics(h, on) bind(C, name='udc_bottom_diagnostics')
      import :: c_int, c_ptr
      type(c_ptr), value :: h
      integer(c_int), value :: on
    end function
    integer(c_int) function udc_bottom_diag_get(h, which, out) bind(C, name='udc_bottom_diag_get')
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: h
      integer(c_int), value :: which
      real(c_double), intent(out) :: out(*)
    end function
    integer(c_int) function udc_set_ibm_conservative(h, on) bind(C, name='udc_set_ibm_conservative')
      import :: c_int, c_ptr
      type(c_ptr), value :: h
      integer(c_int), value :: on
    end function
    integer(c_int) function udc_set_ibm_mask_wrap(h, wrapx, wrapy) bind(C, name='udc_set_ibm_mask_wrap')
      import :: c_int, c_ptr
      type(c_ptr), value :: h
      integer(c_int), value :: wrapx, wrapy
    end function
    integer(c_int) function udc_ibmwallfun(h) bind(C, name='udc_ibmwallfun')
      import :: c_int, c_ptr
      type(c_ptr), value :: h
    end function
    integer(c_int) function udc_ibmnorm(h) bind(C, name='udc_ibmnorm')
      import :: c_int, c_ptr
      type(c_ptr), value :: h
    end function
    integer(c_int) function udc_set_ibm_wallfun(h, iwallmom, prandtlturb, zf, zh) bind(C, name='udc_set_ibm_wallfun')
      import :: c_ptr, c_int, c_double
      type(c_ptr), value :: h
      integer(c_int), value :: iwallmom
      real(c_double), value :: prandtlturb
      real(c_double), intent(in) :: zf(*), zh(*)
    end function udc_set_ibm_wallfun
    integer(c_int) function udc_set_ibm_wallheat(h, iwalltemp) bind(C, name='udc_set_ibm_wallheat')
      import :: c_ptr, c_int
      type(c_ptr), value :: h
      integer(c_int), value :: iwalltemp
    end function udc_set_ibm_wallheat
    integer(c_int) function udc_set_poisson_bczp(h, bczp) bind(C, name='udc_set_poisson_bczp')
      import :: c_int, c_ptr
      type(c_ptr), value :: h
      integer(c_int), value :: bczp
    end function
    integer(c_int) function udc_set_ibm_facet_output(h, nfcts, faca, nsec, fac_u, fac_v, fac_w, fac_c, npres, pcell, parea, pfac) &
        bind(C, name='udc_set_ibm_facet_output')
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: h
      integer(c_int), value :: nfcts, npres
      real(c_double), intent(in) :: faca(*), parea(*)
      integer(c_int), intent(in) :: nsec(4), fac_u(*), fac_v(*), fac_w(*), fac_c(*), pcell(3, *), pfac(*)
    end function
    integer(c_int) function udc_ibm_facet_sample(h, dt) bind(C, name='udc_ibm_facet_sample')
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: h
      real(c_double), value :: dt
    end function
    integer(c_int) function udc_ibm_facet_get(h, out, reset) bind(C, name='udc_ibm_facet_get')
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: h
      real(c_double), intent(out) :: out(*)
      integer(c_int), value :: reset
    end function
    integer(c_int) function udc_set_ibm_wallmoist(h, iwallmoist, n, lgr, qwall, hurel, resc, ress) bind(C, name='udc_set_ibm_wallmoist')
      import :: c_ptr, c_int, c_double
      type(c_ptr), value :: h
      integer(c_int), value :: iwallmoist, n
      integer(c_int), intent(in) :: lgr(*)
      real(c_double), intent(in) :: qwall(*), hurel(*), resc(*), ress(*)
    end function udc_set_ibm_wallmoist
    integer(c_int) function udc_set_ibm_sections(h, grid, n, cell, area, dist, norm, z0, z0h, tsurf, comprec, recpt, recids, tmask) &
        bind(C, name='udc_set_ibm_sections')
      import :: c_ptr, c_int, c_double
      type(c_ptr), value :: h
      integer(c_int), value :: grid, n
      integer(c_int), intent(in) :: cell(*), comprec(*), recids(*)
      real(c_double), intent(in) :: area(*), dist(*), norm(*), z0(*), z0h(*), tsurf(*), recpt(*), tmask(*)
    end function udc_set_ibm_sections
    integer(c_int) function udc_set_scalar_bcx(h, bcxs, svprof, uouttot) bind(C, name='udc_set_scalar_bcx')
      import :: c_ptr, c_int, c_double
      type(c_ptr), value :: h
      integer(c_int), value :: bcxs
      real(c_double), intent(in) :: svprof(*)
      real(c_double), value :: uouttot
    end function udc_set_scalar_bcx
    integer(c_int) function udc_set_scalar_bcx_outflow(h, wlev) bind(C, name='udc_set_scalar_bcx_outflow')
      import :: c_ptr, c_int, c_double
      type(c_ptr), value :: h
      real(c_double), intent(in) :: wlev(*)
    end function udc_set_scalar_bcx_outflow
    integer(c_int) function udc_set_floor_air_temperature(h, thl_kb) bind(C, name='udc_set_floor_air_temperature')
      import :: c_ptr, c_int, c_double
      type(c_ptr), value :: h
      real(c_double), value :: thl_kb
    end function udc_set_floor_air_temperature
    integer(c_int) function udc_set_fkar(h, fkar) bind(C, name='udc_set_fkar')
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: h
      real(c_double), value :: fkar
    end function
    integer(c_int) function udc_set_floor_wf(h, bcbotm, bcbott, thls, z0h, prandtlturb) bind(C, name='udc_set_floor_wf')
      import :: c_ptr, c_int, c_double
      type(c_ptr), value :: h
      integer(c_int), value :: bcbotm, bcbott
      real(c_double), value :: thls, z0h, prandtlturb
    end function udc_set_floor_wf
    integer(c_int) function udc_set_chem(h, lchem, k1, jno2) bind(C, name='udc_set_chem')
      import :: c_ptr, c_int, c_double
      type(c_ptr), value :: h
      integer(c_int), value :: lchem
      real(c_double), value :: k1, jno2
    end function udc_set_chem
    integer(c_int) function udc_set_scalar_top(h, n, bctops, value) bind(C, name='udc_set_scalar_top')
      import :: c_ptr, c_int, c_double
      type(c_ptr), value :: h
      integer(c_int), value :: n, bctops
      real(c_double), value :: value
    end function udc_set_scalar_top
    integer(c_int) function udc_set_moisture(h, iadv_qt, bctopq, wqtop, qt_top, bcbotq, wqsurf) bind(C, name='udc_set_moisture')
      import :: c_ptr, c_int, c_double
      type(c_ptr), value :: h
      integer(c_int), value :: iadv_qt, bctopq, bcbotq
      real(c_double), value :: wqtop, qt_top, wqsurf
    end function udc_set_moisture
    integer(c_int) function udc_set_moist_thermo(h, thls, qts, ps, zf, zh, n, lqlnr) bind(C, name='udc_set_moist_thermo')
      import :: c_ptr, c_int, c_double
      type(c_ptr), value :: h
      real(c_double), value :: thls, qts, ps
      real(c_double), intent(in) :: zf(*), zh(*)
      integer(c_int), value :: n, lqlnr
    end function udc_set_moist_thermo
    integer(c_int) function udc_thermodynamics(h) bind(C, name='udc_thermodynamics')
      import :: c_ptr, c_int
      type(c_ptr), value :: h
    end function udc_thermodynamics
    integer(c_int) function udc_calthv(h) bind(C, name='udc_calthv')
      import :: c_int, c_ptr
      type(c_ptr), value :: h
    end function udc_calthv
    integer(c_int) function udc_set_thl_source(h, thlpcar, n) bind(C, name='udc_set_thl_source')
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: h
      real(c_double), intent(in) :: thlpcar(*)
      integer(c_int), value :: n
    end function
    integer(c_int) function udc_bottom(h) bind(C, name='udc_bottom')
      import :: c_int, c_ptr
      type(c_ptr), value :: h
    end function
    integer(c_int) function udc_forces(h) bind(C, name='udc_forces')
      import :: c_int, c_ptr
      type(c_ptr), value :: h
    end function
    integer(c_int) function udc_poisson(h, rk3step, dt) bind(C, name='udc_poisson')
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: h
      integer(c_int), value :: rk3step
      real(c_double), value :: dt
    end function
    integer(c_int) function udc_tstep_integrate(h, rk3step, dt) bind(C, name='udc_tstep_integrate')
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: h
      integer(c_int), value :: rk3step
      real(c_double), value :: dt
    end function
    integer(c_int) function udc_halos(h) bind(C, name='udc_halos')
      import :: c_int, c_ptr
      type(c_ptr), value :: h
    end function
    integer(c_int) function udc_boundary(h) bind(C, name='udc_boundary')
      import :: c_int, c_ptr
      type(c_ptr), value :: h
    end function
    integer(c_int) function udc_tstep_maxima(h, dt, courtot, diffnrtot) bind(C, name='udc_tstep_maxima')
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: h
      real(c_double), value :: dt
      real(c_double), intent(out) :: courtot, diffnrtot
    end function
    integer(c_int) function udc_substep(h, rk3step, dt, with_forces) bind(C, name='udc_substep')
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: h
      integer(c_int), value :: rk3step, with_forces
      real(c_double), value :: dt
    end function
    integer(c_int) function udc_divergence(h, divmax, divtot) bind(C, name='udc_divergence')
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: h
      real(c_double), intent(out) :: divmax, divtot
    end function
    integer(c_int) function udc_checksim(h, dtmn, d) bind(C, name='udc_checksim')
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: h
      real(c_double), value :: dtmn
      real(c_double), intent(out) :: d(4)
    end function
    integer(c_int) function udc_checksim_begin(h, dtmn) bind(C, name='udc_checksim_begin')
      import :: c_ptr, c_int, c_double
      type(c_ptr), value :: h
      real(c_double), value :: dtmn
    end function udc_checksim_begin
    integer(c_int) function udc_checksim_end(h, d) bind(C, name='udc_checksim_end')
      import :: c_ptr, c_int, c_double
      type(c_ptr), value :: h
      real(c_double) :: d(4)
    end function udc_checksim_end
    integer(c_int) function udc_comm_unique_id(id) bind(C, name='udc_comm_unique_id')
      import :: c_int, c_signed_char
      integer(c_signed_char), intent(out) :: id(128)
    end function
    integer(c_int) function udc_comm_init(h, id) bind(C, name='udc_comm_init')
      import :: c_int, c_ptr, c_signed_char
      type(c_ptr), value :: h
      integer(c_signed_char), intent(in) :: id(128)
    end function
    integer(c_int) function udc_set_coriolis(h, mode, om22, om23, ug, n) bind(C, name='udc_set_coriolis')
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: h
      integer(c_int), value :: mode, n
      real(c_double), value :: om22, om23
      real(c_double), intent(in) :: ug(*)
    end function
    integer(c_int) function udc_coriolis(h) bind(C, name='udc_coriolis')
      import :: c_int, c_ptr
      type(c_ptr), value :: h
    end function
#ifdef UDC_TEST_TRANSPORT
    integer(c_int) function udc_comm_init_shm(h, name) bind(C, name='udc_comm_init_shm')
      import :: c_ptr, c_int, c_char
      type(c_ptr), value :: h
      character(kind=c_char), intent(in) :: name(*)
    end function udc_comm_init_shm
#endif
    integer(c_int) function udc_set_masscorr_outflow(h, lu, uflow) bind(C, name='udc_set_masscorr_outflow')
      import :: c_ptr, c_int, c_double
      type(c_ptr), value :: h
      integer(c_int), value :: lu
      real(c_double), value :: uflow
    end function udc_set_masscorr_outflow
    integer(c_int) function udc_set_masscorr(h, lu, uflow, lv, vflow) bind(C, name='udc_set_masscorr')
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: h
      integer(c_int), value :: lu, lv
      real(c_double), value :: uflow, vflow
    end function
    integer(c_int) function udc_masscorr(h, rk3step, dt) bind(C, name='udc_masscorr')
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: h
      integer(c_int), value :: rk3step
      real(c_double), value :: dt
    end function
    integer(c_int) function udc_set_level_forcing(h, tend, src, A, B, n, when) bind(C, name='udc_set_level_forcing')
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: h
      integer(c_int), value :: tend, src, n, when
      real(c_double), intent(in) :: A(*), B(*)
    end function
    integer(c_int) function udc_level_forcings(h, when) bind(C, name='udc_level_forcings')
      import :: c_int, c_ptr
      type(c_ptr), value :: h
      integer(c_int), value :: when
    end function
    integer(c_int) function udc_slab_averages(h, fields, nf, avg, n) bind(C, name='udc_slab_averages')
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: h
      integer(c_int), intent(in) :: fields(*)
      integer(c_int), value :: nf, n
      real(c_double), intent(out) :: avg(*)
    end function
    integer(c_int) function udc_set_shifted_pbc(h, a, sinx, nx, u0av, nz) bind(C, name='udc_set_shifted_pbc')
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: h
      real(c_double), value :: a
      real(c_double), intent(in) :: sinx(*), u0av(*)
      integer(c_int), value :: nx, nz
    end function
    integer(c_int) function udc_shifted_pbcs(h) bind(C, name='udc_shifted_pbcs')
      import :: c_int, c_ptr
      type(c_ptr), value :: h
    end function
    integer(c_int) function udc_set_scalar_source(h, n, src, lb, ub) bind(C, name='udc_set_scalar_source')
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: h
      integer(c_int), value :: n
      real(c_double), intent(in) :: src(*)
      integer(c_int), intent(in) :: lb(3), ub(3)
    end function
    integer(c_int) function udc_scalsource(h) bind(C, name='udc_scalsource')
      import :: c_int, c_ptr
      type(c_ptr), value :: h
    end function
    integer(c_int) function udc_thermo_state(h, tables, n, set) bind(C, name='udc_thermo_state')
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: h
      real(c_double), intent(inout) :: tables(*)
      integer(c_int), value :: n, set
    end function
    integer(c_int) function udc_set_deferred(h, on) bind(C, name='udc_set_deferred')
      import :: c_int, c_ptr
      type(c_ptr), value :: h
      integer(c_int), value :: on
    end function
    integer(c_int) function udc_flush(h) bind(C, name='udc_flush')
      import :: c_int, c_ptr
      type(c_ptr), value :: h
    end function
    ! statistics on the device (include/udcore.h: tdump accumulators, xytdump profiles)
    integer(c_int) function udc_stats_enable(h, on) bind(C, name='udc_stats_enable')
      import :: c_ptr, c_int
      type(c_ptr), value :: h
      integer(c_int), value :: on
    end function udc_stats_enable
    integer(c_int) function udc_stats_sample(h, tsamplep, tstatsdumpp) bind(C, name='udc_stats_sample')
      import :: c_ptr, c_int, c_double
      type(c_ptr), value :: h
      real(c_double), value :: tsamplep, tstatsdumpp
    end function udc_stats_sample
    integer(c_int) function udc_stats_get(h, id, host, lb, ub) bind(C, name='udc_stats_get')
      import :: c_ptr, c_int, c_double
      type(c_ptr), value :: h
      integer(c_int), value :: id
      real(c_double), intent(inout) :: host(*)
      integer(c_int), intent(in) :: lb(3), ub(3)
    end function udc_stats_get
    integer(c_int) function udc_stats_set_masks(h, bits, counts) bind(C, name='udc_stats_set_masks')
      import :: c_ptr, c_int, c_signed_char
      type(c_ptr), value :: h
      integer(c_signed_char), intent(in) :: bits(*)
      integer(c_int), intent(in) :: counts(*)
    end function udc_stats_set_masks
    integer(c_int) function udc_stats_xyt(h, table) bind(C, name='udc_stats_xyt')
      import :: c_ptr, c_int, c_double
      type(c_ptr), value :: h
      real(c_double), intent(inout) :: table(*)
    end function udc_stats_xyt
    integer(c_int) function udc_deferred_stats(h, fused, unfused) bind(C, name='udc_deferred_stats')
      import :: c_int, c_ptr, c_long
      type(c_ptr), value :: h
      integer(c_long), intent(out) :: fused, unfused
    end function
    integer(c_int) function udc_sync(h) bind(C, name='udc_sync')
      import :: c_int, c_ptr
      type(c_ptr), value :: h
    end function
    ! inflow / outflow in x (&BC BCxm = 2; include/udcore.h, udc_xopen.hip)
    integer(c_int) function udc_create_open_x(cfg, uprof, vprof, h) bind(C, name='udc_create_open_x')
      import :: c_int, c_ptr, c_double, udc_config
      type(udc_config), intent(in) :: cfg
      real(c_double), intent(in) :: uprof(*), vprof(*)
      type(c_ptr), intent(out) :: h
    end function
    integer(c_int) function udc_set_open_x_profile(h, uprof, vprof) bind(C, name='udc_set_open_x_profile')
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: h
      real(c_double), intent(in) :: uprof(*), vprof(*)
    end function
    integer(c_int) function udc_set_open_x_sample_gap(h, on) bind(C, name='udc_set_open_x_sample_gap')
      import :: c_ptr, c_int
      type(c_ptr), value :: h
      integer(c_int), value :: on
    end function udc_set_open_x_sample_gap
    integer(c_int) function udc_set_open_x_outflow(h, wlev, uouttot, hold_first) bind(C, name='udc_set_open_x_outflow')
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: h
      type(c_ptr), value :: wlev      ! c_loc of [ktot] weights, or c_null_ptr: the constant uouttot
      real(c_double), value :: uouttot
      integer(c_int), value :: hold_first
    end function
    integer(c_int) function udc_set_open_x_scalars(h, svprof) bind(C, name='udc_set_open_x_scalars')
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: h
      real(c_double), intent(in) :: svprof(*)
    end function
    integer(c_int) function udc_set_open_x_thl(h, thlprof) bind(C, name='udc_set_open_x_thl')
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: h
      real(c_double), intent(in) :: thlprof(*)
    end function
    integer(c_int) function udc_set_open_x_qt(h, qtprof) bind(C, name='udc_set_open_x_qt')
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: h
      real(c_double), intent(in) :: qtprof(*)
    end function
    integer(c_int) function udc_set_open_x_inlet_scalar(h, field, f0d, fmd, lb, ub) bind(C, name='udc_set_open_x_inlet_scalar')
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: h
      integer(c_int), value :: field
      real(c_double), intent(in) :: f0d(*), fmd(*)
      integer(c_int), intent(in) :: lb(2), ub(2)
    end function
    integer(c_int) function udc_set_open_x_inlet(h, u0d, umd, v0d, vmd, w0d, wmd, lb, ub) bind(C, name='udc_set_open_x_inlet')
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: h
      real(c_double), intent(in) :: u0d(*), umd(*), v0d(*), vmd(*), w0d(*), wmd(*)
      integer(c_int), intent(in) :: lb(2), ub(2)
    end function
    integer(c_int) function udc_set_boundary_rk3coef(h, rk3coef) bind(C, name='udc_set_boundary_rk3coef')
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: h
      real(c_double), value :: rk3coef
    end function
  end interface

contains

  !> The reference's error convention: message on unit 0, stop 1 (e.g. src/modadvection.f90:52-53).
  subroutine udc_check(rc, what)
    integer(c_int), intent(in) :: rc
    character(*), intent(in) :: what
    character(kind=c_char), pointer :: msg(:)
    integer :: n
    if (rc == 0) return
    call c_f_pointer(udc_last_error(), msg, [512])
    n = 1
    do while (n < 512 .and. msg(n) /= c_null_char)
      n = n + 1
    end do
    write (0, *) 'ERROR: libudcore ', what, ': ', msg(1:n - 1)
    stop 1
  end subroutine udc_check

  !> Binding for modibm (INTEGRATION.md section 3): `call udc_set_floor(lbottom, z0)` from initibm,
  !! before the first substep, lets a device-resident run apply `bottom`'s floor wall function on the GPU
  !! (udc_bottom inside the fused substep).
  subroutine udc_set_floor(on, z0)
    logical, intent(in) :: on
    real(c_double), intent(in) :: z0
    if (c_associated(udc_h)) then
      write (0, *) 'ERROR: udc_set_floor must be called before the first advection/subgrid call'
      stop 1
    end if
    udc_floor_on = on
    udc_floor_z0 = z0
  end subroutine udc_set_floor

  !> Create the device mirror once all of initglobal/initfields/initsubgrid/initpois have run.
  subroutine udc_ensure
    use modglobal, only: itot, jtot, ktot, dx, dy, dzf, dzh, kb, ke, kh, numol, prandtlmoli, nsv, &
                         BCtopm, Uinf, Vinf, lles, ltempeq, lbuoyancy, lmoist, iadv_thl, BCtopT, BCbotT, grav, e12min, &
                         iadv_qt, BCtopq, BCbotq, zf, zh, BCbotm, prandtlturb, fkar, BCtops, lchem, k1, JNO2, &
                         lcoriol, lprofforc, om22, om23, luvolflowr, lvvolflowr, uflowrate, vflowrate, &
                         lnudge, igrw_damp, ifixuinf, ds, BCzp, BCxm
    use modsurfdata, only: wttop, thl_top, wtsurf, thvs, wqtop, qt_top, wqsurf, thls, qts, ps, z0h, wsvtop, sv_top
    use modsubgriddata, only: lsmagorinsky, lvreman, loneeqn, ldelta, prandtli, c_vreman, csz, cm, cn, ch1, ch2, ce1, ce2, lbuoycorr, Rigc
    use modfields, only: dpdxl, dpdyl, thlpcar, ug, whls, dthldxls, dthldyls, dqtdxls, dqtdyls, dqtdtls, &
                         dudxls, dudyls, dvdxls, dvdyls
    use modmpi, only: myid, nprocs, comm3d, mpierr
    use decomp_2d, only: zsize, zstart
    use mpi, only: MPI_CHARACTER
    type(udc_config) :: cfg
    integer(c_signed_char) :: nccl_id(128)
    integer :: gpus_per_node, n
    real(c_double), allocatable, target, save :: zf_(:), zh_(:)
    real(c_double), allocatable :: xo_u(:), xo_v(:)
    character(16) :: env
    integer :: stat
#ifdef UDC_TEST_TRANSPORT
    character(64) :: shm_name
    logical :: shm_on = .false.
#endif
    if (c_associated(udc_h)) return
    allocate (zf_(0:ktot + 1), zh_(0:ktot + 1))
    zf_(0:ktot + 1) = dzf(kb - kh:ke + kh)
    zh_(0) = 0.
    zh_(1:ktot + 1) = dzh(kb:ke + kh)
    cfg%itot = itot; cfg%jtot = jtot; cfg%ktot = ktot
    ! the library's layout is y-slabs: x whole on a rank, every rank one slab of jtot / #ranks rows in rank order.  That is what the
    ! decomp_2d of this directory hands out whatever nprocx x nprocy the deck names; with another decomposition library an x-split
    ! deck is refused here (nprocx = 1, nprocy = number of GPUs is then the deck to write)
    if (zsize(1) /= itot .or. zsize(2)*nprocs /= jtot .or. zstart(2) /= myid*zsize(2) + 1) then
      write (0, *) 'ERROR: libudcore decomposes in y only (x whole on a rank, slabs in rank order): link u-dales_amd/fortran/decomp_2d.f90, ', &
                   'or set nprocx = 1, nprocy = number of GPUs.  This rank holds i, j from', zstart(1:2), ' extents', zsize(1:2)
      stop 1
    end if
    ! device: the library deals the ranks round over the node's visible devices (cfg%device = -1) unless UDC_GPUS_PER_NODE says
    ! how many of them to use (1 = every rank on device 0: the one-GPU tests)
    gpus_per_node = 0
    call get_environment_variable('UDC_GPUS_PER_NODE', env, status=stat)
    if (stat == 0) read (env, *, iostat=stat) gpus_per_node
    cfg%nranks = nprocs; cfg%rank = myid
    cfg%device = -1
    if (gpus_per_node > 0) cfg%device = mod(myid, gpus_per_node)
    cfg%dx = dx; cfg%dy = dy
    cfg%dzf = c_loc(zf_); cfg%dzh = c_loc(zh_)
    cfg%numol = numol; cfg%prandtlmoli = prandtlmoli; cfg%prandtli = prandtli
    cfg%c_vreman = c_vreman; cfg%csz = csz(1, kb)
    cfg%sgs = 0
    if (lles) then
      if (lsmagorinsky) then
        cfg%sgs = 1
      else if (lvreman) then
        cfg%sgs = 2
      else if (loneeqn) then
        cfg%sgs = 3          ! one-equation closure: constants follow with udc_set_tke below
      end if
    end if
    cfg%bctopm = BCtopm
    cfg%uinf = Uinf; cfg%vinf = Vinf
    cfg%nsv = nsv
    cfg%lbottom = merge(1, 0, udc_floor_on)
    cfg%z0 = udc_floor_z0
    if (BCxm == 2 .or. BCxm == 3) then      ! inflow (prof.inp's profile, or a precursor run's planes: udc_set_open_x_inlet), convective
      ! outflow: rows with the reference's ghost columns (udc_xopen.hip)
      call open_x_profiles(xo_u, xo_v)
      call udc_check(udc_create_open_x(cfg, xo_u, xo_v, udc_h), 'udc_create_open_x')
    else
      call udc_check(udc_create(cfg, udc_h), 'udc_create')
    end if
    if (BCzp /= 1) call udc_check(udc_set_poisson_bczp(udc_h, int(BCzp, c_int)), 'udc_set_poisson_bczp')      ! cosine transform in z
    if (nprocs > 1) then
      ! RCCL communicator over the y-slab ranks: rank 0 makes the id, MPI carries it (INTEGRATION.md section 4)
#ifdef UDC_TEST_TRANSPORT
      ! test build (udales_full_dropin_mpi_test, linked against libudcore_test.so): with UDC_TEST_SHM=/name the ranks exchange
      ! through shared memory instead of RCCL, so that mpiexec -n P runs on a one-GPU box
      call get_environment_variable('UDC_TEST_SHM', shm_name, status=stat)
      if (stat == 0 .and. len_trim(shm_name) > 0) then
        call udc_check(udc_comm_init_shm(udc_h, trim(shm_name)//c_null_char), 'udc_comm_init_shm')
        shm_on = .true.
      end if
      if (.not. shm_on) then
#endif
      nccl_id = 0
      if (myid == 0) call udc_check(udc_comm_unique_id(nccl_id), 'udc_comm_unique_id')
      call MPI_BCAST(nccl_id, 128, MPI_CHARACTER, 0, comm3d, mpierr)
      call udc_check(udc_comm_init(udc_h, nccl_id), 'udc_comm_init')
#ifdef UDC_TEST_TRANSPORT
      end if
#endif
    end if
    if (ltempeq) then      ! temperature equation; the dry buoyancy term is on the device for device-resident runs
      ! (in residency 0/1 the host's own forces adds it to the pulled tendencies)
      call udc_check(udc_set_tempeq(udc_h, int(iadv_thl, c_int), int(BCtopT, c_int), real(wttop, c_double), &
                                    real(thl_top, c_double), int(BCbotT, c_int), real(wtsurf, c_double)), 'udc_set_tempeq')
    end if
    if (lmoist) then       ! total water; with buoyancy the moist thermodynamics (thermo, diagfld, calthv) too
      call udc_check(udc_set_moisture(udc_h, int(iadv_qt, c_int), int(BCtopq, c_int), real(wqtop, c_double), &
                                      real(qt_top, c_double), int(BCbotq, c_int), real(wqsurf, c_double)), 'udc_set_moisture')
      if (ltempeq .and. (lbuoyancy .or. loneeqn_dev())) then      ! moist buoyancy / calthv's moist dthvdz
        call udc_check(udc_set_moist_thermo(udc_h, real(thls, c_double), real(qts, c_double), real(ps, c_double), &
                                            zf(kb:ke + kh), zh(kb:ke + kh), int(ktot + 1, c_int), &
                                            merge(1_c_int, 0_c_int, udc_lqlnr)), 'udc_set_moist_thermo')
      end if
    end if
    if (ltempeq .and. lbuoyancy) call udc_check(udc_set_buoyancy(udc_h, 1_c_int, real(grav, c_double)), 'udc_set_buoyancy')
    if (cfg%sgs == 2 .and. lbuoyancy .and. lbuoycorr) then      ! Vreman buoyancy correction (src/modsubgrid.f90:332)
      if (.not. ltempeq) then
        write (0, *) 'ERROR: libudcore: lbuoycorr needs the temperature equation (ltempeq)'
        stop 1
      end if
      call udc_check(udc_set_buoycorr(udc_h, 1_c_int, real(Rigc, c_double)), 'udc_set_buoycorr')
    end if
    call udc_check(udc_set_fkar(udc_h, real(fkar, c_double)), 'udc_set_fkar')      ! &WALLS fkar: every wall function's von Karman constant
    if (udc_floor_on .and. (BCbotm == 2 .or. (ltempeq .and. BCbotT == 2))) then   ! wfuno floor (src/modibm.f90:2021-2045)
      call udc_check(udc_set_floor_wf(udc_h, int(BCbotm, c_int), int(BCbotT, c_int), real(thls, c_double), real(z0h, c_double), &
                                      real(prandtlturb, c_double)), 'udc_set_floor_wf')
    end if
    if (lchem) call udc_check(udc_set_chem(udc_h, 1_c_int, real(k1, c_double), real(JNO2, c_double)), 'udc_set_chem')
    if (cfg%sgs == 3) then   ! after udc_set_tempeq: the closure reads thl0 when the temperature equation is on
      call udc_check(udc_set_tke(udc_h, real(cm, c_double), real(cn, c_double), real(ch1, c_double), real(ch2, c_double), &
                                 real(ce1, c_double), real(ce2, c_double), real(e12min, c_double), real(grav, c_double), &
                                 real(thvs, c_double), merge(1_c_int, 0_c_int, ldelta)), 'udc_set_tke')
    end if
    call get_environment_variable('UDC_RESIDENCY', env, status=stat)
    if (stat == 0) read (env, *, iostat=stat) udc_residency
    call get_environment_variable('UDC_PULL_EVERY', env, status=stat)
    if (stat == 0) read (env, *, iostat=stat) udc_pull_every
    call get_environment_variable('UDC_TIMERS', env, status=stat)
    if (stat == 0) udc_timers_on = (trim(env) /= '0')
    call udc_late_setup
    call udc_push_state
  end subroutine udc_ensure

  !> uprof, vprof as the library takes them: [ktot+2] indexed by the reference's k (entry 0 unused; uprof(ke+1) as allocated: zero)
  subroutine open_x_profiles(u, v)
    use modglobal, only: ktot, kb, ke, kh
    use modfields, only: uprof, vprof
    real(c_double), allocatable, intent(out) :: u(:), v(:)
    allocate (u(0:ktot + 1), v(0:ktot + 1))
    u = 0.; v = 0.
    if (allocated(uprof)) u(1:ktot + 1) = uprof(kb:ke + kh)
    if (allocated(vprof)) v(1:ktot + 1) = vprof(kb:ke + kh)
  end subroutine open_x_profiles

  !> What depends on the input files readinitfiles reads (lscale.inp, scalar.inp: dpdxl, dpdyl, thlpcar, ug, whls, the
  !! large-scale gradients, sv_top).  The handle may exist before they are in (readinitfiles itself calls halos /
  !! boundary / thermodynamics), so this runs with every start-up call and a last time when the time loop starts.
  subroutine udc_late_setup
    use modglobal, only: ktot, kb, ke, nsv, ltempeq, BCtops, lcoriol, lprofforc, om22, om23, luvolflowr, lvvolflowr, luoutflowr, &
                         uflowrate, vflowrate, lnudge, igrw_damp, ifixuinf, ds, BCxs, BCxm, BCxT, BCxq, lmoist, kh
    use modsurfdata, only: wsvtop, sv_top
    use modfields, only: thlprof, qtprof, dpdxl, dpdyl, thlpcar, ug, whls, dthldxls, dthldyls, dqtdxls, dqtdyls, dqtdtls, &
                         dudxls, dudyls, dvdxls, dvdyls
    integer :: n
    real(c_double), allocatable :: xo_u(:), xo_v(:)
    call udc_check(udc_set_forcing(udc_h, dpdxl(kb:ke), dpdyl(kb:ke), int(ktot, c_int)), 'udc_set_forcing')
    if (BCxm == 2 .or. BCxm == 3) then      ! (the handle may be older than prof.inp's profiles)
      call open_x_profiles(xo_u, xo_v)
      call udc_check(udc_set_open_x_profile(udc_h, xo_u, xo_v), 'udc_set_open_x_profile')
      if (ltempeq) then      ! BCxT = 2: the temperature enters with its profile (xTi_profile); BCxT = 1 (the reference's default, its
        ! tests/cases/525): it stays periodic in x (halos' xT_periodic) -- the library's own refresh of a handle without an inflow profile;
        ! planes of a precursor run (BCxT = 3) are not taken yet
        if (BCxT /= 1 .and. BCxT /= 2 .and. .not. (BCxT == 3 .and. BCxm == 3)) then
          write (0, *) 'ERROR: libudcore: inflow / outflow in x with the temperature equation needs BCxT = 1 (periodic), 2 (inflow profile) or, with BCxm = 3, 3 (driver planes)'
          stop 1
        end if
        if (BCxT == 2) then
          xo_u = 0.
          if (allocated(thlprof)) xo_u(1:ktot + 1) = thlprof(kb:ke + kh)
          call udc_check(udc_set_open_x_thl(udc_h, xo_u), 'udc_set_open_x_thl')
        end if
      end if
      if (lmoist) then      ! BCxq = 1 periodic; 2 mirrored about the profile (xqi_profile); 3 planes of a precursor run (xqi_driver: driver_inlet)
        if (BCxq /= 1 .and. BCxq /= 2 .and. .not. (BCxq == 3 .and. BCxm == 3)) then
          write (0, *) 'ERROR: libudcore: inflow / outflow in x with moisture needs BCxq = 1 (periodic), 2 (inflow profile) or, with BCxm = 3, 3 (driver planes)'
          stop 1
        end if
        if (BCxq == 2) then
          xo_u = 0.
          if (allocated(qtprof)) xo_u(1:ktot + 1) = qtprof(kb:ke + kh)
          call udc_check(udc_set_open_x_qt(udc_h, xo_u), 'udc_set_open_x_qt')
        end if
      end if
    end if
    if (ltempeq) then
      call udc_check(udc_set_thl_source(udc_h, thlpcar(kb:ke), int(ktot, c_int)), 'udc_set_thl_source')
    end if
    do n = 1, nsv          ! top condition of the scalars (src/modboundary.f90:236-247)
      call udc_check(udc_set_scalar_top(udc_h, int(n - 1, c_int), int(BCtops, c_int), &
                                        real(merge(sv_top(n), wsvtop(n), BCtops == 2), c_double)), 'udc_set_scalar_top')
    end do
    ! coriolis (src/modforces.f90:600-717): lcoriol -> mode 1, lprofforc -> mode 2 (relaxation to ug)
    if (lcoriol) then
      call udc_check(udc_set_coriolis(udc_h, 1_c_int, real(om22, c_double), real(om23, c_double), ug(kb:ke), int(ktot, c_int)), &
                     'udc_set_coriolis')
    else if (lprofforc) then
      call udc_check(udc_set_coriolis(udc_h, 2_c_int, real(om22, c_double), real(om23, c_double), ug(kb:ke), int(ktot, c_int)), &
                     'udc_set_coriolis')
    end if
    ! scalars with an inflow / outflow in x (BCxs = 2: xsi_profile, xso_convective; src/modboundary.f90:844, 983)
    if (nsv > 0 .and. (BCxm == 2 .or. BCxm == 3)) then      ! the scalars enter and leave with the flow: the rows carry their ghost columns
      call open_x_scalars
    else if (nsv > 0 .and. BCxs /= 1) then
      call scalar_bcx_setup
    end if
    ! masscorr, volume-flow branches (src/modforces.f90:389-417, 467-494)
    call udc_check(udc_set_masscorr(udc_h, merge(1_c_int, 0_c_int, luvolflowr), real(uflowrate, c_double), &
                                    merge(1_c_int, 0_c_int, lvvolflowr), real(vflowrate, c_double)), 'udc_set_masscorr')
    ! ... and the u outflow-rate branch (:352-387; it wins over luvolflowr)
    if (luoutflowr) call udc_check(udc_set_masscorr_outflow(udc_h, 1_c_int, real(uflowrate, c_double)), 'udc_set_masscorr_outflow')
    ! does any host routine of the loop read diagfld's slab averages?
    udc_need_avg = lnudge .or. igrw_damp /= 0 .or. ifixuinf /= 0 .or. ds > 0 .or. any(whls /= 0.) .or. &
                   any(dthldxls /= 0.) .or. any(dthldyls /= 0.) .or. any(dqtdxls /= 0.) .or. any(dqtdyls /= 0.) .or. &
                   any(dqtdtls /= 0.) .or. any(dudxls /= 0.) .or. any(dudyls /= 0.) .or. any(dvdxls /= 0.) .or. any(dvdyls /= 0.)
  end subroutine udc_late_setup

  subroutine open_x_scalars
    use modglobal, only: ktot, kb, ke, nsv, BCxs, BCxm
    use modfields, only: svprof
    real(c_double), allocatable :: t(:, :)
    if (BCxs /= 2 .and. .not. (BCxs == 3 .and. BCxm == 3)) then      ! (3: the inlet from a precursor's planes, xsi_driver: driver_inlet)
      write (0, *) 'ERROR: libudcore: inflow / outflow in x with passive scalars needs BCxs = 2 (inflow profile) or, with BCxm = 3, 3 (driver planes)'
      stop 1
    end if
    allocate (t(0:ktot + 1, nsv))
    t = 0.
    t(1:ktot + 1, :) = svprof(kb:ke + 1, 1:nsv)
    call udc_check(udc_set_open_x_scalars(udc_h, t), 'udc_set_open_x_scalars')
  end subroutine open_x_scalars

  subroutine scalar_bcx_setup
    use modglobal, only: ktot, kb, ke, nsv, BCxs, luvolflowr, luoutflowr, dzf, zh
    use modfields, only: svprof
    use modinletdata, only: ubulk
    real(c_double), allocatable :: t(:, :)
    if (BCxs /= 2 .or. luoutflowr) then
      write (0, *) 'ERROR: libudcore: BCxs must be 1 (periodic) or 2 (inflow profile, convective outflow; not with luoutflowr)'
      stop 1
    end if
    allocate (t(0:ktot + 1, nsv))
    t = 0.
    t(1:ktot + 1, :) = svprof(kb:ke + 1, 1:nsv)
    call udc_check(udc_set_scalar_bcx(udc_h, 2_c_int, t, real(ubulk, c_double)), 'udc_set_scalar_bcx')
    if (.not. luvolflowr) &      ! the outlet's speed from diagfld's slab averages (src/modboundary.f90:143-156)
      call udc_check(udc_set_scalar_bcx_outflow(udc_h, real(dzf(kb:ke)/(zh(ke + 1) - zh(kb + 1)), c_double)), 'udc_set_scalar_bcx_outflow')
  end subroutine scalar_bcx_setup

  !> Effective residency: start-up code (before the first tstep_update) works on the host arrays, so every call
  !! there carries the state both ways.
  integer function udc_mode()
    udc_mode = merge(udc_residency, 0, udc_in_loop)
  end function udc_mode

  !> What a drop-in routine does before its device call ...
  subroutine udc_tic(id)
    integer, intent(in) :: id
    if (udc_timers_on) call system_clock(udc_t_start(id))
  end subroutine udc_tic
  subroutine udc_toc(id)
    integer, intent(in) :: id
    integer(8) :: c, rate
    if (.not. udc_timers_on) return
    call system_clock(c, rate)
    udc_t_sum(id) = udc_t_sum(id) + real(c - udc_t_start(id), 8)/real(rate, 8)
    udc_t_cnt(id) = udc_t_cnt(id) + 1
  end subroutine udc_toc
  !> one line per phase: "UDC_TIMER <name> calls=<n> seconds=<s>"
  subroutine udc_timers_report
    use modmpi, only: myid
    integer :: q
    if (.not. udc_timers_on .or. myid /= 0) return
    do q = 1, UDC_T_N
      write (*, '(A,A10,A,I9,A,F12.6)') 'UDC_TIMER ', udc_t_name(q), ' calls=', udc_t_cnt(q), ' seconds=', udc_t_sum(q)
    end do
  end subroutine udc_timers_report

  subroutine udc_begin(tend)
    logical, intent(in) :: tend      !< the routine reads or edits the tendencies
    call udc_ensure
    if (.not. udc_in_loop) then
      call udc_late_setup
      call udc_push_ek      ! start-up: the cold start's ekm = ekh = numol (src/modstartup.f90:1164-1165, 1190) is what the
                            ! flux tops of `boundary` divide by before the first closure has run (src/modboundary.f90:1532)
    end if
    select case (udc_mode())
    case (0)
      call udc_push_state
      if (tend) call udc_push_tend
    case (1)
      if (tend) call udc_push_tend
    end select
  end subroutine udc_begin

  !> ... and after it (routines that only edit tendencies)
  subroutine udc_end_tend
    if (udc_mode() <= 1) call udc_pull_tend
  end subroutine udc_end_tend

  !> First tstep_update: the time loop starts.  In device mode the state goes up once and the routines start recording.
  subroutine udc_enter_loop
    use modglobal, only: ltrees, lpurif, lheatpump
    use modmpi, only: myid
    if (udc_in_loop) return
    call udc_ensure
    call udc_late_setup
    ! vegetation_forcing, purifiers and heatpump are untouched host routines that edit the host's tendencies from the host's fields
    ! inside the loop: only the strict mode carries the state to them and their terms back
    if (udc_residency /= 0 .and. (ltrees .or. lpurif .or. lheatpump)) then
      if (myid == 0) write (6, *) 'libudcore: trees / purifiers / heat pump run on the host inside the time loop: UDC_RESIDENCY=0'
      udc_residency = 0
    end if
    udc_in_loop = .true.
    if (udc_residency == 2) then
      call udc_push_state
      call udc_push_tend
      call udc_push_ek
      call udc_check(udc_set_deferred(udc_h, 1_c_int), 'udc_set_deferred')
      udc_host_fresh = .false.
    end if
  end subroutine udc_enter_loop

  subroutine udc_tab_start(when)
    use modglobal, only: kb, ke, nsv
    integer, intent(in) :: when
    if (.not. allocated(udc_tabA)) then
      allocate (udc_tabA(kb:ke, ROW_SVP + max(nsv, 1), 0:1), udc_tabB(kb:ke, ROW_SVP + max(nsv, 1), 0:1))
    end if
    if (.not. udc_tab_open(when)) then
      udc_tabA(:, :, when) = 0.; udc_tabB(:, :, when) = 0.
    end if
    udc_tab_open(when) = .true.
  end subroutine udc_tab_start

  !> Register the tables built since udc_tab_start(when) and let the device add them to the tendencies.  A substep that
  !! builds none still has to overwrite what an earlier one registered (fixuinf1 acts on RK stage 3 only).
  subroutine udc_tab_apply(when)
    use modglobal, only: kb, ke, nsv, ltempeq, lmoist
    integer, intent(in) :: when
    integer :: n, nlev
    if (.not. udc_tab_open(when)) then
      if (.not. udc_tab_registered(when)) return
      call udc_tab_start(when)          ! all-zero tables
    end if
    udc_tab_open(when) = .false.
    udc_tab_registered(when) = .true.
    nlev = ke - kb + 1
    call udc_begin(.true.)
    call reg(UDC_UP, UDC_U0, ROW_UP)
    call reg(UDC_VP, UDC_V0, ROW_VP)
    call reg(UDC_WP, UDC_W0, ROW_WP)
    if (ltempeq) call reg(UDC_THLP, UDC_THL0, ROW_THLP)
    if (lmoist) call reg(UDC_QTP, UDC_QT0, ROW_QTP)
    do n = 1, nsv
      call reg(UDC_SVP + 3*(n - 1), UDC_SV0 + 3*(n - 1), ROW_SVP + n)
    end do
    call udc_check(udc_level_forcings(udc_h, int(when, c_int)), 'udc_level_forcings')
    call udc_end_tend
  contains
    subroutine reg(tend, src, row)
      integer(c_int), intent(in) :: tend, src
      integer, intent(in) :: row
      integer(c_int) :: s
      ! rows that are and always were zero cost nothing; a row once registered is overwritten from then on
      if (.not. (udc_row_registered(row, when) .or. any(udc_tabA(:, row, when) /= 0.) .or. any(udc_tabB(:, row, when) /= 0.))) return
      udc_row_registered(row, when) = .true.
      s = -1
      if (any(udc_tabB(:, row, when) /= 0.)) s = src
      call udc_check(udc_set_level_forcing(udc_h, tend, s, udc_tabA(:, row, when), udc_tabB(:, row, when), &
                                           int(nlev, c_int), int(when, c_int)), 'udc_set_level_forcing')
    end subroutine reg
  end subroutine udc_tab_apply

  !> diagfld's slab averages (src/modthermodynamics.f90:262-290) from the device: u0av, v0av, thl0av, qt0av, sv0av.
  subroutine udc_refresh_averages
    use modglobal, only: kb, ke, kh, nsv, ltempeq, lmoist
    use modfields, only: u0av, v0av, thl0av, qt0av, sv0av
    integer(c_int) :: ids(16)
    integer :: nf, n, n0, nb, q, nlev
    real(c_double), allocatable :: avg(:, :)
    nlev = ke + kh - kb + 1
    nf = 2; ids(1) = UDC_U0; ids(2) = UDC_V0
    if (ltempeq) then
      nf = nf + 1; ids(nf) = UDC_THL0
    end if
    if (lmoist) then
      nf = nf + 1; ids(nf) = UDC_QT0
    end if
    allocate (avg(nlev, 16))
    call udc_check(udc_slab_averages(udc_h, ids, int(nf, c_int), avg, int(nlev, c_int)), 'udc_slab_averages')
    u0av(kb:ke + kh) = avg(:, 1); v0av(kb:ke + kh) = avg(:, 2)
    q = 2
    if (ltempeq) then
      q = q + 1; thl0av(kb:ke + kh) = avg(:, q)
    end if
    if (lmoist) then
      q = q + 1; qt0av(kb:ke + kh) = avg(:, q)
    end if
    n0 = 1
    do while (n0 <= nsv)                ! the scalars in batches of at most 16 fields per reduction
      nb = min(nsv - n0 + 1, 16)
      do n = 1, nb
        ids(n) = UDC_SV0 + 3*(n0 + n - 2)
      end do
      call udc_check(udc_slab_averages(udc_h, ids, int(nb, c_int), avg, int(nlev, c_int)), 'udc_slab_averages')
      do n = 1, nb
        sv0av(kb:ke + kh, n0 + n - 1) = avg(:, n)
      end do
      n0 = n0 + nb
    end do
    deallocate (avg)
  end subroutine udc_refresh_averages

  subroutine udc_push3(field, a, lb)
    integer(c_int), intent(in) :: field
    real(c_double), intent(in) :: a(:, :, :)
    integer, intent(in) :: lb(3)
    integer(c_int) :: l(3), u(3)
    l = lb; u = lb + shape(a) - 1
    call udc_check(udc_field_upload(udc_h, field, a, l, u), 'upload')
  end subroutine udc_push3

  subroutine udc_pull3(field, a, lb)
    integer(c_int), intent(in) :: field
    real(c_double), intent(inout) :: a(:, :, :)
    integer, intent(in) :: lb(3)
    integer(c_int) :: l(3), u(3)
    l = lb; u = lb + shape(a) - 1
    call udc_check(udc_field_download(udc_h, field, a, l, u), 'download')
  end subroutine udc_pull3

  !> Everything the device needs from the host's prognostic state (bounds: src/modfields.f90:440-474)
  subroutine udc_push_state
    use modglobal, only: ib, jb, kb, ih, jh, kh, ihc, jhc, khc, nsv, ltempeq, lmoist, iadv_thl, iadv_kappa
    use modfields, only: u0, v0, w0, um, vm, wm, pres0, sv0, svm, thl0, thlm, e120, e12m, qt0, qtm, thl0c
    integer :: n
    call udc_tic(UDC_T_UPLOAD)
    call udc_push3(UDC_U0, u0, (/ib - ih, jb - jh, kb - kh/))
    call udc_push3(UDC_V0, v0, (/ib - ih, jb - jh, kb - kh/))
    call udc_push3(UDC_W0, w0, (/ib - ih, jb - jh, kb - kh/))
    call udc_push3(UDC_UM, um, (/ib - ih, jb - jh, kb - kh/))
    call udc_push3(UDC_VM, vm, (/ib - ih, jb - jh, kb - kh/))
    call udc_push3(UDC_WM, wm, (/ib - ih, jb - jh, kb - kh/))
    call udc_push3(UDC_PRES0, pres0, (/ib - ih, jb - jh, kb - kh/))
    if (loneeqn_dev()) then
      call udc_push3(UDC_E120, e120, (/ib - ih, jb - jh, kb - kh/))
      call udc_push3(UDC_E12M, e12m, (/ib - ih, jb - jh, kb - kh/))
    end if
    if (ltempeq) then
      ! kappa on thl reads two lateral ghost cells: those come from the reference's wide copy thl0c (its vertical ghosts
      ! are re-created on the device, udc_set_tempeq), the inner ring and the vertical ghost planes from thl0 itself
      if (iadv_thl == iadv_kappa) call udc_push3(UDC_THL0, thl0c, (/ib - ihc, jb - jhc, kb - khc/))
      call udc_push3(UDC_THL0, thl0, (/ib - ih, jb - jh, kb - kh/))
      call udc_push3(UDC_THLM, thlm, (/ib - ih, jb - jh, kb - kh/))
    end if
    if (lmoist) then
      call udc_push3(UDC_QT0, qt0, (/ib - ih, jb - jh, kb - kh/))
      call udc_push3(UDC_QTM, qtm, (/ib - ih, jb - jh, kb - kh/))
    end if
    do n = 1, nsv
      call udc_push3(UDC_SV0 + 3*(n - 1), sv0(:, :, :, n), (/ib - ihc, jb - jhc, kb - khc/))
      call udc_push3(UDC_SVM + 3*(n - 1), svm(:, :, :, n), (/ib - ihc, jb - jhc, kb - khc/))
    end do
    call udc_toc(UDC_T_UPLOAD)
  end subroutine udc_push_state

  !> the eddy diffusivities as the host has them (start-up values, or a restart file's ekm)
  subroutine udc_push_ek
    use modglobal, only: ib, jb, kb, ih, jh, kh
    use modsubgriddata, only: ekm, ekh
    call udc_tic(UDC_T_UPLOAD)
    call udc_push3(UDC_EKM, ekm, (/ib - ih, jb - jh, kb - kh/))
    call udc_push3(UDC_EKH, ekh, (/ib - ih, jb - jh, kb - kh/))
    call udc_toc(UDC_T_UPLOAD)
  end subroutine udc_push_ek

  subroutine udc_push_tend
    use modglobal, only: ib, jb, kb, ih, jh, ihc, jhc, nsv, ltempeq, lmoist
    use modfields, only: up, vp, wp, svp, thlp, e12p, qtp
    integer :: n
    call udc_tic(UDC_T_UPLOAD)
    call udc_push3(UDC_UP, up, (/ib - ih, jb - jh, kb/))
    call udc_push3(UDC_VP, vp, (/ib - ih, jb - jh, kb/))
    call udc_push3(UDC_WP, wp, (/ib - ih, jb - jh, kb/))
    if (ltempeq) call udc_push3(UDC_THLP, thlp, (/ib - ih, jb - jh, kb/))
    if (lmoist) call udc_push3(UDC_QTP, qtp, (/ib - ih, jb - jh, kb/))
    if (loneeqn_dev()) call udc_push3(UDC_E12P, e12p, (/ib - ih, jb - jh, kb/))
    do n = 1, nsv
      call udc_push3(UDC_SVP + 3*(n - 1), svp(:, :, :, n), (/ib - ihc, jb - jhc, kb/))
    end do
    call udc_toc(UDC_T_UPLOAD)
  end subroutine udc_push_tend

  subroutine udc_pull_tend
    use modglobal, only: ib, jb, kb, ih, jh, ihc, jhc, nsv, ltempeq, lmoist
    use modfields, only: up, vp, wp, svp, thlp, e12p, qtp
    integer :: n
    call udc_pull3(UDC_UP, up, (/ib - ih, jb - jh, kb/))
    call udc_pull3(UDC_VP, vp, (/ib - ih, jb - jh, kb/))
    call udc_pull3(UDC_WP, wp, (/ib - ih, jb - jh, kb/))
    if (ltempeq) call udc_pull3(UDC_THLP, thlp, (/ib - ih, jb - jh, kb/))
    if (lmoist) call udc_pull3(UDC_QTP, qtp, (/ib - ih, jb - jh, kb/))
    if (loneeqn_dev()) call udc_pull3(UDC_E12P, e12p, (/ib - ih, jb - jh, kb/))
    do n = 1, nsv
      call udc_pull3(UDC_SVP + 3*(n - 1), svp(:, :, :, n), (/ib - ihc, jb - jhc, kb/))
    end do
  end subroutine udc_pull_tend

  subroutine udc_pull_vel(with_m)
    use modglobal, only: ib, jb, kb, ih, jh, kh, ihc, jhc, khc, nsv, ltempeq, lmoist, iadv_thl, iadv_kappa
    use modfields, only: u0, v0, w0, um, vm, wm, sv0, svm, thl0, thlm, e120, e12m, qt0, qtm, thl0c
    logical, intent(in) :: with_m
    integer :: n
    call udc_pull3(UDC_U0, u0, (/ib - ih, jb - jh, kb - kh/))
    call udc_pull3(UDC_V0, v0, (/ib - ih, jb - jh, kb - kh/))
    call udc_pull3(UDC_W0, w0, (/ib - ih, jb - jh, kb - kh/))
    if (ltempeq) call udc_pull3(UDC_THL0, thl0, (/ib - ih, jb - jh, kb - kh/))
    if (ltempeq .and. iadv_thl == iadv_kappa) call udc_pull3(UDC_THL0, thl0c, (/ib - ihc, jb - jhc, kb - khc/))   ! the wide copy (src/modboundary.f90:87)
    if (lmoist) call udc_pull3(UDC_QT0, qt0, (/ib - ih, jb - jh, kb - kh/))
    if (loneeqn_dev()) call udc_pull3(UDC_E120, e120, (/ib - ih, jb - jh, kb - kh/))
    do n = 1, nsv
      call udc_pull3(UDC_SV0 + 3*(n - 1), sv0(:, :, :, n), (/ib - ihc, jb - jhc, kb - khc/))
    end do
    if (with_m) then
      call udc_pull3(UDC_UM, um, (/ib - ih, jb - jh, kb - kh/))
      call udc_pull3(UDC_VM, vm, (/ib - ih, jb - jh, kb - kh/))
      call udc_pull3(UDC_WM, wm, (/ib - ih, jb - jh, kb - kh/))
      if (ltempeq) call udc_pull3(UDC_THLM, thlm, (/ib - ih, jb - jh, kb - kh/))
      if (lmoist) call udc_pull3(UDC_QTM, qtm, (/ib - ih, jb - jh, kb - kh/))
      if (loneeqn_dev()) call udc_pull3(UDC_E12M, e12m, (/ib - ih, jb - jh, kb - kh/))
      do n = 1, nsv
        call udc_pull3(UDC_SVM + 3*(n - 1), svm(:, :, :, n), (/ib - ihc, jb - jhc, kb - khc/))
      end do
    end if
  end subroutine udc_pull_vel

  !> one-equation closure active on the device (lles, not overridden by lsmagorinsky / lvreman)
  logical function loneeqn_dev()
    use modglobal, only: lles
    use modsubgriddata, only: lsmagorinsky, lvreman, loneeqn
    loneeqn_dev = lles .and. loneeqn .and. .not. (lsmagorinsky .or. lvreman)
  end function loneeqn_dev

  !> Bring every host array up to date (before output / restart / statistics in device mode).
  subroutine udc_pull_all
    use modglobal, only: ib, jb, kb, ih, jh, kh
    use modfields, only: pres0
    use modsubgriddata, only: ekm, ekh
    if (.not. c_associated(udc_h)) return
    call udc_tic(UDC_T_DOWNLOAD)
    udc_host_fresh = .true.
    call udc_pull_vel(.true.)
    call udc_pull_tend
    call udc_pull3(UDC_PRES0, pres0, (/ib - ih, jb - jh, kb - kh/))
    call udc_pull3(UDC_EKM, ekm, (/ib - ih, jb - jh, kb - kh/))
    call udc_pull3(UDC_EKH, ekh, (/ib - ih, jb - jh, kb - kh/))
    call udc_pull_bottom_diag
    call udc_toc(UDC_T_DOWNLOAD)
  end subroutine udc_pull_all

  !> tau_x, tau_y, tau_z, thl_flux (modfields; the fielddump variables `bottom` leaves behind, src/modibm.f90:2015-2018,
  !! 2094-2097): what the floor added to up, vp, thlp -- nonzero on the k = kb plane only
  subroutine udc_pull_bottom_diag
    use modglobal, only: ib, ie, jb, je, kb, ltempeq
    use modfields, only: tau_x, tau_y, tau_z, thl_flux
    real(c_double), allocatable :: plane(:, :)
    if (.not. udc_bottom_diag_on) return
    if (.not. allocated(tau_x)) return
    allocate (plane(ib:ie, jb:je))
    tau_x = 0.; tau_y = 0.; tau_z = 0.; thl_flux = 0.
    call udc_check(udc_bottom_diag_get(udc_h, 0_c_int, plane), 'udc_bottom_diag_get')
    tau_x(ib:ie, jb:je, kb) = plane
    call udc_check(udc_bottom_diag_get(udc_h, 1_c_int, plane), 'udc_bottom_diag_get')
    tau_y(ib:ie, jb:je, kb) = plane
    if (ltempeq) then
      call udc_check(udc_bottom_diag_get(udc_h, 2_c_int, plane), 'udc_bottom_diag_get')
      thl_flux(ib:ie, jb:je, kb) = plane
    end if
    deallocate (plane)
  end subroutine udc_pull_bottom_diag

end module udc_iface
