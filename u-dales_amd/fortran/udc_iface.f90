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
!!   UDC_RESIDENCY=2 ("device")            nothing moves until udc_pull_all(); for drivers whose
!!                        host-side routines in the loop are no-ops (neutral empty channel).
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
    integer(c_int) function udc_comm_unique_id(id) bind(C, name='udc_comm_unique_id')
      import :: c_int, c_signed_char
      integer(c_signed_char), intent(out) :: id(128)
    end function
    integer(c_int) function udc_comm_init(h, id) bind(C, name='udc_comm_init')
      import :: c_int, c_ptr, c_signed_char
      type(c_ptr), value :: h
      integer(c_signed_char), intent(in) :: id(128)
    end function
    integer(c_int) function udc_sync(h) bind(C, name='udc_sync')
      import :: c_int, c_ptr
      type(c_ptr), value :: h
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
                         iadv_qt, BCtopq, BCbotq, zf, zh, BCbotm, prandtlturb, BCtops, lchem, k1, JNO2
    use modsurfdata, only: wttop, thl_top, wtsurf, thvs, wqtop, qt_top, wqsurf, thls, qts, ps, z0h, wsvtop, sv_top
    use modthermodynamics, only: lqlnr
    use modsubgriddata, only: lsmagorinsky, lvreman, loneeqn, ldelta, prandtli, c_vreman, csz, cm, cn, ch1, ch2, ce1, ce2
    use modfields, only: dpdxl, dpdyl, thlpcar
    use modmpi, only: myid, nprocs, nprocx, comm3d, mpierr
    use mpi, only: MPI_CHARACTER
    type(udc_config) :: cfg
    integer(c_signed_char) :: nccl_id(128)
    integer :: gpus_per_node, n
    real(c_double), allocatable, target, save :: zf_(:), zh_(:)
    character(16) :: env
    integer :: stat
    if (c_associated(udc_h)) return
    allocate (zf_(0:ktot + 1), zh_(0:ktot + 1))
    zf_(0:ktot + 1) = dzf(kb - kh:ke + kh)
    zh_(0) = 0.
    zh_(1:ktot + 1) = dzh(kb:ke + kh)
    cfg%itot = itot; cfg%jtot = jtot; cfg%ktot = ktot
    if (nprocx /= 1) then
      write (0, *) 'ERROR: libudcore decomposes in y only: set nprocx = 1, nprocy = number of GPUs'
      stop 1
    end if
    gpus_per_node = 8
    call get_environment_variable('UDC_GPUS_PER_NODE', env, status=stat)
    if (stat == 0) read (env, *, iostat=stat) gpus_per_node
    cfg%nranks = nprocs; cfg%rank = myid; cfg%device = mod(myid, max(gpus_per_node, 1))
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
    call udc_check(udc_create(cfg, udc_h), 'udc_create')
    if (nprocs > 1) then
      ! RCCL communicator over the y-slab ranks: rank 0 makes the id, MPI carries it (INTEGRATION.md section 4)
      nccl_id = 0
      if (myid == 0) call udc_check(udc_comm_unique_id(nccl_id), 'udc_comm_unique_id')
      call MPI_BCAST(nccl_id, 128, MPI_CHARACTER, 0, comm3d, mpierr)
      call udc_check(udc_comm_init(udc_h, nccl_id), 'udc_comm_init')
    end if
    call udc_check(udc_set_forcing(udc_h, dpdxl(kb:ke), dpdyl(kb:ke), int(ktot, c_int)), 'udc_set_forcing')
    if (ltempeq) then      ! temperature equation; the dry buoyancy term is on the device for device-resident runs
      ! (in residency 0/1 the host's own forces adds it to the pulled tendencies)
      call udc_check(udc_set_tempeq(udc_h, int(iadv_thl, c_int), int(BCtopT, c_int), real(wttop, c_double), &
                                    real(thl_top, c_double), int(BCbotT, c_int), real(wtsurf, c_double)), 'udc_set_tempeq')
      call udc_check(udc_set_thl_source(udc_h, thlpcar(kb:ke), int(ktot, c_int)), 'udc_set_thl_source')
    end if
    if (lmoist) then       ! total water; with buoyancy the moist thermodynamics (thermo, diagfld, calthv) too
      call udc_check(udc_set_moisture(udc_h, int(iadv_qt, c_int), int(BCtopq, c_int), real(wqtop, c_double), &
                                      real(qt_top, c_double), int(BCbotq, c_int), real(wqsurf, c_double)), 'udc_set_moisture')
      if (ltempeq .and. (lbuoyancy .or. loneeqn_dev())) then      ! moist buoyancy / calthv's moist dthvdz
        call udc_check(udc_set_moist_thermo(udc_h, real(thls, c_double), real(qts, c_double), real(ps, c_double), &
                                            zf(kb:ke + kh), zh(kb:ke + kh), int(ktot + 1, c_int), &
                                            merge(1_c_int, 0_c_int, lqlnr)), 'udc_set_moist_thermo')
      end if
    end if
    if (ltempeq .and. lbuoyancy) call udc_check(udc_set_buoyancy(udc_h, 1_c_int, real(grav, c_double)), 'udc_set_buoyancy')
    if (udc_floor_on .and. (BCbotm == 2 .or. (ltempeq .and. BCbotT == 2))) then   ! wfuno floor (src/modibm.f90:2021-2045)
      call udc_check(udc_set_floor_wf(udc_h, int(BCbotm, c_int), int(BCbotT, c_int), real(thls, c_double), real(z0h, c_double), &
                                      real(prandtlturb, c_double)), 'udc_set_floor_wf')
    end if
    if (lchem) call udc_check(udc_set_chem(udc_h, 1_c_int, real(k1, c_double), real(JNO2, c_double)), 'udc_set_chem')
    do n = 1, nsv          ! top condition of the scalars (src/modboundary.f90:236-247)
      call udc_check(udc_set_scalar_top(udc_h, int(n - 1, c_int), int(BCtops, c_int), &
                                        real(merge(sv_top(n), wsvtop(n), BCtops == 2), c_double)), 'udc_set_scalar_top')
    end do
    if (cfg%sgs == 3) then   ! after udc_set_tempeq: the closure reads thl0 when the temperature equation is on
      call udc_check(udc_set_tke(udc_h, real(cm, c_double), real(cn, c_double), real(ch1, c_double), real(ch2, c_double), &
                                 real(ce1, c_double), real(ce2, c_double), real(e12min, c_double), real(grav, c_double), &
                                 real(thvs, c_double), merge(1_c_int, 0_c_int, ldelta)), 'udc_set_tke')
    end if
    call get_environment_variable('UDC_RESIDENCY', env, status=stat)
    if (stat == 0) read (env, *, iostat=stat) udc_residency
    call udc_push_state
  end subroutine udc_ensure

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
  end subroutine udc_push_state

  subroutine udc_push_tend
    use modglobal, only: ib, jb, kb, ih, jh, ihc, jhc, nsv, ltempeq, lmoist
    use modfields, only: up, vp, wp, svp, thlp, e12p, qtp
    integer :: n
    call udc_push3(UDC_UP, up, (/ib - ih, jb - jh, kb/))
    call udc_push3(UDC_VP, vp, (/ib - ih, jb - jh, kb/))
    call udc_push3(UDC_WP, wp, (/ib - ih, jb - jh, kb/))
    if (ltempeq) call udc_push3(UDC_THLP, thlp, (/ib - ih, jb - jh, kb/))
    if (lmoist) call udc_push3(UDC_QTP, qtp, (/ib - ih, jb - jh, kb/))
    if (loneeqn_dev()) call udc_push3(UDC_E12P, e12p, (/ib - ih, jb - jh, kb/))
    do n = 1, nsv
      call udc_push3(UDC_SVP + 3*(n - 1), svp(:, :, :, n), (/ib - ihc, jb - jhc, kb/))
    end do
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
    use modglobal, only: ib, jb, kb, ih, jh, kh, ihc, jhc, khc, nsv, ltempeq, lmoist
    use modfields, only: u0, v0, w0, um, vm, wm, sv0, svm, thl0, thlm, e120, e12m, qt0, qtm
    logical, intent(in) :: with_m
    integer :: n
    call udc_pull3(UDC_U0, u0, (/ib - ih, jb - jh, kb - kh/))
    call udc_pull3(UDC_V0, v0, (/ib - ih, jb - jh, kb - kh/))
    call udc_pull3(UDC_W0, w0, (/ib - ih, jb - jh, kb - kh/))
    if (ltempeq) call udc_pull3(UDC_THL0, thl0, (/ib - ih, jb - jh, kb - kh/))
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
    call udc_pull_vel(.true.)
    call udc_pull_tend
    call udc_pull3(UDC_PRES0, pres0, (/ib - ih, jb - jh, kb - kh/))
    call udc_pull3(UDC_EKM, ekm, (/ib - ih, jb - jh, kb - kh/))
    call udc_pull3(UDC_EKH, ekh, (/ib - ih, jb - jh, kb - kh/))
  end subroutine udc_pull_all

end module udc_iface
