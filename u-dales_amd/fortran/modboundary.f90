!> Drop-in replacement for the reference's module modboundary (src/modboundary.f90).
!! Same module name and public list (src/modboundary.f90:29-30).  halos (:67) and boundary (:115) run on the device
!! (udc_halos, udc_boundary; after a fused substep, which already did both, they find nothing left to do); grwdamp (:1447)
!! builds its per-level tables here and the device adds them.  The library implements the periodic lateral conditions
!! with a free-slip / no-slip top and flux / value tops for the scalars -- what `boundary` does for BCxm = BCym = 1;
!! inflow-outflow decks are refused by initboundary.
!!
!! bcp, bcpup and closurebc (called only from modpois and modsubgrid, both replaced) and the x*/y*_periodic helpers are
!! part of the device kernels; the names stay public so that `use modboundary, only: ...` keeps compiling, and stop
!! with the reference's error convention if something calls them.
module modboundary
  use iso_c_binding, only: c_int, c_double
  implicit none
  save
  private
  public :: driver_inlet, stage3_ahead
  public :: initboundary, boundary, grwdamp, ksp, tqaver, halos, bcp, bcpup, closurebc, &
            xm_periodic, xT_periodic, xq_periodic, xs_periodic, ym_periodic, yT_periodic, yq_periodic, ys_periodic
  integer :: ksp = -1                 !< lowest level of the sponge layer (&DOMAIN ksp; -1 = default)
  logical :: due_now = .false.        !< stage3_ahead's answer for the `halos` of the same RK stage
  logical :: pulled_ahead = .false.   !< inflow / outflow in x: halos refreshed the host arrays from a stage-3 substep that ended ahead of `boundary`
  real :: stat_clock = 0.             !< mirror of statsdump's sampling clock tsamplep (private to modstatsdump, src/modstatsdump.f90:65)
  real, allocatable :: tsc(:)         !< damping coefficients of grwdamp
  real :: rnu0 = 2.75e-3

contains

  !> sponge layer coefficients (src/modboundary.f90:38-65)
  subroutine initboundary
    use modglobal, only: ib, kb, ke, kh, kmax, pi, zf, iplane, BCxm, BCym
    use modinletdata, only: irecy
    real :: zspb, zspt
    integer :: k
    if (BCxm < 1 .or. BCxm > 3 .or. BCym /= 1) then
      write (0, *) 'ERROR: libudcore boundary: BCxm = 1 (periodic), 2 (inflow profile) or 3 (inflow from driver files), BCym = 1'
      stop 1
    end if
    allocate (tsc(kb:ke + kh))
    if (ksp == -1) ksp = (kb - 1) + max(min(3*kmax/4, kmax - 15), 1)
    zspb = zf(ksp); zspt = zf(ke)
    tsc = 0.
    do k = ksp, ke
      tsc(k) = rnu0*sin(0.5*pi*(zf(k) - zspb)/(zspt - zspb))**2
    end do
    tsc(ke + 1) = tsc(ke)
    irecy = ib + iplane
  end subroutine initboundary

  !> periodic ghost cells of the prognostic fields (src/modboundary.f90:67-109)
  subroutine halos
    use udc_iface, only: udc_tic, udc_toc, UDC_T_HALOS
    call udc_tic(UDC_T_HALOS)
    call halos_timed
    call udc_toc(UDC_T_HALOS)
  end subroutine halos

  subroutine halos_timed
    use modglobal, only: rk3step, timeleft, ntrun, timee, lfielddump, tnextfielddump, BCxm
    use udc_iface
    logical :: due, ahead
    call udc_begin(.false.)
    due = .false.
    ahead = .false.
    if (udc_mode() > 1 .and. rk3step == 3) then
      due = due_now                                                 ! (stage3_ahead, called by tstep_integrate before the substep was launched)
      ahead = BCxm /= 1 .and. (due .or. udc_stats_on_device)      ! this stage's substep has ended ahead of `boundary`
    end if
    call udc_check(udc_halos(udc_h), 'udc_halos')
    if (udc_mode() <= 1) then
      call udc_pull_vel(.true.)
    else if (rk3step == 3) then
      ! device mode: checksim, fielddump and statsdump come next (src/program.f90:199-205) and read the host arrays.  They are
      ! refreshed when the run ends, every UDC_PULL_EVERY steps, and -- so that the untouched statsdump samples the state it
      ! would sample in an all-host run -- on exactly the steps on which it takes a sample
      if (timeleft <= 0 .or. due) then
        call udc_pull_all
        pulled_ahead = ahead
      else if (udc_pull_every > 0) then
        if (mod(ntrun, udc_pull_every) == 0) then
          call udc_pull_all
          pulled_ahead = ahead
        end if
      end if
    end if
  end subroutine halos_timed

  !> Device-resident runs, RK stage 3, before the fused substep is launched (the drop-in tstep_integrate calls this): are the host-side
  !! dumps that follow `halos` due on this step?  With inflow / outflow in x the substep then ends ahead of its `boundary`, which overwrites
  !! interior cells there (xTi_profile, src/modboundary.f90:785-791): checksim / fielddump / statsdump see the state ahead of it
  !! (src/program.f90:199-214).
  subroutine stage3_ahead
    use modglobal, only: timee, lfielddump, tnextfielddump, BCxm
    use udc_iface
    due_now = .false.
    if (.not. udc_stats_on_device) due_now = stats_sample_due()      ! (the drop-in statsdump samples on the device)
    if (lfielddump .and. timee >= tnextfielddump) due_now = .true.      ! fielddump's own condition (src/modfielddump.f90:392-396)
    if (due_now .and. BCxm /= 1) call udc_check(udc_set_open_x_sample_gap(udc_h, 1_c_int), 'udc_set_open_x_sample_gap')
  end subroutine stage3_ahead

  !> statsdump's sampling clock (src/modstatsdump.f90:738-741, 797-805, 1394-1397), kept in step here because the module keeps
  !! its own private: true on the steps on which the statsdump call that follows `halos` takes a sample.  Called once per RK
  !! stage 3, like the lines it mirrors.
  logical function stats_sample_due()
    use modglobal, only: timee, dt, tsample, tstatstart, lydump, lytdump, lxydump, lxytdump, ltdump, lmintdump, &
                         lkslicedump, lislicedump, ljslicedump, ltreedump
    stats_sample_due = .false.
    if (timee < tstatstart) return
    if (.not. (lydump .or. lytdump .or. lxydump .or. lxytdump .or. ltdump .or. lmintdump &
               .or. lkslicedump .or. lislicedump .or. ljslicedump .or. ltreedump)) return
    if (stat_clock == 0. .and. tsample <= dt) stat_clock = dt
    if (stat_clock >= tsample) then
      stats_sample_due = .true.
      stat_clock = dt
    else
      stat_clock = stat_clock + dt
    end if
  end function stats_sample_due

  !> w(kb) = 0 and the top ghost planes (src/modboundary.f90:115-247, periodic lateral subset)
  subroutine boundary
    use udc_iface
    use modglobal, only: BCxm
    call udc_begin(.false.)
    if ((BCxm == 2 .or. BCxm == 3) .and. .not. udc_in_loop) call open_x_startup
    if (BCxm == 3) call driver_inlet
    call udc_check(udc_boundary(udc_h), 'udc_boundary')
    if (udc_mode() <= 1) call udc_pull_vel(.true.)
    if (pulled_ahead) then      ! the host arrays hold the state ahead of this `boundary` (the dumps' view): writerestartfiles wants the one after it
      pulled_ahead = .false.
      call udc_pull_all
    end if
  end subroutine boundary

  !> BCxm = 3 (src/modboundary.f90:260-266): where the reference's `boundary` calls drivergen -- RK stage 3 and the start-up -- the
  !! reference's own moddriver interpolates the precursor's planes to the new time on the host, and the six planes go to the device,
  !! which applies them with the next `boundary` (xmi_driver).  Device-resident runs: the fused substep contains that `boundary`, so
  !! the drop-in tstep_integrate calls this first; the call from `boundary` that follows finds the same time and hands over the same planes.
  subroutine driver_inlet
    use udc_iface
    use modglobal, only: rk3step, lchunkread, jb, je, jh, kb, ke, kh, jhc, khc, nsv, ltempeq, lmoist, lhdriver, lqdriver, lsdriver
    use moddriver, only: drivergen, driverchunkread
    use modinletdata, only: u0driver, umdriver, v0driver, vmdriver, w0driver, wmdriver, thl0driver, thlmdriver, qt0driver, qtmdriver, &
                            sv0driver, svmdriver
    integer(c_int) :: lb(2), ub(2)
    integer :: n
    if (.not. (rk3step == 0 .or. rk3step == 3)) return
    if (lchunkread) call driverchunkread
    call drivergen
    lb = (/jb - jh, kb - kh/); ub = (/je + jh, ke + kh/)
    call udc_check(udc_set_open_x_inlet(udc_h, u0driver, umdriver, v0driver, vmdriver, w0driver, wmdriver, lb, ub), 'udc_set_open_x_inlet')
    ! BCxT / BCxq / BCxs = 3: the scalars' planes too (xTi_driver, xqi_driver, xsi_driver; src/modboundary.f90:795-901)
    if (ltempeq .and. lhdriver) call udc_check(udc_set_open_x_inlet_scalar(udc_h, UDC_THL0, thl0driver, thlmdriver, lb, ub), 'udc_set_open_x_inlet_scalar')
    if (lmoist .and. lqdriver) call udc_check(udc_set_open_x_inlet_scalar(udc_h, UDC_QT0, qt0driver, qtmdriver, lb, ub), 'udc_set_open_x_inlet_scalar')
    if (nsv > 0 .and. lsdriver) then
      lb = (/jb - jhc, kb - khc/); ub = (/je + jhc, ke + khc/)
      do n = 1, nsv
        call udc_check(udc_set_open_x_inlet_scalar(udc_h, UDC_SV0 + 3*(n - 1), sv0driver(:, :, n), svmdriver(:, :, n), lb, ub), &
                       'udc_set_open_x_inlet_scalar')
      end do
    end if
  end subroutine driver_inlet

  !> BCxm = 2, the `boundary` of the start-up (src/program.f90:118): the outlet's speed uouttot from the slab averages diagfld has
  !! just formed (src/modboundary.f90:141-160; start-up order src/modstartup.f90:1604) -- the first substep's `boundary` still reads
  !! those, hence hold_first -- and one convective step of the outlet with the start-up's rk3step and dt (:914).  In the loop the
  !! library refreshes uouttot itself from the state every substep starts from.
  subroutine open_x_startup
    use udc_iface
    use iso_c_binding, only: c_loc, c_null_ptr
    use modglobal, only: ib, ie, jb, je, kb, ke, kh, ktot, dzf, zh, dt, rk3step, luvolflowr, luoutflowr
    use modfields, only: uouttot, u0, u0av, IIu, IIus
    use modinletdata, only: ubulk
    use modmpi, only: avexy_ibm
    real(c_double), target, save, allocatable :: wl(:)
    if (luoutflowr) then
      write (0, *) 'ERROR: libudcore boundary: luoutflowr with inflow / outflow in x is not on the device path'
      stop 1
    end if
    if (luvolflowr) then      ! src/modboundary.f90:158-160: a prescribed volume flow convects the outlet with ubulk
      uouttot = ubulk
      call udc_check(udc_set_open_x_outflow(udc_h, c_null_ptr, real(uouttot, c_double), 0_c_int), 'udc_set_open_x_outflow')
    else
      ! diagfld's own line (src/modthermodynamics.f90:271) on the host's start-up fields: the immersed boundary's masks are the host's
      ! at this point (the device takes the point lists with the first ibmwallfun / ibmnorm)
      call avexy_ibm(u0av(kb:ke + kh), u0(ib:ie, jb:je, kb:ke + kh), ib, ie, jb, je, kb, ke, kh, IIu(ib:ie, jb:je, kb:ke + kh), IIus(kb:ke + kh), .false.)
      if (.not. allocated(wl)) allocate (wl(ktot))
      wl = dzf(kb:ke)/(zh(ke + 1) - zh(kb + 1))
      uouttot = sum(u0av(kb:ke)*dzf(kb:ke))/(zh(ke + 1) - zh(kb + 1))
      call udc_check(udc_set_open_x_outflow(udc_h, c_loc(wl), real(uouttot, c_double), 1_c_int), 'udc_set_open_x_outflow')
    end if
    call udc_check(udc_set_boundary_rk3coef(udc_h, real(dt/(4. - real(rk3step)), c_double)), 'udc_set_boundary_rk3coef')
  end subroutine open_x_startup

  !> gravity-wave damping in the sponge layer (src/modboundary.f90:1447-1492): tend -= (field - ref(k)) tsc(k).
  !! Applied after masscorr together with fixuinf1's table; the constant scalar sources registered for a device-mode
  !! run (udc_set_scalar_source) are added here as well, grwdamp being the first drop-in routine after the
  !! reference's scalsource (src/program.f90:181-191).
  subroutine grwdamp
    use modglobal, only: kb, ke, lcoriol, igrw_damp, geodamptime, ltempeq, lmoist
    use modfields, only: ug, vg, thl0av, qt0av, u0av, v0av
    use udc_iface
    integer :: k
    real :: c
    if (udc_scalsrc_on .and. udc_mode() == 2) call udc_check(udc_scalsource(udc_h), 'udc_scalsource')
    select case (igrw_damp)
    case (0)
    case (1, 2, 3)
      call udc_tab_start(1)
      do k = ksp, ke
        if (igrw_damp == 2) then
          call damp(k, ROW_UP, ug(k), tsc(k))
          call damp(k, ROW_VP, vg(k), tsc(k))
        else
          call damp(k, ROW_UP, u0av(k), tsc(k))
          call damp(k, ROW_VP, v0av(k), tsc(k))
        end if
        call damp(k, ROW_WP, 0., tsc(k))
        if (ltempeq) call damp(k, ROW_THLP, thl0av(k), tsc(k))
        if (lmoist) call damp(k, ROW_QTP, qt0av(k), tsc(k))
        if (igrw_damp == 1 .and. lcoriol) then
          c = (1./(geodamptime*rnu0))*tsc(k)
          call damp(k, ROW_UP, ug(k), c)
          call damp(k, ROW_VP, vg(k), c)
        end if
      end do
    case default
      write (0, *) "ERROR: no gravity wave damping option selected"
      stop 1
    end select
    call udc_tab_apply(1)
  contains
    subroutine damp(k, row, ref, coef)
      integer, intent(in) :: k, row
      real, intent(in) :: ref, coef
      udc_tabA(k, row, 1) = udc_tabA(k, row, 1) + ref*coef
      udc_tabB(k, row, 1) = udc_tabB(k, row, 1) - coef
    end subroutine damp
  end subroutine grwdamp

  !> thl, qt and sv at level ke set to their slab averages (src/modboundary.f90:1556-1599; called by readinitfiles
  !! on a warm start): host arithmetic on the host arrays, before the time loop
  subroutine tqaver
    use mpi
    use modmpi, only: comm3d, mpierr, my_real
    use modglobal, only: ib, ie, jb, je, ke, nsv, rslabs
    use modfields, only: thl0, qt0, sv0
    real :: loc(2 + max(nsv, 1)), tot(2 + max(nsv, 1))
    integer :: n
    loc(1) = sum(thl0(ib:ie, jb:je, ke)); loc(2) = sum(qt0(ib:ie, jb:je, ke))
    do n = 1, nsv
      loc(2 + n) = sum(sv0(ib:ie, jb:je, ke, n))
    end do
    call MPI_ALLREDUCE(loc, tot, 2 + nsv, MY_REAL, MPI_SUM, comm3d, mpierr)
    tot = tot/rslabs
    thl0(ib:ie, jb:je, ke) = tot(1); qt0(ib:ie, jb:je, ke) = tot(2)
    do n = 1, nsv
      sv0(ib:ie, jb:je, ke, n) = tot(2 + n)
    end do
  end subroutine tqaver

  subroutine inside(what)
    character(*), intent(in) :: what
    write (0, *) 'ERROR: libudcore modboundary: ', what, ' is part of the device kernels and cannot be called on its own'
    stop 1
  end subroutine inside

  subroutine bcp(p)
    real, intent(inout) :: p(:, :, :)
    call inside('bcp')
  end subroutine bcp

  subroutine bcpup(pup, pvp, pwp, rk3coef)
    real, intent(inout) :: pup(:, :, :), pvp(:, :, :), pwp(:, :, :)
    real, intent(in) :: rk3coef
    call inside('bcpup')
  end subroutine bcpup

  subroutine closurebc
    call inside('closurebc')
  end subroutine closurebc

  subroutine xm_periodic
    call inside('xm_periodic')
  end subroutine xm_periodic
  subroutine xT_periodic
    call inside('xT_periodic')
  end subroutine xT_periodic
  subroutine xq_periodic
    call inside('xq_periodic')
  end subroutine xq_periodic
  subroutine xs_periodic
    call inside('xs_periodic')
  end subroutine xs_periodic
  subroutine ym_periodic
    call inside('ym_periodic')
  end subroutine ym_periodic
  subroutine yT_periodic
    call inside('yT_periodic')
  end subroutine yT_periodic
  subroutine yq_periodic
    call inside('yq_periodic')
  end subroutine yq_periodic
  subroutine ys_periodic
    call inside('ys_periodic')
  end subroutine ys_periodic

end module modboundary
