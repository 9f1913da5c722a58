!> Drop-in replacement for the reference's module modforces (src/modforces.f90).
!! Same module name and public list (src/modforces.f90:41-43).  The routines of the time loop
!! (src/program.f90:144-188: shiftedPBCs, coriolis, forces, lstend, nudge, masscorr, fixuinf2, fixuinf1) hand
!! their work to libudcore; in device mode (UDC_RESIDENCY=2) they only record their call and the drop-in
!! tstep_integrate launches the fused substep (udc_set_deferred, include/udcore.h).
!!
!! lstend, nudge and fixuinf1 add per-level constants to the tendencies (tend(i,j,k) += A(k)): this module builds the
!! tables from diagfld's slab averages (u0av ... of modfields, kept fresh by the drop-in thermodynamics) with the
!! reference's expressions and registers them (udc_set_level_forcing); the device adds them.  One table per tendency
!! and application point: lstend and nudge (applied before masscorr) share one, so lstend starts it and nudge -- the
!! routine the reference's loop calls right after -- registers and applies it (masscorr does so if nudge was skipped).
!!
!! masscorr: the volume-flow branches and the u outflow-rate branch (luoutflowr) on the device.  Not taken over: lvoutflowr (the
!! reference's own call overruns its array there, src/modforces.f90:441-442) and periodicEBcorr (energy balance) -- refused with
!! the reference's error convention.
module modforces
  use iso_c_binding, only: c_int, c_double
  implicit none
  save
  private
  public :: forces, coriolis, lstend, fixuinf1, fixuinf2, fixthetainf, &
            detfreestream, detfreestrtmp, nudge, &
            masscorr, uoutletarea, voutletarea, fluidvolume, calcfluidvolumes, shiftedPBCs, periodicEBcorr

contains

  !> remaining terms of the momentum equations: -dpdxl, -dpdyl, buoyancy, wp(kb) = 0, thlpcar (src/modforces.f90:46-133)
  subroutine forces
    use udc_iface
    call udc_begin(.true.)
    call udc_check(udc_forces(udc_h), 'udc_forces')
    call udc_end_tend
  end subroutine forces

  !> Coriolis terms / relaxation to the geostrophic profile (src/modforces.f90:600-717)
  subroutine coriolis
    use modglobal, only: lcoriol, lprofforc
    use udc_iface
    if (.not. (lcoriol .or. lprofforc)) return
    call udc_begin(.true.)
    call udc_check(udc_coriolis(udc_h), 'udc_coriolis')
    call udc_end_tend
  end subroutine coriolis

  !> large-scale advection and subsidence (src/modforces.f90:719-822)
  subroutine lstend
    use modglobal, only: kb, ke, dzh, nsv, lmomsubs
    use modfields, only: whls, u0av, v0av, thl0av, qt0av, sv0av, &
                         dudxls, dudyls, dvdxls, dvdyls, dthldxls, dthldyls, dqtdxls, dqtdyls, dqtdtls
    use udc_iface, only: udc_need_avg, udc_tab_start, tabA => udc_tabA, ROW_UP, ROW_VP, ROW_THLP, ROW_QTP, ROW_SVP
    integer :: k, n, ku, kl
    real :: w
    if (.not. udc_need_avg) return      ! every input profile is zero: nothing to add
    call udc_tab_start(0)
    do k = kb, ke
      ! downwind differences: towards k+1 for subsiding motion, towards k-1 for rising motion; the lowest level only
      ! feels subsidence (src/modforces.f90:768-781, 790-812)
      if (whls(k + 1) < 0) then
        ku = k + 1; kl = k; w = whls(k + 1)/dzh(k + 1)
      else if (k > kb) then
        ku = k; kl = k - 1; w = whls(k)/dzh(k)
      else
        ku = k; kl = k; w = 0.
      end if
      tabA(k, ROW_THLP, 0) = tabA(k, ROW_THLP, 0) - u0av(k)*dthldxls(k) - v0av(k)*dthldyls(k) - w*(thl0av(ku) - thl0av(kl))
      tabA(k, ROW_QTP, 0) = tabA(k, ROW_QTP, 0) - u0av(k)*dqtdxls(k) - v0av(k)*dqtdyls(k) - w*(qt0av(ku) - qt0av(kl)) + dqtdtls(k)
      tabA(k, ROW_UP, 0) = tabA(k, ROW_UP, 0) - u0av(k)*dudxls(k) - v0av(k)*dudyls(k)
      tabA(k, ROW_VP, 0) = tabA(k, ROW_VP, 0) - u0av(k)*dvdxls(k) - v0av(k)*dvdyls(k)
      if (lmomsubs) then
        tabA(k, ROW_UP, 0) = tabA(k, ROW_UP, 0) - w*(u0av(ku) - u0av(kl))
        tabA(k, ROW_VP, 0) = tabA(k, ROW_VP, 0) - w*(v0av(ku) - v0av(kl))
      end if
      do n = 1, nsv
        tabA(k, ROW_SVP + n, 0) = tabA(k, ROW_SVP + n, 0) - w*(sv0av(ku, n) - sv0av(kl, n))
      end do
    end do
  end subroutine lstend

  !> relaxation of the slab means towards the input profiles above kb + nnudge (src/modforces.f90:824-860)
  subroutine nudge
    use modglobal, only: kb, ke, lmoist, ltempeq, lnudge, lnudgevel, tnudge, nnudge, nsv
    use modfields, only: sv0av, thl0av, qt0av, u0av, v0av, uprof, vprof, thlprof, qtprof, svprof
    use udc_iface, only: udc_tab_start, udc_tab_apply, tabA => udc_tabA, ROW_UP, ROW_VP, ROW_THLP, ROW_QTP, ROW_SVP
    integer :: k, n
    if (lnudge) then
      call udc_tab_start(0)
      do k = kb + nnudge, ke
        if (lnudgevel) then
          tabA(k, ROW_UP, 0) = tabA(k, ROW_UP, 0) - (u0av(k) - uprof(k))/tnudge
          tabA(k, ROW_VP, 0) = tabA(k, ROW_VP, 0) - (v0av(k) - vprof(k))/tnudge
        end if
        do n = 1, nsv
          tabA(k, ROW_SVP + n, 0) = tabA(k, ROW_SVP + n, 0) - (sv0av(k, n) - svprof(k, n))/tnudge
        end do
        if (ltempeq) tabA(k, ROW_THLP, 0) = tabA(k, ROW_THLP, 0) - (thl0av(k) - thlprof(k))/tnudge
        if (lmoist) tabA(k, ROW_QTP, 0) = tabA(k, ROW_QTP, 0) - (qt0av(k) - qtprof(k))/tnudge
      end do
    end if
    call udc_tab_apply(0)
  end subroutine nudge

  !> prescribed volume flow: up += (uflowrate - <um + rk3coef up>)/rk3coef, same for v (src/modforces.f90:389-417, 467-494);
  !! luoutflowr: <.> over the outlet plane i = ie instead (:352-387)
  subroutine masscorr
    use modglobal, only: rk3step, dt, linoutflow, luoutflowr, lvoutflowr, luvolflowr, lvvolflowr
    use udc_iface
    if (udc_tab_open(0)) call udc_tab_apply(0)      ! (lstend's table if the driver skipped nudge)
    if (linoutflow) return
    if (lvoutflowr) then
      write (0, *) 'ERROR: libudcore masscorr: lvoutflowr is not available (src/modforces.f90:441 hands sumy_ibm an array it overruns)'
      stop 1
    end if
    if (.not. (luvolflowr .or. lvvolflowr .or. luoutflowr)) return
    call udc_begin(.true.)
    call udc_check(udc_masscorr(udc_h, int(rk3step, c_int), real(dt, c_double)), 'udc_masscorr')
    call udc_end_tend
  end subroutine masscorr

  subroutine detfreestream(freestream)
    use modglobal, only: ke, lvinf
    use modfields, only: u0av, v0av
    real, intent(out) :: freestream
    freestream = merge(v0av(ke), u0av(ke), lvinf)       ! src/modforces.f90:135-148
  end subroutine detfreestream

  !> mean temperature of the top level (src/modforces.f90:150-172) = diagfld's thl0av(ke) for equidistant x
  subroutine detfreestrtmp(freestrtmp)
    use modglobal, only: ke
    use modfields, only: thl0av
    real, intent(out) :: freestrtmp
    freestrtmp = thl0av(ke)
  end subroutine detfreestrtmp

  !> dp/dx follows an ODE on the running mean of the free-stream velocity (src/modforces.f90:174-218); the drop-in
  !! tstep_integrate advances dpdxl with dgdt (src/modtstep.f90:194-195)
  subroutine fixuinf2
    use modglobal, only: dt, Uinf, ifixuinf, tscale, rk3step, inletav, freestreamav
    use modfields, only: dgdt
    real :: freestream
    if ((ifixuinf == 2) .and. (rk3step == 3)) then
      call detfreestream(freestream)
      freestreamav = freestream*dt/inletav + (1.-dt/inletav)*freestreamav
      dgdt = (1./tscale)*(freestreamav - Uinf)
    end if
  end subroutine fixuinf2

  !> top-level mean pulled back to Uinf (Vinf) on RK stage 3 (src/modforces.f90:220-288): a constant added on every
  !! level, applied after masscorr (`when` = 1, where the drop-in grwdamp applies the tables registered for it)
  subroutine fixuinf1
    use modglobal, only: kb, ke, dt, Uinf, Vinf, ifixuinf, rk3step, lvinf
    use modfields, only: u0av, v0av
    use udc_iface
    if (ifixuinf /= 1 .or. rk3step /= 3) return
    call udc_tab_start(1)
    udc_tabA(kb:ke, ROW_UP, 1) = udc_tabA(kb:ke, ROW_UP, 1) - (1./dt)*(u0av(ke) - Uinf)
    if (lvinf) udc_tabA(kb:ke, ROW_VP, 1) = udc_tabA(kb:ke, ROW_VP, 1) - (1./dt)*(v0av(ke) - Vinf)
  end subroutine fixuinf1

  subroutine fixthetainf      ! (empty in the reference as well, src/modforces.f90:290-326)
  end subroutine fixthetainf

  !> outlet areas (fluid part of the planes i = ie and j = je) and fluid volume, blocks excluded through the c mask of
  !! createmasks (src/modforces.f90:499-598; equidistant x, as there)
  subroutine uoutletarea(area)
    use mpi
    use modglobal, only: ie, jb, je, kb, ke, dy, dzf
    use modfields, only: IIc
    use modmpi, only: comm3d, mpierr, my_real
    real, intent(out) :: area
    real :: loc
    integer :: k
    loc = 0.
    do k = kb, ke
      loc = loc + sum(IIc(ie, jb:je, k))*dy*dzf(k)
    end do
    call MPI_ALLREDUCE(loc, area, 1, MY_REAL, MPI_SUM, comm3d, mpierr)
  end subroutine uoutletarea

  subroutine voutletarea(area)
    use mpi
    use modglobal, only: ib, ie, je, kb, ke, dxf, dzf, jtot, jerank
    use modfields, only: IIc
    use modmpi, only: comm3d, mpierr, my_real
    real, intent(out) :: area
    real :: loc
    integer :: k
    loc = 0.
    if (jerank) then      ! the plane j = jtot lives on the rank that holds the domain's last row (src/modglobal.f90:652-660)
      do k = kb, ke
        loc = loc + sum(IIc(ib:ie, je, k))*dxf(1)*dzf(k)
      end do
    end if
    call MPI_ALLREDUCE(loc, area, 1, MY_REAL, MPI_SUM, comm3d, mpierr)
  end subroutine voutletarea

  subroutine fluidvolume(volume)
    use mpi
    use modglobal, only: ib, ie, jb, je, kb, ke, dy, dxf, dzf
    use modfields, only: IIc
    use modmpi, only: comm3d, mpierr, my_real
    real, intent(out) :: volume
    real :: loc
    integer :: k
    loc = 0.
    do k = kb, ke
      loc = loc + sum(IIc(ib:ie, jb:je, k))*dxf(1)*dy*dzf(k)
    end do
    call MPI_ALLREDUCE(loc, volume, 1, MY_REAL, MPI_SUM, comm3d, mpierr)
  end subroutine fluidvolume

  subroutine calcfluidvolumes
    use modfields, only: uoutarea, voutarea, fluidvol
    call uoutletarea(uoutarea)
    call voutletarea(voutarea)
    call fluidvolume(fluidvol)
  end subroutine calcfluidvolumes

  subroutine periodicEBcorr
    use modglobal, only: lperiodicEBcorr
    if (.not. lperiodicEBcorr) return
    write (0, *) 'ERROR: libudcore: lperiodicEBcorr (energy-balance volume sink) is not available'
    stop 1
  end subroutine periodicEBcorr

  !> spanwise drift in the downstream half of the domain (src/modforces.f90:953-980): the device applies
  !! -vs (phi(j) - phi(j-1))/dy with vs = a u0av(k) sinx(i); u0av is refreshed here every substep
  subroutine shiftedPBCs
    use modglobal, only: itot, kb, ke, xh, ds, xlen, pi
    use modfields, only: u0av
    use udc_iface
    real(c_double), allocatable, save :: sinx(:)
    integer :: ig
    if (.not. (ds > 0)) return
    call udc_ensure
    if (.not. allocated(sinx)) then
      allocate (sinx(itot))
      sinx = 0.
      do ig = itot/2 + 1, itot
        sinx(ig) = sin(pi*(xh(ig) - xh(itot/2))/(0.5*xlen))
      end do
    end if
    call udc_check(udc_set_shifted_pbc(udc_h, real(0.5*pi*ds/(0.5*xlen), c_double), sinx, int(itot, c_int), u0av(kb:ke), &
                                       int(ke - kb + 1, c_int)), 'udc_set_shifted_pbc')
    call udc_begin(.true.)
    call udc_check(udc_shifted_pbcs(udc_h), 'udc_shifted_pbcs')
    call udc_end_tend
  end subroutine shiftedPBCs

end module modforces
