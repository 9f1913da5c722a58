!> Drop-in replacement for the reference's module modtstep (src/modtstep.f90).
!! tstep_update: RK3 stage counter and (adaptive) time step; the two global maxima come from the
!! device (udc_tstep_maxima), the bookkeeping stays here on the host exactly as the driver expects
!! (dt, timee, timeleft, ntimee, ntrun of modglobal).  tstep_integrate: RK3 update, tendency
!! zeroing and the m <- 0 copy on stage 3 on the device; the monitor file is still written here.
module modtstep
  use iso_c_binding, only: c_int, c_double
  implicit none
contains

  subroutine tstep_update
    use modglobal, only: rk3step, timee, dtmax, dt, ntimee, ntrun, courant, diffnr, dt_lim, ladaptive, timeleft
    use udc_iface
    implicit none
    real(c_double) :: courtot, diffnrtot

    if (.not. udc_in_loop) then      ! first call: the time loop starts (src/program.f90:132-134)
      call udc_enter_loop
      if (udc_residency == 2) call register_scalar_sources
    end if
    rk3step = mod(rk3step, 3) + 1
    if (rk3step /= 1) return
    if (ladaptive) then
      call udc_ensure
      if (udc_mode() == 0) call udc_push_state
      call udc_check(udc_tstep_maxima(udc_h, real(dt, c_double), courtot, diffnrtot), 'udc_tstep_maxima')
      if (courtot <= 0) write (6, *) 'courtot=0!'
      if (diffnrtot <= 0) write (6, *) 'diffnrtot=0!'
      dt = min(dtmax, dt*courant/courtot, dt*diffnr/diffnrtot)
      timeleft = timeleft - dt
      dt_lim = timeleft
      timee = timee + dt
    else
      dt = dtmax
      timee = timee + dt
      timeleft = timeleft - dt
    end if
    ntimee = ntimee + 1
    ntrun = ntrun + 1
  end subroutine tstep_update

  !> Device mode: the point / line sources of the scalars (src/modscalsource.f90:379-483) depend on neither time nor
  !! flow, so one call of the reference's own scalsource on a zeroed svp gives them; the device adds the registered
  !! boxes every substep (udc_scalsource, called by the drop-in grwdamp), the host's scalsource calls in the loop then
  !! only touch the host's svp, which device mode ignores.
  subroutine register_scalar_sources
    use modglobal, only: ib, ie, jb, je, kb, ke, nsv, lscasrc, lscasrcl
    use modfields, only: svp
    use modscalsource, only: scalsource
    use udc_iface
    integer :: n, i0, i1, j0, j1, k0, k1, i, j, k
    real(c_double), allocatable :: box(:, :, :)
    integer(c_int) :: lb(3), ub(3)
    if (nsv == 0 .or. .not. (lscasrc .or. lscasrcl)) return
    svp = 0.
    call scalsource
    do n = 1, nsv
      i0 = ie + 1; i1 = ib - 1; j0 = je + 1; j1 = jb - 1; k0 = ke + 1; k1 = kb - 1
      do k = kb, ke
        do j = jb, je
          do i = ib, ie
            if (svp(i, j, k, n) /= 0.) then
              i0 = min(i0, i); i1 = max(i1, i); j0 = min(j0, j); j1 = max(j1, j); k0 = min(k0, k); k1 = max(k1, k)
            end if
          end do
        end do
      end do
      if (i1 < i0) cycle
      allocate (box(i0:i1, j0:j1, k0:k1))
      box = svp(i0:i1, j0:j1, k0:k1, n)
      lb = (/i0, j0, k0/); ub = (/i1, j1, k1/)
      call udc_check(udc_set_scalar_source(udc_h, int(n - 1, c_int), box, lb, ub), 'udc_set_scalar_source')
      deallocate (box)
      udc_scalsrc_on = .true.
    end do
    svp = 0.
  end subroutine register_scalar_sources

  subroutine tstep_integrate
    use udc_iface, only: udc_tic, udc_toc, UDC_T_INTEGRATE
    call udc_tic(UDC_T_INTEGRATE)
    call tstep_integrate_timed
    call udc_toc(UDC_T_INTEGRATE)
  end subroutine tstep_integrate

  subroutine tstep_integrate_timed
    use modglobal, only: rk3step, dt, timee, ifixuinf, lchem, ltempeq, lmoist, iinletgen, idriver, ib, ie, jb, je, kb, ke
    use modfields, only: up, vp, wp, svp, thlp, qtp, e12p, thl0, thl0c, dpdxl, dpdyl, dgdt
    use modsubgriddata, only: loneeqn
    use modmpi, only: myid, cmyid
    use modibm, only: ibm_facet_output
    use modboundary, only: driver_inlet, stage3_ahead
    use modglobal, only: BCxm
    use udc_iface
    implicit none

    call udc_begin(.true.)
    ! BCxm = 3, device-resident: the substep launched below ends with the `boundary` that applies the precursor's planes of the new time
    if (BCxm == 3 .and. udc_mode() == 2) call driver_inlet
    if (rk3step == 3 .and. udc_mode() == 2) call stage3_ahead
    ! (device mode: this launches the recorded routines of the substep, fused)
    call udc_check(udc_tstep_integrate(udc_h, int(rk3step, c_int), real(dt, c_double)), 'udc_tstep_integrate')
    call ibm_facet_output      ! lwritefac: fac.NNN.nc's record, which the reference's ibmwallfun writes (the substep has run only now)
    if (ifixuinf == 2) then      ! src/modtstep.f90:194-195: the dp/dx ODE (dgdt from fixuinf2); `forces` has used the old dpdxl
      dpdxl(:) = dpdxl(:) + dgdt*(dt/(4. - real(rk3step)))
      call udc_check(udc_set_forcing(udc_h, dpdxl(kb:ke), dpdyl(kb:ke), int(ke - kb + 1, c_int)), 'udc_set_forcing')
    end if
    if (udc_mode() <= 1) then
      call udc_pull_vel(rk3step == 3)
      if (ltempeq) thl0c(ib:ie, jb:je, kb:ke) = thl0(ib:ie, jb:je, kb:ke)      ! src/modtstep.f90:249 (halos / boundary fill its ghosts)
    end if

    if ((myid == 0) .and. (rk3step == 3)) then
      open (unit=11, file='monitor'//cmyid//'.txt', position='append')
      write (11, '(13(6e14.6))') timee
      close (11)
    end if

    ! the tendencies the host sees are zeroed as in the reference (src/modtstep.f90:322-328); in device mode the host
    ! arrays are not part of the computation and stay as they are
    if (udc_mode() <= 1) then
      up = 0.; vp = 0.; wp = 0.; thlp = 0.; svp = 0.; e12p = 0.; qtp = 0.
    end if
  end subroutine tstep_integrate_timed

end module modtstep
