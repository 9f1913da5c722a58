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

    rk3step = mod(rk3step, 3) + 1
    if (rk3step /= 1) return
    if (ladaptive) then
      call udc_ensure
      if (udc_residency == 0) call udc_push_state
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

  subroutine tstep_integrate
    use modglobal, only: rk3step, dt, timee, ifixuinf, lchem, ltempeq, lmoist, iinletgen, idriver, ib, ie, jb, je, kb, ke
    use modfields, only: up, vp, wp, svp, thlp, qtp, e12p, thl0, thl0c, dpdxl, dpdyl, dgdt
    use modsubgriddata, only: loneeqn
    use modmpi, only: myid, cmyid
    use udc_iface
    implicit none

    call udc_ensure
    if (ifixuinf == 2) then      ! src/modtstep.f90:194-195: the dp/dx ODE (dgdt from the host's fixuinf2)
      dpdxl(:) = dpdxl(:) + dgdt*(dt/(4. - real(rk3step)))
      call udc_check(udc_set_forcing(udc_h, dpdxl(kb:ke), dpdyl(kb:ke), int(ke - kb + 1, c_int)), 'udc_set_forcing')
    end if
    select case (udc_residency)
    case (0)
      call udc_push_state
      call udc_push_tend
    case (1)
      call udc_push_tend
    end select
    call udc_check(udc_tstep_integrate(udc_h, int(rk3step, c_int), real(dt, c_double)), 'udc_tstep_integrate')
    ! keep the device's own ghosts consistent with what the host's halos/boundary will produce
    call udc_check(udc_halos(udc_h), 'udc_halos')
    call udc_check(udc_boundary(udc_h), 'udc_boundary')
    if (udc_residency <= 1) then
      call udc_pull_vel(rk3step == 3)
      if (ltempeq) thl0c(ib:ie, jb:je, kb:ke) = thl0(ib:ie, jb:je, kb:ke)      ! src/modtstep.f90:249 (the host's halos / boundary fill its ghosts)
    end if

    if ((myid == 0) .and. (rk3step == 3)) then
      open (unit=11, file='monitor'//cmyid//'.txt', position='append')
      write (11, '(13(6e14.6))') timee
      close (11)
    end if

    ! the tendencies the host sees are zeroed as in the reference (src/modtstep.f90:322-328)
    up = 0.; vp = 0.; wp = 0.; thlp = 0.; svp = 0.; e12p = 0.; qtp = 0.
  end subroutine tstep_integrate

end module modtstep
