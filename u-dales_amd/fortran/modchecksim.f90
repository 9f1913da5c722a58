!> Drop-in replacement for the reference's module modchecksim (src/modchecksim.f90): same module name, same public list
!! (initchecksim, checksim), same &NAMCHECKSIM group, same clock, same report lines.
!! Why it is replaced although it only prints: the reference's checksim loops three times over the host arrays on every time
!! step (tcheck defaults to 0) -- ~50 ms at 256^3, fifteen device substeps -- and in a device-resident run those arrays are not
!! the current state.  Here the three numbers come from the device (udc_checksim: calccourant's signed maximum, calcdiffnr,
!! chkdiv, the reference's expressions) in every residency mode.
module modchecksim
  use iso_c_binding, only: c_double
  implicit none
  private
  public initchecksim, checksim
  real :: tcheck = 0.
  real :: tnext = 0.
  real :: dtmn = 0., ndt = 0.
  logical :: pending = .false.
  real :: p_timee = 0., p_dtmn = 0.
  save
contains

  !> src/modchecksim.f90:44-74
  subroutine initchecksim
    use mpi
    use modglobal, only: ifnamopt, fname_options, dtmax, ladaptive, btime
    use modmpi, only: myid, my_real, comm3d, mpierr
    integer :: ierr
    namelist /NAMCHECKSIM/ tcheck
    if (myid == 0) then
      open (ifnamopt, file=fname_options, status='old', iostat=ierr)
      read (ifnamopt, NAMCHECKSIM, iostat=ierr)
      if (ierr > 0) then
        write (0, *) 'ERROR: Problem in namoptions NAMCHECKSIM'
        write (0, *) 'iostat error: ', ierr
        stop 1
      end if
      close (ifnamopt)
      if ((.not. ladaptive) .and. (tcheck < dtmax)) tcheck = dtmax
    end if
    call MPI_BCAST(tcheck, 1, MY_REAL, 0, comm3d, mpierr)
    tnext = tcheck + btime
  end subroutine initchecksim

  !> src/modchecksim.f90:76-100, the three diagnostics (:102-203) taken on the device.
  !! The reductions are queued behind the substep that has just been launched (udc_checksim_begin) and picked up at the NEXT report
  !! (udc_checksim_end), so that a run that reports every time step (tcheck <= dt: the default with a fixed step, :66) never waits
  !! for the device inside its loop; the report keeps the simulation time and mean step it belongs to and appears one interval
  !! later.  The last step of the run -- and every step with UDC_CHECKSIM_SYNC=1 -- reports at once, as the reference does.
  subroutine checksim
    use udc_iface, only: udc_tic, udc_toc, UDC_T_CHECKSIM
    call udc_tic(UDC_T_CHECKSIM)
    call checksim_timed
    call udc_toc(UDC_T_CHECKSIM)
  end subroutine checksim

  subroutine checksim_timed
    use modglobal, only: timee, rk3step, dt, timeleft
    use udc_iface
    character(8) :: env
    integer :: stat
    logical, save :: first = .true., sync_always = .false.
    if (timee == 0.0) return
    if (rk3step /= 3) return
    dtmn = dtmn + dt; ndt = ndt + 1.
    ! the last step of the run is not a report step: the report still waiting from the previous interval appears now, not never
    if (timee < tnext .and. pending .and. timeleft <= 0) call report
    if (timee < tnext) return
    if (first) then
      first = .false.
      call get_environment_variable('UDC_CHECKSIM_SYNC', env, status=stat)
      if (stat == 0) sync_always = (trim(env) /= '0')
    end if
    tnext = tnext + tcheck
    dtmn = dtmn/ndt
    if (pending) call report
    call udc_begin(.false.)
    call udc_check(udc_checksim_begin(udc_h, real(dtmn, c_double)), 'udc_checksim_begin')
    pending = .true.; p_timee = timee; p_dtmn = dtmn
    if (sync_always .or. timeleft <= 0) call report
    dtmn = 0.
    ndt = 0.
  end subroutine checksim_timed

  subroutine report
    use modmpi, only: myid
    use udc_iface
    character(20) :: timeday
    real(c_double) :: d(4)
    call udc_check(udc_checksim_end(udc_h, d), 'udc_checksim_end')
    pending = .false.
    if (myid == 0) then
      call date_and_time(time=timeday)
      write (*, *) '================================================================='
      write (*, '(3A,F15.5,A,F12.9)') 'Time of Day: ', timeday(1:10), '    Time of Simulation: ', p_timee, '    dt: ', p_dtmn
      write (*, '(A,ES10.2)') 'Courant numbers (x,y,z,tot):', d(1)
      write (6, '(A,ES10.2)') 'Diffusion number:', d(2)
      write (6, '(A,2ES11.2)') 'divmax, divtot = ', d(3), d(4)
    end if
  end subroutine report

end module modchecksim
