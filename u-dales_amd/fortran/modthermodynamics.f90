!> Drop-in replacement for the reference's module modthermodynamics (src/modthermodynamics.f90).
!! Same module name and public entities (initthermodynamics, thermodynamics, calc_halflev, exitthermodynamics, lqlnr,
!! th0av, chi_half, thv0).  `thermodynamics` (:57-124) runs on the device (udc_thermodynamics: thermo, diagfld, fromztop,
!! calc_halflev, thermo on the half levels, calthv and the slab average thvh -- everything `forces`' buoyancy term and the
!! one-equation closure then read, both of which are device routines too).  What comes back to the host are the 1-D
!! products other host code reads: diagfld's slab averages u0av, v0av, thl0av, qt0av, sv0av (only when some routine of the
!! loop needs them: lstend, nudge, grwdamp, fixuinf, shiftedPBCs) and, with moisture, presf, presh, exnf, exnh, thvh,
!! ql0av, th0av (udc_thermo_state).  The 3-D work arrays thl0h, qt0h, ql0h, thv0h, dthvdz and thv0 are not mirrored.
!! In dry air the device's buoyancy kernel takes thv0h = thl0h and its slab average itself (src/modforces.f90:73-84).
module modthermodynamics
  use iso_c_binding, only: c_int, c_double
  implicit none
  public :: thermodynamics, calc_halflev
  public :: lqlnr
  logical :: lqlnr = .false.        !< switch for ql calc. with Newton-Raphson (on/off)
  real, allocatable :: th0av(:)
  real :: chi_half = 0.5
  real, allocatable :: thv0(:, :, :)

contains

  subroutine initthermodynamics
    use modglobal, only: kb, ke, kh, ib, ie, jb, je
    use udc_iface, only: udc_lqlnr
    allocate (th0av(kb:ke + kh))
    allocate (thv0(ib:ie, jb:je, kb:ke + kh))
    th0av = 0.
    udc_lqlnr = lqlnr
  end subroutine initthermodynamics

  subroutine thermodynamics
    use udc_iface, only: udc_tic, udc_toc, UDC_T_THERMO
    call udc_tic(UDC_T_THERMO)
    call thermodynamics_timed
    call udc_toc(UDC_T_THERMO)
  end subroutine thermodynamics

  subroutine thermodynamics_timed
    use modglobal, only: kb, ke, kh, lmoist, ltempeq, lbuoyancy, rk3step, timee, tnextrestart, timeleft
    use modfields, only: presf, presh, exnf, exnh, thvh, thl0av, qt0av, ql0av
    use udc_iface
    real(c_double), allocatable :: t(:, :)
    integer :: n
    logical :: moist_dev
    call udc_begin(.false.)
    moist_dev = lmoist .and. ltempeq .and. (lbuoyancy .or. loneeqn_dev())      ! udc_set_moist_thermo was called
    ! start-up (src/modstartup.f90:1601: before program.f90:118's `boundary`): calthv's dthvdz at the top level is built from the
    ! ghost plane as it is NOW; the device keeps that plane for the first closure (udc_calthv; udc_thermodynamics does it itself)
    if (.not. udc_in_loop .and. ltempeq .and. .not. moist_dev) call udc_check(udc_calthv(udc_h), 'udc_calthv')
    if (moist_dev) then
      call udc_check(udc_thermodynamics(udc_h), 'udc_thermodynamics')
      if (udc_mode() <= 1 .or. udc_need_avg) then
        n = ke + kh - kb + 1
        allocate (t(n, 9))
        call udc_check(udc_thermo_state(udc_h, t, int(n, c_int), 0_c_int), 'udc_thermo_state')
        presf(kb:ke + kh) = t(:, 1); presh(kb:ke + kh) = t(:, 2); exnf(kb:ke + kh) = t(:, 3); exnh(kb:ke + kh) = t(:, 4)
        thvh(kb:ke + kh) = t(:, 5); ql0av(kb:ke + kh) = t(:, 8); th0av(kb:ke + kh) = t(:, 9)
        deallocate (t)
      end if
    end if
    if (udc_need_avg) call udc_refresh_averages
    ! device mode: writerestartfiles comes next (src/program.f90:220) and reads the host arrays when a restart is due
    ! (src/modsave.f90:74-78)
    if (udc_mode() == 2 .and. rk3step == 3) then
      if (timee >= tnextrestart .or. timeleft <= 0) call udc_pull_all
    end if
  end subroutine thermodynamics_timed

  !> thl0h, qt0h (src/modthermodynamics.f90:508-539) are read by `thermodynamics` alone, which computes them on the device
  subroutine calc_halflev
  end subroutine calc_halflev

  subroutine exitthermodynamics
    if (allocated(th0av)) deallocate (th0av)
  end subroutine exitthermodynamics

end module modthermodynamics
