!> Drop-in replacement for the reference's module modibm (src/modibm.f90), floor part.
!! Same module name; public procedures initibm, createmasks, bottom, ibmwallfun, ibmnorm and the namelist variables
!! of src/modibm.f90:30-62.  `bottom` (:1998-2100: floor wall function wfmneutral / wfuno for momentum and temperature,
!! flux floors for thl, qt and the scalars, the e120 floor ghost) runs on the device (udc_bottom).  The immersed
!! boundary itself (libm: ibmnorm, ibmwallfun and the *_corr routines they drive) is not taken over yet: a deck with
!! libm = .true. stops in initibm with the reference's error convention.  Without IBM, createmasks (:2103-2135)
!! fills the masks with ones and the slab counts with the cells per slab.
!! The tau_x / tau_y / tau_z / thl_flux diagnostics of `bottom` (:2015-2018, 2094-2097: the tendency increments of the
!! floor, read by the statistics) are not produced.
module modibm
  use iso_c_binding, only: c_int, c_double
  use modibmdata
  implicit none
  save
  public :: initibm, ibmnorm, ibmwallfun, bottom, lbottom, createmasks, &
            nsolpts_u, nsolpts_v, nsolpts_w, nsolpts_c, &
            nbndpts_u, nbndpts_v, nbndpts_w, nbndpts_c, &
            nfctsecs_u, nfctsecs_v, nfctsecs_w, nfctsecs_c, &
            mask_u, mask_v, mask_w, mask_c

  logical :: lbottom = .false.
  logical :: lnorec = .false.
  integer :: nsolpts_u = 0, nsolpts_v = 0, nsolpts_w = 0, nsolpts_c = 0, &
             nbndpts_u = 0, nbndpts_v = 0, nbndpts_w = 0, nbndpts_c = 0, &
             nfctsecs_u = 0, nfctsecs_v = 0, nfctsecs_w = 0, nfctsecs_c = 0
  real, allocatable, target, dimension(:, :, :) :: mask_u, mask_v, mask_w, mask_c

contains

  subroutine initibm
    use modglobal, only: libm, ib, ie, ih, jb, je, jh, kb, ke, kh, ltempeq, lmoist, nsv, BCbotm, BCbotT, BCbotq, BCbots
    use modsurfdata, only: z0
    use udc_iface, only: udc_set_floor
    if (libm) then
      write (0, *) 'ERROR: libudcore modibm: immersed boundaries (libm) are not available in this build'
      stop 1
    end if
    if (lbottom) then      ! what `bottom` would stop on at its first call (src/modibm.f90:2027-2090)
      if (BCbotm /= 2 .and. BCbotm /= 3) then
        write (0, *) 'ERROR: bottom boundary type for momentum undefined'
        stop 1
      end if
      if (ltempeq .and. BCbotT /= 1 .and. BCbotT /= 2) then
        write (0, *) 'ERROR: bottom boundary type for temperature undefined'
        stop 1
      end if
      if (lmoist .and. BCbotq /= 1) then
        write (0, *) 'ERROR: bottom boundary type for moisture undefined'
        stop 1
      end if
      if (nsv > 0 .and. BCbots /= 1) then
        write (0, *) 'ERROR: bottom boundary type for scalars undefined'
        stop 1
      end if
    end if
    allocate (mask_u(ib - ih:ie + ih, jb - jh:je + jh, kb - kh:ke + kh)); mask_u = 1.
    allocate (mask_v(ib - ih:ie + ih, jb - jh:je + jh, kb - kh:ke + kh)); mask_v = 1.
    allocate (mask_w(ib - ih:ie + ih, jb - jh:je + jh, kb - kh:ke + kh)); mask_w = 1.
    allocate (mask_c(ib - ih:ie + ih, jb - jh:je + jh, kb - kh:ke + kh)); mask_c = 1.
    call udc_set_floor(lbottom, real(z0, c_double))
  end subroutine initibm

  !> masks and slab cell counts without blocks (src/modibm.f90:2121-2141)
  subroutine createmasks
    use modglobal, only: libm, jtot, rslabs
    use modfields, only: IIc, IIu, IIv, IIw, IIuw, IIvw, IIuv, IIcs, IIus, IIvs, IIws, IIuws, IIvws, IIuvs, &
                         IIct, IIut, IIvt, IIwt, IIuwt
    if (libm) then
      write (0, *) 'ERROR: libudcore modibm: immersed boundaries (libm) are not available in this build'
      stop 1
    end if
    IIc = 1; IIu = 1; IIv = 1; IIw = 1; IIuw = 1; IIvw = 1; IIuv = 1
    IIcs = nint(rslabs); IIus = nint(rslabs); IIvs = nint(rslabs); IIws = nint(rslabs)
    IIuws = nint(rslabs); IIvws = nint(rslabs); IIuvs = nint(rslabs)
    IIct = jtot; IIut = jtot; IIvt = jtot; IIwt = jtot; IIuwt = jtot
  end subroutine createmasks

  !> floor of the domain (src/modibm.f90:1998-2100)
  subroutine bottom
    use modsubgriddata, only: loneeqn
    use udc_iface
    if (.not. (lbottom .or. loneeqn_dev())) return
    call udc_begin(.true.)
    call udc_check(udc_bottom(udc_h), 'udc_bottom')
    call udc_end_tend
    if (udc_mode() == 0 .and. loneeqn_dev()) call udc_pull_vel(.true.)     ! e120, e12m floor ghosts
  end subroutine bottom

  subroutine ibmwallfun      ! no-ops without libm (src/modibm.f90:1167, 697)
  end subroutine ibmwallfun

  subroutine ibmnorm
  end subroutine ibmnorm

end module modibm
