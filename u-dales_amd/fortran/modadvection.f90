!> Drop-in replacement for the reference's module modadvection (src/modadvection.f90).
!! Same module name, same public procedure `advection` (called at src/program.f90:142);
!! the body hands the work to libudcore (udc_advection: fused cd2 momentum advection incl.
!! -grad(pres0), kappa-scheme scalars) instead of looping on the host.
module modadvection
  implicit none
contains

  subroutine advection
    use udc_iface, only: udc_tic, udc_toc, UDC_T_RECORD
    call udc_tic(UDC_T_RECORD)
    call advection_timed
    call udc_toc(UDC_T_RECORD)
  end subroutine advection

  subroutine advection_timed
    use iso_c_binding, only: c_int
    use modglobal, only: iadv_mom, iadv_cd2, iadv_thl, iadv_qt, iadv_kappa, ltempeq, lmoist
    use modsubgriddata, only: loneeqn
    use udc_iface
    implicit none

    ! the device library implements what every BASELINE configuration uses
    ! (src/modadvection.f90:46-99 dispatches on the same switches)
    if (iadv_mom /= iadv_cd2) then
      write (0, *) 'ERROR: Unknown advection scheme'
      stop 1
    end if
    if (lmoist .and. iadv_qt /= iadv_cd2) then     ! qt: advecc_2nd only (src/modadvection.f90:78-86)
      write (0, *) 'ERROR: Unknown advection scheme'
      stop 1
    end if
    if (ltempeq .and. iadv_thl /= iadv_cd2 .and. iadv_thl /= iadv_kappa) then   ! src/modadvection.f90:64-76
      write (0, *) 'ERROR: Unknown advection scheme'
      stop 1
    end if

    call udc_begin(.true.)
    call udc_check(udc_advection(udc_h), 'udc_advection')
    if (udc_mode() <= 1) call udc_pull_tend
  end subroutine advection_timed

end module modadvection
