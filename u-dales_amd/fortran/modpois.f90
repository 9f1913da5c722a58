!> Drop-in replacement for the reference's module modpois (src/modpois.f90).
!! Public surface as at src/modpois.f90:41.  `poisson` = fillps + bcpup + FFT(x,y) +
!! tridiagonal(z) + tderive + bcp, all on the device (udc_poisson: rocFFT-backed, the 8 pencil
!! transposes of the reference become 0 on one GPU / 2 RCCL all-to-alls on a node).
!! The public work arrays that nothing outside modpois references in the reference
!! (pup, pvp, pwp, rhs, dpupdx, dpvpdy, dpwpdz, Fxy, Fxyz, xyzrt) are kept as zero-size
!! allocatables so that `use modpois, only: ...` still links; `p` is a real mirror.
module modpois
  use iso_c_binding, only: c_int, c_double
  implicit none
  private
  public :: initpois, poisson, exitpois, p, pup, pvp, pwp, rhs, dpupdx, dpvpdy, dpwpdz, xyzrt, sp, Fxy, Fxyz, &
            dpdztop, pij
  save

  real, allocatable, target :: p(:, :, :)
  real, allocatable, target :: pup(:, :, :), pvp(:, :, :), pwp(:, :, :)
  real, allocatable, target :: rhs(:, :, :), dpupdx(:, :, :), dpvpdy(:, :, :), dpwpdz(:, :, :)
  real, allocatable, target :: Fxy(:, :, :), Fxyz(:, :, :), xyzrt(:, :, :)
  real, allocatable :: dpdztop(:, :), pij(:)
  integer :: sp = 0      ! the reference's DECOMP_INFO for the unused 2decomp FFT branches

contains

  subroutine initpois
    use modglobal, only: ib, ie, ih, jb, je, jh, kb, ke, kh, ipoiss, POISS_FFT2D, BCxm, BCym, BCzp
    implicit none
    if (ipoiss /= POISS_FFT2D) then
      write (0, *) 'Invalid choice for Poisson solver'     ! as src/modpois.f90:897-898
      stop 1
    end if
    if (BCxm < 1 .or. BCxm > 3 .or. BCym /= 1 .or. (BCzp /= 1 .and. BCzp /= 2)) then
      write (0, *) 'ERROR: libudcore poisson: BCxm 1 (periodic) or 2 (Neumann: cosine transform in x), periodic y; BCzp 1 (tridiagonal z) or 2 (cosine transform in z)'
      stop 1
    end if
    allocate (p(ib - ih:ie + ih, jb - jh:je + jh, kb - kh:ke + kh)); p = 0.
    allocate (pup(0, 0, 0), pvp(0, 0, 0), pwp(0, 0, 0), rhs(0, 0, 0), dpupdx(0, 0, 0), dpvpdy(0, 0, 0), &
              dpwpdz(0, 0, 0), Fxy(0, 0, 0), Fxyz(0, 0, 0), xyzrt(0, 0, 0))
    allocate (dpdztop(ib:ie, jb:je), pij(kb:ke + kh))
  end subroutine initpois

  subroutine poisson
    use udc_iface, only: udc_tic, udc_toc, UDC_T_RECORD
    call udc_tic(UDC_T_RECORD)
    call poisson_timed
    call udc_toc(UDC_T_RECORD)
  end subroutine poisson

  subroutine poisson_timed
    use modglobal, only: ib, jb, kb, ih, jh, kh, rk3step, dt
    use modfields, only: pres0
    use udc_iface
    implicit none
    call udc_begin(.true.)
    call udc_check(udc_poisson(udc_h, int(rk3step, c_int), real(dt, c_double)), 'udc_poisson')
    if (udc_mode() <= 1) then
      call udc_pull_tend
      call udc_pull3(UDC_P, p, (/ib - ih, jb - jh, kb - kh/))
      call udc_pull3(UDC_PRES0, pres0, (/ib - ih, jb - jh, kb - kh/))
    end if
  end subroutine poisson_timed

  subroutine exitpois
    implicit none
    if (allocated(p)) deallocate (p)
  end subroutine exitpois

end module modpois
