!> Drop-in replacement for the reference's module modsubgrid (src/modsubgrid.f90).
!! Public surface as at src/modsubgrid.f90:41: subgrid, initsubgrid, exitsubgrid, subgridnamelist;
!! re-exports modsubgriddata (modstatsdump does `use modsubgrid, only: ekh, ekm`, src/modstatsdump.f90:538).
!! closure + closurebc + diffu/diffv/diffw/diffc run on the device (udc_subgrid).
module modsubgrid
  use modsubgriddata
  implicit none
  save
  public :: subgrid, initsubgrid, exitsubgrid, subgridnamelist

contains

  !> Host-side set-up the rest of the code relies on: ekm/ekh exist with the reference's bounds
  !! (src/modsubgrid.f90:54-61) and the model constants are derived (:64-77).
  subroutine initsubgrid
    use modglobal, only: ih, ib, ie, jh, jb, je, kb, ke, kh, pi
    implicit none
    real :: ceps, ch

    call subgridnamelist

    allocate (ekm(ib - ih:ie + ih, jb - jh:je + jh, kb - kh:ke + kh))
    allocate (ekh(ib - ih:ie + ih, jb - jh:je + jh, kb - kh:ke + kh))
    allocate (zlt(ib - ih:ie + ih, jb - jh:je + jh, kb:ke + kh))
    allocate (sbdiss(ib - ih:ie + ih, jb - jh:je + jh, kb:ke + kh))
    allocate (sbshr(ib - ih:ie + ih, jb - jh:je + jh, kb:ke + kh))
    allocate (sbbuo(ib - ih:ie + ih, jb - jh:je + jh, kb:ke + kh))
    allocate (csz(ib - ih:ie + ih, kb:ke + kh))
    allocate (damp(ib:ie, jb:je, kb:ke))
    ekm = 0.; ekh = 0.; zlt = 0.; sbdiss = 0.; sbshr = 0.; sbbuo = 0.
    damp = 1.

    ! Kolmogorov-constant relations of the Smagorinsky/TKE family
    cm = cf/(2.*pi)*(1.5*alpha_kolm)**(-1.5)
    ch = prandtl
    ch2 = ch - ch1
    ceps = 2.*pi/cf*(1.5*alpha_kolm)**(-1.5)
    ce1 = (cn**2)*(cm/Rigc - ch1*cm)
    ce2 = ceps - ce1
    if (cs == -1.) then
      csz(:, :) = (cm**3/ceps)**0.25
    else
      csz(:, :) = cs
    end if
  end subroutine initsubgrid

  !> &NAMSUBGRID, same variables as the reference (src/modsubgrid.f90:89-90), rank 0 reads, all get it.
  subroutine subgridnamelist
    use modglobal, only: ifnamopt, fname_options, lles
    use modmpi, only: myid, comm3d, mpierr, my_real, mpi_logical
    implicit none
    integer :: ierr
    namelist /NAMSUBGRID/ ldelta, lmason, cf, cn, Rigc, Prandtl, lsmagorinsky, lvreman, loneeqn, &
      c_vreman, cs, nmason, lbuoycorr

    if (myid == 0) then
      open (ifnamopt, file=fname_options, status='old', iostat=ierr)
      read (ifnamopt, NAMSUBGRID, iostat=ierr)
      if (ierr > 0) then
        write (0, *) 'ERROR: Problem in namoptions NAMSUBGRID'
        write (0, *) 'iostat error: ', ierr
        stop 1
      end if
      close (ifnamopt)
    end if
    call MPI_BCAST(ldelta, 1, MPI_LOGICAL, 0, comm3d, mpierr)
    call MPI_BCAST(lmason, 1, MPI_LOGICAL, 0, comm3d, mpierr)
    call MPI_BCAST(lsmagorinsky, 1, MPI_LOGICAL, 0, comm3d, mpierr)
    call MPI_BCAST(lvreman, 1, MPI_LOGICAL, 0, comm3d, mpierr)
    call MPI_BCAST(lbuoycorr, 1, MPI_LOGICAL, 0, comm3d, mpierr)
    call MPI_BCAST(loneeqn, 1, MPI_LOGICAL, 0, comm3d, mpierr)
    call bcast_real(nmason); call bcast_real(c_vreman); call bcast_real(cs); call bcast_real(cf)
    call bcast_real(cn); call bcast_real(Rigc); call bcast_real(Prandtl)
    prandtli = 1./Prandtl
    if (lsmagorinsky .or. lvreman .or. loneeqn) lles = .true.
  contains
    subroutine bcast_real(x)
      real, intent(inout) :: x
      call MPI_BCAST(x, 1, MY_REAL, 0, comm3d, mpierr)
    end subroutine bcast_real
  end subroutine subgridnamelist

  subroutine subgrid
    use udc_iface, only: udc_tic, udc_toc, UDC_T_RECORD
    call udc_tic(UDC_T_RECORD)
    call subgrid_timed
    call udc_toc(UDC_T_RECORD)
  end subroutine subgrid

  subroutine subgrid_timed
    use modglobal, only: ib, jb, kb, ih, jh, kh, ltempeq, lmoist
    use udc_iface
    implicit none
    call udc_begin(.true.)
    call udc_check(udc_subgrid(udc_h), 'udc_subgrid')
    if (udc_mode() <= 1) then
      call udc_pull_tend
      ! ekm/ekh are read by the host (statistics, tstep_update, IBM wall functions)
      call udc_pull3(UDC_EKM, ekm, (/ib - ih, jb - jh, kb - kh/))
      call udc_pull3(UDC_EKH, ekh, (/ib - ih, jb - jh, kb - kh/))
    end if
    if (udc_mode() == 0) call udc_pull_vel(.true.)   ! top ghost rows re-imposed by closurebc
  end subroutine subgrid_timed

  subroutine exitsubgrid
    implicit none
    deallocate (ekm, ekh, zlt, sbdiss, sbbuo, sbshr, csz)
  end subroutine exitsubgrid

end module modsubgrid
