!> Drop-in replacement of the reference's `modstatsdump` (src/modstatsdump.f90) for device-resident runs: the time-averaged
!! statistics -- tdump (:286-333, 1557-1663), mintdump (:335-355, 1670-1684), xytdump (:247-283, 1404-1460), ytdump (:160-213,
!! 1466-1551) -- are accumulated by libudcore on the device (udc_stats_sample: the sampling lines :812-1213) and reach the host
!! only when a record is written, instead of every field crossing the bus on every sample.  Same public routines, same clocks
!! (tsamplep, tstatsdumpp: :797-811, 1393-1399, 1723-1729), same files, variables, attributes and record order, written through
!! the reference's own modstat_nc.
!!
!! The instantaneous tables written at every sample -- xydump (:215-245, 1326-1349), ydump (:131-158, 1294-1323) -- are the device's
!! tables of the last sample (udc_stats_xy / udc_stats_y).  The slice dumps (kslicedump / islicedump / jslicedump, :418-502, 1352-1389:
!! um, vm, wm, thlm, qtm on one plane every tsample) download that plane alone.
!!
!! Not taken over (stop 1, the reference's error convention; link the reference's own modstatsdump instead -- the drop-in
!! modboundary then refreshes the host arrays on exactly the steps on which it samples, u-dales_amd/fortran/Makefile):
!! the TKE budget (whose sampling call the reference has commented out), the tree dump, and tdump's PSS defect with lchem.
module modstatsdump
  use iso_c_binding, only: c_int, c_double, c_ptr
  implicit none
  private
  public :: initstatsdump, statsdump, exitstatsdump
  save

  integer, parameter :: nstatt = 32, nstatmint = 6, nstatxyt = 23, nstatyt = 34, nstatxy = 15, nstaty = 14
  integer :: ncidt = -1, ncidmint = -1, ncidxyt = -1, ncidyt = -1, ncidsl(3) = -1, ncidxy = -1, ncidy = -1
  integer :: nrect = 0, nrecmint = 0, nrecxyt = 0, nrecyt = 0, nrecsl(3) = 0, nrecxy = 0, nrecy = 0
  character(80) :: ncstatxy(nstatxy, 4), ncstaty(nstaty, 4)
  character(80) :: ncstatsl(5, 4, 3)
  logical :: slice_on(3) = .false., slices = .false.
  character(80) :: tvar(1, 4)
  character(80) :: ncstatt(nstatt, 4), ncstatmint(nstatmint, 4), ncstatxyt(nstatxyt, 4), ncstatyt(nstatyt, 4)
  real :: tsamplep = 0., tstatsdumpp = 0.
  logical :: active = .false., device_ready = .false.

  ! name | long name | unit | dimensions, in the reference's record order
  character(*), parameter :: meta_t(nstatt) = [character(64) :: &
    'ut|Streamwise velocity|m/s|mttt', 'vt|Spanwise velocity|m/s|tmtt', 'wt|Vertical velocity|m/s|ttmt', 'thlt|Temperature|K|tttt', &
    'qtt|Moisture|kg/kg|tttt', 'pt|Pressure|m^2/s^2|tttt', 'sca1t|Concentration field 1|g/m^3|tttt', &
    'sca2t|Concentration field 2|g/m^3|tttt', 'sca3t|Concentration field 3|g/m^3|tttt', 'sca4t|Concentration field 4|g/m^3|tttt', &
    'PSS|PSS defect|gm/s|tttt', 'upwpt|Turbulent momentum flux|m^2/s^2|mtmt', 'vpwpt|Turbulent momentum flux|m^2/s^2|tmmt', &
    'upvpt|Turbulent momentum flux|m^2/s^2|mmtt', 'wpthlpt|Turbulent heat flux|K m/s|ttmt', 'wpsca1pt|Turbulent flux 1|gm/s|ttmt', &
    'wpsca2pt|Turbulent flux 2|gm/s|ttmt', 'wpsca3pt|Turbulent flux 3|gm/s|ttmt', 'wpsca4pt|Turbulent flux 4|gm/s|ttmt', &
    'thlpthlpt|Temperature variance|K^2|tttt', 'upuptc|u variance|m^2/s^2|tttt', 'vpvptc|v variance|m^2/s^2|tttt', &
    'wpwptc|w variance|m^2/s^2|tttt', 'tketc|TKE|m^2/s^2|tttt', 'sca1psca1pt|Concentration variance 1|g^2/m^6|tttt', &
    'sca2psca2pt|Concentration variance 2|g^2/m^6|tttt', 'sca3psca3pt|Concentration variance 3|g^2/m^6|tttt', &
    'sca4psca4pt|Concentration variance 4|g^2/m^6|tttt', 'sv1sgs|SGS flux 1|gm/s|ttmt', 'sv2sgs|SGS flux 2|gm/s|ttmt', &
    'sv3sgs|SGS flux 3|gm/s|ttmt', 'sv4sgs|SGS flux 4|gm/s|ttmt']
  character(*), parameter :: meta_xyt(nstatxyt) = [character(64) :: &
    'uxyt|Streamwise velocity|m/s|tt', 'vxyt|Spanwise velocity|m/s|tt', 'wxyt|Vertical velocity|m/s|mt', 'thlxyt|Temperature|K|tt', &
    'qtxyt|Moisture|kg/kg|tt', 'pxyt|Pressure|m^2/s^2|tt', 'upwpxyt|Turbulent mom. flux|m^2/s^2|mt', 'wpthlpxyt|Turbulent heat flux|K m/s|mt', &
    'vpwpxyt|Turbulent mom. flux|m^2/s^2|mt', 'upvpxyt|Turbulent mom. flux|m^2/s^2|mt', 'uwxyt|Kinematic mom. flux|m^2/s^2|mt', &
    'wthlxyt|Kinematic heat flux|K m/s|mt', 'uvxyt|Kinematic mom. flux|m^2/s^2|mt', 'vwxyt|Kinematic mom. flux|m^2/s^2|mt', &
    'wwxyt|Kinematic mom. flux|m^2/s^2|mt', 'usgsxyt|SGS mom. flux|m^2/s^2|mt', 'thlsgsxyt|SGS heat flux|K m/s|mt', &
    'vsgsxyt|SGS mom. flux|K m/s|mt', 'thlpthlptxy|Temp. variance|K^2|tt', 'upuptxyc|u variance|m^2/s^2|tt', &
    'vpvptxyc|v variance|m^2/s^2|tt', 'wpwptxyc|w variance|m^2/s^2|tt', 'tketxyc|tke|m^2/s^2|tt']
  character(*), parameter :: meta_yt(nstatyt) = [character(64) :: &
    'uyt|Streamwise velocity|m/s|m0tt', 'vyt|Spanwise velocity|m/s|t0tt', 'wyt|Vertical velocity|m/s|t0mt', 'thlyt|Temperature|K|t0tt', &
    'qtyt|Moisture|kg/kg|t0tt', 'sca1yt|Scalar field 1|kg/m^3|t0tt', 'sca2yt|Scalar field 2|kg/m^3|t0tt', 'sca3yt|Scalar field 3|kg/m^3|t0tt', &
    'upwpyt|Turbulent mom. flux|m^2/s^2|m0mt', 'wpthlpyt|Turbulent heat flux|K m/s|t0mt', 'wpqtpyt|Turbulent moisture flux|kg/kg m/s|t0mt', &
    'wpsca1tpyt|Turbulent scalar flux|M m/s|t0mt', 'wpsca2tpyt|Turbulent scalar flux|M m/s|t0mt', 'wpsca3tpyt|Turbulent scalar flux|M m/s|t0mt', &
    'uwyt|Kinematic mom. flux|m^2/s^2|m0mt', 'wthlyt|Kinematic heat flux|K m/s|t0mt', 'wqtyt|Kinematic moisture flux|K m/s|t0mt', &
    'wsca1yt|Kinematic scalar flux|K m/s|t0mt', 'wsca2yt|Kinematic scalar flux|K m/s|t0mt', 'wsca3yt|Kinematic scalar flux|K m/s|t0mt', &
    'upupyt|mom. variance|m^2/s^2|t0tt', 'wpwpyt|mom. variance|m^2/s^2|t0tt', 'thlpthlpyt|temp. variance|K^2|t0tt', &
    'qtpqtpyt|moisture. variance|kg^2/kg^2|t0tt', 'sca1tpsca1pyt|scalar. variance|M^2|t0tt', 'sca2tpsca2pyt|scalar. variance|M^2|t0tt', &
    'sca3tpsca3pyt|scalar. variance|M^2|t0tt', 'usgsyt|SGS mom. flux|m^2/s^2|m0mt', 'wsgsyt|SGS mom. flux|m^2/s^2|t0mt', &
    'thlsgsyt|SGS heat flux|K m/s|t0mt', 'qtsgsyt|SGS moisture flux|kg/kg m/s|t0mt', 'sca1sgsyt|SGS scalar flux|M m/s|t0mt', &
    'sca2sgsyt|SGS scalar flux|M m/s|t0mt', 'sca3sgsyt|SGS scalar flux|M m/s|t0mt']

  character(*), parameter :: meta_xy(nstatxy) = [character(64) :: &
    'uxy|Streamwise velocity|m/s|tt', 'vxy|Spanwise velocity|m/s|tt', 'wxy|Vertical velocity|m/s|mt', 'thlxy|Temperature|K|tt', &
    'qtxy|Moisture|kg/kg|tt', 'pxy|Pressure|m^2/s^2|tt', 'upwpxy|Mom. flux|m^2/s^2|mt', 'wpthlpxy|Heat flux|Km/s|mt', 'vpwpxy|Mom. flux|Km/s|mt', &
    'usgsxy|SGS mom. flux|m^2/s^2|mt', 'thlsgsxy|SGS heat flux|Km/s|mt', 'vsgsxy|SGS mom. flux|m^2/s^2|mt', &
    'uwxyik|Advective mom. flux|m^2/s^2|mt', 'wthlxy|Advective heat flux|K m/s|mt', 'vwxy|Advective mom. flux|m^2/s^2|mt']
  character(*), parameter :: meta_y(nstaty) = [character(64) :: &
    'uy|Streamwise velocity|m/s|m0tt', 'vy|Spanwise velocity|m/s|t0tt', 'wy|Vertical velocity|m/s|t0mt', 'thly|Temperature|K|t0tt', &
    'qty|Moisture|kg/kg|t0tt', 'sca1y|Scalar field 1|kg/m^3|t0tt', 'sca2y|Scalar field 2|kg/m^3|t0tt', 'sca3y|Scalar field 3|kg/m^3|t0tt', &
    'upwpy|Turbulent mom. flux|m^2/s^2|m0mt', 'wpthlpy|Turbulent heat flux|K m/s|t0mt', 'usgsy|SGS mom. flux|m^2/s^2|m0mt', &
    'thlsgsy|SGS heat flux|K m/s|t0mt', 'uwyik|Advective mom. flux|m^2/s^2|m0mt', 'wthlyk|Advective heat flux|K m/s|t0mt']

  ! slices: plane k = kslice (1), i = islice (2), j = jslice (3); the dimension string has a 0 where the plane cuts
  character(*), parameter :: slice_tag(3) = ['k', 'i', 'j']
  character(*), parameter :: slice_dims(5, 3) = reshape([character(4) :: 'mt0t', 'tm0t', 'tt0t', 'tt0t', 'tt0t', &
                                                         '0ttt', '0mtt', '0tmt', '0ttt', '0ttt', &
                                                         'm0tt', 't0tt', 't0mt', 't0tt', 't0tt'], [5, 3])
  character(*), parameter :: slice_var(5) = [character(4) :: 'u', 'v', 'w', 'thl', 'qt']
  character(*), parameter :: slice_long(5) = [character(24) :: 'Streamwise velocity', 'Spanwise velocity', 'Vertical velocity', &
                                              'Potential temperature', 'Specific humidity']

  ! accumulators of the device (include/udcore.h, UDC_ST_*)
  integer(c_int), parameter :: ST_UMT = 0, ST_VMT = 1, ST_WMT = 2, ST_PT = 3, ST_UTC = 4, ST_VTC = 5, ST_WTC = 6, ST_UUTC = 7, ST_VVTC = 8, &
                               ST_WWTC = 9, ST_UWTIK = 10, ST_VWTJK = 11, ST_UVTIJ = 12, ST_UTIK = 13, ST_WTIK = 14, ST_VTJK = 15, &
                               ST_WTJK = 16, ST_UTIJ = 17, ST_VTIJ = 18, ST_THL = 19, ST_QT = 23, ST_SV = 27, ST_SV_STRIDE = 5

  interface
    integer(c_int) function udc_stats_enable(h, on) bind(C, name='udc_stats_enable')
      import :: c_ptr, c_int
      type(c_ptr), value :: h
      integer(c_int), value :: on
    end function udc_stats_enable
    integer(c_int) function udc_stats_sample(h, tsamplep, tstatsdumpp) bind(C, name='udc_stats_sample')
      import :: c_ptr, c_int, c_double
      type(c_ptr), value :: h
      real(c_double), value :: tsamplep, tstatsdumpp
    end function udc_stats_sample
    integer(c_int) function udc_stats_get(h, id, host, lb, ub) bind(C, name='udc_stats_get')
      import :: c_ptr, c_int, c_double
      type(c_ptr), value :: h
      integer(c_int), value :: id
      real(c_double), intent(out) :: host(*)
      integer(c_int), intent(in) :: lb(3), ub(3)
    end function udc_stats_get
    integer(c_int) function udc_stats_set_masks(h, bits, counts) bind(C, name='udc_stats_set_masks')
      import :: c_ptr, c_int
      type(c_ptr), value :: h
      integer(1), intent(in) :: bits(*)
      integer(c_int), intent(in) :: counts(*)
    end function udc_stats_set_masks
    integer(c_int) function udc_stats_set_forced(h, forced) bind(C, name='udc_stats_set_forced')
      import :: c_ptr, c_int
      type(c_ptr), value :: h
      integer(c_int), intent(in) :: forced(7)
    end function udc_stats_set_forced
    integer(c_int) function udc_stats_xyt(h, table) bind(C, name='udc_stats_xyt')
      import :: c_ptr, c_int, c_double
      type(c_ptr), value :: h
      real(c_double), intent(out) :: table(*)
    end function udc_stats_xyt
    integer(c_int) function udc_stats_yt(h, table) bind(C, name='udc_stats_yt')
      import :: c_ptr, c_int, c_double
      type(c_ptr), value :: h
      real(c_double), intent(out) :: table(*)
    end function udc_stats_yt
    integer(c_int) function udc_stats_xy(h, table) bind(C, name='udc_stats_xy')
      import :: c_ptr, c_int, c_double
      type(c_ptr), value :: h
      real(c_double), intent(out) :: table(*)
    end function udc_stats_xy
    integer(c_int) function udc_stats_y(h, table) bind(C, name='udc_stats_y')
      import :: c_ptr, c_int, c_double
      type(c_ptr), value :: h
      real(c_double), intent(out) :: table(*)
    end function udc_stats_y
  end interface

contains

  !> "name|long name|unit|dims" -> the four strings modstat_nc's define_nc wants
  subroutine unpack_meta(packed, out)
    character(*), intent(in) :: packed(:)
    character(80), intent(out) :: out(:, :)
    integer :: n, q, a, b
    do n = 1, size(packed)
      a = 1
      do q = 1, 4
        b = index(packed(n)(a:), '|')
        if (b == 0 .or. q == 4) then
          out(n, q) = trim(packed(n)(a:))
        else
          out(n, q) = packed(n)(a:a + b - 2)
          a = a + b
        end if
      end do
    end do
  end subroutine unpack_meta

  !> the files of the enabled sets, defined through the reference's modstat_nc like its own initstatsdump (:160-355)
  subroutine initstatsdump
    use modglobal, only: lydump, lytdump, ltkedump, lxydump, lxytdump, ltdump, lmintdump, ltreedump, lkslicedump, lislicedump, ljslicedump, &
                         lchem, imax, jmax, kb, ke, cexpnr, islice, jslice, islicerank, isliceloc, jslicerank, jsliceloc
    use decomp_2d, only: zstart, zend
    use modmpi, only: myid, cmyidx, cmyidy
    use modstat_nc, only: open_nc, define_nc, writestat_dims_nc
    use udc_iface, only: udc_stats_on_device
    character(80) :: fname
    integer :: nk, q, n
    if (ltkedump .or. ltreedump) then
      write (0, *) 'ERROR: libudcore statsdump: tkedump and treedump are not taken over by the device ', &
        'statistics; link the reference''s modstatsdump instead (u-dales_amd/fortran/Makefile)'
      stop 1
    end if
    if (ltdump .and. lchem) then
      write (0, *) 'ERROR: libudcore statsdump: tdump''s PSS defect (lchem) is not accumulated on the device'
      stop 1
    end if
    active = ltdump .or. lmintdump .or. lxytdump .or. lytdump .or. lxydump .or. lydump
    slices = lkslicedump .or. lislicedump .or. ljslicedump
    udc_stats_on_device = active .or. slices
    tsamplep = 0.; tstatsdumpp = 0.
    if (.not. (active .or. slices)) return
    nk = ke - kb + 1
    tvar(1, :) = [character(80) :: 'time', 'Time', 's', 'time']
    if (lydump) then
      call unpack_meta(meta_y, ncstaty)
      if (myid == 0) then
        fname = 'ydump.'//cexpnr//'.nc'
        call start_file(fname, ncidy, nrecy, nstaty, ncstaty, n1=imax, n3=nk)
      end if
    end if
    if (lxydump) then
      call unpack_meta(meta_xy, ncstatxy)
      if (myid == 0) then
        fname = 'xydump.'//cexpnr//'.nc'
        call start_file(fname, ncidxy, nrecxy, nstatxy, ncstatxy, n3=nk)
      end if
    end if
    if (lytdump) then
      call unpack_meta(meta_yt, ncstatyt)
      if (myid == 0) then
        fname = 'ytdump.'//cexpnr//'.nc'
        tvar(1, 2) = 'Sampling time'
        call start_file(fname, ncidyt, nrecyt, nstatyt, ncstatyt, n1=imax, n3=nk)
        tvar(1, 2) = 'Time'
      end if
    end if
    if (lxytdump) then
      call unpack_meta(meta_xyt, ncstatxyt)
      if (myid == 0) then
        fname = 'xytdump.'//cexpnr//'.nc'
        call start_file(fname, ncidxyt, nrecxyt, nstatxyt, ncstatxyt, n3=nk)
      end if
    end if
    if (ltdump) then        ! (one file per rank: each holds its own block)
      call unpack_meta(meta_t, ncstatt)
      fname = 'tdump.'//cmyidx//'.'//cmyidy//'.'//cexpnr//'.nc'
      call start_file(fname, ncidt, nrect, nstatt, ncstatt, n1=imax, n2=jmax, n3=nk)
    end if
    if (lmintdump) then
      call unpack_meta(meta_t(1:nstatmint), ncstatmint)
      fname = 'mintdump.'//cmyidx//'.'//cmyidy//'.'//cexpnr//'.nc'
      call start_file(fname, ncidmint, nrecmint, nstatmint, ncstatmint, n1=imax, n2=jmax, n3=nk)
    end if
    ! slices: every rank that holds the plane writes its own piece (k: all of them; i, j: the ranks the index falls on)
    islicerank = islice >= zstart(1) .and. islice <= zend(1); isliceloc = islice - zstart(1) + 1
    jslicerank = jslice >= zstart(2) .and. jslice <= zend(2); jsliceloc = jslice - zstart(2) + 1
    slice_on = [lkslicedump, lislicedump .and. islicerank, ljslicedump .and. jslicerank]
    do q = 1, 3
      if (.not. slice_on(q)) cycle
      do n = 1, 5
        ncstatsl(n, :, q) = [character(80) :: trim(slice_var(n))//'_'//slice_tag(q)//'slice', &
                             trim(slice_long(n))//' at '//slice_tag(q)//'slice', '-', slice_dims(n, q)]
      end do
      fname = slice_tag(q)//'slicedump.'//cmyidx//'.'//cmyidy//'.'//cexpnr//'.nc'
      select case (q)
      case (1); call start_file(fname, ncidsl(q), nrecsl(q), 5, ncstatsl(:, :, q), n1=imax, n2=jmax)
      case (2); call start_file(fname, ncidsl(q), nrecsl(q), 5, ncstatsl(:, :, q), n2=jmax, n3=nk)
      case (3); call start_file(fname, ncidsl(q), nrecsl(q), 5, ncstatsl(:, :, q), n1=imax, n3=nk)
      end select
    end do
  contains
    subroutine start_file(name, ncid, nrec, nvar, vars, n1, n2, n3)
      character(*), intent(in) :: name
      integer, intent(out) :: ncid
      integer, intent(inout) :: nrec
      integer, intent(in) :: nvar
      character(80), intent(in) :: vars(nvar, 4)
      integer, intent(in), optional :: n1, n2, n3
      call open_nc(name, ncid, nrec, n1=n1, n2=n2, n3=n3)
      if (nrec == 0) then
        call define_nc(ncid, 1, tvar)
        call writestat_dims_nc(ncid)
      end if
      call define_nc(ncid, nvar, vars)
    end subroutine start_file
  end subroutine initstatsdump

  !> first call inside the loop (the handle exists by then): switch the device accumulators on and hand over createmasks' masks
  subroutine device_setup
    use modglobal, only: ib, ie, jb, je, kb, ke, imax, jmax, lxytdump, lytdump, lxydump, lydump, libm
    use modfields, only: IIu, IIv, IIw, IIc, IIuw, IIvw, IIuv, IIus, IIvs, IIws, IIcs, IIuws, IIvws, IIuvs
    use udc_iface, only: udc_h, udc_check
    integer(1), allocatable :: bits(:, :, :)
    integer(c_int), allocatable :: counts(:, :)
    integer(c_int) :: forced(7)
    integer :: nk, q
    call udc_check(udc_stats_enable(udc_h, int(1 + merge(2, 0, lxytdump) + merge(4, 0, lytdump) + merge(8, 0, lxydump) + merge(16, 0, lydump), &
                                               c_int)), 'udc_stats_enable')
    device_ready = .true.
    if (.not. (libm .and. (lxytdump .or. lytdump .or. lxydump .or. lydump))) return
    ! one byte per cell, bit order IIu, IIv, IIw, IIc, IIuw, IIvw, IIuv, levels kb..ke; the global fluid counts per level.
    ! avexy_ibm's rule for a first level without fluid points (src/modmpi.f90:646-649: the unmasked sum over the count of level
    ! ke) is applied here, as udc_stats_set_masks asks: that level gets its bit set everywhere and the count of the last level
    nk = ke - kb + 1
    allocate (bits(imax, jmax, nk), counts(nk, 7))
    counts(:, 1) = IIus(kb:ke); counts(:, 2) = IIvs(kb:ke); counts(:, 3) = IIws(kb:ke); counts(:, 4) = IIcs(kb:ke)
    counts(:, 5) = IIuws(kb:ke); counts(:, 6) = IIvws(kb:ke); counts(:, 7) = IIuvs(kb:ke)
    bits = int(IIu(ib:ie, jb:je, kb:ke) + 2*IIv(ib:ie, jb:je, kb:ke) + 4*IIw(ib:ie, jb:je, kb:ke) + 8*IIc(ib:ie, jb:je, kb:ke) &
               + 16*IIuw(ib:ie, jb:je, kb:ke) + 32*IIvw(ib:ie, jb:je, kb:ke) + 64*IIuv(ib:ie, jb:je, kb:ke), 1)
    forced = 0
    do q = 1, 7
      if (counts(1, q) == 0) then
        forced(q) = 1
        counts(1, q) = counts(nk, q)
        bits(:, :, 1) = ior(bits(:, :, 1), int(2**(q - 1), 1))
      end if
    end do
    call udc_check(udc_stats_set_masks(udc_h, bits, counts), 'udc_stats_set_masks')
    call udc_check(udc_stats_set_forced(udc_h, forced), 'udc_stats_set_forced')
  end subroutine device_setup

  !> one accumulator of the device as a block (ib:ie, jb:je, kb:ke)
  subroutine fetch(id, a)
    use modglobal, only: ib, ie, jb, je, kb, ke
    use udc_iface, only: udc_h, udc_check
    integer(c_int), intent(in) :: id
    real, intent(out) :: a(ib:ie, jb:je, kb:ke)
    call udc_check(udc_stats_get(udc_h, id, a, int([ib, jb, kb], c_int), int([ie, je, ke], c_int)), 'udc_stats_get')
  end subroutine fetch

  !> the reference's two clocks (src/modstatsdump.f90:738-741, 797-811, 1393-1399, 1400, 1723-1729); a sample is one sweep on the
  !! device, a record is what crosses the bus
  subroutine statsdump
    use udc_iface, only: udc_tic, udc_toc, UDC_T_STATS
    call udc_tic(UDC_T_STATS)
    call statsdump_timed
    call udc_toc(UDC_T_STATS)
  end subroutine statsdump

  subroutine statsdump_timed
    use modglobal, only: rk3step, timee, dt, tsample, tstatsdump, tstatstart, lxytdump, lytdump, ltdump, lmintdump, lxydump, lydump
    use udc_iface, only: udc_h, udc_check, udc_begin
    use modglobal, only: BCxm
    if (.not. (active .or. slices)) return
    if (active .and. BCxm /= 1 .and. .not. device_ready) then
      ! inflow / outflow in x: a fused stage-3 substep of a handle with the statistics on ends ahead of `boundary`, where the reference samples
      ! (src/program.f90:199-214) -- so they are switched on at the first call (RK stage 1 of the first step), not at the first sample
      call udc_begin(.false.)
      call device_setup
    end if
    if (timee < tstatstart) return
    if (rk3step /= 3) return
    if (tsamplep == 0. .and. tsample <= dt) tsamplep = dt
    if (tstatsdumpp == 0. .and. tsample <= dt) tstatsdumpp = dt
    if (tsamplep >= tsample) then
      call udc_begin(.false.)
      if (active) then
        if (.not. device_ready) call device_setup
        call udc_check(udc_stats_sample(udc_h, real(tsamplep, c_double), real(tstatsdumpp, c_double)), 'udc_stats_sample')
        if (lydump) call write_y
        if (lxydump) call write_xy
      end if
      if (slices) call write_slices
      tsamplep = dt
    else
      tsamplep = tsamplep + dt
    end if
    if (tstatsdumpp >= tstatsdump) then
      if (active .and. .not. device_ready) then
        call udc_begin(.false.)
        call device_setup
      end if
      if (lxytdump) call write_xyt
      if (lytdump) call write_yt
      if (ltdump) call write_t
      if (lmintdump) call write_mint
      tstatsdumpp = dt
    else
      tstatsdumpp = tstatsdumpp + dt
    end if
  end subroutine statsdump_timed

  !> um, vm, wm, thlm, qtm on the planes of the slice dumps (:1352-1389): only the planes come over (two where a velocity is
  !! brought to the cell centre across the plane); fields the deck does not carry are read from the host arrays, where they never change
  subroutine write_slices
    use modglobal, only: ib, ie, jb, je, kb, ke, imax, jmax, kslice, isliceloc, jsliceloc, ltempeq, lmoist, timee
    use modfields, only: thlm, qtm
    use modstat_nc, only: writestat_nc
    use udc_iface, only: udc_pull3, UDC_UM, UDC_VM, UDC_WM, UDC_THLM, UDC_QTM
    real, allocatable :: v(:, :, :), a(:, :, :)
    integer :: nk
    nk = ke - kb + 1
    if (slice_on(1)) then
      allocate (v(imax, jmax, 5), a(ib:ie, jb:je, kslice:kslice + 1))
      call udc_pull3(UDC_UM, a(:, :, kslice:kslice), [ib, jb, kslice]); v(:, :, 1) = a(:, :, kslice)
      call udc_pull3(UDC_VM, a(:, :, kslice:kslice), [ib, jb, kslice]); v(:, :, 2) = a(:, :, kslice)
      call udc_pull3(UDC_WM, a, [ib, jb, kslice]); v(:, :, 3) = 0.5*(a(:, :, kslice) + a(:, :, kslice + 1))
      v(:, :, 4) = thlm(ib:ie, jb:je, kslice); v(:, :, 5) = qtm(ib:ie, jb:je, kslice)
      if (ltempeq) then
        call udc_pull3(UDC_THLM, a(:, :, kslice:kslice), [ib, jb, kslice]); v(:, :, 4) = a(:, :, kslice)
      end if
      if (lmoist) then
        call udc_pull3(UDC_QTM, a(:, :, kslice:kslice), [ib, jb, kslice]); v(:, :, 5) = a(:, :, kslice)
      end if
      call writestat_nc(ncidsl(1), 1, tvar, (/timee/), nrecsl(1), .true.)
      call writestat_nc(ncidsl(1), 5, ncstatsl(:, :, 1), v, nrecsl(1), imax, jmax)
      deallocate (v, a)
    end if
    if (slice_on(2)) then
      allocate (v(jmax, nk, 5), a(isliceloc:isliceloc + 1, jb:je, kb:ke))
      call udc_pull3(UDC_UM, a, [isliceloc, jb, kb]); v(:, :, 1) = 0.5*(a(isliceloc, :, :) + a(isliceloc + 1, :, :))
      call udc_pull3(UDC_VM, a(isliceloc:isliceloc, :, :), [isliceloc, jb, kb]); v(:, :, 2) = a(isliceloc, :, :)
      call udc_pull3(UDC_WM, a(isliceloc:isliceloc, :, :), [isliceloc, jb, kb]); v(:, :, 3) = a(isliceloc, :, :)
      v(:, :, 4) = thlm(isliceloc, jb:je, kb:ke); v(:, :, 5) = qtm(isliceloc, jb:je, kb:ke)
      if (ltempeq) then
        call udc_pull3(UDC_THLM, a(isliceloc:isliceloc, :, :), [isliceloc, jb, kb]); v(:, :, 4) = a(isliceloc, :, :)
      end if
      if (lmoist) then
        call udc_pull3(UDC_QTM, a(isliceloc:isliceloc, :, :), [isliceloc, jb, kb]); v(:, :, 5) = a(isliceloc, :, :)
      end if
      call writestat_nc(ncidsl(2), 1, tvar, (/timee/), nrecsl(2), .true.)
      call writestat_nc(ncidsl(2), 5, ncstatsl(:, :, 2), v, nrecsl(2), jmax, nk)
      deallocate (v, a)
    end if
    if (slice_on(3)) then
      allocate (v(imax, nk, 5), a(ib:ie, jsliceloc:jsliceloc + 1, kb:ke))
      call udc_pull3(UDC_UM, a(:, jsliceloc:jsliceloc, :), [ib, jsliceloc, kb]); v(:, :, 1) = a(:, jsliceloc, :)
      call udc_pull3(UDC_VM, a, [ib, jsliceloc, kb]); v(:, :, 2) = 0.5*(a(:, jsliceloc, :) + a(:, jsliceloc + 1, :))
      call udc_pull3(UDC_WM, a(:, jsliceloc:jsliceloc, :), [ib, jsliceloc, kb]); v(:, :, 3) = a(:, jsliceloc, :)
      v(:, :, 4) = thlm(ib:ie, jsliceloc, kb:ke); v(:, :, 5) = qtm(ib:ie, jsliceloc, kb:ke)
      if (ltempeq) then
        call udc_pull3(UDC_THLM, a(:, jsliceloc:jsliceloc, :), [ib, jsliceloc, kb]); v(:, :, 4) = a(:, jsliceloc, :)
      end if
      if (lmoist) then
        call udc_pull3(UDC_QTM, a(:, jsliceloc:jsliceloc, :), [ib, jsliceloc, kb]); v(:, :, 5) = a(:, jsliceloc, :)
      end if
      call writestat_nc(ncidsl(3), 1, tvar, (/timee/), nrecsl(3), .true.)
      call writestat_nc(ncidsl(3), 5, ncstatsl(:, :, 3), v, nrecsl(3), imax, nk)
      deallocate (v, a)
    end if
  end subroutine write_slices

  !> the last sample's own profiles / x-z fields (every tsample, :1294-1349)
  subroutine write_xy
    use modglobal, only: kb, ke, timee
    use modmpi, only: myid
    use modstat_nc, only: writestat_nc, writestat_1D_nc
    use udc_iface, only: udc_h, udc_check
    real, allocatable :: tab(:, :)
    allocate (tab(ke - kb + 1, nstatxy))
    call udc_check(udc_stats_xy(udc_h, tab), 'udc_stats_xy')
    if (myid /= 0) return
    call writestat_nc(ncidxy, 1, tvar, (/timee/), nrecxy, .true.)
    call writestat_1D_nc(ncidxy, nstatxy, ncstatxy, tab, nrecxy, ke - kb + 1)
  end subroutine write_xy

  subroutine write_y
    use modglobal, only: kb, ke, imax, timee
    use modmpi, only: myid
    use modstat_nc, only: writestat_nc
    use udc_iface, only: udc_h, udc_check
    real, allocatable :: tab(:, :, :)
    allocate (tab(imax, ke - kb + 1, nstaty))
    call udc_check(udc_stats_y(udc_h, tab), 'udc_stats_y')
    if (myid /= 0) return
    call writestat_nc(ncidy, 1, tvar, (/timee/), nrecy, .true.)
    call writestat_nc(ncidy, nstaty, ncstaty, tab, nrecy, imax, ke - kb + 1)
  end subroutine write_y

  subroutine write_xyt
    use modglobal, only: kb, ke, timee
    use modmpi, only: myid
    use modstat_nc, only: writestat_nc, writestat_1D_nc
    use udc_iface, only: udc_h, udc_check
    real, allocatable :: tab(:, :)
    allocate (tab(ke - kb + 1, nstatxyt))
    call udc_check(udc_stats_xyt(udc_h, tab), 'udc_stats_xyt')      ! (every rank: the slab sums are all-reduced)
    if (myid /= 0) return
    call writestat_nc(ncidxyt, 1, tvar, (/timee/), nrecxyt, .true.)
    call writestat_1D_nc(ncidxyt, nstatxyt, ncstatxyt, tab, nrecxyt, ke - kb + 1)
  end subroutine write_xyt

  subroutine write_yt
    use modglobal, only: kb, ke, imax, timee, ltempeq, lmoist, nsv
    use modmpi, only: myid
    use modstat_nc, only: writestat_nc
    use udc_iface, only: udc_h, udc_check
    real, allocatable :: tab(:, :, :)
    integer :: n
    allocate (tab(imax, ke - kb + 1, nstatyt))
    call udc_check(udc_stats_yt(udc_h, tab), 'udc_stats_yt')
    if (myid /= 0) return
    ! fields the deck does not carry stay at the zeros the reference's never-touched arrays hold: temperature rows 4, 10, 16, 23, 30,
    ! moisture one further, scalar n another n further
    if (.not. ltempeq) tab(:, :, [4, 10, 16, 23, 30]) = 0.
    if (.not. lmoist) tab(:, :, [5, 11, 17, 24, 31]) = 0.
    do n = nsv + 1, 3
      tab(:, :, [5, 11, 17, 24, 31] + n) = 0.
    end do
    call writestat_nc(ncidyt, 1, tvar, (/timee/), nrecyt, .true.)
    call writestat_nc(ncidyt, nstatyt, ncstatyt, tab, nrecyt, imax, ke - kb + 1)
  end subroutine write_yt

  !> tdump's 32 variables from the accumulators: means as they are, fluxes and variances as <ab> - <a><b> (:1557-1600)
  subroutine write_t
    use modglobal, only: ib, ie, jb, je, kb, ke, imax, jmax, nsv, ltempeq, lmoist, timee
    use modstat_nc, only: writestat_nc
    real, allocatable :: v(:, :, :, :), a(:, :, :), b(:, :, :), wmt(:, :, :)
    integer :: n
    integer(c_int) :: s
    allocate (v(ib:ie, jb:je, kb:ke, nstatt), a(ib:ie, jb:je, kb:ke), b(ib:ie, jb:je, kb:ke), wmt(ib:ie, jb:je, kb:ke))
    v = 0.
    call fetch(ST_UMT, v(:, :, :, 1)); call fetch(ST_VMT, v(:, :, :, 2)); call fetch(ST_WMT, wmt); v(:, :, :, 3) = wmt
    call fetch(ST_PT, v(:, :, :, 6))
    call fetch(ST_UWTIK, v(:, :, :, 12)); call fetch(ST_UTIK, a); call fetch(ST_WTIK, b); v(:, :, :, 12) = v(:, :, :, 12) - a*b
    call fetch(ST_VWTJK, v(:, :, :, 13)); call fetch(ST_VTJK, a); call fetch(ST_WTJK, b); v(:, :, :, 13) = v(:, :, :, 13) - a*b
    call fetch(ST_UVTIJ, v(:, :, :, 14)); call fetch(ST_UTIJ, a); call fetch(ST_VTIJ, b); v(:, :, :, 14) = v(:, :, :, 14) - a*b
    call fetch(ST_UUTC, v(:, :, :, 21)); call fetch(ST_UTC, a); v(:, :, :, 21) = v(:, :, :, 21) - a*a
    call fetch(ST_VVTC, v(:, :, :, 22)); call fetch(ST_VTC, a); v(:, :, :, 22) = v(:, :, :, 22) - a*a
    call fetch(ST_WWTC, v(:, :, :, 23)); call fetch(ST_WTC, a); v(:, :, :, 23) = v(:, :, :, 23) - a*a
    v(:, :, :, 24) = 0.5*(v(:, :, :, 21) + v(:, :, :, 22) + v(:, :, :, 23))
    if (ltempeq) then
      call fetch(ST_THL, v(:, :, :, 4))
      call fetch(ST_THL + 2_c_int, v(:, :, :, 15)); call fetch(ST_THL + 1_c_int, a); v(:, :, :, 15) = v(:, :, :, 15) - wmt*a
      call fetch(ST_THL + 3_c_int, v(:, :, :, 20)); v(:, :, :, 20) = v(:, :, :, 20) - v(:, :, :, 4)*v(:, :, :, 4)
    end if
    if (lmoist) call fetch(ST_QT, v(:, :, :, 5))
    do n = 1, min(nsv, 4)
      s = ST_SV + ST_SV_STRIDE*int(n - 1, c_int)
      call fetch(s, v(:, :, :, 6 + n))
      call fetch(s + 2_c_int, v(:, :, :, 15 + n)); call fetch(s + 1_c_int, a); v(:, :, :, 15 + n) = v(:, :, :, 15 + n) - wmt*a
      call fetch(s + 3_c_int, v(:, :, :, 24 + n)); v(:, :, :, 24 + n) = v(:, :, :, 24 + n) - v(:, :, :, 6 + n)*v(:, :, :, 6 + n)
      call fetch(s + 4_c_int, v(:, :, :, 28 + n))
    end do
    call writestat_nc(ncidt, 1, tvar, (/timee/), nrect, .true.)
    call writestat_nc(ncidt, nstatt, ncstatt, v, nrect, imax, jmax, ke - kb + 1)
  end subroutine write_t

  subroutine write_mint
    use modglobal, only: ib, ie, jb, je, kb, ke, imax, jmax, ltempeq, lmoist, timee
    use modstat_nc, only: writestat_nc
    real, allocatable :: v(:, :, :, :)
    allocate (v(ib:ie, jb:je, kb:ke, nstatmint))
    v = 0.
    call fetch(ST_UMT, v(:, :, :, 1)); call fetch(ST_VMT, v(:, :, :, 2)); call fetch(ST_WMT, v(:, :, :, 3))
    if (ltempeq) call fetch(ST_THL, v(:, :, :, 4))
    if (lmoist) call fetch(ST_QT, v(:, :, :, 5))
    call fetch(ST_PT, v(:, :, :, 6))
    call writestat_nc(ncidmint, 1, tvar, (/timee/), nrecmint, .true.)
    call writestat_nc(ncidmint, nstatmint, ncstatmint, v, nrecmint, imax, jmax, ke - kb + 1)
  end subroutine write_mint

  !> (the reference's does nothing either: src/modstatsdump.f90:2148-2169)
  subroutine exitstatsdump
    use udc_iface, only: udc_timers_report
    call udc_timers_report      ! (UDC_TIMERS=1: the host-side phase clock of the run)
  end subroutine exitstatsdump

end module modstatsdump
