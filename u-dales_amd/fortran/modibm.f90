!> Drop-in replacement for the reference's module modibm (src/modibm.f90).
!! Same module name; public procedures initibm, createmasks, bottom, ibmwallfun, ibmnorm and the namelist variables
!! of src/modibm.f90:30-62.
!!   bottom (:1998-2100: floor wall function wfmneutral / wfuno for momentum and temperature, flux floors for thl, qt and
!!       the scalars, the e120 floor ghost)                                        -> udc_bottom
!!   initibm (:131): reads solid_{u,v,w,c}.txt and fluid_boundary_{u,v,w,c}.txt with the reference's own read_sparse_ijk
!!       (src/readinput.f90) and hands the global lists to the device (udc_set_ibm_points, udc_ibm_commit); builds the
!!       real masks mask_u .. mask_c as :150-186 does (modfielddump reads them)
!!   ibmwallfun (:1167) without facet wall functions: diffu/v/w/c_corr             -> udc_ibmwallfun
!!   ibmnorm (:697): solid                                                         -> udc_ibmnorm
!!   createmasks (:2103): the integer masks II* and their slab / column counts, on the host as in the reference
!!   facet wall functions (iwallmom = 2, 3: wallfunmom :1286; iwalltemp = 2: wallfunheat :1436, sensible part): the section
!!       tables of initibmwallfun (:273-644) are built here from facet_sections_*.txt and the reference's own initfac data
!!       (readfacetfiles stays the driver's call) and handed to the device once (udc_set_ibm_sections)
!!   facet output (lwritefac, :196-243, 1246-1282): the wall-function kernels add each section's stress / heat transfer coefficient /
!!       pressure to its facet on RK stage 3 and keep the time integrals (udc_set_ibm_facet_output, udc_ibm_facet_sample); fac.NNN.nc is
!!       written here through the reference's modstat_nc, after the substep has run (ibm_facet_output, called by the drop-in
!!       tstep_integrate: with deferred execution the substep runs there)
!! The tau_x / tau_y / thl_flux planes `bottom` leaves behind (src/modibm.f90:2015-2018, 2094-2097: what the floor added to the
!! tendencies) are filled from the device whenever the host fields are refreshed (udc_iface: udc_bottom_diagnostics).
!! Without wall functions (iwallmom = 1) the fluid-boundary points of the velocity grids are not handed over: the reference never
!! reads them then (:166-179) and diffu/v/w_corr act on no points.
module modibm
  use iso_c_binding, only: c_int, c_double
  use modibmdata
  implicit none
  save
  public :: initibm, ibmnorm, ibmwallfun, bottom, lbottom, createmasks, ibm_facet_output, &
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
  ! this rank's solid points (local indices), kept for createmasks
  integer, allocatable :: sol_u(:, :), sol_v(:, :), sol_w(:, :), sol_c(:, :)

  ! The reference's modibm has no `private` (src/modibm.f90:24-34), and its in-solver self test reads the point tables of
  ! initibm through these two types (src/tests.f90:50-54, 139, 177: this rank's count, file indices and local i, j, k).  Kept
  ! with the reference's component names for the parts that exist here; the device holds the working copies.
  type solid_info_type
    integer :: nsolpts = 0
    integer, allocatable :: solpts(:, :)          ! (nsolpts, 3): global i, j, k as read
    integer, allocatable :: solptsrank(:)         ! file indices of the points on this rank
    integer :: nsolptsrank = 0
    integer, allocatable :: solpts_loc(:, :)      ! (nsolptsrank, 3): local i, j, k
  end type solid_info_type
  type bound_info_type
    integer :: nbndpts = 0
    integer, allocatable :: bndpts(:, :)
    integer, allocatable :: bndptsrank(:)
    integer :: nbndptsrank = 0
    integer, allocatable :: bndpts_loc(:, :)
  end type bound_info_type
  type(solid_info_type) :: solid_info_u, solid_info_v, solid_info_w, solid_info_c
  type(bound_info_type) :: bound_info_u, bound_info_v, bound_info_w, bound_info_c

  type ibm_lists
    integer(c_int), allocatable :: sol(:, :), bnd(:, :)      ! (3, n): global i, j, k as in the input files
    logical :: given = .false.
  end type ibm_lists
  type(ibm_lists) :: lists(0:3)
  logical :: ibm_pending = .false.
  ! latent part of wallfunheat: per kept c-grid section, between grid_sections(3) and udc_set_ibm_wallmoist
  logical :: latent = .false.
  ! lwritefac: the facet of every kept section of a grid (grid_sections), the output file (names as src/modibm.f90:109-112)
  type fac_ids
    integer(c_int), allocatable :: fac(:)
  end type fac_ids
  type(fac_ids) :: secfac(0:3)
  integer :: nstatfac = 7, ncidfac, nrecfac = 0
  character(80), allocatable :: ncstatfac(:, :)
  character(80) :: facname = 'fac.xxx.nc'
  character(80), dimension(1, 4) :: tncstatfac
  integer(c_int), allocatable :: wm_lgr(:)
  real(c_double), allocatable :: wm_q(:), wm_hurel(:), wm_resc(:), wm_ress(:)

contains

  subroutine initibm
    use modglobal, only: libm, ib, ie, ih, jb, je, jh, kb, ke, kh, ltempeq, lmoist, lbuoyancy, nsv, BCbotm, BCbotT, BCbotq, BCbots, &
                         iwallmom, iwalltemp, iwallmoist, lwritefac
    use modibmdata, only: bctfxm, bctfxp, bctfym, bctfyp, bctfz, bcqfxm, bcqfxp, bcqfym, bcqfyp, bcqfz
    use modsurfdata, only: z0
    use udc_iface, only: udc_set_floor
    logical :: need_c
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
    call udc_set_floor(lbottom, real(z0, c_double))
    allocate (mask_u(ib - ih:ie + ih, jb - jh:je + jh, kb - kh:ke + kh)); mask_u = 1.
    allocate (mask_v(ib - ih:ie + ih, jb - jh:je + jh, kb - kh:ke + kh)); mask_v = 1.
    allocate (mask_w(ib - ih:ie + ih, jb - jh:je + jh, kb - kh:ke + kh)); mask_w = 1.
    allocate (mask_c(ib - ih:ie + ih, jb - jh:je + jh, kb - kh:ke + kh)); mask_c = 1.
    if (.not. libm) return
    if (iwallmom < 1 .or. iwallmom > 3 .or. (iwallmom == 2 .and. .not. ltempeq)) then
      write (0, *) 'ERROR: libudcore modibm: iwallmom must be 1, 2 (with ltempeq: the stability functions read the air temperature) or 3'
      stop 1
    end if
    ! wallfunheat (src/modibm.f90:1436): sensible part prescribed (iwalltemp = 1) or from the facet temperatures (2); latent part on
    ! the vegetated facets prescribed (iwallmoist = 1) or from the facets' humidity (2, with the heat transfer coefficient of iwalltemp = 2)
    if ((ltempeq .and. iwalltemp /= 1 .and. iwalltemp /= 2) .or. (lmoist .and. iwallmoist /= 1 .and. iwallmoist /= 2)) then
      write (0, *) 'ERROR: libudcore modibm: iwalltemp / iwallmoist must be 1 or 2'
      stop 1
    end if
    if (lmoist .and. iwallmoist == 2 .and. .not. (ltempeq .and. iwalltemp == 2)) then
      write (0, *) 'ERROR: libudcore modibm: iwallmoist = 2 takes its aerodynamic resistance from the heat transfer coefficient of iwalltemp = 2'
      stop 1
    end if
    if (lmoist .and. .not. ltempeq .and. any((/bcqfxm, bcqfxp, bcqfym, bcqfyp, bcqfz/) /= 0.)) then
      write (0, *) 'ERROR: libudcore modibm: wall moisture fluxes: wallfunheat runs with the temperature equation (ltempeq)'
      stop 1
    end if
    need_c = nsv > 0 .or. ltempeq .or. lmoist .or. lwritefac          ! src/modibm.f90:181
    mask_w(:, :, kb) = 0.                     ! src/modibm.f90:154-157
    mask_u(:, :, kb - kh) = 0.; mask_v(:, :, kb - kh) = 0.; mask_w(:, :, kb - kh) = 0.; mask_c(:, :, kb - kh) = 0.
    call grid_lists(0, 'solid_u.txt', nsolpts_u, 'fluid_boundary_u.txt', nbndpts_u, mask_u, sol_u, solid_info_u, bound_info_u)
    call grid_lists(1, 'solid_v.txt', nsolpts_v, 'fluid_boundary_v.txt', nbndpts_v, mask_v, sol_v, solid_info_v, bound_info_v)
    call grid_lists(2, 'solid_w.txt', nsolpts_w, 'fluid_boundary_w.txt', nbndpts_w, mask_w, sol_w, solid_info_w, bound_info_w)
    if (need_c) call grid_lists(3, 'solid_c.txt', nsolpts_c, 'fluid_boundary_c.txt', nbndpts_c, mask_c, sol_c, solid_info_c, bound_info_c)
    ibm_pending = .true.
    if (lwritefac) call init_facet_file
  end subroutine initibm

  !> fac.NNN.nc as the reference defines it (src/modibm.f90:228-246)
  subroutine init_facet_file
    use modglobal, only: cexpnr, nfcts
    use modmpi, only: myid
    use modstat_nc, only: open_nc, define_nc, ncinfo, writestat_dims_nc
    facname(5:7) = cexpnr
    allocate (ncstatfac(nstatfac, 4))
    call ncinfo(tncstatfac(1, :), 't', 'Time', 's', 'time')
    call ncinfo(ncstatfac(1, :), 'tau_x', 'tau_x', 'm^2/s^2', 'ft')
    call ncinfo(ncstatfac(2, :), 'tau_y', 'tau_y', 'm^2/s^2', 'ft')
    call ncinfo(ncstatfac(3, :), 'tau_z', 'tau_z', 'm^2/s^2', 'ft')
    call ncinfo(ncstatfac(4, :), 'pres', 'pressure', 'm^2/s^2', 'ft')
    call ncinfo(ncstatfac(5, :), 'htc', 'heat transfer coefficient', '', 'ft')
    call ncinfo(ncstatfac(6, :), 'cth', 'heat transfer coefficient (Ivo)', '', 'ft')
    call ncinfo(ncstatfac(7, :), 'pres_flc', 'pressure fluctuation', '', 'ft')
    if (myid == 0) then
      call open_nc(facname, ncidfac, nrecfac, nfcts=nfcts)
      if (nrecfac == 0) then
        call define_nc(ncidfac, 1, tncstatfac)
        call writestat_dims_nc(ncidfac)
      end if
      call define_nc(ncidfac, nstatfac, ncstatfac)
    end if
  end subroutine init_facet_file

  !> fac.NNN.nc's record (src/modibm.f90:1255-1280), once the substep whose ibmwallfun sampled the facets has run: called by the
  !! drop-in tstep_integrate.  Rank 0 alone, like the reference: every rank's device holds the all-reduced integrals.
  subroutine ibm_facet_output
    use modglobal, only: libm, lwritefac, rk3step, timee, tnextfac, tfac, dtfac, nfcts
    use modmpi, only: myid
    use modstat_nc, only: writestat_nc, writestat_1D_nc
    use udc_iface
    real(c_double), allocatable :: acc(:, :)
    real, allocatable :: varsfac(:, :)
    if (.not. (libm .and. lwritefac) .or. rk3step /= 3 .or. myid /= 0) return
    if (timee < tnextfac) return
    allocate (acc(nfcts, 7), varsfac(nfcts, nstatfac))
    call udc_check(udc_ibm_facet_get(udc_h, acc, 1_c_int), 'udc_ibm_facet_get')
    tfac = timee - tfac
    varsfac(:, 1) = acc(:, 1)/tfac
    varsfac(:, 2) = acc(:, 2)/tfac
    varsfac(:, 3) = acc(:, 3)/tfac
    varsfac(:, 4) = acc(:, 4)/tfac
    varsfac(:, 5) = acc(:, 6)/tfac
    varsfac(:, 6) = acc(:, 7)/tfac
    varsfac(:, 7) = acc(:, 5)/tfac - (acc(:, 4)/dtfac*acc(:, 4)/tfac)
    call writestat_nc(ncidfac, 1, tncstatfac, (/timee/), nrecfac, .true.)
    call writestat_1D_nc(ncidfac, nstatfac, ncstatfac, varsfac, nrecfac, nfcts)
    tfac = timee
    tnextfac = NINT((timee + dtfac))*1.0
  end subroutine ibm_facet_output

  !> one grid: read both lists (read_sparse_ijk: this rank's points with local indices + the global list), zero the real
  !! mask at this rank's solid points, keep the global lists for the device (udc_ensure runs after all init* routines)
  subroutine grid_lists(grid, fsol, nsol, fbnd, nbnd, mask, sol_loc, sinfo, binfo)
    use readinput, only: read_sparse_ijk
    integer, intent(in) :: grid, nsol, nbnd
    character(*), intent(in) :: fsol, fbnd
    real, intent(inout) :: mask(:, :, :)
    integer, allocatable, intent(out) :: sol_loc(:, :)
    type(solid_info_type), intent(out) :: sinfo
    type(bound_info_type), intent(out) :: binfo
    integer, allocatable :: ids(:), loc(:, :), glob(:, :)
    integer :: n, nloc, lb(3)
    call read_sparse_ijk(fsol, nsol, nloc, ids, sol_loc, nskip=1, pts_glob_out=glob)
    lb = lbound_of_mask()
    do n = 1, nloc
      mask(sol_loc(n, 1) - lb(1) + 1, sol_loc(n, 2) - lb(2) + 1, sol_loc(n, 3) - lb(3) + 1) = 0.
    end do
    allocate (lists(grid)%sol(3, nsol))
    lists(grid)%sol = transpose(glob)
    sinfo%nsolpts = nsol; sinfo%nsolptsrank = nloc
    sinfo%solpts = glob; sinfo%solptsrank = ids; sinfo%solpts_loc = sol_loc
    deallocate (glob, ids)
    call read_sparse_ijk(fbnd, nbnd, nloc, ids, loc, nskip=1, pts_glob_out=glob)
    allocate (lists(grid)%bnd(3, nbnd))
    lists(grid)%bnd = transpose(glob)
    lists(grid)%given = .true.
    binfo%nbndpts = nbnd; binfo%nbndptsrank = nloc
    binfo%bndpts = glob; binfo%bndptsrank = ids; binfo%bndpts_loc = loc
  contains
    function lbound_of_mask() result(l)
      use modglobal, only: ib, ih, jb, jh, kb, kh
      integer :: l(3)
      l = (/ib - ih, jb - jh, kb - kh/)
    end function lbound_of_mask
  end subroutine grid_lists

  !> hand the lists to the device once its handle exists (first call of ibmwallfun / ibmnorm, or of anything else)
  subroutine ibm_to_device
    use udc_iface
    use modmpi, only : nprocx, nprocy
    use modglobal, only : lconservativeibm, iwallmom
    integer :: q, nbnd
    integer(c_int) :: none(3)
    if (.not. ibm_pending) return
    call udc_ensure
    none = 0
    ! initibm's masks get their ghost cells from exchange_halo_z alone, which wraps a direction only when this run splits
    ! it over ranks (periodic_bc, src/modstartup.f90:662-672): the device looks neighbours up the same way
    call udc_check(udc_set_ibm_mask_wrap(udc_h, merge(1_c_int, 0_c_int, nprocx > 1), merge(1_c_int, 0_c_int, nprocy > 1)), &
                   'udc_set_ibm_mask_wrap')
    call udc_check(udc_set_ibm_conservative(udc_h, merge(1_c_int, 0_c_int, lconservativeibm)), 'udc_set_ibm_conservative')
    do q = 0, 3
      if (lists(q)%given) then
        ! without wall functions the reference never reads the fluid-boundary points of the velocity grids (initibmwallfun for
        ! u, v, w sits under `iwallmom > 1`, src/modibm.f90:166-179): diffu/v/w_corr then loop over no points
        nbnd = size(lists(q)%bnd, 2)
        if (iwallmom == 1 .and. q < 3) nbnd = 0
        call udc_check(udc_set_ibm_points(udc_h, int(q, c_int), lists(q)%sol, int(size(lists(q)%sol, 2), c_int), &
                                          lists(q)%bnd, int(nbnd, c_int)), 'udc_set_ibm_points')
      end if
    end do
    call udc_check(udc_ibm_commit(udc_h), 'udc_ibm_commit')
    call sections_to_device
    ibm_pending = .false.
  end subroutine ibm_to_device

  !> facet wall functions (wallfunmom :1286, wallfunheat :1436): level coordinates and the facet sections of each grid
  subroutine sections_to_device
    use udc_iface
    use modglobal, only: iwallmom, iwalltemp, iwallmoist, ltempeq, lmoist, prandtlturb, zf, zh, kb, ke, kh, lwritefac
    use modibmdata, only: bctfxm, bctfxp, bctfym, bctfyp, bctfz, bcqfxm, bcqfxp, bcqfym, bcqfyp, bcqfz
    logical :: heat
    latent = lmoist .and. (iwallmoist == 2 .or. (iwallmoist == 1 .and. any((/bcqfxm, bcqfxp, bcqfym, bcqfyp, bcqfz/) /= 0.)))
    heat = ltempeq .and. (iwalltemp == 2 .or. (iwalltemp == 1 .and. any((/bctfxm, bctfxp, bctfym, bctfyp, bctfz/) /= 0.)) .or. latent)
    if (iwallmom <= 1 .and. .not. heat) then
      if (lwritefac) call facets_to_device
      return
    end if
    call udc_check(udc_set_ibm_wallfun(udc_h, int(iwallmom, c_int), real(prandtlturb, c_double), real(zf(kb:ke + kh), c_double), &
                                       real(zh(kb:ke + kh), c_double)), 'udc_set_ibm_wallfun')
    if (iwallmom > 1) then
      call grid_sections(0, 'facet_sections_u.txt', nfctsecs_u)
      call grid_sections(1, 'facet_sections_v.txt', nfctsecs_v)
      call grid_sections(2, 'facet_sections_w.txt', nfctsecs_w)
    end if
    if (heat) then
      call grid_sections(3, 'facet_sections_c.txt', nfctsecs_c)
      call udc_check(udc_set_ibm_wallheat(udc_h, int(iwalltemp, c_int)), 'udc_set_ibm_wallheat')
      if (latent) then      ! the tables grid_sections(3) left: vegetated?, prescribed flux / saturation humidity, humidity, resistances
        call udc_check(udc_set_ibm_wallmoist(udc_h, int(iwallmoist, c_int), int(size(wm_lgr), c_int), wm_lgr, wm_q, wm_hurel, wm_resc, wm_ress), &
                       'udc_set_ibm_wallmoist')
        deallocate (wm_lgr, wm_q, wm_hurel, wm_resc, wm_ress)
      end if
    end if
    if (lwritefac) call facets_to_device
  end subroutine sections_to_device

  !> lwritefac: the facet of every section the device holds (grid_sections kept them), the facets' areas (initfac's faca, read from
  !! facetarea.inp when lwritefac) and ALL sections of the c grid for the pressure (wallfunheat adds pres0 of the boundary cell before it
  !! looks at lskipsec, src/modibm.f90:1475-1478)
  subroutine facets_to_device
    use udc_iface
    use modglobal, only: ifinput, nfcts, itot, jtot
    use initfac, only: faca
    integer(c_int) :: nsec(0:3), none(1)
    integer(c_int), allocatable :: pcell(:, :), pfac(:)
    real(c_double), allocatable :: parea(:)
    integer :: q, n, m, fac, bid
    real :: a, dst
    character(80) :: chmess
    none = 0
    do q = 0, 3
      if (.not. allocated(secfac(q)%fac)) allocate (secfac(q)%fac(0))
      nsec(q) = size(secfac(q)%fac)
    end do
    allocate (pcell(3, max(nfctsecs_c, 1)), pfac(max(nfctsecs_c, 1)), parea(max(nfctsecs_c, 1)))
    m = 0
    if (nfctsecs_c > 0 .and. lists(3)%given) then
      open (ifinput, file='facet_sections_c.txt')
      read (ifinput, '(a80)') chmess
      do n = 1, nfctsecs_c
        read (ifinput, *) fac, a, bid, dst
        if (lists(3)%bnd(1, bid) < 1 .or. lists(3)%bnd(1, bid) > itot .or. lists(3)%bnd(2, bid) < 1 .or. lists(3)%bnd(2, bid) > jtot) cycle
        m = m + 1
        pcell(:, m) = lists(3)%bnd(:, bid); pfac(m) = fac; parea(m) = a
      end do
      close (ifinput)
    end if
    call udc_check(udc_set_ibm_facet_output(udc_h, int(nfcts, c_int), real(faca(1:nfcts), c_double), nsec, secfac(0)%fac, secfac(1)%fac, &
                                            secfac(2)%fac, secfac(3)%fac, int(m, c_int), pcell, parea, pfac), 'udc_set_ibm_facet_output')
  end subroutine facets_to_device

  !> One grid's section table as initibmwallfun (src/modibm.f90:273-644) leaves it: which sections act (:366-373), where the
  !! velocity is taken -- the boundary cell, or a reconstruction point along the facet normal when log(dist / z0) <= 1
  !! (:375-424) -- and the cells around that point on the four grids (:426-482).  Facet data from the reference's initfac.
  subroutine grid_sections(grid, fname, nsec)
    use udc_iface
    use modglobal, only: ifinput, ib, itot, ih, jb, jtot, jh, kb, ke, kh, xf, xh, yf, yh, zf, zh, dx, dy, dzf, eps1, iwalltemp, iwallmoist
    use modibmdata, only: bctfxm, bctfxp, bctfyp, bctfz, bcqfxm, bcqfxp, bcqfym, bcqfyp, bcqfz
    use initfac, only: facnorm, facz0, facz0h, facT, faclGR, facqsat, fachurel, facf
    use decomp_2d, only: zstart, zend
    integer, intent(in) :: grid, nsec
    character(*), intent(in) :: fname
    integer(c_int), allocatable :: cell(:, :), comprec(:), recids(:, :, :), t_lgr(:), t_fac(:)
    real(c_double), allocatable :: area(:), dist(:), norm(:, :), z0(:), z0h(:), ts(:), recpt(:, :), tmask(:, :), t_q(:), t_hurel(:), t_resc(:), t_ress(:)
    integer :: n, m, fac, bid, dalign, q, pos, i, j, k, di, dj, dk, li, lj
    real :: a, dst, xc, yc, zc, p0(3), p1(3), nrm(3), inter(6, 3), idist(6), planes(6, 3), pn(6, 3)
    integer :: chk(6)
    logical :: skip
    character(80) :: chmess
    allocate (cell(3, nsec), comprec(nsec), recids(3, 4, nsec), area(nsec), dist(nsec), norm(3, nsec), z0(nsec), z0h(nsec), ts(nsec), &
              recpt(3, nsec), tmask(2, nsec))
    recids = 1; recpt = 0.; tmask = 1.
    allocate (t_fac(nsec)); t_fac = 0
    if (grid == 3 .and. latent) then
      allocate (t_lgr(nsec), t_q(nsec), t_hurel(nsec), t_resc(nsec), t_ress(nsec))
      t_lgr = 0; t_q = 0.; t_hurel = 0.; t_resc = 0.; t_ress = 0.
    end if
    dalign = merge(0, grid + 1, grid == 3)
    di = merge(1, 0, grid == 0); dj = merge(1, 0, grid == 1); dk = merge(1, 0, grid == 2)
    m = 0
    if (nsec > 0) then
      open (ifinput, file=fname)
      read (ifinput, '(a80)') chmess
      do n = 1, nsec
        read (ifinput, *) fac, a, bid, dst
        nrm = facnorm(fac, :)
        if ((dalign /= 0 .and. dalign == alignment(nrm)) .or. facz0(fac) < eps1) cycle          ! :366-373
        i = lists(grid)%bnd(1, bid); j = lists(grid)%bnd(2, bid); k = lists(grid)%bnd(3, bid)
        ! (a section of a boundary point that no rank owns acts nowhere: lfctsecsrank, src/modibm.f90:340-350; the lists of the
        !  reference's tests/cases/526 reach beyond the domain)
        if (i < 1 .or. i > itot .or. j < 1 .or. j > jtot) cycle
        m = m + 1
        cell(:, m) = (/i, j, k/); area(m) = a; dist(m) = dst; norm(:, m) = nrm; t_fac(m) = fac
        z0(m) = facz0(fac); z0h(m) = facz0h(fac)
        ts(m) = 0.
        if (allocated(facT)) ts(m) = facT(fac, 1)
        if (grid == 3 .and. iwalltemp == 1) then      ! the prescribed flux of the facet's direction (:1508-1524) rides in this slot
          select case (alignment(nrm))
          case (1); ts(m) = bctfxp
          case (-1); ts(m) = bctfxm
          case (2); ts(m) = bctfyp
          case (-2); ts(m) = bctfxm      ! (the reference's own assignment, :1518)
          case (3); ts(m) = bctfz
          case default
            write (0, *) 'ERROR: libudcore modibm: iwalltemp = 1: the reference defines the wall heat flux for facets facing +-x, +-y, +z only'
            stop 1
          end select
        end if
        if (grid == 3 .and. latent) then      ! latent part (:1556-1600): the vegetated facets only
          t_lgr(m) = merge(1, 0, faclGR(fac)); t_q(m) = 0.; t_hurel(m) = 0.; t_resc(m) = 0.; t_ress(m) = 0.
          if (faclGR(fac) .and. iwallmoist == 1) then
            select case (alignment(nrm))
            case (1); t_q(m) = bcqfxp
            case (-1); t_q(m) = bcqfxm
            case (2); t_q(m) = bcqfyp
            case (-2); t_q(m) = bcqfym
            case (3); t_q(m) = bcqfz
            case default
              write (0, *) 'ERROR: libudcore modibm: iwallmoist = 1: the reference defines the wall moisture flux for facets facing +-x, +-y, +z only'
              stop 1
            end select
          else if (iwallmoist == 2) then
            t_q(m) = facqsat(fac); t_hurel(m) = fachurel(fac); t_resc(m) = facf(fac, 4); t_ress(m) = facf(fac, 5)
          end if
        end if
        comprec(m) = 1
        if (.not. (log(dst/facz0(fac)) > 1. .or. lnorec)) then                                       ! :375-378
          comprec(m) = 0
          select case (grid)
          case (0); xc = xh(i); yc = yf(j); zc = zf(k)
          case (1); xc = xf(i); yc = yh(j); zc = zf(k)
          case (2); xc = xf(i); yc = yf(j); zc = zh(k)
          case default; xc = xf(i); yc = yf(j); zc = zf(k)
          end select
          p0 = (/xc, yc, zc/)
          p1 = p0 + nrm*sqrt(3.)*(dx*dy*dzf(1))**(1./3.)
          pn = 0.; pn(1:2, 1) = 1.; pn(3:4, 2) = 1.; pn(5:6, 3) = 1.
          planes(1, :) = (/xc - dx/2., yc, zc/); planes(2, :) = (/xc + dx/2., yc, zc/)
          planes(3, :) = (/xc, yc - dy/2., zc/); planes(4, :) = (/xc, yc + dy/2., zc/)
          planes(5, :) = (/xc, yc, zc - dzf(1)/2./); planes(6, :) = (/xc, yc, zc + dzf(1)/2./)
          pos = 0
          do q = 1, 6
            call plane_line(pn(q, :), planes(q, :), p0, p1, inter(q, :), chk(q), idist(q))
            if (chk(q) == 1) then
              if (pos == 0) then
                pos = q
              else if (idist(q) < idist(pos)) then
                pos = q
              end if
            end if
          end do
          if (pos == 0) then
            write (0, *) 'ERROR: no intersection found'
            stop 1
          end if
          recpt(:, m) = inter(pos, :)
          recids(1, 1, m) = lastle(xh, recpt(1, m)); recids(2, 1, m) = lastle(yf, recpt(2, m)); recids(3, 1, m) = lastle(zf, recpt(3, m))
          recids(1, 2, m) = lastle(xf, recpt(1, m)); recids(2, 2, m) = lastle(yh, recpt(2, m)); recids(3, 2, m) = lastle(zf, recpt(3, m))
          recids(1, 3, m) = lastle(xf, recpt(1, m)); recids(2, 3, m) = lastle(yf, recpt(2, m)); recids(3, 3, m) = lastle(zh, recpt(3, m))
          recids(1, 4, m) = lastle(xf, recpt(1, m)); recids(2, 4, m) = lastle(yf, recpt(2, m)); recids(3, 4, m) = lastle(zf, recpt(3, m))
          skip = .false.
          do q = 1, 4                                                                                  ! :447-482
            if (recids(1, q, m) < ib .or. recids(1, q, m) + 1 > itot + ih .or. recids(2, q, m) < jb .or. recids(2, q, m) + 1 > jtot + jh .or. &
                recids(3, q, m) < kb .or. recids(3, q, m) + 1 > ke + kh) skip = .true.
          end do
          if (skip) then
            m = m - 1
            cycle
          end if
        end if
        ! interp_temperature_* (:1794-1830) reads mask_c at the cell and its lower neighbour along the grid's direction
        if (grid < 3 .and. allocated(mask_c) .and. i >= zstart(1) .and. i <= zend(1) .and. j >= zstart(2) .and. j <= zend(2)) then
          li = i - zstart(1) + 1; lj = j - zstart(2) + 1
          tmask(1, m) = mask_c(li, lj, k); tmask(2, m) = mask_c(li - di, lj - dj, k - dk)
        end if
      end do
      close (ifinput)
    end if
    call udc_check(udc_set_ibm_sections(udc_h, int(grid, c_int), int(m, c_int), cell, area, dist, norm, z0, z0h, ts, comprec, recpt, recids, &
                                        tmask), 'udc_set_ibm_sections')
    if (allocated(secfac(grid)%fac)) deallocate (secfac(grid)%fac)
    allocate (secfac(grid)%fac(m)); secfac(grid)%fac = t_fac(1:m)
    if (grid == 3 .and. latent) then
      allocate (wm_lgr(m), wm_q(m), wm_hurel(m), wm_resc(m), wm_ress(m))
      wm_lgr = t_lgr(1:m); wm_q = t_q(1:m); wm_hurel = t_hurel(1:m); wm_resc = t_resc(1:m); wm_ress = t_ress(1:m)
    end if
  contains
    !> findloc(x >= g, .true., 1, back = .true.): the last index of g (from ib, jb or kb on) with g <= x, 0 when there is none
    integer function lastle(g, x)
      real, intent(in) :: g(:), x
      integer :: t
      lastle = 0
      do t = 1, size(g)
        if (x >= g(t)) lastle = t
      end do
    end function lastle
  end subroutine grid_sections

  !> src/modibm.f90:1683-1706
  integer function alignment(n)
    use modglobal, only: eps1
    real, intent(in) :: n(3)
    integer :: d
    real :: e(3)
    alignment = 0
    do d = 1, 3
      e = 0.; e(d) = 1.
      if (all(abs(n - e) < eps1)) alignment = d
      if (all(abs(n + e) < eps1)) alignment = -d
    end do
  end function alignment

  !> src/modibm.f90:647-694
  subroutine plane_line(norm, V0, P0, P1, I, check, dist)
    use modglobal, only: eps1
    real, intent(in) :: norm(3), V0(3), P0(3), P1(3)
    real, intent(out) :: I(3), dist
    integer, intent(out) :: check
    real :: u(3), w(3), D, N, sI
    I = 0.; dist = 0.
    w = P0 - V0; u = P1 - P0
    D = dot_product(norm, u); N = -dot_product(norm, w)
    if (abs(D) < eps1) then
      check = merge(2, 0, abs(N) < eps1)
      return
    end if
    sI = N/D
    I = P0 + sI*u
    dist = norm2(I - P0)
    check = merge(3, 1, (sI < 0.) .or. (sI > 1.))
  end subroutine plane_line

  !> integer masks and their counts (src/modibm.f90:2103-2234)
  subroutine createmasks
    use mpi
    use modglobal, only: libm, ib, ie, jb, je, kb, ke, khc, jtot, rslabs
    use modfields, only: IIc, IIu, IIv, IIw, IIuw, IIvw, IIuv, IIcs, IIus, IIvs, IIws, IIuws, IIvws, IIuvs, &
                         IIct, IIut, IIvt, IIwt, IIuwt
    use modmpi, only: comm3d, mpierr
    integer :: i, j, k, n
    integer, allocatable :: loc(:), tot(:), cl(:, :), ct(:, :)
    IIc = 1; IIu = 1; IIv = 1; IIw = 1; IIuw = 1; IIvw = 1; IIuv = 1
    if (.not. libm) then
      IIcs = nint(rslabs); IIus = nint(rslabs); IIvs = nint(rslabs); IIws = nint(rslabs)
      IIuws = nint(rslabs); IIvws = nint(rslabs); IIuvs = nint(rslabs)
      IIct = jtot; IIut = jtot; IIvt = jtot; IIwt = jtot; IIuwt = jtot
      return
    end if
    if (allocated(sol_u)) then
      do n = 1, size(sol_u, 1)
        IIu(sol_u(n, 1), sol_u(n, 2), sol_u(n, 3)) = 0
      end do
    end if
    if (allocated(sol_v)) then
      do n = 1, size(sol_v, 1)
        IIv(sol_v(n, 1), sol_v(n, 2), sol_v(n, 3)) = 0
      end do
    end if
    if (allocated(sol_w)) then
      do n = 1, size(sol_w, 1)
        IIw(sol_w(n, 1), sol_w(n, 2), sol_w(n, 3)) = 0
      end do
    end if
    if (allocated(sol_c)) then
      do n = 1, size(sol_c, 1)
        IIc(sol_c(n, 1), sol_c(n, 2), sol_c(n, 3)) = 0
      end do
    end if
    IIw(:, :, kb) = 0; IIuw(:, :, kb) = 0; IIvw(:, :, kb) = 0
    ! edge masks: an edge point counts as fluid only when the four staggered points around it do (src/modibm.f90:2182-2192) -- the
    ! two points of each component that meet at the edge.  uv edges exist on every level, uw / vw edges from kb + 1 up (below them
    ! lies the floor, set solid above).  Whole-array products over this rank's interior:
    IIuv(ib:ie, jb:je, kb:ke) = min(IIu(ib:ie, jb:je, kb:ke), IIu(ib:ie, jb - 1:je - 1, kb:ke), &
                                    IIv(ib:ie, jb:je, kb:ke), IIv(ib - 1:ie - 1, jb:je, kb:ke))
    IIuw(ib:ie, jb:je, kb + 1:ke) = min(IIu(ib:ie, jb:je, kb + 1:ke), IIu(ib:ie, jb:je, kb:ke - 1), &
                                        IIw(ib:ie, jb:je, kb + 1:ke), IIw(ib - 1:ie - 1, jb:je, kb + 1:ke))
    IIvw(ib:ie, jb:je, kb + 1:ke) = min(IIv(ib:ie, jb:je, kb + 1:ke), IIv(ib:ie, jb:je, kb:ke - 1), &
                                        IIw(ib:ie, jb:je, kb + 1:ke), IIw(ib:ie, jb - 1:je - 1, kb + 1:ke))
    allocate (loc(kb:ke + khc), tot(kb:ke + khc))
    call slabcount(IIc, IIcs); call slabcount(IIu, IIus); call slabcount(IIv, IIvs); call slabcount(IIw, IIws)
    call slabcount(IIuw, IIuws); call slabcount(IIvw, IIvws); call slabcount(IIuv, IIuvs)
    allocate (cl(ib:ie, kb:ke), ct(ib:ie, kb:ke))
    call colcount(IIw, IIwt); call colcount(IIc, IIct); call colcount(IIuw, IIuwt); call colcount(IIu, IIut); call colcount(IIv, IIvt)
  contains
    subroutine slabcount(II, IIs)
      integer, intent(in) :: II(:, :, :)
      integer, intent(inout) :: IIs(kb:)
      integer :: kk
      ! (II arrays: (ib-ihc:ie+ihc, jb-jhc:je+jhc, kb:ke+khc), src/modfields.f90:574-580)
      do kk = kb, ke + khc
        loc(kk) = sum(II(lbi():ubi(), lbj():ubj(), kk - kb + 1))
      end do
      call MPI_ALLREDUCE(loc, tot, ke + khc - kb + 1, MPI_INTEGER, MPI_SUM, comm3d, mpierr)
      IIs(kb:ke + khc) = tot
    end subroutine slabcount
    subroutine colcount(II, IIt)
      integer, intent(in) :: II(:, :, :)
      integer, intent(inout) :: IIt(ib:, kb:)
      cl = sum(II(lbi():ubi(), lbj():ubj(), 1:ke - kb + 1), DIM=2)
      call MPI_ALLREDUCE(cl, ct, size(cl), MPI_INTEGER, MPI_SUM, comm3d, mpierr)
      IIt(ib:ie, kb:ke) = ct
    end subroutine colcount
    integer function lbi()
      use modglobal, only: ihc
      lbi = 1 + ihc
    end function lbi
    integer function ubi()
      use modglobal, only: ihc
      ubi = ie - ib + 1 + ihc
    end function ubi
    integer function lbj()
      use modglobal, only: jhc
      lbj = 1 + jhc
    end function lbj
    integer function ubj()
      use modglobal, only: jhc
      ubj = je - jb + 1 + jhc
    end function ubj
  end subroutine createmasks

  !> floor of the domain (src/modibm.f90:1998-2100)
  subroutine bottom
    use udc_iface
    use modglobal, only: ltempeq, BCbotm, ib, jb, kb
    use modfields, only: thl0
    logical, save :: air_set = .false.
    if (.not. (lbottom .or. loneeqn_dev())) return
    call udc_begin(.true.)
    if (lbottom .and. BCbotm == 2 .and. .not. ltempeq .and. .not. air_set) then
      ! temperature equation off: thl0 keeps its start-up profile for ever and wfuno still reads its first level (:2022)
      call udc_check(udc_set_floor_air_temperature(udc_h, real(thl0(ib, jb, kb), c_double)), 'udc_set_floor_air_temperature')
      air_set = .true.
    end if
    if (lbottom .and. .not. udc_bottom_diag_on) then      ! tau_x, tau_y, thl_flux: kept on the device, pulled with the fields
      call udc_check(udc_bottom_diagnostics(udc_h, 1_c_int), 'udc_bottom_diagnostics')
      udc_bottom_diag_on = .true.
    end if
    call udc_check(udc_bottom(udc_h), 'udc_bottom')
    call udc_end_tend
    if (udc_mode() == 0 .and. loneeqn_dev()) call udc_pull_vel(.true.)     ! e120, e12m floor ghosts
  end subroutine bottom

  !> immersed boundary forcing, shear part (src/modibm.f90:1167): the diffusion corrections at the fluid-boundary points
  subroutine ibmwallfun
    use modglobal, only: libm, lwritefac, rk3step, dt
    use udc_iface
    if (.not. libm) return
    call ibm_to_device
    call udc_begin(.true.)
    ! facet output: the sums of this call's sections enter the time integrals on RK stage 3 (src/modibm.f90:1246-1254, 1417, 1595)
    if (lwritefac .and. rk3step == 3) call udc_check(udc_ibm_facet_sample(udc_h, real(dt, c_double)), 'udc_ibm_facet_sample')
    call udc_check(udc_ibmwallfun(udc_h), 'udc_ibmwallfun')
    call udc_end_tend
  end subroutine ibmwallfun

  !> immersed boundary forcing, normal part (src/modibm.f90:697): solid points
  subroutine ibmnorm
    use modglobal, only: libm
    use udc_iface
    if (.not. libm) return
    call ibm_to_device
    call udc_begin(.true.)
    call udc_check(udc_ibmnorm(udc_h), 'udc_ibmnorm')
    call udc_end_tend
    if (udc_mode() <= 1) call udc_pull_vel(.true.)      ! um, vm, wm, svm were edited
  end subroutine ibmnorm

end module modibm
