! decomp_2d / decomp_2d_fft -- the interface of the 2DECOMP&FFT library as the uDALES solver uses it (SURVEY.md section 2.4:
! decomp_2d_init, decomp_info_init, alloc_x / _y / _z with halo levels, exchange_halo_z on arrays allocated WITH halos, the
! transposes, zsize / zstart / zend ..., DECOMP_2D_COMM_CART_Z), provided for the device-resident run.  The reference's submodule
! `2decomp-fft/` is empty and unpinned (`/root/reference/.gitmodules`), so whoever builds the driver has to supply these symbols; this
! is the layout libudcore wants: **y-slabs only** -- x stays whole on a rank (one GPU), y is split over ALL ranks.
!
! A deck that asks for p_row x p_col pencils (`nprocx`, `nprocy` of &RUN, src/modstartup.f90:676) is run as p_row * p_col y-slabs:
! the solver takes its local extents from zsize / zstart (src/modglobal.f90:622-662), decides "am I at a domain edge" from them
! (:640-660) and wraps a direction itself where one rank holds both edges (src/modboundary.f90:95-107), so nothing of it has to know.
! The 19 of the reference's 28 shipped decks that say nprocx > 1 therefore run unedited (jtot must divide by the number of ranks) --
! refused: an x-split deck with purifiers, a driver inflow or a warm start from per-pencil files (refuse_what_slabs_misplace), and the
! per-rank output files are per slab (tdump.000.RRR...), not per pencil.
! Host-side data movement only; the device path (libudcore) does its own exchanges over RCCL.
!   exchange_halo_z : whole padded rows to the two y neighbours -- always between neighbouring slabs, from the last slab to the first
!                     only where y is periodic (BCym of the deck; read from the deck file where the deck itself did not split y) --
!                     after wrapping the x ghosts where the DECK split x periodically (the solver then expects them from here);
!   transposes      : z <-> y is an exchange of equal blocks with every rank (pairwise MPI_SENDRECV rounds; jtot and ktot divisible by
!                     the ranks), y <-> x a copy -- used by
!                     the reference's own modpois only (the CPU baseline of oracle/Makefile; the drop-in modpois never calls them).
! The all-reference MPI builds of oracle/Makefile (the CPU baseline) link this file too: the reference's own program then runs an
! x-split deck as slabs as well (tests/test_oracle_mpi.py: 2 x 1 and 2 x 2 decks against the one-rank build); rank 0 says so in a line.
module decomp_2d
  use mpi
  implicit none
  integer, parameter :: mytype = kind(0.d0)
  integer, save :: nrank = 0, nproc = 1
  integer, save :: nx_global, ny_global, nz_global
  integer, save, dimension(3) :: xstart, xend, xsize
  integer, save, dimension(3) :: ystart, yend, ysize
  integer, save, dimension(3) :: zstart, zend, zsize
  integer, save :: DECOMP_2D_COMM_CART_X = 0, DECOMP_2D_COMM_CART_Y = 0, DECOMP_2D_COMM_CART_Z = 0
  integer, save :: pcol = 1, mycol = 0, nbr_prev = 0, nbr_next = 0
  logical, save :: periodic_y = .false., wrap_x = .false.

  type DECOMP_INFO
    integer, dimension(3) :: xst, xen, xsz
    integer, dimension(3) :: yst, yen, ysz
    integer, dimension(3) :: zst, zen, zsz
    integer, dimension(3) :: xlevel = (/0, 0, 0/), ylevel = (/0, 0, 0/), zlevel = (/0, 0, 0/)
  end type DECOMP_INFO
  type(DECOMP_INFO), save :: decomp_main

  interface alloc_x
    module procedure alloc_x_real
  end interface
  interface alloc_y
    module procedure alloc_y_real
  end interface
  interface alloc_z
    module procedure alloc_z_real
  end interface
  interface transpose_x_to_y
    module procedure copy_real, copy_complex
  end interface
  interface transpose_y_to_x
    module procedure copy_real, copy_complex
  end interface
  interface transpose_z_to_y
    module procedure z_to_y_real, unsupported_complex
  end interface
  interface transpose_y_to_z
    module procedure y_to_z_real, unsupported_complex
  end interface
  interface exchange_halo_z
    module procedure exchange_halo_z_real
  end interface
  interface exchange_halo_x
    module procedure exchange_unsupported
  end interface
  interface exchange_halo_y
    module procedure exchange_unsupported
  end interface

contains

  subroutine decomp_2d_init(nx, ny, nz, p_row, p_col, periodic_bc)
    integer, intent(in) :: nx, ny, nz, p_row, p_col
    logical, dimension(3), intent(in), optional :: periodic_bc
    integer :: ierr, dims(2), coords(2)
    logical :: periods(2)
    call MPI_COMM_RANK(MPI_COMM_WORLD, nrank, ierr)
    call MPI_COMM_SIZE(MPI_COMM_WORLD, nproc, ierr)
    if (p_row*p_col /= nproc .or. mod(ny, nproc) /= 0) then
      if (nrank == 0) write (0, *) 'ERROR: decomp_2d (y-slabs): nprocx * nprocy must be the number of ranks and divide jtot:', &
        p_row, p_col, nproc, ny
      call MPI_ABORT(MPI_COMM_WORLD, 1, ierr)
    end if
    if (p_row > 1 .and. nrank == 0) write (*, '(A,I0,A,I0,A,I0,A)') ' decomp_2d: the deck asks for ', p_row, ' x ', p_col, &
      ' pencils; run as ', nproc, ' y-slabs (x stays whole on a rank)'
    if (p_row > 1) call refuse_what_slabs_misplace
    nx_global = nx; ny_global = ny; nz_global = nz
    pcol = nproc
    dims = (/1, nproc/); periods = (/.true., .true./)
    call MPI_CART_CREATE(MPI_COMM_WORLD, 2, dims, periods, .false., DECOMP_2D_COMM_CART_Z, ierr)
    call MPI_CART_COORDS(DECOMP_2D_COMM_CART_Z, nrank, 2, coords, ierr)
    mycol = coords(2)
    call MPI_CART_SHIFT(DECOMP_2D_COMM_CART_Z, 1, 1, nbr_prev, nbr_next, ierr)
    DECOMP_2D_COMM_CART_X = DECOMP_2D_COMM_CART_Z; DECOMP_2D_COMM_CART_Y = DECOMP_2D_COMM_CART_Z
    periodic_y = .false.; wrap_x = .false.
    if (present(periodic_bc)) then
      ! (periodic_bc(d) is set by the solver only where the DECK splits direction d, src/modstartup.f90:662-672)
      wrap_x = periodic_bc(1)                    ! the deck split x: the solver expects x's periodic ghosts from the exchange
      if (p_col > 1) then
        periodic_y = periodic_bc(2)              ! the deck split y: this is BCym == periodic
      else if (p_row > 1) then
        ! y is split here but not in the deck: the solver meant to wrap y itself if it is periodic (src/modboundary.f90:95-107) and
        ! cannot any more, and periodic_bc(2) is .false. whatever BCym says -- the deck has to be asked
        periodic_y = deck_bcym() == 1
      end if
    end if
    ! rows travel between neighbouring slabs whatever the lateral condition; only the wrap from the last slab to the first is y's
    ! periodicity (2DECOMP's own exchange does the same on a non-periodic Cartesian communicator)
    if (.not. periodic_y) then
      if (mycol == 0) nbr_prev = MPI_PROC_NULL
      if (mycol == nproc - 1) nbr_next = MPI_PROC_NULL
    end if
    call decomp_info_init(nx, ny, nz, decomp_main)
    xstart = decomp_main%xst; xend = decomp_main%xen; xsize = decomp_main%xsz
    ystart = decomp_main%yst; yend = decomp_main%yen; ysize = decomp_main%ysz
    zstart = decomp_main%zst; zend = decomp_main%zen; zsize = decomp_main%zsz
  end subroutine decomp_2d_init

  !> The value of `key` (lower case) in the deck named on the command line, as text; `dflt` where the deck does not set it.  (The
  !! solver's own variables live in modglobal, which uses this module.)  Aborts when the deck cannot be read: a guess would be a
  !! silent wrong answer.
  function deck_value(key, dflt) result(val)
    character(*), intent(in) :: key, dflt
    character(64) :: val
    character(256) :: fname, line
    integer :: u, ios, p, q, ierr
    val = dflt
    if (command_argument_count() < 1) then
      if (nrank == 0) write (0, *) 'ERROR: decomp_2d (y-slabs): an x-split deck is run as y-slabs and needs ', key, &
        ' from the deck, and no deck is named on the command line'
      call MPI_ABORT(MPI_COMM_WORLD, 1, ierr)
    end if
    call get_command_argument(1, fname)
    open (newunit=u, file=trim(fname), status='old', action='read', iostat=ios)
    if (ios /= 0) then
      if (nrank == 0) write (0, *) 'ERROR: decomp_2d (y-slabs): cannot read ', key, ' from ', trim(fname)
      call MPI_ABORT(MPI_COMM_WORLD, 1, ierr)
    end if
    do
      read (u, '(a)', iostat=ios) line
      if (ios /= 0) exit
      p = index(line, '!')
      if (p > 0) line(p:) = ' '
      do q = 1, len(line)
        if (line(q:q) >= 'A' .and. line(q:q) <= 'Z') line(q:q) = achar(iachar(line(q:q)) + 32)
      end do
      line = adjustl(line)
      if (line(1:len(key)) /= key) cycle
      q = verify(line(len(key) + 1:), ' ')                    ! the first character after the name must be '='
      if (q == 0) cycle
      if (line(len(key) + q:len(key) + q) /= '=') cycle
      val = adjustl(line(len(key) + q + 1:))
      p = scan(val, ' ,/')
      if (p > 0) val(p:) = ' '
    end do
    close (u)
  end function deck_value

  !> &BC BCym of the deck: 1 = periodic, the reference's default (src/modglobal.f90:151)
  integer function deck_bcym()
    integer :: ios
    read (deck_value('bcym', '1'), *, iostat=ios) deck_bcym
    if (ios /= 0) deck_bcym = 1
  end function deck_bcym

  !> What an x-split deck run as slabs cannot serve: reference modules that combine myidy (0 .. nproc-1 here) with the DECK's nprocy --
  !! the purifiers' offsets (jl = purif - myidy * jtot / nprocy, src/modpurifiers.f90), the driver inflow's rank id
  !! (mod(myidy, nprocy), src/moddriver.f90) -- and a warm start from the per-pencil restart files of a true pencil run.
  subroutine refuse_what_slabs_misplace
    character(64) :: v
    integer :: ierr, n, ios
    logical :: bad
    bad = .false.
    v = deck_value('lpurif', '.false.'); bad = bad .or. index(v, 't') > 0
    v = deck_value('lwarmstart', '.false.'); bad = bad .or. index(v, 't') > 0
    v = deck_value('idriver', '0'); read (v, *, iostat=ios) n; bad = bad .or. (ios == 0 .and. n /= 0)
    if (bad) then
      if (nrank == 0) write (0, *) 'ERROR: decomp_2d (y-slabs): an x-split deck with purifiers, a driver inflow or a warm start from ', &
        'per-pencil restart files cannot be run as y-slabs (their rank arithmetic / file names are those of the pencils): ', &
        'set nprocx = 1 in the deck'
      call MPI_ABORT(MPI_COMM_WORLD, 1, ierr)
    end if
  end subroutine refuse_what_slabs_misplace

  subroutine decomp_2d_finalize
  end subroutine decomp_2d_finalize

  subroutine decomp_info_init(nx, ny, nz, info)
    integer, intent(in) :: nx, ny, nz
    type(DECOMP_INFO), intent(inout) :: info
    integer :: nyl, nzl
    nyl = ny/pcol; nzl = nz/pcol
    ! z-pencil: (nx, ny/P, nz);  y- and x-pencils: (nx, ny, nz/P)
    info%zst = (/1, mycol*nyl + 1, 1/); info%zsz = (/nx, nyl, nz/); info%zen = info%zst + info%zsz - 1
    info%yst = (/1, 1, mycol*nzl + 1/); info%ysz = (/nx, ny, nzl/); info%yen = info%yst + info%ysz - 1
    info%xst = info%yst; info%xsz = info%ysz; info%xen = info%yen
  end subroutine decomp_info_init

  subroutine alloc_any(var, sz, lev)
    real(mytype), allocatable, dimension(:, :, :) :: var
    integer, intent(in) :: sz(3), lev(3)
    allocate (var(1 - lev(1):sz(1) + lev(1), 1 - lev(2):sz(2) + lev(2), 1 - lev(3):sz(3) + lev(3)))
    var = 0.
  end subroutine alloc_any

  subroutine alloc_x_real(var, opt_decomp, opt_global, opt_xlevel)
    real(mytype), allocatable, dimension(:, :, :) :: var
    type(DECOMP_INFO), intent(in), optional :: opt_decomp
    logical, intent(in), optional :: opt_global
    integer, intent(in), optional :: opt_xlevel(3)
    integer :: lev(3)
    lev = decomp_main%xlevel; if (present(opt_xlevel)) lev = opt_xlevel
    call alloc_any(var, decomp_main%xsz, lev)
  end subroutine alloc_x_real

  subroutine alloc_y_real(var, opt_decomp, opt_global, opt_ylevel)
    real(mytype), allocatable, dimension(:, :, :) :: var
    type(DECOMP_INFO), intent(in), optional :: opt_decomp
    logical, intent(in), optional :: opt_global
    integer, intent(in), optional :: opt_ylevel(3)
    integer :: lev(3)
    lev = decomp_main%ylevel; if (present(opt_ylevel)) lev = opt_ylevel
    call alloc_any(var, decomp_main%ysz, lev)
  end subroutine alloc_y_real

  subroutine alloc_z_real(var, opt_decomp, opt_global, opt_zlevel)
    real(mytype), allocatable, dimension(:, :, :) :: var
    type(DECOMP_INFO), intent(in), optional :: opt_decomp
    logical, intent(in), optional :: opt_global
    integer, intent(in), optional :: opt_zlevel(3)
    integer :: lev(3)
    lev = decomp_main%zlevel; if (present(opt_zlevel)) lev = opt_zlevel
    call alloc_any(var, decomp_main%zsz, lev)
  end subroutine alloc_z_real

  subroutine copy_real(src, dst, opt_decomp)
    real(mytype), dimension(:, :, :), intent(in) :: src
    real(mytype), dimension(:, :, :), intent(out) :: dst
    type(DECOMP_INFO), intent(in), optional :: opt_decomp
    dst = src
  end subroutine copy_real

  subroutine copy_complex(src, dst, opt_decomp)
    complex(mytype), dimension(:, :, :), intent(in) :: src
    complex(mytype), dimension(:, :, :), intent(out) :: dst
    type(DECOMP_INFO), intent(in), optional :: opt_decomp
    dst = src
  end subroutine copy_complex

  subroutine unsupported_complex(src, dst, opt_decomp)
    complex(mytype), dimension(:, :, :), intent(in) :: src
    complex(mytype), dimension(:, :, :), intent(out) :: dst
    type(DECOMP_INFO), intent(in), optional :: opt_decomp
    write (0, *) 'ERROR: complex z<->y transposes are not provided by this y-slab decomp_2d'
    stop 1
  end subroutine unsupported_complex

  ! z-pencil (nx, ny/P, nz) -> y-pencil (nx, ny, nz/P): block d holds my rows, z-slab of rank d
  !> Equal blocks to every rank.  Pairwise MPI_SENDRECV rounds by default -- in round r a rank sends to mycol + r and receives from
  !! mycol - r, one pair of blocks in flight per rank --; UDC_DECOMP_ALLTOALL=1: one MPI_ALLTOALL (MPICH 3.3's shared-memory all-to-all
  !! of these 0.1 - 1 MB blocks stops scaling at 16 ranks of a 128-core host: the CPU baseline of bench.py fell from 16 to 32 ranks).
  subroutine blocks_to_all(sbuf, rbuf, blk)
    real(mytype), intent(in) :: sbuf(:)
    real(mytype), intent(out) :: rbuf(:)
    integer, intent(in) :: blk
    integer :: r, to, from, ierr, st(MPI_STATUS_SIZE), stat
    character(8) :: env
    logical, save :: first = .true., collective = .false.
    if (first) then
      call get_environment_variable('UDC_DECOMP_ALLTOALL', env, status=stat)
      collective = stat == 0 .and. trim(env) == '1'
      first = .false.
    end if
    if (collective) then
      call MPI_ALLTOALL(sbuf, blk, MPI_DOUBLE_PRECISION, rbuf, blk, MPI_DOUBLE_PRECISION, DECOMP_2D_COMM_CART_Z, ierr)
      return
    end if
    rbuf(mycol*blk + 1:(mycol + 1)*blk) = sbuf(mycol*blk + 1:(mycol + 1)*blk)
    do r = 1, pcol - 1
      to = mod(mycol + r, pcol); from = mod(mycol - r + pcol, pcol)
      call MPI_SENDRECV(sbuf(to*blk + 1), blk, MPI_DOUBLE_PRECISION, to, 7, rbuf(from*blk + 1), blk, MPI_DOUBLE_PRECISION, from, 7, &
                        DECOMP_2D_COMM_CART_Z, st, ierr)
    end do
  end subroutine blocks_to_all

  subroutine z_to_y_real(src, dst, opt_decomp)
    real(mytype), dimension(:, :, :), intent(in) :: src
    real(mytype), dimension(:, :, :), intent(out) :: dst
    type(DECOMP_INFO), intent(in), optional :: opt_decomp
    real(mytype), allocatable :: sbuf(:), rbuf(:)
    integer :: nx, nyl, nzl, d, k, j, blk, ierr, o
    nx = nx_global; nyl = ny_global/pcol; nzl = nz_global/pcol
    if (mod(nz_global, pcol) /= 0) then
      write (0, *) 'ERROR: decomp_2d (y-slabs): the transposes need ktot divisible by the number of ranks'
      call MPI_ABORT(MPI_COMM_WORLD, 1, ierr)
    end if
    blk = nx*nyl*nzl
    allocate (sbuf(blk*pcol), rbuf(blk*pcol))
    do d = 0, pcol - 1
      o = d*blk
      do k = 1, nzl
        do j = 1, nyl
          sbuf(o + 1:o + nx) = src(:, j, d*nzl + k)
          o = o + nx
        end do
      end do
    end do
    call blocks_to_all(sbuf, rbuf, blk)
    do d = 0, pcol - 1
      o = d*blk
      do k = 1, nzl
        do j = 1, nyl
          dst(:, d*nyl + j, k) = rbuf(o + 1:o + nx)
          o = o + nx
        end do
      end do
    end do
    deallocate (sbuf, rbuf)
  end subroutine z_to_y_real

  subroutine y_to_z_real(src, dst, opt_decomp)
    real(mytype), dimension(:, :, :), intent(in) :: src
    real(mytype), dimension(:, :, :), intent(out) :: dst
    type(DECOMP_INFO), intent(in), optional :: opt_decomp
    real(mytype), allocatable :: sbuf(:), rbuf(:)
    integer :: nx, nyl, nzl, d, k, j, blk, ierr, o
    nx = nx_global; nyl = ny_global/pcol; nzl = nz_global/pcol
    blk = nx*nyl*nzl
    allocate (sbuf(blk*pcol), rbuf(blk*pcol))
    do d = 0, pcol - 1
      o = d*blk
      do k = 1, nzl
        do j = 1, nyl
          sbuf(o + 1:o + nx) = src(:, d*nyl + j, k)
          o = o + nx
        end do
      end do
    end do
    call blocks_to_all(sbuf, rbuf, blk)
    do d = 0, pcol - 1
      o = d*blk
      do k = 1, nzl
        do j = 1, nyl
          dst(:, j, d*nzl + k) = rbuf(o + 1:o + nx)
          o = o + nx
        end do
      end do
    end do
    deallocate (sbuf, rbuf)
  end subroutine y_to_z_real

  ! in-place halo exchange of a z-pencil array allocated WITH halos (uDALES-fork convention); the
  ! halo widths come from the array's own extents: (nx + 2hi, ny/P + 2hj, nz + hk_lo + hk_hi).
  subroutine exchange_halo_z_real(var, opt_decomp, opt_xlevel, opt_ylevel, opt_zlevel)
    real(mytype), dimension(:, :, :), intent(inout) :: var
    type(DECOMP_INFO), intent(in), optional :: opt_decomp
    integer, intent(in), optional :: opt_xlevel(3), opt_ylevel(3), opt_zlevel(3)
    integer :: hi, hj, n1, n2, n3, nyl, cnt, ierr, st(MPI_STATUS_SIZE)
    real(mytype), allocatable :: s1(:, :, :), s2(:, :, :), r1(:, :, :), r2(:, :, :)
    n1 = size(var, 1); n2 = size(var, 2); n3 = size(var, 3)
    if (wrap_x) then        ! x ghosts first, over the interior rows; the rows then travel with their x ghosts (corners consistent)
      hi = (n1 - nx_global)/2
      if (hi >= 1) then
        var(1:hi, :, :) = var(n1 - 2*hi + 1:n1 - hi, :, :)
        var(n1 - hi + 1:n1, :, :) = var(hi + 1:2*hi, :, :)
      end if
    end if
    if (pcol == 1) return       ! (one slab: the solver wraps y itself where it is periodic, src/modboundary.f90:95-107)
    nyl = ny_global/pcol
    hj = (n2 - nyl)/2
    if (hj < 1) return
    allocate (s1(n1, hj, n3), s2(n1, hj, n3), r1(n1, hj, n3), r2(n1, hj, n3))
    s1 = var(:, hj + 1:2*hj, :)                 ! my lowest interior rows  -> previous rank's upper ghosts
    s2 = var(:, n2 - 2*hj + 1:n2 - hj, :)       ! my highest interior rows -> next rank's lower ghosts
    cnt = n1*hj*n3
    call MPI_SENDRECV(s1, cnt, MPI_DOUBLE_PRECISION, nbr_prev, 1, r2, cnt, MPI_DOUBLE_PRECISION, nbr_next, 1, &
                      DECOMP_2D_COMM_CART_Z, st, ierr)
    call MPI_SENDRECV(s2, cnt, MPI_DOUBLE_PRECISION, nbr_next, 2, r1, cnt, MPI_DOUBLE_PRECISION, nbr_prev, 2, &
                      DECOMP_2D_COMM_CART_Z, st, ierr)
    if (nbr_prev /= MPI_PROC_NULL) var(:, 1:hj, :) = r1             ! (a domain edge of a non-periodic y keeps what the solver put there)
    if (nbr_next /= MPI_PROC_NULL) var(:, n2 - hj + 1:n2, :) = r2
    deallocate (s1, s2, r1, r2)
  end subroutine exchange_halo_z_real

  subroutine exchange_unsupported(var, opt_decomp, opt_xlevel, opt_ylevel, opt_zlevel)
    real(mytype), dimension(:, :, :), intent(inout) :: var
    type(DECOMP_INFO), intent(in), optional :: opt_decomp
    integer, intent(in), optional :: opt_xlevel(3), opt_ylevel(3), opt_zlevel(3)
    write (0, *) 'ERROR: exchange_halo_x/y are not provided by this y-slab decomp_2d'
    stop 1
  end subroutine exchange_unsupported

  subroutine update_halo(in, out, level, opt_decomp, opt_global)
    real(mytype), dimension(:, :, :), intent(in) :: in
    real(mytype), allocatable, dimension(:, :, :), intent(out) :: out
    integer, intent(in) :: level
    type(DECOMP_INFO), intent(in), optional :: opt_decomp
    logical, intent(in), optional :: opt_global
    write (0, *) 'ERROR: update_halo not provided by this y-slab decomp_2d'
    stop 1
  end subroutine update_halo

end module decomp_2d

module decomp_2d_fft
  use decomp_2d
  implicit none
  integer(8), save :: plan(-1:2, 3) = 0
  interface decomp_2d_fft_3d
    module procedure fft_3d_r2c, fft_3d_c2r
  end interface
contains
  subroutine fft_unavailable
    write (0, *) 'ERROR: decomp_2d_fft is not provided by this y-slab decomp_2d (ipoiss must be 0)'
    stop 1
  end subroutine fft_unavailable
  subroutine decomp_2d_fft_init(pencil)
    integer, intent(in) :: pencil
    call fft_unavailable
  end subroutine decomp_2d_fft_init
  subroutine r2c_1m_x(input, output)
    real(mytype), dimension(:, :, :), intent(in) :: input
    complex(mytype), dimension(:, :, :), intent(out) :: output
    call fft_unavailable
  end subroutine r2c_1m_x
  subroutine c2r_1m_x(input, output)
    complex(mytype), dimension(:, :, :), intent(in) :: input
    real(mytype), dimension(:, :, :), intent(out) :: output
    call fft_unavailable
  end subroutine c2r_1m_x
  subroutine c2c_1m_y(inout, isign, plan1)
    complex(mytype), dimension(:, :, :), intent(inout) :: inout
    integer, intent(in) :: isign
    integer(8), intent(in) :: plan1
    call fft_unavailable
  end subroutine c2c_1m_y
  subroutine fft_3d_r2c(in_r, out_c)
    real(mytype), dimension(:, :, :), intent(in) :: in_r
    complex(mytype), dimension(:, :, :), intent(out) :: out_c
    call fft_unavailable
  end subroutine fft_3d_r2c
  subroutine fft_3d_c2r(in_c, out_r)
    complex(mytype), dimension(:, :, :), intent(in) :: in_c
    real(mytype), dimension(:, :, :), intent(out) :: out_r
    call fft_unavailable
  end subroutine fft_3d_c2r
end module decomp_2d_fft
