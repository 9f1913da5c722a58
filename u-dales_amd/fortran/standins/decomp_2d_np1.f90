! Stand-in of the one-rank builds (this directory's Makefile; the test infrastructure's all-reference builds use it too).
!
! Single-rank stand-in for the 2DECOMP&FFT library (uDALES fork).  The real
! library is an un-vendored, unpinned git submodule of the reference
! (.gitmodules:1-8; `2decomp-fft/` is empty in /root/reference), so its source is
! absent.  This file supplies, for ONE rank only, the symbols the reference's
! hot-path modules import (SURVEY.md section 2.4):
!   * pencil extents (x/y/z pencils all equal the whole domain when np=1),
!   * alloc_{x,y,z}: allocation WITH halos, lower bound 1-h  (src/modfields.f90:469,
!     src/modpois.f90:83-89,441-443) -- arrays are zero-filled here, which is the
!     convention the test oracle and the device library both adopt,
!   * exchange_halo_z: no-op (with one rank 2DECOMP's periodic_bc is .false.,
!     src/modstartup.f90:662-672, and the solver wraps periodicity itself,
!     src/modboundary.f90:95-107),
!   * transpose_*: plain copies.
! It moves data only; it contains no arithmetic.
module decomp_2d
  implicit none
  integer, parameter :: mytype = kind(0.d0)
  integer, save :: nrank = 0, nproc = 1
  integer, save :: nx_global, ny_global, nz_global
  integer, save, dimension(3) :: xstart, xend, xsize
  integer, save, dimension(3) :: ystart, yend, ysize
  integer, save, dimension(3) :: zstart, zend, zsize
  integer, save :: DECOMP_2D_COMM_CART_X = 0, DECOMP_2D_COMM_CART_Y = 0, DECOMP_2D_COMM_CART_Z = 0
  integer, save :: mycol = 0          ! y coordinate of this rank (always 0 here)

  type DECOMP_INFO
    integer, dimension(3) :: xst, xen, xsz
    integer, dimension(3) :: yst, yen, ysz
    integer, dimension(3) :: zst, zen, zsz
    integer, dimension(3) :: xlevel = (/0, 0, 0/), ylevel = (/0, 0, 0/), zlevel = (/0, 0, 0/)
  end type DECOMP_INFO

  type(DECOMP_INFO), save :: decomp_main

  interface alloc_x
    module procedure alloc_x_real, alloc_x_complex
  end interface
  interface alloc_y
    module procedure alloc_y_real, alloc_y_complex
  end interface
  interface alloc_z
    module procedure alloc_z_real, alloc_z_complex
  end interface
  interface transpose_x_to_y
    module procedure copy_real, copy_complex
  end interface
  interface transpose_y_to_z
    module procedure copy_real, copy_complex
  end interface
  interface transpose_z_to_y
    module procedure copy_real, copy_complex
  end interface
  interface transpose_y_to_x
    module procedure copy_real, copy_complex
  end interface
  interface exchange_halo_x
    module procedure exchange_noop
  end interface
  interface exchange_halo_y
    module procedure exchange_noop
  end interface
  interface exchange_halo_z
    module procedure exchange_noop
  end interface

contains

  subroutine decomp_2d_init(nx, ny, nz, p_row, p_col, periodic_bc)
    integer, intent(in) :: nx, ny, nz, p_row, p_col
    logical, dimension(3), intent(in), optional :: periodic_bc
    if (p_row /= 1 .or. p_col /= 1) then
      write (0, *) 'ERROR: the one-rank decomp_2d stand-in is single-rank only'
      stop 1
    end if
    nx_global = nx; ny_global = ny; nz_global = nz
    call decomp_info_init(nx, ny, nz, decomp_main)
    xstart = decomp_main%xst; xend = decomp_main%xen; xsize = decomp_main%xsz
    ystart = decomp_main%yst; yend = decomp_main%yen; ysize = decomp_main%ysz
    zstart = decomp_main%zst; zend = decomp_main%zen; zsize = decomp_main%zsz
  end subroutine decomp_2d_init

  subroutine decomp_2d_finalize
  end subroutine decomp_2d_finalize

  subroutine decomp_info_init(nx, ny, nz, info)
    integer, intent(in) :: nx, ny, nz
    type(DECOMP_INFO), intent(inout) :: info
    info%xst = 1; info%yst = 1; info%zst = 1
    info%xen = (/nx, ny, nz/); info%yen = info%xen; info%zen = info%xen
    info%xsz = info%xen; info%ysz = info%xen; info%zsz = info%xen
  end subroutine decomp_info_init

  subroutine alloc_any_real(var, sz, lev)
    real(mytype), allocatable, dimension(:, :, :) :: var
    integer, intent(in) :: sz(3), lev(3)
    allocate (var(1 - lev(1):sz(1) + lev(1), 1 - lev(2):sz(2) + lev(2), 1 - lev(3):sz(3) + lev(3)))
    var = 0.
  end subroutine alloc_any_real

  subroutine alloc_any_complex(var, sz, lev)
    complex(mytype), allocatable, dimension(:, :, :) :: var
    integer, intent(in) :: sz(3), lev(3)
    allocate (var(1 - lev(1):sz(1) + lev(1), 1 - lev(2):sz(2) + lev(2), 1 - lev(3):sz(3) + lev(3)))
    var = 0.
  end subroutine alloc_any_complex

  subroutine alloc_x_real(var, opt_decomp, opt_global, opt_xlevel)
    real(mytype), allocatable, dimension(:, :, :) :: var
    type(DECOMP_INFO), intent(in), optional :: opt_decomp
    logical, intent(in), optional :: opt_global
    integer, intent(in), optional :: opt_xlevel(3)
    integer :: lev(3), sz(3)
    sz = decomp_main%xsz; if (present(opt_decomp)) sz = opt_decomp%xsz
    lev = decomp_main%xlevel; if (present(opt_xlevel)) lev = opt_xlevel
    call alloc_any_real(var, sz, lev)
  end subroutine alloc_x_real

  subroutine alloc_y_real(var, opt_decomp, opt_global, opt_ylevel)
    real(mytype), allocatable, dimension(:, :, :) :: var
    type(DECOMP_INFO), intent(in), optional :: opt_decomp
    logical, intent(in), optional :: opt_global
    integer, intent(in), optional :: opt_ylevel(3)
    integer :: lev(3), sz(3)
    sz = decomp_main%ysz; if (present(opt_decomp)) sz = opt_decomp%ysz
    lev = decomp_main%ylevel; if (present(opt_ylevel)) lev = opt_ylevel
    call alloc_any_real(var, sz, lev)
  end subroutine alloc_y_real

  subroutine alloc_z_real(var, opt_decomp, opt_global, opt_zlevel)
    real(mytype), allocatable, dimension(:, :, :) :: var
    type(DECOMP_INFO), intent(in), optional :: opt_decomp
    logical, intent(in), optional :: opt_global
    integer, intent(in), optional :: opt_zlevel(3)
    integer :: lev(3), sz(3)
    sz = decomp_main%zsz; if (present(opt_decomp)) sz = opt_decomp%zsz
    lev = decomp_main%zlevel; if (present(opt_zlevel)) lev = opt_zlevel
    call alloc_any_real(var, sz, lev)
  end subroutine alloc_z_real

  subroutine alloc_x_complex(var, opt_decomp, opt_global, opt_xlevel)
    complex(mytype), allocatable, dimension(:, :, :) :: var
    type(DECOMP_INFO), intent(in), optional :: opt_decomp
    logical, intent(in), optional :: opt_global
    integer, intent(in), optional :: opt_xlevel(3)
    integer :: lev(3), sz(3)
    sz = decomp_main%xsz; if (present(opt_decomp)) sz = opt_decomp%xsz
    lev = decomp_main%xlevel; if (present(opt_xlevel)) lev = opt_xlevel
    call alloc_any_complex(var, sz, lev)
  end subroutine alloc_x_complex

  subroutine alloc_y_complex(var, opt_decomp, opt_global, opt_ylevel)
    complex(mytype), allocatable, dimension(:, :, :) :: var
    type(DECOMP_INFO), intent(in), optional :: opt_decomp
    logical, intent(in), optional :: opt_global
    integer, intent(in), optional :: opt_ylevel(3)
    integer :: lev(3), sz(3)
    sz = decomp_main%ysz; if (present(opt_decomp)) sz = opt_decomp%ysz
    lev = decomp_main%ylevel; if (present(opt_ylevel)) lev = opt_ylevel
    call alloc_any_complex(var, sz, lev)
  end subroutine alloc_y_complex

  subroutine alloc_z_complex(var, opt_decomp, opt_global, opt_zlevel)
    complex(mytype), allocatable, dimension(:, :, :) :: var
    type(DECOMP_INFO), intent(in), optional :: opt_decomp
    logical, intent(in), optional :: opt_global
    integer, intent(in), optional :: opt_zlevel(3)
    integer :: lev(3), sz(3)
    sz = decomp_main%zsz; if (present(opt_decomp)) sz = opt_decomp%zsz
    lev = decomp_main%zlevel; if (present(opt_zlevel)) lev = opt_zlevel
    call alloc_any_complex(var, sz, lev)
  end subroutine alloc_z_complex

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

  subroutine exchange_noop(var, opt_decomp, opt_xlevel, opt_ylevel, opt_zlevel)
    real(mytype), dimension(:, :, :), intent(inout) :: var
    type(DECOMP_INFO), intent(in), optional :: opt_decomp
    integer, intent(in), optional :: opt_xlevel(3), opt_ylevel(3), opt_zlevel(3)
  end subroutine exchange_noop

  subroutine update_halo(in, out, level, opt_decomp, opt_global)
    real(mytype), dimension(:, :, :), intent(in) :: in
    real(mytype), allocatable, dimension(:, :, :), intent(out) :: out
    integer, intent(in) :: level
    type(DECOMP_INFO), intent(in), optional :: opt_decomp
    logical, intent(in), optional :: opt_global
    write (0, *) 'ERROR: update_halo not provided by the one-rank stand-in'
    stop 1
  end subroutine update_halo

end module decomp_2d

! The non-default Poisson branches (ipoiss = 2, 3; src/modpois.f90:228-353,714-880)
! reference these symbols; they only need to exist for the module to compile.
module decomp_2d_fft
  use decomp_2d
  implicit none
  integer(8), save :: plan(-1:2, 3) = 0
  interface decomp_2d_fft_3d
    module procedure fft_3d_r2c, fft_3d_c2r
  end interface
contains
  subroutine fft_unavailable
    write (0, *) 'ERROR: decomp_2d_fft is not provided by the one-rank stand-in (ipoiss must be 0)'
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
