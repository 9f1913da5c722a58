! Stand-in of every build of the reference's program in this image (the drop-in program and the all-reference builds).
!
! Stand-in for the NetCDF Fortran-90 module, which this image does not have: the nf90_* generics that src/modstat_nc.f90 and
! src/initfac.f90:267-270 call, over netcdf_rec_io.c (tables in memory, nf90_put_var appended to the file as a flat
! float64 record stream that tests/refdump.py reads).  With it the reference's WHOLE src/ tree -- program.f90, modstartup.f90,
! the statistics and dump modules -- compiles and runs unmodified.  Re-opening an existing file (a run continued in the
! directory of an earlier one, src/modstat_nc.f90:129-164) and reading view factors (lEB without lvfsparse) are not provided:
! nf90_open returns an error status for anything but a file this process created itself (src/modfielddump.f90:313-321 opens its
! file twice).
module netcdf
  use iso_c_binding
  implicit none
  private
  integer, parameter, public :: nf90_noerr = 0, nf90_global = 0, nf90_unlimited = 0
  integer, parameter, public :: nf90_nowrite = 0, nf90_write = 1, nf90_share = 2048, nf90_netcdf4 = 4096
  integer, parameter, public :: nf90_byte = 1, nf90_char = 2, nf90_short = 3, nf90_int = 4, nf90_float = 5, nf90_double = 6
  integer, parameter :: err_unsupported = -200

  public :: nf90_create, nf90_open, nf90_close, nf90_sync, nf90_enddef, nf90_redef, nf90_strerror
  public :: nf90_def_dim, nf90_def_var, nf90_inq_dimid, nf90_inq_varid, nf90_inquire, nf90_inquire_dimension
  public :: nf90_put_att, nf90_put_var, nf90_get_var

  interface
    integer(c_int) function udnc_create(path, n) bind(C)
      import; character(kind=c_char) :: path(*); integer(c_int), value :: n
    end function
    integer(c_int) function udnc_reopen(path, n) bind(C)
      import; character(kind=c_char) :: path(*); integer(c_int), value :: n
    end function
    integer(c_int) function udnc_time_values(ncid, varid, n, out) bind(C)
      import; integer(c_int), value :: ncid, varid, n; real(c_double) :: out(*)
    end function
    integer(c_int) function udnc_def_dim(ncid, name, n, len) bind(C)
      import; integer(c_int), value :: ncid, n, len; character(kind=c_char) :: name(*)
    end function
    integer(c_int) function udnc_inq_dimid(ncid, name, n) bind(C)
      import; integer(c_int), value :: ncid, n; character(kind=c_char) :: name(*)
    end function
    integer(c_int) function udnc_dim_len(ncid, dimid) bind(C)
      import; integer(c_int), value :: ncid, dimid
    end function
    integer(c_int) function udnc_unlimited(ncid) bind(C)
      import; integer(c_int), value :: ncid
    end function
    integer(c_int) function udnc_def_var(ncid, name, n, xtype, ndims, dimids) bind(C)
      import; integer(c_int), value :: ncid, n, xtype, ndims; character(kind=c_char) :: name(*); integer(c_int) :: dimids(*)
    end function
    integer(c_int) function udnc_inq_varid(ncid, name, n) bind(C)
      import; integer(c_int), value :: ncid, n; character(kind=c_char) :: name(*)
    end function
    integer(c_int) function udnc_put(ncid, varid, nstart, start, rank, shp, v) bind(C)
      import; integer(c_int), value :: ncid, varid, nstart, rank; integer(c_int) :: start(*), shp(*); real(c_double) :: v(*)
    end function
    integer(c_int) function udnc_sync(ncid) bind(C)
      import; integer(c_int), value :: ncid
    end function
    integer(c_int) function udnc_close(ncid) bind(C)
      import; integer(c_int), value :: ncid
    end function
  end interface

  interface nf90_def_var
    module procedure def_var_0, def_var_n
  end interface
  interface nf90_put_att
    module procedure put_att_c, put_att_r4, put_att_r8, put_att_i
  end interface
  interface nf90_put_var
    module procedure put_r0, put_r1, put_r2, put_r3, put_i3s
  end interface
  interface nf90_get_var
    module procedure get_r1, get_r2
  end interface

contains

  integer function nf90_create(path, cmode, ncid)
    character(*), intent(in) :: path
    integer, intent(in) :: cmode
    integer, intent(out) :: ncid
    ncid = udnc_create(path, len(path))
    nf90_create = merge(nf90_noerr, err_unsupported, ncid > 0)
  end function

  integer function nf90_open(path, mode, ncid)
    character(*), intent(in) :: path
    integer, intent(in) :: mode
    integer, intent(out) :: ncid
    ncid = udnc_reopen(path, len(path))
    nf90_open = nf90_noerr
    if (ncid < 1) then
      nf90_open = err_unsupported
      write (0, *) 'netcdf stand-in: cannot open a file of an earlier run: ', trim(path)
    end if
  end function

  integer function nf90_close(ncid)
    integer, intent(in) :: ncid
    nf90_close = udnc_close(ncid)
  end function

  integer function nf90_sync(ncid)
    integer, intent(in) :: ncid
    nf90_sync = udnc_sync(ncid)
  end function

  integer function nf90_enddef(ncid)
    integer, intent(in) :: ncid
    nf90_enddef = nf90_noerr
  end function

  integer function nf90_redef(ncid)
    integer, intent(in) :: ncid
    nf90_redef = nf90_noerr
  end function

  function nf90_strerror(status)
    integer, intent(in) :: status
    character(len=80) :: nf90_strerror
    write (nf90_strerror, '(a,i0)') 'netcdf stand-in: status ', status
  end function

  integer function nf90_def_dim(ncid, name, len, dimid)
    integer, intent(in) :: ncid, len
    character(*), intent(in) :: name
    integer, intent(out) :: dimid
    dimid = udnc_def_dim(ncid, name, len_trim(name), len)
    nf90_def_dim = merge(nf90_noerr, err_unsupported, dimid > 0)
  end function

  integer function nf90_inq_dimid(ncid, name, dimid)
    integer, intent(in) :: ncid
    character(*), intent(in) :: name
    integer, intent(out) :: dimid
    dimid = udnc_inq_dimid(ncid, name, len_trim(name))
    nf90_inq_dimid = merge(nf90_noerr, -46, dimid > 0)
  end function

  integer function nf90_inq_varid(ncid, name, varid)
    integer, intent(in) :: ncid
    character(*), intent(in) :: name
    integer, intent(out) :: varid
    varid = udnc_inq_varid(ncid, name, len_trim(name))
    nf90_inq_varid = merge(nf90_noerr, -49, varid > 0)
  end function

  integer function nf90_inquire(ncid, nDimensions, nVariables, nAttributes, unlimitedDimId, formatNum)
    integer, intent(in) :: ncid
    integer, intent(out), optional :: nDimensions, nVariables, nAttributes, unlimitedDimId, formatNum
    if (present(nDimensions)) nDimensions = 0
    if (present(nVariables)) nVariables = 0
    if (present(nAttributes)) nAttributes = 0
    if (present(formatNum)) formatNum = 0
    if (present(unlimitedDimId)) unlimitedDimId = udnc_unlimited(ncid)
    nf90_inquire = nf90_noerr
  end function

  integer function nf90_inquire_dimension(ncid, dimid, name, len)
    integer, intent(in) :: ncid, dimid
    character(*), intent(out), optional :: name
    integer, intent(out), optional :: len
    integer :: l
    l = udnc_dim_len(ncid, dimid)
    if (present(name)) name = ''
    if (present(len)) len = l
    nf90_inquire_dimension = merge(nf90_noerr, -46, l >= 0)
  end function

  integer function def_var_0(ncid, name, xtype, dimids, varid)
    integer, intent(in) :: ncid, xtype, dimids
    character(*), intent(in) :: name
    integer, intent(out) :: varid
    integer :: d(1)
    d(1) = dimids
    varid = udnc_def_var(ncid, name, len_trim(name), xtype, 1, d)
    def_var_0 = merge(nf90_noerr, err_unsupported, varid > 0)
  end function

  integer function def_var_n(ncid, name, xtype, dimids, varid)
    integer, intent(in) :: ncid, xtype, dimids(:)
    character(*), intent(in) :: name
    integer, intent(out) :: varid
    integer :: d(size(dimids))
    d = dimids
    varid = udnc_def_var(ncid, name, len_trim(name), xtype, size(dimids), d)
    def_var_n = merge(nf90_noerr, err_unsupported, varid > 0)
  end function

  integer function put_att_c(ncid, varid, name, values)
    integer, intent(in) :: ncid, varid
    character(*), intent(in) :: name, values
    put_att_c = nf90_noerr
  end function
  integer function put_att_r4(ncid, varid, name, values)
    integer, intent(in) :: ncid, varid
    character(*), intent(in) :: name
    real(4), intent(in) :: values
    put_att_r4 = nf90_noerr
  end function
  integer function put_att_r8(ncid, varid, name, values)
    integer, intent(in) :: ncid, varid
    character(*), intent(in) :: name
    real(8), intent(in) :: values
    put_att_r8 = nf90_noerr
  end function
  integer function put_att_i(ncid, varid, name, values)
    integer, intent(in) :: ncid, varid
    character(*), intent(in) :: name
    integer, intent(in) :: values
    put_att_i = nf90_noerr
  end function

  integer function put_any(ncid, varid, v, shp, start)
    integer, intent(in) :: ncid, varid, shp(:)
    real(8), intent(in) :: v(*)
    integer, intent(in), optional :: start(:)
    integer :: s(8), sh(8), ns
    ns = 0
    if (present(start)) then
      ns = size(start); s(1:ns) = start
    end if
    sh(1:size(shp)) = shp
    put_any = udnc_put(ncid, varid, ns, s, size(shp), sh, v)
  end function

  integer function put_r0(ncid, varid, values, start)
    integer, intent(in) :: ncid, varid
    real(8), intent(in) :: values
    integer, intent(in), optional :: start(:)
    real(8) :: v(1)
    integer :: shp(0)
    v(1) = values
    put_r0 = put_any(ncid, varid, v, shp, start)
  end function

  integer function put_r1(ncid, varid, values, start, count)
    integer, intent(in) :: ncid, varid
    real(8), intent(in) :: values(:)
    integer, intent(in), optional :: start(:), count(:)
    real(8) :: v(size(values))
    v = values
    put_r1 = put_any(ncid, varid, v, shape(values), start)
  end function

  integer function put_r2(ncid, varid, values, start, count)
    integer, intent(in) :: ncid, varid
    real(8), intent(in) :: values(:, :)
    integer, intent(in), optional :: start(:), count(:)
    real(8), allocatable :: v(:)
    v = reshape(values, (/size(values)/))
    put_r2 = put_any(ncid, varid, v, shape(values), start)
  end function

  integer function put_r3(ncid, varid, values, start, count)
    integer, intent(in) :: ncid, varid
    real(8), intent(in) :: values(:, :, :)
    integer, intent(in), optional :: start(:), count(:)
    real(8), allocatable :: v(:)
    v = reshape(values, (/size(values)/))
    put_r3 = put_any(ncid, varid, v, shape(values), start)
  end function

  integer function put_i3s(ncid, varid, values, start, count)
    integer, intent(in) :: ncid, varid
    integer(kind=selected_int_kind(4)), intent(in) :: values(:, :, :)
    integer, intent(in), optional :: start(:), count(:)
    real(8), allocatable :: v(:)
    v = reshape(real(values, 8), (/size(values)/))
    put_i3s = put_any(ncid, varid, v, shape(values), start)
  end function

  integer function get_r1(ncid, varid, values)
    integer, intent(in) :: ncid, varid
    real(8), intent(out) :: values(:)
    get_r1 = udnc_time_values(ncid, varid, size(values), values)
    if (get_r1 /= 0) get_r1 = err_unsupported
  end function

  integer function get_r2(ncid, varid, values)
    integer, intent(in) :: ncid, varid
    real(8), intent(out) :: values(:, :)
    values = 0.
    get_r2 = err_unsupported
  end function
end module netcdf
