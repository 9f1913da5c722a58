! Stand-in of the one-rank builds (this directory's Makefile; the test infrastructure's all-reference builds use it too).
!
! Single-rank stand-in for the `mpi` module so that the reference's unmodified
! hot-path Fortran (src/modmpi.f90:34 `use mpi`) can be compiled in a container
! that has no Fortran MPI module usable by flang.  It declares only the handles
! and constants that the 18 hot-path modules reference; the MPI_* procedures
! themselves stay implicit-interface externals and are resolved by
! mpi_np1_externals.c (np=1: broadcasts are no-ops, reductions copy).
!
! No arithmetic happens here: with one rank every MPI call in the hot path is
! an identity on the data.
module mpi
  implicit none
  integer, parameter :: MPI_COMM_WORLD = 0
  integer, parameter :: MPI_COMM_NULL = -1
  integer, parameter :: MPI_PROC_NULL = -2
  integer, parameter :: MPI_STATUS_SIZE = 5
  ! datatype handles: value = size in bytes * 100 + tag, decoded in mpi_np1_externals.c
  integer, parameter :: MPI_CHARACTER        = 101
  integer, parameter :: MPI_INTEGER          = 402
  integer, parameter :: MPI_LOGICAL          = 403
  integer, parameter :: MPI_REAL             = 404
  integer, parameter :: MPI_REAL4            = 405
  integer, parameter :: MPI_DOUBLE_PRECISION = 806
  integer, parameter :: MPI_REAL8            = 807
  integer, parameter :: MPI_INTEGER8         = 808
  integer, parameter :: MPI_SUM = 1, MPI_MAX = 2, MPI_MIN = 3, MPI_LOR = 4, MPI_LAND = 5
  integer, parameter :: MPI_INFO_NULL = 0
  interface
    function MPI_Wtime() result(t)
      real(8) :: t
    end function MPI_Wtime
  end interface
end module mpi
