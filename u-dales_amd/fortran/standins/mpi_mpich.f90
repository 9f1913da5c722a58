! Wrapper of the builds over a real MPI (MPICH): used by the drop-in MPI program and the all-reference MPI baseline build.
! The image's /opt/conda/include/mpi.mod was written by gfortran and cannot be read by flang;
! MPICH's Fortran-77 header can.  This wrapper gives the reference's `use mpi` (src/modmpi.f90:34)
! the real MPICH constants and lets the MPI_* calls resolve to libmpifort.
module mpi
  implicit none
  include 'mpif.h'
end module mpi
