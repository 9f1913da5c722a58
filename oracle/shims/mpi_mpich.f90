! TEST INFRASTRUCTURE (oracle/_ref MPI build only) -- not part of the product.
! The image's /opt/conda/include/mpi.mod was written by gfortran and cannot be read by flang;
! MPICH's Fortran-77 header can.  This wrapper gives the reference's `use mpi` (src/modmpi.f90:34)
! the real MPICH constants and lets the MPI_* calls resolve to libmpifort.
module mpi
  implicit none
  include 'mpif.h'
end module mpi
