! TEST INFRASTRUCTURE (oracle/_ref build only) -- not part of the product.
! Constants of FFTW3's legacy Fortran include file that src/modpois.f90:34
! pulls in; values are those of the public fftw3.h (api/fftw3.h: FFTW_MEASURE = 0,
! FFTW_ESTIMATE = 64, r2r kinds REDFT01 = 4, REDFT10 = 5).
      INTEGER FFTW_MEASURE
      PARAMETER (FFTW_MEASURE=0)
      INTEGER FFTW_ESTIMATE
      PARAMETER (FFTW_ESTIMATE=64)
      INTEGER FFTW_REDFT01
      PARAMETER (FFTW_REDFT01=4)
      INTEGER FFTW_REDFT10
      PARAMETER (FFTW_REDFT10=5)
      INTEGER FFTW_FORWARD
      PARAMETER (FFTW_FORWARD=-1)
      INTEGER FFTW_BACKWARD
      PARAMETER (FFTW_BACKWARD=+1)
