/* TEST INFRASTRUCTURE (oracle) -- not part of the product.
 *
 * Plain-C discrete Fourier / cosine transforms with FFTW's *unnormalised*
 * conventions.  FFTW3 is a system dependency of the reference that is absent from
 * /root/reference and from this image (CMakeLists.txt:97-98; call sites
 * src/modpois.f90:110-111,120-121,134-135,144-145,190-191,481,497,525,541,564,587,
 * 622,644,676,686).  The published definitions restated here (FFTW manual,
 * "What FFTW Really Computes"):
 *   r2c     : Y[k] = sum_j X[j] exp(-2 pi i j k / n),            k = 0..n/2
 *   c2r     : X[j] = sum_k Y[k] exp(+2 pi i j k / n)  (Hermitian Y, unnormalised)
 *   REDFT10 : Y[k] = 2 sum_j X[j] cos(pi (j+1/2) k / n)          (DCT-II)
 *   REDFT01 : Y[k] = X[0] + 2 sum_{j>=1} X[j] cos(pi j (k+1/2) / n)  (DCT-III)
 * Pinned against numpy.fft / scipy.fft golden vectors in tests/golden/fft_*.npz.
 */
#ifndef UDC_ORACLE_FFT_REF_H
#define UDC_ORACLE_FFT_REF_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct fft_ref_plan fft_ref_plan;

/* complex data is interleaved (re, im) doubles */
fft_ref_plan *fft_ref_plan_create(int n);
void fft_ref_plan_destroy(fft_ref_plan *p);

/* in: n reals; out: (n/2+1) complex */
void fft_ref_r2c(fft_ref_plan *p, const double *in, double *out);
/* in: (n/2+1) complex; out: n reals (unnormalised) */
void fft_ref_c2r(fft_ref_plan *p, const double *in, double *out);
/* complex -> complex, sign = -1 forward, +1 backward, unnormalised; in != out */
void fft_ref_c2c(fft_ref_plan *p, const double *in, double *out, int sign);
/* DCT-II / DCT-III (FFTW REDFT10 / REDFT01), O(n^2) */
void fft_ref_redft10(int n, const double *in, double *out);
void fft_ref_redft01(int n, const double *in, double *out);

#ifdef __cplusplus
}
#endif
#endif
