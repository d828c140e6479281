/*
 * oracle/orc_fft.h -- TEST INFRASTRUCTURE ONLY (see oracle/README.md).
 *
 * Minimal 3-D real<->half-complex FFT for the CPU oracle.  The reference links FFTW3
 * (barlib/src/fftwrapper.cc:26-119, 281-333), which is not present in this image, so the
 * oracle carries its own transform with FFTW's conventions:
 *   - r2c: forward, exponent sign -1, unnormalised, output N1 x N2 x (N3/2+1) interleaved re/im,
 *          row-major with the halved axis fastest (fftw_plan_dft_r2c_3d layout);
 *   - c2r: backward, exponent sign +1, unnormalised, input may be destroyed.
 * The 1/N of the reference's inverse (fftwrapper.cc:99-101, FOURIER_DEF_2) is applied by the caller.
 */
#ifndef ORC_FFT_H
#define ORC_FFT_H
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* in: N1*N2*N3 doubles; out: N1*N2*(N3/2+1) complex (2 doubles each). in and out must not alias. */
void orc_fft_r2c_3d(unsigned N1, unsigned N2, unsigned N3, const double *in, double *out);
/* in: N1*N2*(N3/2+1) complex, DESTROYED; out: N1*N2*N3 doubles (unnormalised). */
void orc_fft_c2r_3d(unsigned N1, unsigned N2, unsigned N3, double *in, double *out);

#ifdef __cplusplus
}
#endif
#endif
