/*
 * oracle/orc_fft.c -- TEST INFRASTRUCTURE ONLY (see oracle/README.md).
 *
 * Self-contained double-precision 3-D r2c / c2r FFT with FFTW's layout and sign conventions
 * (stand-in for the FFTW3 calls at /root/reference/barlib/src/fftwrapper.cc:88-119).
 * Radix-2 for power-of-two lengths, plain O(n^2) DFT otherwise (only used for tiny odd-sized tests).
 * Twiddles come from sin/cos per index (no recurrences) to stay at fp64 round-off.
 */
#include "orc_fft.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct {
  unsigned n;
  int pow2;
  double *tw;    /* pow2: n/2 entries (cos, -sin) of exp(-2 pi i k/n); else n entries */
  unsigned *rev; /* pow2: bit-reversal permutation */
} fft1d;

static void fft1d_init(fft1d *p, unsigned n) {
  p->n = n;
  p->pow2 = (n & (n - 1)) == 0;
  unsigned nt = p->pow2 ? (n / 2 ? n / 2 : 1) : n;
  p->tw = (double *)malloc(sizeof(double) * 2 * nt);
  for (unsigned k = 0; k < nt; k++) {
    double a = -2.0 * M_PI * (double)k / (double)n;
    p->tw[2 * k] = cos(a);
    p->tw[2 * k + 1] = sin(a);
  }
  p->rev = NULL;
  if (p->pow2) {
    p->rev = (unsigned *)malloc(sizeof(unsigned) * n);
    unsigned bits = 0;
    while ((1u << bits) < n) bits++;
    for (unsigned i = 0; i < n; i++) {
      unsigned r = 0;
      for (unsigned b = 0; b < bits; b++)
        if (i & (1u << b)) r |= 1u << (bits - 1 - b);
      p->rev[i] = r;
    }
  }
}

static void fft1d_free(fft1d *p) {
  free(p->tw);
  free(p->rev);
}

/* x: n interleaved complex, transformed in place. sign = -1 forward, +1 backward. scratch: 2n doubles. */
static void fft1d_exec(const fft1d *p, double *x, int sign, double *scratch) {
  const unsigned n = p->n;
  if (n == 1) return;
  if (p->pow2) {
    for (unsigned i = 0; i < n; i++) {
      unsigned r = p->rev[i];
      if (r > i) {
        double tr = x[2 * i], ti = x[2 * i + 1];
        x[2 * i] = x[2 * r];
        x[2 * i + 1] = x[2 * r + 1];
        x[2 * r] = tr;
        x[2 * r + 1] = ti;
      }
    }
    const double s = (sign < 0) ? 1.0 : -1.0; /* table holds exp(-i a): flip imag for backward */
    for (unsigned len = 2; len <= n; len <<= 1) {
      const unsigned half = len >> 1, step = n / len;
      for (unsigned i = 0; i < n; i += len) {
        for (unsigned k = 0; k < half; k++) {
          const double wr = p->tw[2 * k * step], wi = s * p->tw[2 * k * step + 1];
          double *a = x + 2 * (i + k), *b = x + 2 * (i + k + half);
          const double br = b[0] * wr - b[1] * wi, bi = b[0] * wi + b[1] * wr;
          b[0] = a[0] - br;
          b[1] = a[1] - bi;
          a[0] += br;
          a[1] += bi;
        }
      }
    }
  } else {
    const double s = (sign < 0) ? 1.0 : -1.0;
    for (unsigned k = 0; k < n; k++) {
      double sr = 0., si = 0.;
      for (unsigned j = 0; j < n; j++) {
        unsigned idx = (unsigned)(((unsigned long)j * k) % n);
        const double wr = p->tw[2 * idx], wi = s * p->tw[2 * idx + 1];
        sr += x[2 * j] * wr - x[2 * j + 1] * wi;
        si += x[2 * j] * wi + x[2 * j + 1] * wr;
      }
      scratch[2 * k] = sr;
      scratch[2 * k + 1] = si;
    }
    memcpy(x, scratch, sizeof(double) * 2 * n);
  }
}

static size_t maxu(size_t a, size_t b) { return a > b ? a : b; }

/* Transform along axis 2 (N2 direction, stride N3h) and axis 1 (N1 direction, stride N2*N3h) of a
 * N1 x N2 x N3h complex array in place. */
static void fft_axes_xy(unsigned N1, unsigned N2, unsigned N3h, double *c, int sign, const fft1d *p1,
                        const fft1d *p2) {
  const size_t plane = (size_t)N2 * N3h;
  /* y */
#pragma omp parallel
  {
    double *tmp = (double *)malloc(sizeof(double) * 2 * (plane + maxu(N1, N2)));
    double *scr = tmp + 2 * plane;
#pragma omp for schedule(static)
    for (long i = 0; i < (long)N1; i++) {
      double *pl = c + 2 * plane * (size_t)i;
      for (unsigned j = 0; j < N2; j++)
        for (unsigned k = 0; k < N3h; k++) {
          tmp[2 * ((size_t)k * N2 + j)] = pl[2 * ((size_t)j * N3h + k)];
          tmp[2 * ((size_t)k * N2 + j) + 1] = pl[2 * ((size_t)j * N3h + k) + 1];
        }
      for (unsigned k = 0; k < N3h; k++) fft1d_exec(p2, tmp + 2 * (size_t)k * N2, sign, scr);
      for (unsigned j = 0; j < N2; j++)
        for (unsigned k = 0; k < N3h; k++) {
          pl[2 * ((size_t)j * N3h + k)] = tmp[2 * ((size_t)k * N2 + j)];
          pl[2 * ((size_t)j * N3h + k) + 1] = tmp[2 * ((size_t)k * N2 + j) + 1];
        }
    }
    free(tmp);
  }
  /* x */
#pragma omp parallel
  {
    const size_t slab = (size_t)N1 * N3h;
    double *tmp = (double *)malloc(sizeof(double) * 2 * (slab + maxu(N1, N2)));
    double *scr = tmp + 2 * slab;
#pragma omp for schedule(static)
    for (long j = 0; j < (long)N2; j++) {
      for (unsigned i = 0; i < N1; i++) {
        const double *src = c + 2 * (plane * i + (size_t)j * N3h);
        for (unsigned k = 0; k < N3h; k++) {
          tmp[2 * ((size_t)k * N1 + i)] = src[2 * k];
          tmp[2 * ((size_t)k * N1 + i) + 1] = src[2 * k + 1];
        }
      }
      for (unsigned k = 0; k < N3h; k++) fft1d_exec(p1, tmp + 2 * (size_t)k * N1, sign, scr);
      for (unsigned i = 0; i < N1; i++) {
        double *dst = c + 2 * (plane * i + (size_t)j * N3h);
        for (unsigned k = 0; k < N3h; k++) {
          dst[2 * k] = tmp[2 * ((size_t)k * N1 + i)];
          dst[2 * k + 1] = tmp[2 * ((size_t)k * N1 + i) + 1];
        }
      }
    }
    free(tmp);
  }
}

void orc_fft_r2c_3d(unsigned N1, unsigned N2, unsigned N3, const double *in, double *out) {
  const unsigned N3h = N3 / 2 + 1;
  fft1d p1, p2, p3;
  fft1d_init(&p1, N1);
  fft1d_init(&p2, N2);
  fft1d_init(&p3, N3);
  const long rows = (long)N1 * N2;
#pragma omp parallel
  {
    double *tmp = (double *)malloc(sizeof(double) * 4 * N3);
    double *scr = tmp + 2 * N3;
#pragma omp for schedule(static)
    for (long r = 0; r < rows; r++) {
      const double *src = in + (size_t)r * N3;
      for (unsigned k = 0; k < N3; k++) {
        tmp[2 * k] = src[k];
        tmp[2 * k + 1] = 0.;
      }
      fft1d_exec(&p3, tmp, -1, scr);
      memcpy(out + 2 * (size_t)r * N3h, tmp, sizeof(double) * 2 * N3h);
    }
    free(tmp);
  }
  fft_axes_xy(N1, N2, N3h, out, -1, &p1, &p2);
  fft1d_free(&p1);
  fft1d_free(&p2);
  fft1d_free(&p3);
}

void orc_fft_c2r_3d(unsigned N1, unsigned N2, unsigned N3, double *in, double *out) {
  const unsigned N3h = N3 / 2 + 1;
  fft1d p1, p2, p3;
  fft1d_init(&p1, N1);
  fft1d_init(&p2, N2);
  fft1d_init(&p3, N3);
  fft_axes_xy(N1, N2, N3h, in, +1, &p1, &p2);
  const long rows = (long)N1 * N2;
#pragma omp parallel
  {
    double *tmp = (double *)malloc(sizeof(double) * 4 * N3);
    double *scr = tmp + 2 * N3;
#pragma omp for schedule(static)
    for (long r = 0; r < rows; r++) {
      const double *src = in + 2 * (size_t)r * N3h;
      /* Hermitian extension of the half row; like FFTW's c2r the imaginary parts of the
       * self-conjugate elements (k=0 and, for even N3, k=N3/2) do not reach the real output. */
      for (unsigned k = 0; k < N3h; k++) {
        tmp[2 * k] = src[2 * k];
        tmp[2 * k + 1] = src[2 * k + 1];
      }
      for (unsigned k = N3h; k < N3; k++) {
        tmp[2 * k] = src[2 * (N3 - k)];
        tmp[2 * k + 1] = -src[2 * (N3 - k) + 1];
      }
      tmp[1] = 0.;
      if ((N3 & 1u) == 0) tmp[2 * (N3 / 2) + 1] = 0.;
      fft1d_exec(&p3, tmp, +1, scr);
      double *dst = out + (size_t)r * N3;
      for (unsigned k = 0; k < N3; k++) dst[k] = tmp[2 * k];
    }
    free(tmp);
  }
  fft1d_free(&p1);
  fft1d_free(&p2);
  fft1d_free(&p3);
}
