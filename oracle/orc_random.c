/*
 * oracle/orc_random.c -- TEST INFRASTRUCTURE ONLY (see oracle/README.md).
 *
 * The reference's momentum draw, restated: draw_momenta (barlib/src/HMC_momenta.cc:42-94),
 * create_GARFIELD (barlib/src/random.cpp:48-511) and resolution_independent_random_grid_FS
 * (barlib/include/random.hpp:35-120), with the serial random stream the reference takes from GSL.
 *
 * THIRD-PARTY DEPENDENCY ABSENT FROM /root/reference AND FROM THIS IMAGE: GSL (un-pinned upstream, CMakeLists.txt:36,
 * "GSL" in README.md).  Its published algorithms are restated here:
 *   * gsl_rng_mt19937 (rng/mt.c): Matsumoto & Nishimura's MT19937 with the 2002 seeding
 *     mt[i] = 1812433253 (mt[i-1] ^ (mt[i-1] >> 30)) + i (seed 0 -> 4357), tempering 11 / 7 (0x9d2c5680) / 15
 *     (0xefc60000) / 18; gsl_rng_uniform = genrand_int32 / 2^32; gsl_rng_uniform_pos redraws on 0.
 *     Known-answer check: seed 5489's first outputs 3499211612, 581869302 (the reference implementation's published
 *     vector; tests/test_oracle_random.py).
 *   * gsl_ran_gaussian (randist/gauss.c): polar Box-Muller -- x, y = -1 + 2 uniform_pos until 0 < r2 = x^2 + y^2 <= 1,
 *     return sigma * y * sqrt(-2 log(r2) / r2); gsl_ran_ugaussian = gsl_ran_gaussian(r, 1).
 * PARITY UNPINNED against a real GSL build (none is available); the stream is pinned to the MT19937 known answers only.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "bchmc_oracle.h"
#include "orc_fft.h"

/* ---- gsl_rng_mt19937 ---- */
typedef struct {
  uint32_t mt[624];
  int mti;
} orc_rng;

static void rng_set(orc_rng *r, unsigned long s) {
  if (s == 0) s = 4357; /* the default seed of rng/mt.c */
  r->mt[0] = (uint32_t)(s & 0xffffffffUL);
  for (int i = 1; i < 624; i++) r->mt[i] = 1812433253U * (r->mt[i - 1] ^ (r->mt[i - 1] >> 30)) + (uint32_t)i;
  r->mti = 624;
}

static uint32_t rng_get(orc_rng *r) {
  uint32_t *mt = r->mt;
  if (r->mti >= 624) {
    int kk;
    for (kk = 0; kk < 624 - 397; kk++) {
      uint32_t y = (mt[kk] & 0x80000000U) | (mt[kk + 1] & 0x7fffffffU);
      mt[kk] = mt[kk + 397] ^ (y >> 1) ^ ((y & 1U) ? 0x9908b0dfU : 0U);
    }
    for (; kk < 623; kk++) {
      uint32_t y = (mt[kk] & 0x80000000U) | (mt[kk + 1] & 0x7fffffffU);
      mt[kk] = mt[kk + (397 - 624)] ^ (y >> 1) ^ ((y & 1U) ? 0x9908b0dfU : 0U);
    }
    {
      uint32_t y = (mt[623] & 0x80000000U) | (mt[0] & 0x7fffffffU);
      mt[623] = mt[396] ^ (y >> 1) ^ ((y & 1U) ? 0x9908b0dfU : 0U);
    }
    r->mti = 0;
  }
  uint32_t k = mt[r->mti++];
  k ^= (k >> 11);
  k ^= (k << 7) & 0x9d2c5680U;
  k ^= (k << 15) & 0xefc60000U;
  k ^= (k >> 18);
  return k;
}

static double rng_uniform(orc_rng *r) { return rng_get(r) / 4294967296.0; }
static double rng_uniform_pos(orc_rng *r) {
  double x;
  do x = rng_uniform(r);
  while (x == 0);
  return x;
}

/* gsl_ran_gaussian, randist/gauss.c (polar Box-Muller) */
static double ran_gaussian(orc_rng *r, double sigma) {
  double x, y, r2;
  do {
    x = -1 + 2 * rng_uniform_pos(r);
    y = -1 + 2 * rng_uniform_pos(r);
    r2 = x * x + y * y;
  } while (r2 > 1.0 || r2 == 0);
  return sigma * y * sqrt(-2.0 * log(r2) / r2);
}

/* hooks for the tests */
void orc_mt19937_stream(unsigned long seed, uint32_t *out, size_t n) {
  orc_rng r;
  rng_set(&r, seed);
  for (size_t i = 0; i < n; i++) out[i] = rng_get(&r);
}
void orc_ugaussian_stream(unsigned long seed, double *out, size_t n) {
  orc_rng r;
  rng_set(&r, seed);
  for (size_t i = 0; i < n; i++) out[i] = ran_gaussian(&r, 1.0);
}

/* random.hpp:23-31: real part first, then the imaginary part */
static void complex_gaussian(orc_rng *r, double *cell) {
  cell[0] = ran_gaussian(r, 1.0);
  cell[1] = ran_gaussian(r, 1.0);
}

/* resolution_independent_random_grid_FS (random.hpp:35-120), half_size = false as create_GARFIELD calls it
 * (random.cpp:77): the cube is filled layer by layer from the eight corners inwards, so that the grid of size n is the
 * low-k part of the grid of size 2 n for the same seed.  out: n^3 interleaved complex. */
static void random_grid_FS(unsigned n, orc_rng *rng, double *out) {
  const size_t js = n, is = (size_t)n * n;
#define CELL(a, b, c) (out + 2 * ((size_t)(a) * is + (size_t)(b) * js + (size_t)(c)))
  for (unsigned i = 0; i < n / 2; i++) {
    const unsigned m = n - 1;
    for (unsigned k = 0; k < i + 1; k++) { /* the two "walls" */
      for (unsigned j = 0; j < i; j++) {   /* "slim" side, corners 1-8 in the reference's order (random.hpp:74-85) */
        complex_gaussian(rng, CELL(i, j, k));
        complex_gaussian(rng, CELL(m - i, j, k));
        complex_gaussian(rng, CELL(i, m - j, k));
        complex_gaussian(rng, CELL(m - i, m - j, k));
        complex_gaussian(rng, CELL(i, j, m - k));
        complex_gaussian(rng, CELL(m - i, j, m - k));
        complex_gaussian(rng, CELL(i, m - j, m - k));
        complex_gaussian(rng, CELL(m - i, m - j, m - k));
      }
      for (unsigned j = 0; j < i + 1; j++) { /* "broad" side (random.hpp:87-100) */
        complex_gaussian(rng, CELL(j, i, k));
        complex_gaussian(rng, CELL(m - j, i, k));
        complex_gaussian(rng, CELL(j, m - i, k));
        complex_gaussian(rng, CELL(m - j, m - i, k));
        complex_gaussian(rng, CELL(j, i, m - k));
        complex_gaussian(rng, CELL(m - j, i, m - k));
        complex_gaussian(rng, CELL(j, m - i, m - k));
        complex_gaussian(rng, CELL(m - j, m - i, m - k));
      }
    }
    for (unsigned j = 0; j < i; j++) /* the "roof" (random.hpp:102-116) */
      for (unsigned k = 0; k < i; k++) {
        complex_gaussian(rng, CELL(j, k, i));
        complex_gaussian(rng, CELL(m - j, k, i));
        complex_gaussian(rng, CELL(j, m - k, i));
        complex_gaussian(rng, CELL(m - j, m - k, i));
        complex_gaussian(rng, CELL(j, k, m - i));
        complex_gaussian(rng, CELL(m - j, k, m - i));
        complex_gaussian(rng, CELL(j, m - k, m - i));
        complex_gaussian(rng, CELL(m - j, m - k, m - i));
      }
  }
#undef CELL
}

/* create_GARFIELD (random.cpp:48-511), FOURIER_DEF_2.  The reference spells out 27 index classes of (i, j, k) in
 * [0, N/2]^3; they are one rule: an axis index a is self-conjugate iff a is 0 or N/2, the Hermitian partner of
 * (a, b, c) is (-a, -b, -c) mod N.  Per (i, j, k):
 *   no free axis  (8 points): DC -> 0 (351-357); the others real: re *= sqrt(2) sigma, im = 0 (264-270, 458-498);
 *   otherwise the representatives are (i, j, k) itself and, with THREE free axes, its three single-axis mirrors
 *   (109-142), with TWO free axes the mirror of the FIRST free axis only (145-201, 272-318: the mirror of the second is
 *   the partner of that one), with ONE free axis none; each representative keeps its own random number times sigma,
 *   its partner gets the conjugate.  sigma = sqrt(N^2 / V * Power[k + N3 (j + N2 i)] / 2) for all of them (line 106). */
static void garfield(unsigned n, double vol, const double *power, orc_rng *rng, double *delta) {
  const size_t N = (size_t)n * n * n, nh = n / 2 + 1;
  double *G = (double *)calloc(2 * N, sizeof(double));
  random_grid_FS(n, rng, G);
  const double ps2dft_amp = (double)N * (double)N / vol; /* random.cpp:88-90 */
  const unsigned h = n / 2;
#define IX(a, b, c) (2 * ((size_t)(c) + (size_t)n * ((size_t)(b) + (size_t)n * (size_t)(a))))
  for (unsigned i = 0; i <= h; i++)
    for (unsigned j = 0; j <= h; j++)
      for (unsigned k = 0; k <= h; k++) {
        const double sigma = sqrt(ps2dft_amp * power[k + (size_t)n * (j + (size_t)n * i)] / 2.);
        const unsigned idx[3] = {i, j, k};
        int freeax[3], nfree = 0;
        for (int a = 0; a < 3; a++)
          if (idx[a] > 0 && idx[a] < h) freeax[nfree++] = a;
        if (nfree == 0) {
          double *c = G + IX(i, j, k);
          if (i == 0 && j == 0 && k == 0) {
            c[0] = 0.;
            c[1] = 0.;
          } else {
            c[0] *= sqrt(2.) * sigma;
            c[1] = 0.;
          }
          continue;
        }
        const int nrep = nfree == 3 ? 4 : (nfree == 2 ? 2 : 1);
        for (int r = 0; r < nrep; r++) {
          unsigned a[3] = {i, j, k};
          if (r > 0) { /* mirror one free axis: the r-th of three, or the first of two */
            const int ax = nfree == 3 ? freeax[r - 1] : freeax[0];
            a[ax] = n - a[ax];
          }
          unsigned b[3];
          for (int t = 0; t < 3; t++) b[t] = (n - a[t]) % n;
          double *rep = G + IX(a[0], a[1], a[2]), *par = G + IX(b[0], b[1], b[2]);
          rep[0] *= sigma;
          rep[1] *= sigma;
          par[0] = rep[0];
          par[1] = -rep[1];
        }
      }
  /* half-complex copy and fftC2R with its 1/N (random.cpp:497-511, fftwrapper.cc:44-46) */
  double *H = (double *)calloc(2 * (size_t)n * n * nh, sizeof(double));
  for (unsigned i = 0; i < n; i++)
    for (unsigned j = 0; j < n; j++)
      for (unsigned k = 0; k < nh; k++) {
        const size_t ih = 2 * ((size_t)k + nh * ((size_t)j + (size_t)n * i));
        H[ih] = G[IX(i, j, k)];
        H[ih + 1] = G[IX(i, j, k) + 1];
      }
#undef IX
  orc_fft_c2r_3d(n, n, n, H, delta);
  for (size_t t = 0; t < N; t++) delta[t] *= 1. / (double)N;
  free(G);
  free(H);
}

int orc_create_GARFIELD(unsigned n, double L, const double *power, unsigned long seed, double *delta) {
  if (!power || !delta || n < 2 || (n & 1)) return ORC_ERR_ARG;
  orc_rng r;
  rng_set(&r, seed);
  garfield(n, L * L * L, power, &r, delta);
  return ORC_OK;
}

/* draw_momenta (HMC_momenta.cc:42-74) + draw_real_space_momenta (76-94): the Fourier-space part first, then -- from
 * the SAME stream -- sqrt(mass_r) times one gaussian per cell in (i, j, k) order.  mass_f / mass_r may be NULL when the
 * mass type does not use them. */
int orc_draw_momenta(unsigned n, double L, int mass_fs, int mass_rs, const double *mass_f, const double *mass_r,
                     unsigned long seed, double *momenta) {
  if (!momenta || n < 2 || (n & 1) || (mass_fs && !mass_f) || (mass_rs && !mass_r)) return ORC_ERR_ARG;
  const size_t N = (size_t)n * n * n;
  orc_rng r;
  rng_set(&r, seed);
  if (mass_fs)
    garfield(n, L * L * L, mass_f, &r, momenta);
  else
    memset(momenta, 0, N * sizeof(double));
  if (mass_rs)
    for (size_t t = 0; t < N; t++) momenta[t] += sqrt(mass_r[t]) * ran_gaussian(&r, 1.);
  return ORC_OK;
}
