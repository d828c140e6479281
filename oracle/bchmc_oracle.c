/*
 * oracle/bchmc_oracle.c -- TEST INFRASTRUCTURE ONLY.  Not part of the product; see oracle/README.md.
 *
 * CPU restatement (plain C99) of Barcode's HMC leapfrog hot path.  Every function names the reference
 * file:line (under /root/reference/) it restates and keeps that function's loop structure, operation
 * order, scratch-buffer use and quirks, so that differences against the reference are bounded by FFT
 * round-off.  It is deliberately UNFUSED (12 FFTs per leapfrog step like the reference) and doubles as
 * the timed CPU baseline (`bench.py` cpu_baseline, kind "port").
 *
 * PARITY UNPINNED: no reference golden vectors exist for this path and the reference cannot be built in
 * this image (needs FFTW3 + GSL, both absent).  Self-checks that stand in: an independent numpy
 * restatement (oracle/np_restatement.py), finite-difference force checks, reversibility/energy tests.
 *
 * Build: see oracle/Makefile.  OpenMP (-fopenmp) is optional; golden vectors are made WITHOUT it
 * (deterministic scatter order, SURVEY section 7 "Hard parts").
 */
#include "bchmc_oracle.h"
#include "orc_fft.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define EPS_KSQ 1.e-14 /* define_opt.h:84 "eps" */

struct orc_hamil {
  orc_config c;
  unsigned N1, N2, N3;
  size_t N, Nhalf;
  double L1, L2, L3, vol, d1, d2, d3;
  int mass_fs, mass_rs;
  /* arrays (owned, N doubles each) */
  double *arr[ORC_F_COUNT];
  /* plan_pkg scratch: R2Cplan->{R,C}, C2Rplan->{R,C} (fftwrapper.h:28-46, init_par.cc:418-426) */
  double *R2C_R, *R2C_C, *C2R_R, *C2R_C;
  /* SPH stencil (init_par.cc:382-388) */
  int ncells;
  int *ci, *cj, *ck;
  /* kept for tests */
  double *grad_prior, *grad_like;
  double psi_prior, psi_likeli;
};

#define A(h, f) ((h)->arr[(f)])

/* ------------------------------------------------------------------------------------------------
 * small helpers (convenience.cc:20-229)
 * ---------------------------------------------------------------------------------------------- */
static double *dalloc(size_t n) { return (double *)calloc(n ? n : 1, sizeof(double)); }

static void multiply_factor_array(double f, const double *in, double *out, size_t n) {
#pragma omp parallel for
  for (long i = 0; i < (long)n; i++) out[i] = f * in[i];
}

static void copyArray(const double *in, double *out, size_t n) {
  if (in != out) memcpy(out, in, n * sizeof(double));
}

/* scale_space.cpp:41-51 */
static double calc_ki(unsigned i, double Li, unsigned Ni) {
  double kfac = 2. * M_PI / Li;
  if (i <= Ni / 2) return kfac * (double)i;
  return -kfac * (double)(Ni - i);
}

/* pacman.cpp:20-28 */
static void pacman_coordinate(double *x, double L) {
  if (*x < 0.) {
    *x = fmod(*x, L);
    *x += L;
  }
  if (*x >= L) *x = fmod(*x, L);
}

/* cosmo.cc:26-31 */
static double E_Hubble_a(double a, double OM, double OL) {
  double OK = 1. - OM - OL;
  return sqrt(OM / (a * a * a) + OK / (a * a) + OL);
}

/* cosmo.cc:182-217 */
double orc_fgrow(double a, double OM, double OL, int term) {
  double E = E_Hubble_a(a, OM, OL);
  double Omega = OM / ((E * E) * (a * a * a));
  double f = 0.;
  switch (term) {
    case 1: f = pow(Omega, 5. / 9.); break;
    case 2: f = 2. * pow(Omega, 6. / 11.); break;
    case 3: f = 3. * pow(Omega, 13. / 24.); break;
  }
  return f;
}

/* cosmo.cc:220-235 */
double orc_c_pecvel(double a, double OM, double OL, int term) {
  double H0 = 100.;
  double f = orc_fgrow(a, OM, OL, term);
  double E = E_Hubble_a(a, OM, OL);
  return f * H0 * E * a;
}

/* ------------------------------------------------------------------------------------------------
 * planned FFTs (fftwrapper.cc:88-119): forward unnormalised, inverse followed by 1/N
 * ---------------------------------------------------------------------------------------------- */
static void fftR2Cplanned(orc_hamil *h, const double *in, double *out) {
  if (in != h->R2C_R) copyArray(in, h->R2C_R, h->N);
  orc_fft_r2c_3d(h->N1, h->N2, h->N3, h->R2C_R, h->R2C_C);
  if (out != h->R2C_C) copyArray(h->R2C_C, out, 2 * h->Nhalf);
}

static void fftC2Rplanned(orc_hamil *h, const double *in, double *out) {
  if (in != h->C2R_C) copyArray(in, h->C2R_C, 2 * h->Nhalf);
  orc_fft_c2r_3d(h->N1, h->N2, h->N3, h->C2R_C, h->C2R_R);
  double fac = 1 / (double)h->N;
  multiply_factor_array(fac, h->C2R_R, out, h->N);
}

/* ------------------------------------------------------------------------------------------------
 * SPH stencil (SPH_kernel.cpp:62-102, 110-139)
 * ---------------------------------------------------------------------------------------------- */
static void SPH_kernel_3D_cells(orc_hamil *h) {
  double reach = h->c.particle_kernel_h * 2; /* SPH_kernel.cpp:16-28 */
  int r1 = (int)(reach / h->d1) + 1, r2 = (int)(reach / h->d2) + 1, r3 = (int)(reach / h->d3) + 1;
  double reach_sq = reach * reach;
  int cap = (2 * r1 + 1) * (2 * r2 + 1) * (2 * r3 + 1);
  h->ci = (int *)malloc(sizeof(int) * cap);
  h->cj = (int *)malloc(sizeof(int) * cap);
  h->ck = (int *)malloc(sizeof(int) * cap);
  int n = 0;
  for (int i1 = -r1; i1 <= r1; ++i1)
    for (int i2 = -r2; i2 <= r2; ++i2)
      for (int i3 = -r3; i3 <= r3; ++i3) {
        double dx = (fabs((double)i1) - 0.5) * h->d1;
        double dy = (fabs((double)i2) - 0.5) * h->d2;
        double dz = (fabs((double)i3) - 0.5) * h->d3;
        double r_sq = dx * dx + dy * dy + dz * dz;
        if (r_sq <= reach_sq) {
          h->ci[n] = i1;
          h->cj[n] = i2;
          h->ck[n] = i3;
          ++n;
        }
      }
  h->ncells = n;
}

typedef struct {
  int n;
  int *i, *j, *kb, *kl;
} hull_t;

static void SPH_kernel_3D_cells_hull_1(const orc_hamil *h, hull_t *u) {
  int N = h->ncells;
  u->i = (int *)malloc(sizeof(int) * N);
  u->j = (int *)malloc(sizeof(int) * N);
  u->kb = (int *)malloc(sizeof(int) * N);
  u->kl = (int *)malloc(sizeof(int) * N);
  u->n = 0;
  for (int ix = 0; ix < N; ++ix) {
    int dup = -1;
    for (int m = 0; m < u->n; ++m)
      if (u->i[m] == h->ci[ix] && u->j[m] == h->cj[ix]) {
        dup = m;
        break;
      }
    if (dup < 0) {
      u->i[u->n] = h->ci[ix];
      u->j[u->n] = h->cj[ix];
      u->kb[u->n] = h->ck[ix];
      u->kl[u->n] = h->ck[ix];
      u->n++;
    } else {
      if (u->kb[dup] > h->ck[ix]) u->kb[dup] = h->ck[ix];
      if (u->kl[dup] < h->ck[ix]) u->kl[dup] = h->ck[ix];
    }
  }
}

static void hull_free(hull_t *u) {
  free(u->i);
  free(u->j);
  free(u->kb);
  free(u->kl);
}

/* ------------------------------------------------------------------------------------------------
 * lifecycle
 * ---------------------------------------------------------------------------------------------- */
size_t orc_sizeof_config(void) { return sizeof(orc_config); }

/* Thread count of the OpenMP build (no-op in the serial one).  The harness passes the host's CPU SHARE: on a GPU box
 * whose cgroup grants 16 CPUs of 256 visible ones, 256 threads spin against the quota and a 64^3 step takes 18 s
 * instead of 0.3 s. */
#ifdef _OPENMP
#include <omp.h>
void orc_set_threads(int n) { if (n > 0) omp_set_num_threads(n); }
int orc_get_max_threads(void) { return omp_get_max_threads(); }
#else
void orc_set_threads(int n) { (void)n; }
int orc_get_max_threads(void) { return 1; }
#endif

int orc_create(const orc_config *cfg, orc_hamil **out) {
  if (!cfg || !out || cfg->N1 < 2 || !(cfg->L1 > 0)) return ORC_ERR_ARG;
  orc_hamil *h = (orc_hamil *)calloc(1, sizeof(orc_hamil));
  h->c = *cfg;
  h->N1 = h->N2 = h->N3 = cfg->N1;
  h->N = (size_t)h->N1 * h->N2 * h->N3;
  h->Nhalf = (size_t)h->N1 * h->N2 * (h->N3 / 2 + 1);
  h->L1 = h->L2 = h->L3 = cfg->L1;
  h->vol = h->L1 * h->L2 * h->L3;
  h->d1 = h->L1 / (double)h->N1;
  h->d2 = h->L2 / (double)h->N2;
  h->d3 = h->L3 / (double)h->N3;
  switch (cfg->mass_type) { /* struct_hamil.h:272-313 */
    case 0: case 6: case 60: h->mass_rs = 1; h->mass_fs = 0; break;
    case 1: case 2: case 3: case 4: h->mass_rs = 0; h->mass_fs = 1; break;
    case 5: h->mass_rs = 1; h->mass_fs = 1; break;
    default: free(h); return ORC_ERR_MASS_TYPE;
  }
  for (int f = 0; f < ORC_F_COUNT; f++) A(h, f) = dalloc(h->N);
  h->R2C_R = dalloc(h->N);
  h->C2R_R = dalloc(h->N);
  h->R2C_C = dalloc(2 * h->Nhalf);
  h->C2R_C = dalloc(2 * h->Nhalf);
  h->grad_prior = dalloc(h->N);
  h->grad_like = dalloc(h->N);
  SPH_kernel_3D_cells(h);
  *out = h;
  return ORC_OK;
}

void orc_destroy(orc_hamil *h) {
  if (!h) return;
  for (int f = 0; f < ORC_F_COUNT; f++) free(A(h, f));
  free(h->R2C_R); free(h->C2R_R); free(h->R2C_C); free(h->C2R_C);
  free(h->grad_prior); free(h->grad_like);
  free(h->ci); free(h->cj); free(h->ck);
  free(h);
}

int orc_set_array(orc_hamil *h, int field, const double *src) {
  if (!h || field < 0 || field >= ORC_F_COUNT || !src) return ORC_ERR_ARG;
  memcpy(A(h, field), src, h->N * sizeof(double));
  return ORC_OK;
}

double *orc_get_array(orc_hamil *h, int field) {
  if (!h || field < 0 || field >= ORC_F_COUNT) return NULL;
  return A(h, field);
}

int orc_stencil(orc_hamil *h, int *n, const int **ci, const int **cj, const int **ck) {
  *n = h->ncells; *ci = h->ci; *cj = h->cj; *ck = h->ck;
  return ORC_OK;
}

double *orc_last_grad_prior(orc_hamil *h) { return h->grad_prior; }
double *orc_last_grad_like(orc_hamil *h) { return h->grad_like; }

/* ------------------------------------------------------------------------------------------------
 * a3: HMC_help.cc:16-64   out = IFFT[ normFS / C(k) * FFT[signal] ],  C<=0 -> 0
 * NB the spectrum array is a full N-grid indexed k + N3*(j + N2*i) with k <= N3/2 (line 44).
 * ---------------------------------------------------------------------------------------------- */
int orc_convolveInvCorrFuncWithSignal(orc_hamil *h, const double *signal, double *out, const double *corr) {
  const unsigned N3half = h->N3 / 2 + 1;
  const double normFS = h->vol / (double)h->N; /* FOURIER_DEF_2 */
  fftR2Cplanned(h, signal, h->R2C_C);
#pragma omp parallel for
  for (long i = 0; i < (long)h->N1; i++)
    for (unsigned j = 0; j < h->N2; j++)
      for (unsigned k = 0; k < N3half; ++k) {
        size_t ix = k + (size_t)h->N3 * (j + (size_t)h->N2 * i);
        size_t ix_C = k + (size_t)N3half * (j + (size_t)h->N2 * i);
        double invC_normFS;
        if (corr[ix] > 0.0)
          invC_normFS = normFS / corr[ix];
        else
          invC_normFS = 0.;
        h->R2C_C[2 * ix_C] *= invC_normFS;
        h->R2C_C[2 * ix_C + 1] *= invC_normFS;
      }
  fftC2Rplanned(h, h->R2C_C, out);
  return ORC_OK;
}

/* a4: hmc/prior/gaussian.cpp:15-18 */
int orc_grad_log_prior(orc_hamil *h, const double *signal, double *out) {
  return orc_convolveInvCorrFuncWithSignal(h, signal, out, A(h, ORC_F_SIGNAL_PS));
}

/* a4: hmc/prior/gaussian.cpp:20-35 */
int orc_log_prior(orc_hamil *h, const double *signal, double *value) {
  double *dummy = dalloc(h->N);
  orc_convolveInvCorrFuncWithSignal(h, signal, dummy, A(h, ORC_F_SIGNAL_PS));
  double psi_prior = 0.;
#pragma omp parallel for reduction(+ : psi_prior)
  for (long i = 0; i < (long)h->N; i++) psi_prior += 0.5 * signal[i] * dummy[i];
  free(dummy);
  *value = psi_prior;
  return ORC_OK;
}

/* ------------------------------------------------------------------------------------------------
 * a7: EqSolvers.cc:168-277 (zeropad=false, norm=false -> cpecvel = 1)
 * ---------------------------------------------------------------------------------------------- */
int orc_theta2vel(orc_hamil *h, const double *delta, double *vex, double *vey, double *vez) {
  const unsigned N1 = h->N1, N2 = h->N2, N3 = h->N3;
  const unsigned N3half = N3 / 2 + 1;
  const double cpecvel = 1.;
  double *velx = dalloc(2 * h->Nhalf), *vely = dalloc(2 * h->Nhalf);
  double *C = h->R2C_C;
  fftR2Cplanned(h, delta, C);
#pragma omp parallel for
  for (long i = 0; i < (long)N1; i++) {
    double kx = calc_ki((unsigned)i, h->L1, N1);
    for (unsigned j = 0; j < N2; j++) {
      double ky = calc_ki(j, h->L2, N2);
      for (unsigned k = 0; k < N3half; ++k) {
        double kz = calc_ki(k, h->L3, N3);
        double ksq = kx * kx + ky * ky + kz * kz;
        size_t ix = k + (size_t)N3half * (j + (size_t)N2 * i);
        if (ksq > EPS_KSQ) {
          double fac_kern = cpecvel / ksq;
          double dr = C[2 * ix], di = C[2 * ix + 1];
          double fx = fac_kern * kx;
          velx[2 * ix] = fx * di;
          velx[2 * ix + 1] = fx * -dr;
          double fy = fac_kern * ky;
          vely[2 * ix] = fy * di;
          vely[2 * ix + 1] = fy * -dr;
          double fz = fac_kern * kz;
          C[2 * ix] = fz * di;
          C[2 * ix + 1] = fz * -dr;
        } else {
          velx[2 * ix] = velx[2 * ix + 1] = 0;
          vely[2 * ix] = vely[2 * ix + 1] = 0;
          C[2 * ix] = C[2 * ix + 1] = 0;
        }
        if (((unsigned)i == N1 / 2) || (j == N2 / 2) || (k == N3 / 2)) { /* Nyquist planes, 254-265 */
          velx[2 * ix] = velx[2 * ix + 1] = 0.;
          vely[2 * ix] = vely[2 * ix + 1] = 0.;
          C[2 * ix] = C[2 * ix + 1] = 0;
        }
      }
    }
  }
  fftC2Rplanned(h, C, vez); /* must go first, C2R destroys input (274) */
  fftC2Rplanned(h, velx, vex);
  fftC2Rplanned(h, vely, vey);
  free(velx);
  free(vely);
  return ORC_OK;
}

/* a8: disp_part.cc:34-157 (facL=1, reggrid=true, periodic=true) */
static void disp_part(orc_hamil *h, double *posx, double *posy, double *posz, const double *psix, const double *psiy,
                      const double *psiz) {
  const unsigned N1 = h->N1, N2 = h->N2, N3 = h->N3;
  for (unsigned i = 0; i < N1; i++)
    for (unsigned j = 0; j < N2; j++)
      for (unsigned k = 0; k < N3; k++) {
        size_t jj = k + (size_t)N3 * (j + (size_t)N2 * i);
        double rx = 0.5 * h->d1, ry = 0.5 * h->d2, rz = 0.5 * h->d3;
        posx[jj] = h->d1 * (double)i + rx;
        posy[jj] = h->d2 * (double)j + ry;
        posz[jj] = h->d3 * (double)k + rz;
      }
  for (size_t n = 0; n < h->N; n++) { /* add_to_array, 103-106 */
    posx[n] += psix[n];
    posy[n] += psiy[n];
    posz[n] += psiz[n];
  }
  if (h->c.periodic) /* module-level `periodic = true`, disp_part.cc:28 */
    for (size_t n = 0; n < h->N; n++) {
      pacman_coordinate(&posx[n], h->L1);
      pacman_coordinate(&posy[n], h->L2);
      pacman_coordinate(&posz[n], h->L3);
    }
}

/* a9: rsd.cc:18-68 */
static int calc_pos_rsd(orc_hamil *h, double *x, double *y, double *z, const double *vx, const double *vy,
                        const double *vz) {
  (void)vx; (void)vy; (void)x; (void)y;
  const double ascale = h->c.ascale, OM = h->c.OM, OL = h->c.OL;
  double Omega_C = 1. - OM - OL;
  double Hub = 100. * sqrt(OM / ascale / ascale / ascale + OL + Omega_C / ascale / ascale);
  if (!h->c.planepar) return ORC_ERR_RSD_NOT_PLANEPAR; /* periodic && !planepar throws (60-62) */
#pragma omp parallel for
  for (long i = 0; i < (long)h->N; ++i) {
    double v_norm = 1. / Hub / ascale;
    double ruxv = vz[i] * v_norm;
    z[i] = z[i] + ruxv;
    if (h->c.periodic) pacman_coordinate(&z[i], h->L3);
  }
  return ORC_OK;
}

/* a10: massFunctions.cc:366-384 */
static double SPH_kernel_3D(double r, double hh) {
  double result = 0.;
  double q = r / hh;
  if (q <= 1.)
    result = 1. / M_PI / (hh * hh * hh) * (1 - 3. / 2 * q * q + 3. / 4 * q * q * q);
  else if (q <= 2.)
    result = 1. / M_PI / (hh * hh * hh) * (1. / 4 * ((2. - q) * (2. - q) * (2. - q)));
  return result;
}

static inline void atomic_add(double *p, double v) {
#pragma omp atomic
  *p += v;
}

/* a10: massFunctions.cc:392-495 (weightmass with unit masses) */
static void getDensity_SPH(orc_hamil *h, const double *xp, const double *yp, const double *zp, double *delta) {
  const size_t N1 = h->N1, N2 = h->N2, N3 = h->N3;
  const double d1 = h->d1, d2 = h->d2, d3 = h->d3, kernel_h = h->c.particle_kernel_h;
  const double min1 = h->c.min1, min2 = h->c.min2, min3 = h->c.min3;
  memset(delta, 0, sizeof(double) * h->N);
  int reach1 = (int)(2 * kernel_h / d1) + 1, reach2 = (int)(2 * kernel_h / d2) + 1, reach3 = (int)(2 * kernel_h / d3) + 1;
#pragma omp parallel for
  for (long n = 0; n < (long)h->N; n++) {
    if ((xp[n] >= min1 && xp[n] < min1 + h->L1) && (yp[n] >= min2 && yp[n] < min2 + h->L2) &&
        (zp[n] >= min3 && zp[n] < min3 + h->L3)) {
      double mass = 1.; /* dummyL == 1 (Lag2Eul.cc:102-103) */
      size_t ix = (size_t)(xp[n] / d1), iy = (size_t)(yp[n] / d2), iz = (size_t)(zp[n] / d3);
      double ccx = ((double)ix + 0.5) * d1, ccy = ((double)iy + 0.5) * d2, ccz = ((double)iz + 0.5) * d3;
      for (int i1 = -reach1; i1 <= reach1; ++i1)
        for (int i2 = -reach2; i2 <= reach2; ++i2)
          for (int i3 = -reach3; i3 <= reach3; ++i3) {
            double cx = ccx + (double)i1 * d1, cy = ccy + (double)i2 * d2, cz = ccz + (double)i3 * d3;
            size_t kx = ((size_t)((long)N1 + i1) + ix) % N1;
            size_t ky = ((size_t)((long)N2 + i2) + iy) % N2;
            size_t kz = ((size_t)((long)N3 + i3) + iz) % N3;
            size_t index = kz + N3 * (ky + N2 * kx);
            double dx = xp[n] - cx, dy = yp[n] - cy, dz = zp[n] - cz;
            double r = sqrt(dx * dx + dy * dy + dz * dz);
            if (r / kernel_h <= 2.) atomic_add(&delta[index], SPH_kernel_3D(r, kernel_h) * mass);
          }
    }
  }
}

/* interpolate_grid.cpp:27-50, 55-79 */
static void getCICcells(orc_hamil *h, double x, double y, double z, size_t *c1, size_t *c2) {
  double xpos = x - 0.5 * h->d1, ypos = y - 0.5 * h->d2, zpos = z - 0.5 * h->d3;
  pacman_coordinate(&xpos, h->L1);
  pacman_coordinate(&ypos, h->L2);
  pacman_coordinate(&zpos, h->L3);
  c1[0] = (size_t)(xpos / h->d1);
  c1[1] = (size_t)(ypos / h->d2);
  c1[2] = (size_t)(zpos / h->d3);
  c1[0] = (c1[0] + h->N1) % h->N1;
  c1[1] = (c1[1] + h->N2) % h->N2;
  c1[2] = (c1[2] + h->N3) % h->N3;
  c2[0] = (c1[0] + 1) % h->N1;
  c2[1] = (c1[1] + 1) % h->N2;
  c2[2] = (c1[2] + 1) % h->N3;
}

static void getCICweights(orc_hamil *h, double x, double y, double z, const size_t *c1, double *dx, double *tx) {
  double xpos = x - 0.5 * h->d1, ypos = y - 0.5 * h->d2, zpos = z - 0.5 * h->d3;
  pacman_coordinate(&xpos, h->L1);
  pacman_coordinate(&ypos, h->L2);
  pacman_coordinate(&zpos, h->L3);
  dx[0] = xpos / h->d1 - (double)c1[0];
  dx[1] = ypos / h->d2 - (double)c1[1];
  dx[2] = zpos / h->d3 - (double)c1[2];
  tx[0] = 1. - dx[0];
  tx[1] = 1. - dx[1];
  tx[2] = 1. - dx[2];
}

/* massFunctions.cc:100-164 */
static void getDensity_CIC(orc_hamil *h, const double *xp, const double *yp, const double *zp, double *delta) {
  const size_t N2 = h->N2, N3 = h->N3;
  const double min1 = h->c.min1, min2 = h->c.min2, min3 = h->c.min3;
  memset(delta, 0, sizeof(double) * h->N);
#define DELTA(a, b, c) delta[(c)[2] + N3 * ((b)[1] + N2 * (a)[0])]
#pragma omp parallel for
  for (long n = 0; n < (long)h->N; n++) {
    double dx[3], tx[3];
    size_t i[3], ii[3];
    if ((xp[n] >= min1 && xp[n] < min1 + h->L1) && (yp[n] >= min2 && yp[n] < min2 + h->L2) &&
        (zp[n] >= min3 && zp[n] < min3 + h->L3)) {
      getCICcells(h, xp[n], yp[n], zp[n], i, ii);
      getCICweights(h, xp[n], yp[n], zp[n], i, dx, tx);
      double mass = 1.;
      atomic_add(&DELTA(i, i, i), mass * tx[0] * tx[1] * tx[2]);
      atomic_add(&DELTA(ii, i, i), mass * dx[0] * tx[1] * tx[2]);
      atomic_add(&DELTA(i, ii, i), mass * tx[0] * dx[1] * tx[2]);
      atomic_add(&DELTA(i, i, ii), mass * tx[0] * tx[1] * dx[2]);
      atomic_add(&DELTA(ii, ii, i), mass * dx[0] * dx[1] * tx[2]);
      atomic_add(&DELTA(ii, i, ii), mass * dx[0] * tx[1] * dx[2]);
      atomic_add(&DELTA(i, ii, ii), mass * tx[0] * dx[1] * dx[2]);
      atomic_add(&DELTA(ii, ii, ii), mass * dx[0] * dx[1] * dx[2]);
    }
  }
#undef DELTA
}

/* massFunctions.cc:49-98 */
static void getDensity_NGP(orc_hamil *h, const double *xp, const double *yp, const double *zp, double *delta) {
  const unsigned N1 = h->N1, N2 = h->N2, N3 = h->N3;
  const double min1 = h->c.min1, min2 = h->c.min2, min3 = h->c.min3;
  memset(delta, 0, sizeof(double) * h->N);
#pragma omp parallel for
  for (long n = 0; n < (long)h->N; n++) {
    if ((xp[n] >= min1 && xp[n] < min1 + h->L1) && (yp[n] >= min2 && yp[n] < min2 + h->L2) &&
        (zp[n] >= min3 && zp[n] < min3 + h->L3)) {
      unsigned i = (unsigned)floor((xp[n] - min1) / h->d1);
      unsigned j = (unsigned)floor((yp[n] - min2) / h->d2);
      unsigned k = (unsigned)floor((zp[n] - min3) / h->d3);
      i = (unsigned)fmod((double)i, (double)N1);
      j = (unsigned)fmod((double)j, (double)N2);
      k = (unsigned)fmod((double)k, (double)N3);
      atomic_add(&delta[k + (size_t)N3 * (j + (size_t)N2 * i)], 1.);
    }
  }
}

/* massFunctions.cc:167-364 (27 cloud weights; note the inclusive `<= min+L` domain test at 195) */
static void getDensity_TSC(orc_hamil *h, const double *xp, const double *yp, const double *zp, double *delta) {
  const unsigned N1 = h->N1, N2 = h->N2, N3 = h->N3;
  const double min1 = h->c.min1, min2 = h->c.min2, min3 = h->c.min3;
  memset(delta, 0, sizeof(double) * h->N);
#pragma omp parallel for
  for (long n = 0; n < (long)h->N; n++) {
    if ((xp[n] >= min1 && xp[n] <= min1 + h->L1) && (yp[n] >= min2 && yp[n] <= min2 + h->L2) &&
        (zp[n] >= min3 && zp[n] <= min3 + h->L3)) {
      unsigned c[3][3]; /* [axis][0: -1, 1: 0, 2: +1] */
      double w[3][3];
      const double pos[3] = {(xp[n] - min1) / h->d1, (yp[n] - min2) / h->d2, (zp[n] - min3) / h->d3};
      const unsigned NN[3] = {N1, N2, N3};
      for (int a = 0; a < 3; a++) {
        unsigned i = (unsigned)floor(pos[a]);
        i = (unsigned)fmod((double)i, (double)NN[a]);
        c[a][1] = i;
        c[a][2] = (unsigned)fmod((double)(i + 1), (double)NN[a]);
        c[a][0] = (unsigned)fmod((double)(i - 1 + NN[a]), (double)NN[a]);
        double xc = (double)(i + 0.5);
        double dx = pos[a] - xc;
        w[a][1] = 0.75 - dx * dx;
        w[a][2] = 0.5 * (0.5 + dx) * (0.5 + dx);
        w[a][0] = 0.5 * (0.5 - dx) * (0.5 - dx);
      }
      double mass = 1.;
      for (int a = 0; a < 3; a++)
        for (int b = 0; b < 3; b++)
          for (int d = 0; d < 3; d++)
            atomic_add(&delta[c[2][d] + (size_t)N3 * (c[1][b] + (size_t)N2 * c[0][a])],
                       mass * w[0][a] * w[1][b] * w[2][d]);
    }
  }
}

int orc_getDensity(orc_hamil *h, int mk, const double *xp, const double *yp, const double *zp, double *rho) {
  switch (mk) { /* Lag2Eul.cc:114-128 */
    case 0: getDensity_NGP(h, xp, yp, zp, rho); break;
    case 1: getDensity_CIC(h, xp, yp, zp, rho); break;
    case 2: getDensity_TSC(h, xp, yp, zp, rho); break;
    case 3: getDensity_SPH(h, xp, yp, zp, rho); break;
    default: return ORC_ERR_ARG;
  }
  return ORC_OK;
}

/* a11: massFunctions.cc:30-47 */
void orc_overdens(orc_hamil *h, const double *in, double *out) {
  double nmeanD = 0.;
#pragma omp parallel for reduction(+ : nmeanD)
  for (long i = 0; i < (long)h->N; i++) nmeanD += in[i];
  double nmean = nmeanD / (double)h->N;
#pragma omp parallel for
  for (long i = 0; i < (long)h->N; i++) out[i] = in[i] / nmean - 1.;
}

int orc_alpt_displacement(orc_hamil *h, const double *in, double *psix, double *psiy, double *psiz);

/* ------------------------------------------------------------------------------------------------
 * a6: Lag2Eul.cc:69-132 (Zel'dovich), 338-424 (Zel'dovich + plane-parallel RSD), dispatcher 318-332.
 * `in` may alias C2R_R (as in HMC_models.cc:383-405); `out` receives delta_x.
 * ---------------------------------------------------------------------------------------------- */
int orc_Lag2Eul(orc_hamil *h, const double *in, double *out, double *posx, double *posy, double *posz, int use_rsd) {
  const size_t N = h->N;
  int rc = ORC_OK;
  double *psix = dalloc(N), *psiy = dalloc(N), *psiz = dalloc(N);
  if (!use_rsd && h->c.sfmodel != 1) { /* Lag2Eul_non_zeldovich, Lag2Eul.cc:138-312 (dispatcher 325-331) */
    copyArray(in, out, N); /* 166: `dummy` is the output array */
    orc_alpt_displacement(h, out, psix, psiy, psiz);
    disp_part(h, posx, posy, posz, psix, psiy, psiz);
  } else if (!use_rsd) {
    multiply_factor_array(-h->c.D1, in, out, N); /* 88 */
    orc_theta2vel(h, out, psix, psiy, psiz);
    disp_part(h, posx, posy, posz, psix, psiy, psiz);
  } else {
    double *vex = dalloc(N), *vey = dalloc(N), *vez = dalloc(N);
    multiply_factor_array(-h->c.D1, in, h->R2C_R, N); /* 361 */
    orc_theta2vel(h, h->R2C_R, psix, psiy, psiz);
    double cpecvel = orc_c_pecvel(h->c.ascale, h->c.OM, h->c.OL, 1);
    multiply_factor_array(cpecvel, psix, vex, N);
    multiply_factor_array(cpecvel, psiy, vey, N);
    multiply_factor_array(cpecvel, psiz, vez, N);
    disp_part(h, posx, posy, posz, psix, psiy, psiz);
    rc = calc_pos_rsd(h, posx, posy, posz, vex, vey, vez);
    free(vex); free(vey); free(vez);
  }
  if (rc == ORC_OK) rc = orc_getDensity(h, h->c.mk, posx, posy, posz, out);
  if (rc == ORC_OK) orc_overdens(h, out, out);
  free(psix); free(psiy); free(psiz);
  return rc;
}

/* ------------------------------------------------------------------------------------------------
 * a13: per-cell d(-log L)/d delta_x
 * ---------------------------------------------------------------------------------------------- */
int orc_partial_f_delta_x_log_like(orc_hamil *h, const double *deltaX, double *out) {
  const double *window = A(h, ORC_F_WINDOW), *nobs = A(h, ORC_F_NOBS), *noise = A(h, ORC_F_NOISE);
  const double rho_c = h->c.rho_c, biasP = h->c.biasP, biasE = h->c.biasE;
  switch (h->c.likelihood) {
    case 1: /* gaussian_independent.cpp:24-42 */
#pragma omp parallel for
      for (long i = 0; i < (long)h->N; i++) {
        double Lambda = window[i] * rho_c * pow(1. + biasP * deltaX[i], biasE);
        if ((window[i] > 0.) && (Lambda > 0.0)) {
          double resid = nobs[i] - Lambda;
          out[i] = resid / (noise[i] * noise[i]);
        } else
          out[i] = 0.0;
      }
      break;
    case 0: /* poissonian.cpp:19-34 */
      for (size_t i = 0; i < h->N; i++) {
        double dens = 1. + biasP * deltaX[i];
        double Lambda = window[i] * rho_c * pow(dens, biasE);
        if ((window[i] > 0.0) && (dens > 0.0))
          out[i] = (1 - nobs[i] / Lambda) * rho_c * biasE * biasP * pow(dens, biasE - 1);
        else
          out[i] = 0.0;
      }
      break;
    case 2: /* lognormal_independent.cpp:40-55 */
#pragma omp parallel for
      for (long i = 0; i < (long)h->N; i++) {
        double Lambda = log(rho_c * pow(1. + biasP * deltaX[i], biasE));
        if (window[i] > 0.)
          out[i] = (nobs[i] - Lambda) / (noise[i] * noise[i]);
        else
          out[i] = 0.0;
      }
      break;
    case 3: /* gaussian_random_field.cpp:21-23: empty body, output untouched */
      break;
    default: return ORC_ERR_ARG;
  }
  return ORC_OK;
}

/* a14: pacman.cpp:73-86 */
static void pad_array_pacman(const double *input, unsigned N1_in, double *out, unsigned padding) {
  unsigned N1_out = N1_in + 2 * padding;
  for (unsigned io = 0; io < N1_out; ++io)
    for (unsigned jo = 0; jo < N1_out; ++jo)
      for (unsigned ko = 0; ko < N1_out; ++ko) {
        size_t ix_out = ko + (size_t)N1_out * (jo + (size_t)N1_out * io);
        unsigned ii = (unsigned)((int)(io + N1_in) - (int)padding) % N1_in;
        unsigned ji = (unsigned)((int)(jo + N1_in) - (int)padding) % N1_in;
        unsigned ki = (unsigned)((int)(ko + N1_in) - (int)padding) % N1_in;
        size_t ix_in = ki + (size_t)N1_in * (ji + (size_t)N1_in * ii);
        out[ix_out] = input[ix_in];
      }
}

/* SPH_kernel.cpp:148-208 */
static inline void grad_SPH_kernel_3D_h_units(double x_h, double y_h, double z_h, double norm, double *ox, double *oy,
                                              double *oz) {
  double q_sq = x_h * x_h + y_h * y_h + z_h * z_h;
  double partial;
  if (q_sq > 4)
    partial = 0.;
  else if (q_sq > 1) {
    double q = sqrt(q_sq);
    double qmin2 = q - 2;
    partial = -0.75 * qmin2 * qmin2 * norm / q;
  } else {
    double q = sqrt(q_sq);
    partial = (2.25 * q - 3) * norm;
  }
  *ox = partial * x_h;
  *oy = partial * y_h;
  *oz = partial * z_h;
}

/* a14: HMC_models.cc:200-303 with inner loop 77-128.  out_x may alias part_like (the reference passes
 * R2Cplan->R for both, 336-337): the padded copy is taken first. */
int orc_likelihood_calc_V_SPH(orc_hamil *h, const double *part_like, const double *posx, const double *posy,
                              const double *posz, double *out_x, double *out_y, double *out_z) {
  hull_t u;
  SPH_kernel_3D_cells_hull_1(h, &u);
  int maxi = h->ci[0];
  for (int m = 1; m < h->ncells; m++)
    if (h->ci[m] > maxi) maxi = h->ci[m];
  const unsigned padding = (unsigned)maxi;
  const unsigned N2 = h->N2, N3 = h->N3;
  const size_t N3pad = N3 + 2 * padding, N2pad = N2 + 2 * padding;
  double *padded = dalloc((size_t)(h->N1 + 2 * padding) * N2pad * N3pad);
  pad_array_pacman(part_like, h->N1, padded, padding);

  const double normalize = h->c.rho_c * h->L1 * h->L2 * h->L3 / (double)((size_t)h->N1 * h->N2 * h->N3);
  double f1 = 0.;
  const int rsd_model = h->c.rsd_model;
  if (rsd_model) f1 = orc_fgrow(h->c.ascale, h->c.OM, h->c.OL, 1);
  if (rsd_model && !h->c.planepar) {
    free(padded);
    hull_free(&u);
    return ORC_ERR_RSD_NOT_PLANEPAR;
  }
  const double hh = h->c.particle_kernel_h;
  const double h_sq = hh * hh, h_inv = 1. / hh;
  const double norm = 1. / (M_PI * h_sq * h_sq);
  const double d1 = h->d1, d2 = h->d2, d3 = h->d3;
  const double d1_h = d1 * h_inv, d2_h = d2 * h_inv, d3_h = d3 * h_inv;

#pragma omp parallel for
  for (long j = 0; j < (long)h->N; j++) {
    double px = posx[j], py = posy[j], pz = posz[j];
    /* Not in the reference: a non-finite position (blown-up trajectory) would index out of bounds below
     * (undefined behaviour upstream); the oracle and the engine both give such a particle V = 0. */
    if (!(px >= 0. && px <= h->L1 && py >= 0. && py <= h->L2 && pz >= 0. && pz <= h->L3)) {
      out_x[j] = out_y[j] = out_z[j] = 0.;
      continue;
    }
    int ix = (int)(px / d1), iy = (int)(py / d2), iz = (int)(pz / d3);
    double ccx_h = ((double)ix + 0.5) * d1_h, ccy_h = ((double)iy + 0.5) * d2_h, ccz_h = ((double)iz + 0.5) * d3_h;
    double dpcx_h = px * h_inv - ccx_h, dpcy_h = py * h_inv - ccy_h, dpcz_h = pz * h_inv - ccz_h;
    double ox = 0., oy = 0., oz = 0.;
    unsigned ix_pad = (unsigned)ix + padding, iy_pad = (unsigned)iy + padding, iz_pad = (unsigned)iz + padding;
    for (int m = 0; m < u.n; ++m) {
      int i1 = u.i[m], i2 = u.j[m];
      unsigned kx = (unsigned)((int)ix_pad + i1), ky = (unsigned)((int)iy_pad + i2);
      double diff_x_h = dpcx_h - (double)i1 * d1_h;
      double diff_y_h = dpcy_h - (double)i2 * d2_h;
      size_t index_xy_part = N3pad * (ky + N2pad * kx);
      int kz_begin = u.kb[m], kz_last = u.kl[m];
      size_t index_begin = (size_t)(kz_begin + (long)(index_xy_part + iz_pad));
      size_t index_end = index_begin + (size_t)(kz_last - kz_begin);
      double diff_z_h = dpcz_h - (double)kz_begin * d3_h;
      for (size_t index = index_begin; index <= index_end; ++index) {
        double common_part = padded[index];
        double gx, gy, gz;
        grad_SPH_kernel_3D_h_units(diff_x_h, diff_y_h, diff_z_h, norm, &gx, &gy, &gz);
        ox += common_part * gx;
        oy += common_part * gy;
        oz += common_part * gz;
        diff_z_h -= d3_h;
      }
    }
    out_x[j] = normalize * ox;
    out_y[j] = normalize * oy;
    out_z[j] = normalize * oz;
    if (rsd_model) out_z[j] += f1 * out_z[j];
  }
  free(padded);
  hull_free(&u);
  return ORC_OK;
}

/* interpolate_grid.cpp:134-191 -- bug-for-bug: wx[2] and wy[2] use dz (lines 166-168) */
static double interpolate_TSC_one(orc_hamil *h, double xp, double yp, double zp, const double *field) {
  const unsigned N1 = h->N1, N2 = h->N2, N3 = h->N3;
  double output = 0;
  double xk = xp / h->d1, yk = yp / h->d2, zk = zp / h->d3;
  unsigned cx = (unsigned)xk, cy = (unsigned)yk, cz = (unsigned)zk;
  double dx = xk - ((double)cx + 0.5), dy = yk - ((double)cy + 0.5), dz = zk - ((double)cz + 0.5);
  double wx[3], wy[3], wz[3];
  wx[1] = 0.75 - dx * dx;
  wy[1] = 0.75 - dy * dy;
  wz[1] = 0.75 - dz * dz;
  wx[0] = 0.5 * ((1.5 - fabs(dx + 1)) * (1.5 - fabs(dx + 1)));
  wy[0] = 0.5 * ((1.5 - fabs(dy + 1)) * (1.5 - fabs(dy + 1)));
  wz[0] = 0.5 * ((1.5 - fabs(dz + 1)) * (1.5 - fabs(dz + 1)));
  wx[2] = 0.5 * ((1.5 - fabs(dz - 1)) * (1.5 - fabs(dz - 1)));
  wy[2] = 0.5 * ((1.5 - fabs(dz - 1)) * (1.5 - fabs(dz - 1)));
  wz[2] = 0.5 * ((1.5 - fabs(dz - 1)) * (1.5 - fabs(dz - 1)));
  unsigned ixx[3], ixy[3], ixz[3];
  ixx[1] = cx; ixy[1] = cy; ixz[1] = cz;
  ixx[0] = (cx - 1 + N1) % N1; ixy[0] = (cy - 1 + N2) % N2; ixz[0] = (cz - 1 + N3) % N3;
  ixx[2] = (cx + 1) % N1; ixy[2] = (cy + 1) % N2; ixz[2] = (cz + 1) % N3;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
      for (int k = 0; k < 3; ++k) {
        size_t ix_f = ((size_t)ixx[i] * N2 + ixy[j]) * N3 + ixz[k];
        output += wx[i] * wy[j] * wz[k] * field[ix_f];
      }
  return output;
}

static void interpolate_TSC(orc_hamil *h, const double *xp, const double *yp, const double *zp, const double *field,
                            double *interp) {
#pragma omp parallel for
  for (long n = 0; n < (long)h->N; n++) interp[n] = interpolate_TSC_one(h, xp[n], yp[n], zp[n], field);
}

/* a15: HMC_models_testing.cpp:54-188.  Positions are taken from the handle (posx/posy/posz). */
int orc_likelihood_calc_V_SPH_fourier_TSC(orc_hamil *h, const double *part_like, double *out_x, double *out_y,
                                          double *out_z) {
  const double hh = h->c.particle_kernel_h;
  const double norm_kernel = 24. / (hh * hh * hh);
  const double norm_density = h->c.rho_c * h->L1 * h->L2 * h->L3 / (double)((size_t)h->N1 * h->N2 * h->N3);
  const double norm = norm_kernel * norm_density;
  double *conv_y_F = dalloc(2 * h->Nhalf);
  fftR2Cplanned(h, part_like, h->R2C_C);
  const unsigned N1 = h->N1, N2 = h->N2, N3 = h->N3, N3half = N3 / 2 + 1;
  double *PF = h->R2C_C, *CX = h->C2R_C;
#pragma omp parallel for
  for (long i = 0; i < (long)N1; ++i) {
    double kx = calc_ki((unsigned)i, h->L1, N1);
    for (unsigned j = 0; j < N2; ++j) {
      double ky = calc_ki(j, h->L2, N2);
      for (unsigned k = 0; k < N3half; ++k) {
        double kz = calc_ki(k, h->L3, N3);
        double k_sq = kx * kx + ky * ky + kz * kz;
        double SPH_kernel_F;
        if (k_sq == 0.) {
          SPH_kernel_F = 1. / (hh * hh * hh);
        } else {
          double kk = sqrt(k_sq);
          double ksink = kk * sin(kk);
          SPH_kernel_F = norm * (3 + cos(2 * kk) - ksink + cos(kk) * (ksink - 4)) / (k_sq * k_sq * k_sq);
        }
        size_t ix = k + (size_t)N3half * (j + (size_t)N2 * i);
        CX[2 * ix] = hh * kx * -PF[2 * ix + 1] * SPH_kernel_F;
        CX[2 * ix + 1] = hh * kx * PF[2 * ix] * SPH_kernel_F;
        conv_y_F[2 * ix] = hh * ky * -PF[2 * ix + 1] * SPH_kernel_F;
        conv_y_F[2 * ix + 1] = hh * ky * PF[2 * ix] * SPH_kernel_F;
        double dummy = PF[2 * ix];
        PF[2 * ix] = hh * kz * -PF[2 * ix + 1] * SPH_kernel_F;
        PF[2 * ix + 1] = hh * kz * dummy * SPH_kernel_F;
      }
    }
  }
  const double *px = A(h, ORC_F_POSX), *py = A(h, ORC_F_POSY), *pz = A(h, ORC_F_POSZ);
  fftC2Rplanned(h, h->C2R_C, h->C2R_R);
  interpolate_TSC(h, px, py, pz, h->C2R_R, out_x);
  fftC2Rplanned(h, conv_y_F, h->C2R_R);
  interpolate_TSC(h, px, py, pz, h->C2R_R, out_y);
  fftC2Rplanned(h, h->R2C_C, h->C2R_R);
  interpolate_TSC(h, px, py, pz, h->C2R_R, out_z);
  free(conv_y_F);
  if (h->c.rsd_model) {
    if (!h->c.planepar) return ORC_ERR_RSD_NOT_PLANEPAR;
    double f1 = orc_fgrow(h->c.ascale, h->c.OM, h->c.OL, 1);
#pragma omp parallel for
    for (long ix = 0; ix < (long)h->N; ++ix) out_z[ix] += f1 * out_z[ix];
  }
  return ORC_OK;
}

/* a16: gradient.cpp:157-211 (rfft = true, in place) */
static void grad_inv_lap_FS(orc_hamil *h, double *c, unsigned index) {
  const unsigned N1 = h->N1, N2 = h->N2, N3 = h->N3, kz_max = N3 / 2 + 1;
#pragma omp parallel for
  for (long i = 0; i < (long)N1; i++)
    for (unsigned j = 0; j < N2; j++)
      for (unsigned k = 0; k < kz_max; ++k) {
        size_t ii = k + (size_t)kz_max * (j + (size_t)N2 * i);
        double kx = calc_ki((unsigned)i, h->L1, N1), ky = calc_ki(j, h->L2, N2), kz = calc_ki(k, h->L3, N3);
        double kmod = kx * kx + ky * ky + kz * kz;
        double fac_kmod = 0.;
        if (kmod > 0) fac_kmod = 1 / kmod;
        double ki_over_kmod = 0.;
        switch (index) {
          case 1: ki_over_kmod = kx * fac_kmod; break;
          case 2: ki_over_kmod = ky * fac_kmod; break;
          case 3: ki_over_kmod = kz * fac_kmod; break;
        }
        double dummy = c[2 * ii];
        c[2 * ii] = ki_over_kmod * c[2 * ii + 1];
        c[2 * ii + 1] = -ki_over_kmod * dummy;
        if (((unsigned)i == N1 / 2) || (j == N2 / 2) || (k == N3 / 2)) {
          c[2 * ii] = 0.;
          c[2 * ii + 1] = 0.;
        }
      }
}

static void add_to_array(const double *in, double *out, size_t n) {
#pragma omp parallel for
  for (long i = 0; i < (long)n; i++) out[i] += in[i];
}


/* gradient.cpp:22-78: spectral gradient i k_dim f^, Nyquist planes zeroed (unplanned FFTs, fftwrapper.cc:26-79) */
static void gradfft(orc_hamil *h, const double *in, double *out, unsigned dim) {
  const unsigned N1 = h->N1, N2 = h->N2, N3 = h->N3, N3half = N3 / 2 + 1;
  double *AUX = dalloc(2 * h->Nhalf);
  orc_fft_r2c_3d(N1, N2, N3, in, AUX);
#pragma omp parallel for
  for (long i = 0; i < (long)N1; i++)
    for (unsigned j = 0; j < N2; j++)
      for (unsigned k = 0; k < N3half; ++k) {
        double kl = dim == 1 ? calc_ki((unsigned)i, h->L1, N1) : (dim == 2 ? calc_ki(j, h->L2, N2) : calc_ki(k, h->L3, N3));
        size_t ll = k + (size_t)N3half * (j + (size_t)N2 * i);
        double dummy = AUX[2 * ll];
        AUX[2 * ll] = -kl * AUX[2 * ll + 1];
        AUX[2 * ll + 1] = kl * dummy;
        if (((unsigned)i == N1 / 2) || (j == N2 / 2) || (k == N3 / 2)) {
          AUX[2 * ll] = 0.;
          AUX[2 * ll + 1] = 0.;
        }
      }
  orc_fft_c2r_3d(N1, N2, N3, AUX, out);
  multiply_factor_array(1 / (double)h->N, out, out, h->N);
  free(AUX);
}

/* gradient.cpp:81-154: 4th-order central differences, periodic */
static void gradfindif(orc_hamil *h, const double *in, double *out, unsigned dim) {
  const int N1 = (int)h->N1;
  const double fac = N1 / (2. * h->L1);
#pragma omp parallel for
  for (long x = 0; x < N1; x++)
    for (int y = 0; y < N1; y++)
      for (int z = 0; z < N1; z++) {
        int c[3] = {(int)x, y, z};
        int l[3] = {c[0], c[1], c[2]}, ll[3] = {c[0], c[1], c[2]}, r[3] = {c[0], c[1], c[2]}, rr[3] = {c[0], c[1], c[2]};
        const int a = (int)dim - 1;
        r[a] = c[a] + 1; l[a] = c[a] - 1; rr[a] = c[a] + 2; ll[a] = c[a] - 2;
        if (r[a] >= N1) r[a] -= N1;
        if (rr[a] >= N1) rr[a] -= N1;
        if (l[a] < 0) l[a] += N1;
        if (ll[a] < 0) ll[a] += N1;
#define IX(v) ((size_t)(v)[2] + (size_t)N1 * ((size_t)(v)[1] + (size_t)N1 * (size_t)(v)[0]))
        out[IX(c)] = -(fac * ((4.0 / 3) * (in[IX(l)] - in[IX(r)]) - (1.0 / 6) * (in[IX(ll)] - in[IX(rr)])));
#undef IX
      }
}

/* ================================================================================================
 * f-3: ALPT forward model (Lag2Eul_non_zeldovich, Lag2Eul.cc:138-312; default build: GFINDIFF, no TRANSF)
 * ============================================================================================== */
static double k_squared_full(const orc_hamil *h, unsigned i, unsigned j, unsigned k) {
  double kx = calc_ki(i, h->L1, h->N1), ky = calc_ki(j, h->L2, h->N2), kz = calc_ki(k, h->L3, h->N3);
  return kx * kx + ky * ky + kz * kz;
}

/* EqSolvers.cc:29-64: Pot = IFFT[-delta^ / k^2], k = 0 -> 0 */
int orc_PoissonSolver(orc_hamil *h, const double *delta, double *Pot) {
  const unsigned N1 = h->N1, N2 = h->N2, N3half = h->N3 / 2 + 1;
  double *C = dalloc(2 * h->Nhalf), *R = dalloc(h->N);
  copyArray(delta, R, h->N);
  orc_fft_r2c_3d(h->N1, h->N2, h->N3, R, C);
#pragma omp parallel for
  for (long i = 0; i < (long)N1; i++)
    for (unsigned j = 0; j < N2; j++)
      for (unsigned k = 0; k < N3half; ++k) {
        size_t ix = k + (size_t)N3half * (j + (size_t)N2 * i);
        double kmod2 = k_squared_full(h, (unsigned)i, j, k);
        double fackern = 0.;
        if (kmod2 > 0.) fackern = -1. / kmod2;
        C[2 * ix] *= fackern;
        C[2 * ix + 1] *= fackern;
      }
  orc_fft_c2r_3d(h->N1, h->N2, h->N3, C, Pot);
  multiply_factor_array(1 / (double)h->N, Pot, Pot, h->N); /* fftC2R, fftwrapper.cc:44-46 */
  free(C);
  free(R);
  return ORC_OK;
}

/* EqSolvers.cc:373-422, GFINDIFF branch (cmake/Modules/Options.cmake:92-93: GFFT OFF, GFINDIFF ON) */
int orc_calc_m2v_mem(orc_hamil *h, const double *phiv, double *m2v) {
  const size_t N = h->N;
  double *xx = dalloc(N), *yy = dalloc(N), *zz = dalloc(N), *xy = dalloc(N), *xz = dalloc(N), *yz = dalloc(N);
  double *dummy = dalloc(N);
  gradfindif(h, phiv, dummy, 1);
  gradfindif(h, dummy, xx, 1);
  gradfindif(h, dummy, xy, 2);
  gradfindif(h, dummy, xz, 3);
  gradfindif(h, phiv, dummy, 2);
  gradfindif(h, dummy, yy, 2);
  gradfindif(h, dummy, yz, 3);
  gradfindif(h, phiv, dummy, 3);
  gradfindif(h, dummy, zz, 3);
#pragma omp parallel for
  for (long i = 0; i < (long)N; i++)
    m2v[i] = xx[i] * yy[i] - xy[i] * xy[i] + xx[i] * zz[i] - xz[i] * xz[i] + yy[i] * zz[i] - yz[i] * yz[i];
  free(xx); free(yy); free(zz); free(xy); free(xz); free(yz); free(dummy);
  return ORC_OK;
}

/* convolution.cpp:224-324 with filtertype 1 (barcoderunner.cc:371-374): K(k) = exp(-k^2 smol^2 / 2) on the FULL
 * n^3 grid, divided by the sum of its inverse transform (= K(0) = 1 up to round-off).  The reference writes this
 * table to `auxkernelr<int(smol)>` and convcomp reads it back; the file is a raw dump, so nothing changes. */
int orc_kernelcomp(orc_hamil *h, double smol, double *out) {
  const unsigned N1 = h->N1, N2 = h->N2, N3 = h->N3, N3half = N3 / 2 + 1;
  const double rS2 = smol * smol;
#pragma omp parallel for
  for (long i = 0; i < (long)N1; i++)
    for (unsigned j = 0; j < N2; j++)
      for (unsigned k = 0; k < N3; k++)
        out[k + (size_t)N3 * (j + (size_t)N2 * i)] = exp(-k_squared_full(h, (unsigned)i, j, k) * rS2 / 2.);
  /* FFT3d(to_Rspace) of the (real, even) table and the sum of the result: the table is Hermitian, so its
   * half-complex part transformed with c2r and scaled by 1/N is the same real array. */
  double *C = dalloc(2 * h->Nhalf), *R = dalloc(h->N);
  for (unsigned i = 0; i < N1; i++)
    for (unsigned j = 0; j < N2; j++)
      for (unsigned k = 0; k < N3half; k++)
        C[2 * (k + (size_t)N3half * (j + (size_t)N2 * i))] = out[k + (size_t)N3 * (j + (size_t)N2 * i)];
  orc_fft_c2r_3d(N1, N2, N3, C, R);
  double wtot = 0.;
  for (size_t i = 0; i < h->N; i++) wtot += R[i] / (double)h->N;
  for (size_t i = 0; i < h->N; i++) out[i] /= wtot;
  free(C);
  free(R);
  return ORC_OK;
}

/* convolution.cpp:327-377: out = IFFT[FFT[in] * kernel]; the reference uses full complex transforms of the real
 * input, of which the half-complex part carries everything. */
int orc_convcomp(orc_hamil *h, const double *in, double *out, double smol) {
  const unsigned N1 = h->N1, N2 = h->N2, N3 = h->N3, N3half = N3 / 2 + 1;
  double *kern = dalloc(h->N), *C = dalloc(2 * h->Nhalf), *R = dalloc(h->N);
  orc_kernelcomp(h, smol, kern);
  copyArray(in, R, h->N);
  orc_fft_r2c_3d(N1, N2, N3, R, C);
#pragma omp parallel for
  for (long i = 0; i < (long)N1; i++)
    for (unsigned j = 0; j < N2; j++)
      for (unsigned k = 0; k < N3half; k++) {
        size_t ix = k + (size_t)N3half * (j + (size_t)N2 * i);
        double kv = kern[k + (size_t)N3 * (j + (size_t)N2 * i)];
        C[2 * ix] *= kv;
        C[2 * ix + 1] *= kv;
      }
  orc_fft_c2r_3d(N1, N2, N3, C, out);
  multiply_factor_array(1 / (double)h->N, out, out, h->N); /* FFT3dC2R, fftwrapper.cc:238 */
  free(kern); free(C); free(R);
  return ORC_OK;
}

/* EqSolvers.cc:280-368 with zeropad = false, norm = false (cpecvel = 1) */
int orc_theta2velcomp(orc_hamil *h, const double *delta, double *vei, int comp) {
  const unsigned N1 = h->N1, N2 = h->N2, N3 = h->N3, N3half = N3 / 2 + 1;
  if (comp < 1 || comp > 3) return ORC_ERR_ARG;
  double *C = dalloc(2 * h->Nhalf), *R = dalloc(h->N);
  copyArray(delta, R, h->N);
  orc_fft_r2c_3d(N1, N2, N3, R, C);
#pragma omp parallel for
  for (long i = 0; i < (long)N1; i++)
    for (unsigned j = 0; j < N2; j++)
      for (unsigned k = 0; k < N3half; k++) {
        size_t ix = k + (size_t)N3half * (j + (size_t)N2 * i);
        double kx = calc_ki((unsigned)i, h->L1, N1), ky = calc_ki(j, h->L2, N2), kz = calc_ki(k, h->L3, N3);
        double kl = comp == 1 ? kx : (comp == 2 ? ky : kz);
        double kmod2 = kx * kx + ky * ky + kz * kz;
        double fackern = 0.0; /* linearvel3d, EqSolvers.cc:130-165 */
        if (kmod2 > EPS_KSQ) fackern = kl / kmod2;
        double dr = C[2 * ix], di = C[2 * ix + 1];
        C[2 * ix] = fackern * di;
        C[2 * ix + 1] = fackern * -dr;
        if (((unsigned)i == N1 / 2) || (j == N2 / 2) || (k == N3 / 2)) C[2 * ix] = C[2 * ix + 1] = 0.;
      }
  orc_fft_c2r_3d(N1, N2, N3, C, vei);
  multiply_factor_array(1 / (double)h->N, vei, vei, h->N);
  free(C); free(R);
  return ORC_OK;
}

/* massFunctions.cc:588-658: vi[l] <- (vi[l] + vi[l - (1,1,1)]) / 2, periodic */
int orc_cellboundcomp(orc_hamil *h, double *vi) {
  const int N1 = (int)h->N1, N2 = (int)h->N2, N3 = (int)h->N3;
  double *viout = dalloc(h->N);
#pragma omp parallel for
  for (long i = 0; i < N1; i++)
    for (int j = 0; j < N2; j++)
      for (int k = 0; k < N3; k++) {
        size_t l = (size_t)k + (size_t)N3 * ((size_t)j + (size_t)N2 * (size_t)i);
        int im = (int)i > 0 ? (int)i - 1 : N1 - 1, jm = j > 0 ? j - 1 : N2 - 1, km = k > 0 ? k - 1 : N3 - 1;
        size_t m = (size_t)km + (size_t)N3 * ((size_t)jm + (size_t)N2 * (size_t)im);
        viout[l] = 0.5 * (vi[m] + vi[l]);
      }
  copyArray(viout, vi, h->N);
  free(viout);
  return ORC_OK;
}

/* Lag2Eul.cc:160-267: Psi^tot = K o Psi^2LPT + Psi^SC - K o Psi^SC per component, then cellboundcomp.
 * `in` is delta^(1); it is left untouched here (the reference works on a copy, 166). */
int orc_alpt_displacement(orc_hamil *h, const double *in, double *psix, double *psiy, double *psiz) {
  const size_t N = h->N;
  const double D1 = h->c.D1, D2 = h->c.D2, kth = h->c.kth, kthsc = h->c.kth;
  double *dummy = dalloc(N), *dummy2 = dalloc(N), *dummy3 = dalloc(N), *dummy4 = dalloc(N);
  copyArray(in, dummy, N);
  orc_PoissonSolver(h, dummy, dummy2);   /* delta(1) -> Phi(1)   169 */
  orc_calc_m2v_mem(h, dummy2, dummy3);   /* Phi(1)  -> delta(2)  171 */
  for (size_t i = 0; i < N; i++) dummy2[i] = D1 * dummy[i] - D2 * dummy3[i]; /* div Psi^2LPT, 199-200 */
  orc_convcomp(h, dummy2, dummy2, kth);  /* K o div Psi^2LPT, 203 */
  for (size_t i = 0; i < N; i++) {       /* div Psi^SC, 212-226 */
    double psilin = -D1 * dummy[i];
    double psisc = 0.;
    if (1. + 2. / 3. * psilin > 0.)
      psisc = 3. * (sqrt(1. + 2. / 3. * psilin) - 1.);
    else
      psisc = -3.;
    psisc *= -1.;
    dummy4[i] = psisc;
  }
  double *psi[3] = {psix, psiy, psiz};
  for (int c = 1; c <= 3; c++) {         /* 240-281 */
    orc_theta2velcomp(h, dummy2, dummy3, c);  /* K o Psi^2LPT_c */
    orc_theta2velcomp(h, dummy4, dummy, c);   /* Psi^SC_c */
    add_to_array(dummy, dummy3, N);           /* K o Psi^2LPT_c + Psi^SC_c */
    orc_convcomp(h, dummy, dummy, kthsc);     /* K o Psi^SC_c */
    for (size_t i = 0; i < N; i++) psi[c - 1][i] = dummy3[i] - dummy[i]; /* subtract_arrays */
    orc_cellboundcomp(h, psi[c - 1]);
  }
  free(dummy); free(dummy2); free(dummy3); free(dummy4);
  return ORC_OK;
}

/* ================================================================================================
 * f-4: measure_spectrum, field_statistics.cpp:20-90.  The reference transforms the real field with a full
 * complex FFT and visits all N1 N2 N3 modes; F(-k) = conj F(k), so the half-complex transform holds every value.
 * ============================================================================================== */
int orc_measure_spectrum(orc_hamil *h, const double *signal, double *kmode, double *power, uint64_t N_bin) {
  const unsigned N1 = h->N1, N2 = h->N2, N3 = h->N3, N3half = N3 / 2 + 1;
  if (N_bin == 0) return ORC_ERR_ARG;
  uint64_t *nmode = (uint64_t *)calloc(N_bin, sizeof(uint64_t));
  for (uint64_t l = 0; l < N_bin; l++) kmode[l] = power[l] = 0.;
  double *C = dalloc(2 * h->Nhalf), *R = dalloc(h->N);
  copyArray(signal, R, h->N);
  orc_fft_r2c_3d(N1, N2, N3, R, C);
  const double kmax = sqrt(k_squared_full(h, N1 / 2, N2 / 2, N3 / 2));
  const double dk = kmax / (double)N_bin;
  for (unsigned i = 0; i < N1; i++)
    for (unsigned j = 0; j < N2; j++)
      for (unsigned k = 0; k < N3; k++) {
        double ktot = sqrt(k_squared_full(h, i, j, k));
        uint64_t nbin = (uint64_t)(ktot / dk);
        if (nbin < N_bin) {
          size_t ix;
          if (k < N3half)
            ix = k + (size_t)N3half * (j + (size_t)N2 * i);
          else /* conjugate partner */
            ix = (N3 - k) + (size_t)N3half * (((N2 - j) % N2) + (size_t)N2 * ((N1 - i) % N1));
          double akl = C[2 * ix], bkl = C[2 * ix + 1];
          kmode[nbin] += 1 * ktot;
          power[nbin] += (akl * akl + bkl * bkl);
          nmode[nbin] += 1;
        }
      }
  const double NORM = h->L1 * h->L2 * h->L3 / (double)h->N / (double)h->N; /* FOURIER_DEF_2 */
  for (uint64_t l = 0; l < N_bin; l++)
    if (nmode[l] > 0) {
      kmode[l] = kmode[l] / (double)nmode[l];
      power[l] = power[l] / (double)nmode[l] * NORM;
    }
  free(nmode); free(C); free(R);
  return ORC_OK;
}

/* *_likelihood_grad_f_delta_x_comp: gaussian_independent.cpp:43-50 (gradfft), poissonian.cpp:37-42 (gradfindif),
 * lognormal_independent.cpp:71-91 (gradfindif of log(rho_c (1 + max(delta, delta_min)))) */
static int grad_f_delta_x_comp(orc_hamil *h, const double *deltaX, double *out, unsigned comp) {
  switch (h->c.likelihood) {
    case 1: gradfft(h, deltaX, out, comp); return ORC_OK;
    case 0: gradfindif(h, deltaX, out, comp); return ORC_OK;
    case 2: {
      double *f = dalloc(h->N);
      for (size_t i = 0; i < h->N; i++) {
        double d = deltaX[i];
        if (d < h->c.delta_min) d = h->c.delta_min;
        f[i] = log(h->c.rho_c * (1. + d));
      }
      gradfindif(h, f, out, comp);
      free(f);
      return ORC_OK;
    }
  }
  return ORC_ERR_UNSUPPORTED;
}

/* HMC_models_testing.cpp:25-50 (calc_h = 0, labelled WRONG upstream but still selectable) */
static int likelihood_calc_h(orc_hamil *h, const double *deltaX, double *out) {
  double *partLike = dalloc(h->N), *dummy = dalloc(h->N);
  double *outC = dalloc(2 * h->Nhalf), *dummyC = dalloc(2 * h->Nhalf);
  int rc = orc_partial_f_delta_x_log_like(h, deltaX, partLike);
  for (unsigned i = 1; i <= 3 && rc == ORC_OK; i++) {
    rc = grad_f_delta_x_comp(h, deltaX, dummy, i);
    if (rc != ORC_OK) break;
    for (size_t m = 0; m < h->N; m++) dummy[m] = partLike[m] * dummy[m];
    orc_fft_r2c_3d(h->N1, h->N2, h->N3, dummy, dummyC);
    grad_inv_lap_FS(h, dummyC, i);
    add_to_array(dummyC, outC, 2 * h->Nhalf);
  }
  if (rc == ORC_OK) {
    orc_fft_c2r_3d(h->N1, h->N2, h->N3, outC, out);
    multiply_factor_array(1 / (double)h->N, out, out, h->N);
  }
  free(partLike); free(dummy); free(outC); free(dummyC);
  return rc;
}

/* a12: HMC_models.cc:312-372 */
int orc_likelihood_calc_h_SPH(orc_hamil *h, const double *deltaX, double *out) {
  if (!(h->c.mk == 3)) return ORC_ERR_MK_NOT_SPH;
  int rc;
  double *V_y = dalloc(h->N), *V_z = dalloc(h->N);
  rc = orc_partial_f_delta_x_log_like(h, deltaX, h->R2C_R);
  if (rc == ORC_OK) {
    switch (h->c.calc_h) {
      case 2:
        rc = orc_likelihood_calc_V_SPH(h, h->R2C_R, A(h, ORC_F_POSX), A(h, ORC_F_POSY), A(h, ORC_F_POSZ), h->R2C_R, V_y,
                                       V_z);
        break;
      case 3:
        rc = orc_likelihood_calc_V_SPH_fourier_TSC(h, h->R2C_R, h->R2C_R, V_y, V_z);
        break;
    }
  }
  if (rc == ORC_OK) {
    fftR2Cplanned(h, h->R2C_R, h->C2R_C);
    grad_inv_lap_FS(h, h->C2R_C, 1);
    fftR2Cplanned(h, V_y, h->R2C_C);
    grad_inv_lap_FS(h, h->R2C_C, 2);
    add_to_array(h->R2C_C, h->C2R_C, 2 * h->Nhalf);
    fftR2Cplanned(h, V_z, h->R2C_C);
    grad_inv_lap_FS(h, h->R2C_C, 3);
    add_to_array(h->R2C_C, h->C2R_C, 2 * h->Nhalf);
    fftC2Rplanned(h, h->C2R_C, out);
  }
  free(V_y);
  free(V_z);
  return rc;
}

/* a5: HMC_models.cc:377-471 */
int orc_likelihood_grad_log_like(orc_hamil *h, const double *delta, double *out) {
  int rc;
  if (h->c.deltaQ_factor != 1.)
    multiply_factor_array(h->c.deltaQ_factor, delta, h->C2R_R, h->N);
  else
    copyArray(delta, h->C2R_R, h->N);
  rc = orc_Lag2Eul(h, h->C2R_R, A(h, ORC_F_DELTAX), A(h, ORC_F_POSX), A(h, ORC_F_POSY), A(h, ORC_F_POSZ),
                   h->c.rsd_model);
  if (rc != ORC_OK) return rc;
  switch (h->c.calc_h) {
    case 0: {
      double *tmp = dalloc(h->N); /* the reference writes into C2Rplan->R while the FFTs inside use their own buffers */
      rc = likelihood_calc_h(h, A(h, ORC_F_DELTAX), tmp);
      copyArray(tmp, h->C2R_R, h->N);
      free(tmp);
    } break;
    case 1: rc = orc_partial_f_delta_x_log_like(h, A(h, ORC_F_DELTAX), h->C2R_R); break;
    case 2:
    case 3: rc = orc_likelihood_calc_h_SPH(h, A(h, ORC_F_DELTAX), h->C2R_R); break;
    default: return ORC_ERR_ARG;
  }
  if (rc != ORC_OK) return rc;
  double norm = 1.;
  double zeldovich_norm = -1.;
  norm *= zeldovich_norm;
  norm *= h->c.deltaQ_factor;
  if (h->c.correct_delta) norm *= h->c.D1;
  multiply_factor_array(norm, h->C2R_R, out, h->N);
  return ORC_OK;
}

/* gaussian_random_field.cpp:25-37 */
static void grf_likelihood_grad_log_like(orc_hamil *h, const double *delta, double *out) {
  const double *window = A(h, ORC_F_WINDOW), *nobs = A(h, ORC_F_NOBS), *noise = A(h, ORC_F_NOISE);
#pragma omp parallel for
  for (long i = 0; i < (long)h->N; i++)
    if (window[i] > 0.)
      out[i] = (delta[i] - nobs[i]) / (noise[i] * noise[i]);
    else
      out[i] = 0;
}

/* a18: *_log_like */
int orc_log_like(orc_hamil *h, const double *signal, double *value) {
  const double *window = A(h, ORC_F_WINDOW), *nobs = A(h, ORC_F_NOBS), *noise = A(h, ORC_F_NOISE);
  const double rho_c = h->c.rho_c, biasP = h->c.biasP, biasE = h->c.biasE;
  double *deltaX = A(h, ORC_F_DELTAX);
  double out = 0.;
  int rc = ORC_OK;
  switch (h->c.likelihood) {
    case 1: { /* gaussian_independent.cpp:51-92 */
      double *delta_growing = dalloc(h->N);
      multiply_factor_array(h->c.deltaQ_factor, signal, delta_growing, h->N);
      rc = orc_Lag2Eul(h, delta_growing, deltaX, A(h, ORC_F_POSX), A(h, ORC_F_POSY), A(h, ORC_F_POSZ), h->c.rsd_model);
      free(delta_growing);
      if (rc != ORC_OK) return rc;
#pragma omp parallel for reduction(+ : out)
      for (long i = 0; i < (long)h->N; i++) {
        double Lambda = window[i] * rho_c * pow(1. + biasP * deltaX[i], biasE);
        if ((window[i] > 0.) && (Lambda > 0.0)) {
          double t = (Lambda - nobs[i]) / noise[i];
          out += 0.5 * (t * t);
        }
      }
    } break;
    case 0: /* poissonian.cpp:44-74 -- ignores rsd_model and deltaQ_factor */
      rc = orc_Lag2Eul(h, signal, deltaX, A(h, ORC_F_POSX), A(h, ORC_F_POSY), A(h, ORC_F_POSZ), 0);
      if (rc != ORC_OK) return rc;
#pragma omp parallel for reduction(+ : out)
      for (long i = 0; i < (long)h->N; i++) {
        double dens = 1. + biasP * deltaX[i];
        double Lambda = window[i] * rho_c * pow(dens, biasE);
        if ((window[i] > 0.) && (Lambda > 0.0)) out += Lambda - nobs[i] * log(Lambda);
      }
      break;
    case 2: /* lognormal_independent.cpp:93-125 with :57-69 (delta_min clamp, no bias) */
      rc = orc_Lag2Eul(h, signal, deltaX, A(h, ORC_F_POSX), A(h, ORC_F_POSY), A(h, ORC_F_POSZ), 0);
      if (rc != ORC_OK) return rc;
#pragma omp parallel for reduction(+ : out)
      for (long i = 0; i < (long)h->N; i++) {
        double dx = deltaX[i];
        if (dx < h->c.delta_min) dx = h->c.delta_min;
        double Lambda = log(rho_c * (1. + dx));
        if (window[i] > 0.) {
          double resid = Lambda - nobs[i];
          out += 0.5 * resid * resid / (noise[i] * noise[i]);
        }
      }
      break;
    case 3: /* gaussian_random_field.cpp:39-52 */
#pragma omp parallel for reduction(+ : out)
      for (long i = 0; i < (long)h->N; i++)
        if (window[i] > 0.) {
          double t = (signal[i] - nobs[i]) / noise[i];
          out += 0.5 * (t * t);
        }
      break;
    default: return ORC_ERR_ARG;
  }
  *value = out;
  return ORC_OK;
}

/* a2: HMC.cc:146-206 (debug conjugate / times-i paths are not restated) */
int orc_gradient_psi(orc_hamil *h, const double *signal) {
  int rc = orc_grad_log_prior(h, signal, h->grad_prior);
  if (rc != ORC_OK) return rc;
  if (h->c.likelihood == 3)
    grf_likelihood_grad_log_like(h, signal, h->grad_like);
  else {
    rc = orc_likelihood_grad_log_like(h, signal, h->grad_like);
    if (rc != ORC_OK) return rc;
  }
  multiply_factor_array(h->c.grad_psi_prior_factor, h->grad_prior, h->grad_prior, h->N);
  multiply_factor_array(h->c.grad_psi_likeli_factor, h->grad_like, h->grad_like, h->N);
  double *g = A(h, ORC_F_GRADPSI);
#pragma omp parallel for
  for (long i = 0; i < (long)h->N; i++) g[i] = h->grad_prior[i] + h->grad_like[i];
  return ORC_OK;
}

/* a18: HMC.cc:64-121 (MASKING off) */
int orc_kinetic_term(orc_hamil *h, const double *momenta, double *value) {
  double *dummy = dalloc(h->N);
  if (h->mass_fs) orc_convolveInvCorrFuncWithSignal(h, momenta, dummy, A(h, ORC_F_MASS_F));
  if (h->mass_rs) {
    const double *mass_r = A(h, ORC_F_MASS_R);
#pragma omp parallel for
    for (long i = 0; i < (long)h->N; i++) {
      double invM = 0.;
      if (mass_r[i] > 0.0) invM = 1. / mass_r[i];
      dummy[i] += invM * momenta[i];
    }
  }
  double v = 0.;
#pragma omp parallel for reduction(+ : v)
  for (long i = 0; i < (long)h->N; i++) v += 0.5 * momenta[i] * dummy[i];
  free(dummy);
  *value = v;
  return ORC_OK;
}

/* a18: HMC.cc:124-143 */
int orc_psi(orc_hamil *h, const double *signal, double *psi_prior, double *psi_like) {
  int rc = orc_log_prior(h, signal, psi_prior);
  if (rc != ORC_OK) return rc;
  rc = orc_log_like(h, signal, psi_like);
  h->psi_prior = *psi_prior;
  h->psi_likeli = *psi_like;
  return rc;
}

/* a18: HMC.cc:209-248 */
int orc_delta_Hamiltonian(orc_hamil *h, const double *qi, const double *pi, const double *qf, const double *pf,
                          double *dH, double out[6]) {
  int rc;
  double Hkini, Hkinf, ppi, pli, ppf, plf;
  if ((rc = orc_kinetic_term(h, pi, &Hkini))) return rc;
  if ((rc = orc_psi(h, qi, &ppi, &pli))) return rc;
  double Hami = Hkini + (ppi + pli);
  if ((rc = orc_kinetic_term(h, pf, &Hkinf))) return rc;
  if ((rc = orc_psi(h, qf, &ppf, &plf))) return rc;
  double Hamf = Hkinf + (ppf + plf);
  double dHam = Hamf - Hami;
  if (h->c.div_dH_by_N) dHam /= (double)h->N;
  *dH = dHam;
  out[0] = Hkini; out[1] = ppi; out[2] = pli;
  out[3] = Hkinf; out[4] = ppf; out[5] = plf;
  return ORC_OK;
}

/* a1: HMC.cc:251-369 with Neps and epsilon forced */
int orc_Hamiltonian_EoM(orc_hamil *h, const double *qi, const double *pi, double *qf, double *pf, double epsilon,
                        uint64_t Neps, uint64_t *steps_done) {
  const size_t N = h->N;
  int rc;
  if (epsilon > 2.) epsilon = 2.; /* 263-264 */
  double *dummy = dalloc(N);
  double *g = A(h, ORC_F_GRADPSI);
  copyArray(qi, qf, N);
  copyArray(pi, pf, N);
  memset(g, 0, N * sizeof(double));
  if ((rc = orc_gradient_psi(h, qf))) { free(dummy); return rc; }
  uint64_t done = 0;
  for (uint64_t jj = 0; jj < Neps; jj++) {
#pragma omp parallel for
    for (long i = 0; i < (long)N; i++) pf[i] -= 0.5 * epsilon * g[i];
    if (h->mass_fs)
      orc_convolveInvCorrFuncWithSignal(h, pf, dummy, A(h, ORC_F_MASS_F));
    else
      memset(dummy, 0, N * sizeof(double));
    if (h->mass_rs) {
      const double *mass_r = A(h, ORC_F_MASS_R);
#pragma omp parallel for
      for (long i = 0; i < (long)N; i++) {
        double invM = 0.;
        if (mass_r[i] > 0.0) invM = 1. / mass_r[i];
        dummy[i] += pf[i] * invM;
      }
    }
#pragma omp parallel for
    for (long i = 0; i < (long)N; i++) qf[i] += epsilon * dummy[i];
    memset(g, 0, N * sizeof(double));
    if ((rc = orc_gradient_psi(h, qf))) { free(dummy); return rc; }
#pragma omp parallel for
    for (long i = 0; i < (long)N; i++) pf[i] -= 0.5 * epsilon * g[i];
    done = jj + 1;
    if (fabs(pf[0]) > 1e50) jj = Neps; /* 360-364: stop the loop */
  }
  if (steps_done) *steps_done = done;
  free(dummy);
  return ORC_OK;
}
