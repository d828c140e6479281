/*
 * oracle/bchmc_oracle.h -- TEST INFRASTRUCTURE ONLY.  Not part of the product; see oracle/README.md.
 *
 * CPU restatement (plain C) of Barcode's HMC leapfrog hot path, function by function, each citing the
 * reference file:line it follows.  PARITY UNPINNED: the reference ships no golden vectors for this path
 *  and cannot be built in this image (FFTW3 and GSL headers/libraries absent),
 * so this restatement has not been checked against reference output.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 */
#ifndef BCHMC_ORACLE_H
#define BCHMC_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Scalars of HAMIL_NUMERICAL / HAMIL_DATA that the hot path reads
 * (/root/reference/barlib/include/struct_hamil.h:51-144, 146-222). */
typedef struct orc_config {
  uint32_t N1;            /* cubic only: N1 == N2 == N3 (init_par.cc:116-118) */
  double L1;              /* box side */
  double min1, min2, min3;
  double xobs, yobs, zobs;
  int32_t planepar, periodic;
  int32_t mk;             /* masskernel 0 NGP, 1 CIC, 2 TSC, 3 SPH */
  int32_t calc_h;         /* 0..3 */
  int32_t likelihood;     /* 0 Poisson, 1 Gaussian, 2 log-normal, 3 GRF */
  int32_t sfmodel;        /* 1 Zel'dovich; anything else = ALPT (Lag2Eul_non_zeldovich) unless rsd_model */
  int32_t rsd_model;
  int32_t mass_type;      /* -> mass_fs/mass_rs as struct_hamil.h:272-313 */
  int32_t correct_delta;
  int32_t div_dH_by_N;
  double particle_kernel_h;
  double grad_psi_prior_factor, grad_psi_likeli_factor, deltaQ_factor;
  double rho_c, delta_min, biasP, biasE;
  double ascale, D1, D2, OM, OL;
  double kth;             /* ALPT split scale = slength (struct_hamil.h:259); used when sfmodel != 1 && !rsd_model */
} orc_config;

enum {
  ORC_OK = 0,
  ORC_ERR_ARG = 1,
  ORC_ERR_MK_NOT_SPH = 2,      /* HMC_models.cc:316-319 */
  ORC_ERR_RSD_NOT_PLANEPAR = 3,/* HMC_models.cc:296-298, rsd.cc:60-62 */
  ORC_ERR_MASS_TYPE = 4,       /* struct_hamil.h:309-312 */
  ORC_ERR_UNSUPPORTED = 5
};

enum { /* array slots for orc_set_array / orc_get_array */
  ORC_F_SIGNAL_PS = 0, ORC_F_MASS_F, ORC_F_MASS_R, ORC_F_NOBS, ORC_F_NOISE, ORC_F_WINDOW,
  ORC_F_GRADPSI, ORC_F_DELTAX, ORC_F_POSX, ORC_F_POSY, ORC_F_POSZ,
  ORC_F_COUNT
};

typedef struct orc_hamil orc_hamil;

size_t orc_sizeof_config(void);
void orc_set_threads(int n);    /* OpenMP build: omp_set_num_threads; serial build: no-op */
int orc_get_max_threads(void);
int orc_create(const orc_config *cfg, orc_hamil **out);
void orc_destroy(orc_hamil *h);
int orc_set_array(orc_hamil *h, int field, const double *src); /* copies N doubles in */
double *orc_get_array(orc_hamil *h, int field);                /* borrowed pointer, N doubles */
int orc_stencil(orc_hamil *h, int *n, const int **ci, const int **cj, const int **ck);

/* a3 */ int orc_convolveInvCorrFuncWithSignal(orc_hamil *h, const double *signal, double *out, const double *corr);
/* a7 */ int orc_theta2vel(orc_hamil *h, const double *delta, double *vex, double *vey, double *vez);
/* a6 */ int orc_Lag2Eul(orc_hamil *h, const double *in, double *out, double *posx, double *posy, double *posz,
                        int use_rsd);
/* a10 */ int orc_getDensity(orc_hamil *h, int mk, const double *xp, const double *yp, const double *zp, double *rho);
/* a11 */ void orc_overdens(orc_hamil *h, const double *in, double *out);
/* a13 */ int orc_partial_f_delta_x_log_like(orc_hamil *h, const double *deltaX, double *out);
/* a14 */ int orc_likelihood_calc_V_SPH(orc_hamil *h, const double *part_like, const double *posx, const double *posy,
                                       const double *posz, double *out_x, double *out_y, double *out_z);
/* a15 */ int orc_likelihood_calc_V_SPH_fourier_TSC(orc_hamil *h, const double *part_like, double *out_x, double *out_y,
                                                   double *out_z);
/* a12 */ int orc_likelihood_calc_h_SPH(orc_hamil *h, const double *deltaX, double *out);
/* a5 */ int orc_likelihood_grad_log_like(orc_hamil *h, const double *delta, double *out);
/* a4 */ int orc_grad_log_prior(orc_hamil *h, const double *signal, double *out);
/* a4 */ int orc_log_prior(orc_hamil *h, const double *signal, double *value);
/* a18 */ int orc_log_like(orc_hamil *h, const double *signal, double *value);
/* a2 */ int orc_gradient_psi(orc_hamil *h, const double *signal); /* -> gradpsi; also leaves grad_prior/grad_like */
double *orc_last_grad_prior(orc_hamil *h);
double *orc_last_grad_like(orc_hamil *h);
/* a18 */ int orc_kinetic_term(orc_hamil *h, const double *momenta, double *value);
/* a18 */ int orc_psi(orc_hamil *h, const double *signal, double *psi_prior, double *psi_like);
/* a18: out[0..5] = H_kin_i, psi_prior_i, psi_likeli_i, H_kin_f, psi_prior_f, psi_likeli_f; returns dH in *dH */
int orc_delta_Hamiltonian(orc_hamil *h, const double *qi, const double *pi, const double *qf, const double *pf,
                          double *dH, double out[6]);
/* a1: Neps and epsilon are forced by the caller; steps_done mirrors the runaway guard HMC.cc:360-364 */
int orc_Hamiltonian_EoM(orc_hamil *h, const double *qi, const double *pi, double *qf, double *pf, double epsilon,
                        uint64_t Neps, uint64_t *steps_done);

/* scalars of cosmo.cc used by the path */
/* f-3: the pieces of Lag2Eul_non_zeldovich, Lag2Eul.cc:138-312 */
int orc_PoissonSolver(orc_hamil *h, const double *delta, double *Pot);             /* EqSolvers.cc:29-64 */
int orc_calc_m2v_mem(orc_hamil *h, const double *phiv, double *m2v);               /* EqSolvers.cc:373-422 (GFINDIFF) */
int orc_kernelcomp(orc_hamil *h, double smol, double *kernel_full);                /* convolution.cpp:224-324, Gaussian */
int orc_convcomp(orc_hamil *h, const double *in, double *out, double smol);        /* convolution.cpp:327-377 */
int orc_theta2velcomp(orc_hamil *h, const double *delta, double *vei, int comp);   /* EqSolvers.cc:280-368 */
int orc_cellboundcomp(orc_hamil *h, double *vi);                                   /* massFunctions.cc:588-658 */
int orc_alpt_displacement(orc_hamil *h, const double *in, double *psix, double *psiy, double *psiz);

/* f-4: field_statistics.cpp:20-90 (FOURIER_DEF_2) */
int orc_measure_spectrum(orc_hamil *h, const double *signal, double *kmode, double *power, uint64_t N_bin);

/* oracle/orc_random.c: the reference's momentum draw with GSL's MT19937 + polar Box-Muller stream restated */
int orc_create_GARFIELD(unsigned n, double L, const double *power, unsigned long seed, double *delta); /* random.cpp:48-511 */
int orc_draw_momenta(unsigned n, double L, int mass_fs, int mass_rs, const double *mass_f, const double *mass_r,
                     unsigned long seed, double *momenta);                                            /* HMC_momenta.cc:42-94 */
void orc_mt19937_stream(unsigned long seed, uint32_t *out, size_t n);   /* gsl_rng_mt19937 raw outputs */
void orc_ugaussian_stream(unsigned long seed, double *out, size_t n);   /* gsl_ran_ugaussian */

double orc_fgrow(double a, double OM, double OL, int term);    /* cosmo.cc:182-217 */
double orc_c_pecvel(double a, double OM, double OL, int term); /* cosmo.cc:220-235 */

#ifdef __cplusplus
}
#endif
#endif
