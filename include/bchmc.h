/*
 * bchmc.h -- C ABI of libbarcode_hip.so: MI355X-native engine for Barcode's HMC leapfrog hot path.
 *
 * This is the drop-in boundary.  Each entry point replaces a piece of the reference's C++ interface
 * (paths under /root/reference/).  The engine owns all device state (rocFFT plans, grids in HBM);
 * the caller owns every host array passed in and out.  One handle = one GPU = one Markov chain;
 * calls on one handle are not re-entrant, different handles are independent.
 *
 * Error convention: every function returns 0 on success or a BCHMC_ERR_* code; bchmc_strerror() gives
 * the text, bchmc_last_error() the detail of the last failure on a handle.  The reference throws
 * std::runtime_error in the same situations (HMC_models.cc:316-319, 296-298; struct_hamil.h:309-312);
 * the reference-side shim turns non-zero codes back into exceptions.
 */
#ifndef BCHMC_H
#define BCHMC_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BCHMC_ABI_VERSION 4

/* Scalars of HAMIL_NUMERICAL / HAMIL_DATA read by the path (barlib/include/struct_hamil.h:51-222);
 * filled by the shim from the HAMIL_DATA that call_hamil.cc:42 builds.  Cubic grids only, like the
 * reference (init_par.cc:116-118). */
typedef struct bchmc_config {
  uint32_t abi_version;      /* must be BCHMC_ABI_VERSION */
  uint32_t Nx;               /* N1 = N2 = N3 */
  double L;                  /* L1 = L2 = L3 [Mpc/h] */
  double min1, min2, min3;   /* xllc, yllc, zllc */
  double xobs, yobs, zobs;
  int32_t planepar, periodic;
  int32_t mk;                /* masskernel: 0 NGP, 1 CIC, 2 TSC, 3 SPH */
  int32_t calc_h;            /* 0 (legacy), 1, 2 (SPH adjoint, default) or 3 (Fourier + TSC) */
  int32_t likelihood;        /* 0 Poisson, 1 Gaussian, 2 log-normal, 3 GRF (init_par.cc:534-559) */
  int32_t sfmodel;           /* 1 Zel'dovich; anything else: ALPT (Lag2Eul_non_zeldovich, Lag2Eul.cc:138-312, dispatcher
                              * 325-331) unless rsd_model is set, which always takes the Zel'dovich + RSD model */
  int32_t rsd_model;
  int32_t mass_type;         /* 0,1,2,3,4,5,6,60 -> mass_fs/mass_rs as struct_hamil.h:272-313 */
  int32_t correct_delta;
  int32_t div_dH_by_N;
  double particle_kernel_h;  /* SPH scale length, = particle_kernel_h_rel * cell size */
  double grad_psi_prior_factor, grad_psi_likeli_factor, deltaQ_factor;
  double rho_c, delta_min, biasP, biasE;
  double ascale, D1, D2, OM, OL;
  double kth;                /* ALPT split scale [Mpc/h] = n->kth = slength (struct_hamil.h:102,259; input.par:121) */
  int32_t precision;         /* 0: fp64 field arrays (reference DOUBLE_PREC); 1: fp32 field arrays (SINGLE_PREC-like:
                              * storage + particle-mesh arithmetic in float, k-space arithmetic and reductions in
                              * double).  The ABI's arrays are double in both modes. */
  int32_t device;            /* HIP device ordinal */
  int32_t deterministic;     /* 1: bitwise repeatable results -- the mass assignment accumulates in 64-bit fixed point
                              * (integer adds are order-independent), the density is converted and summed in a fixed
                              * order; one extra pass over the grid per force evaluation.  0: hardware float atomics,
                              * last bits vary from run to run like the reference's OpenMP build (barcode/main.cc:86-90).
                              * BCHMC_DETERMINISTIC=1 in the environment switches it on for every handle.
                              * Range: contributions are scaled to 2^46 per maximal one (W(0), or weight 1), so a cell
                              * holds 2^17 of them before the 63-bit sum wraps; a cell that passed 2^16 (or came out
                              * negative) makes the next synchronising call return BCHMC_ERR_STATE. */
  int32_t reserved0;
} bchmc_config;

enum {
  BCHMC_OK = 0,
  BCHMC_ERR_ARG = 1,
  BCHMC_ERR_MK_NOT_SPH = 2,       /* calc_h 2/3 need masskernel 3 (HMC_models.cc:316-319) */
  BCHMC_ERR_RSD_NOT_PLANEPAR = 3, /* HMC_models.cc:296-298, rsd.cc:60-62 */
  BCHMC_ERR_MASS_TYPE = 4,        /* struct_hamil.h:309-312 */
  BCHMC_ERR_UNSUPPORTED = 5,
  BCHMC_ERR_HIP = 6,
  BCHMC_ERR_ROCFFT = 7,
  BCHMC_ERR_NOMEM = 8,
  BCHMC_ERR_STATE = 9             /* e.g. an input array was never uploaded */
};

/* Arrays of HAMIL_DATA (struct_hamil.h:146-166).  Inputs are uploaded once per chain (or when
 * Hamiltonian_mass recomputes the mass, HMC.cc:400-423); outputs are fetched on demand. */
typedef enum bchmc_field {
  BCHMC_F_SIGNAL_PS = 0, /* in : prior power spectrum on the full N^3 grid (hd->signal_PS) */
  BCHMC_F_MASS_F = 1,    /* in : Fourier-space mass (hd->mass_f), full N^3 grid */
  BCHMC_F_MASS_R = 2,    /* in : real-space mass (hd->mass_r) */
  BCHMC_F_NOBS = 3,      /* in : hd->nobs */
  BCHMC_F_NOISE = 4,     /* in : hd->noise */
  BCHMC_F_WINDOW = 5,    /* in : hd->window */
  BCHMC_F_DELTAX = 6,    /* out: hd->deltaX of the last force / energy evaluation */
  BCHMC_F_POSX = 7,      /* out: hd->posx */
  BCHMC_F_POSY = 8,
  BCHMC_F_POSZ = 9,
  BCHMC_F_RHO = 10,      /* out (diagnostic): density before overdens() */
  BCHMC_F_PART_LIKE = 11,/* out (diagnostic): partial_f_delta_x_log_like */
  BCHMC_F_VX = 12,       /* out (diagnostic): likelihood_calc_V_SPH */
  BCHMC_F_VY = 13,
  BCHMC_F_VZ = 14,
  BCHMC_F_PSIX = 15,     /* out (diagnostic): theta2vel displacement */
  BCHMC_F_PSIY = 16,
  BCHMC_F_PSIZ = 17,
  BCHMC_F_GRAD_PRIOR = 18, /* out: prior term of the last bchmc_gradient (after its test factor) */
  BCHMC_F_GRAD_LIKE = 19,  /* out: likelihood term of the last bchmc_gradient (after its test factor) */
  BCHMC_F_COUNT = 20
} bchmc_field;

typedef struct bchmc_handle bchmc_handle;

/* Lifecycle.  Replaces plan_pkg construction (fftwrapper.cc:281-324, init_par.cc:418-426) and the
 * per-sample HAMIL_DATA setup (call_hamil.cc:38-42). */
int bchmc_create(const bchmc_config *cfg, bchmc_handle **out);
void bchmc_destroy(bchmc_handle *h);
const char *bchmc_strerror(int code);
const char *bchmc_last_error(const bchmc_handle *h);

/* Host -> HBM copy of one input array of N = Nx^3 doubles. */
int bchmc_upload(bchmc_handle *h, bchmc_field field, const double *host, size_t n);
/* HBM -> host copy of one output array of N doubles (state of the last force / energy evaluation). */
int bchmc_fetch(bchmc_handle *h, bchmc_field field, double *host, size_t n);

/* Hamiltonian_EoM (HMC.cc:251-369): `neps` leapfrog steps of size `eps` from (q0, p0) to (q1, p1).
 * The shim draws neps and eps from the caller's gsl_rng exactly as HMC.cc:260-264 and passes them in.
 * *steps_done < neps iff the runaway-momentum guard |p[0]| > 1e50 (HMC.cc:360-364) fired. */
int bchmc_leapfrog(bchmc_handle *h, const double *q0, const double *p0, double *q1, double *p1, double eps,
                   uint64_t neps, uint64_t *steps_done);

/* Hamiltonian_EoM (HMC.cc:251-369) and delta_Hamiltonian (HMC.cc:209-248) of the same four arrays in ONE pass --
 * what HamiltonianMC does with consecutive calls at HMC.cc:455 and 459.  Same (q1, p1, *steps_done) as bchmc_leapfrog;
 * *dH and terms as bchmc_delta_hamiltonian would return for (q0, p0, q1, p1): K and psi_prior of both ends are
 * Parseval sums of the k-space state, -log L of both ends comes from the trajectory's own first and last force
 * evaluation where log_like's forward model is the force's one (else, and for real-space masses and the GRF
 * likelihood, the energies are evaluated around the trajectory) -- four array uploads and two forward models fewer
 * than the two separate calls.  The reuse is the CALLER's statement, made by choosing this entry point: the engine
 * keeps no memory of earlier calls, and bchmc_delta_hamiltonian always evaluates what it is given.  Leaves
 * psi(q1)'s deltaX / pos* in the handle (HMC.cc:225).  (ABI version 4; replaces the pointer + sampled-content
 * cache version 3 kept inside bchmc_delta_hamiltonian.) */
int bchmc_leapfrog_dh(bchmc_handle *h, const double *q0, const double *p0, double *q1, double *p1, double eps,
                      uint64_t neps, uint64_t *steps_done, double *dH, double terms[6]);

/* kinetic_term + psi (HMC.cc:64-143): out = { H_kin, psi_prior, psi_likeli } at (q, p).  Leaves
 * deltaX / pos* of this evaluation in the handle like the reference's log_like does. */
int bchmc_energies(bchmc_handle *h, const double *q, const double *p, double out[3]);

/* kinetic_term (HMC.cc:64-121) and psi (HMC.cc:124-143) on their own: one transform and one reduction for the
 * kinetic term (needs mass_f / mass_r only); psi_out = { log_prior, log_like } with the forward model of `q` left in
 * the handle (deltaX / pos*), like the reference's log_like leaves it in HAMIL_DATA. */
int bchmc_kinetic_term(bchmc_handle *h, const double *p, double *out);
int bchmc_psi(bchmc_handle *h, const double *q, double psi_out[2]);

/* delta_Hamiltonian (HMC.cc:209-248): terms = { H_kin_i, psi_prior_i, psi_likeli_i, H_kin_f,
 * psi_prior_f, psi_likeli_f }, *dH includes div_dH_by_N.  Always a full evaluation of the arrays passed in, against
 * the inputs uploaded at the time of the call (no result of an earlier call is reused). */
int bchmc_delta_hamiltonian(bchmc_handle *h, const double *qi, const double *pi, const double *qf, const double *pf,
                            double *dH, double terms[6]);

/* gradient_psi (HMC.cc:146-206): g = prior_factor * S^-1 q + likeli_factor * d(-log L)/dq. */
int bchmc_gradient(bchmc_handle *h, const double *q, double *g);

/* Forward model only (Lag2Eul, Lag2Eul.cc:318-332 / 338-424): leaves deltaX and pos* in the handle.
 * use_rsd < 0 means "as configured". */
int bchmc_forward(bchmc_handle *h, const double *q, int use_rsd);

/* ---- device-resident variants: same semantics, all pointers are HBM addresses on the handle's device,
 * work is enqueued on the handle's stream and NOT synchronised (call bchmc_sync). ---- */
int bchmc_leapfrog_device(bchmc_handle *h, const double *d_q0, const double *d_p0, double *d_q1, double *d_p1,
                          double eps, uint64_t neps);
int bchmc_steps_done(bchmc_handle *h, uint64_t *steps_done); /* synchronises */
int bchmc_energies_device(bchmc_handle *h, const double *d_q, const double *d_p, double out[3]); /* synchronises */
/* Waits for the handle's stream.  Also the point where the engine adapts its working storage to the field it has just
 * seen: if a force evaluation overflowed the binning's per-tile record slots (that evaluation itself was still exact,
 * through the two-pass sort), they are doubled here, so a trajectory never pays for a reallocation.  bchmc_steps_done,
 * bchmc_forward and the host-array energy / gradient calls do the same. */
int bchmc_sync(bchmc_handle *h);
void *bchmc_stream(bchmc_handle *h); /* hipStream_t the engine launches on */

/* ---- device-resident chain ("next" components, built on the same path) ----
 * The current sample q and the momenta stay in HBM between attempts, so one HamiltonianMC attempt
 * (HMC.cc:445-498) costs no 4 x N-double PCIe round trip, and the -log L of both trajectory ends is taken
 * from the force evaluations the trajectory performs anyway instead of two extra forward models.
 * Host keeps the control flow: (Neps, epsilon) draws, u < exp(-dH) test, epsilon adaptation. */
int bchmc_chain_set_state(bchmc_handle *h, const double *q);   /* hd->x -> HBM */
int bchmc_chain_get_state(bchmc_handle *h, double *q);
int bchmc_chain_set_momenta(bchmc_handle *h, const double *p); /* host-drawn momenta (reference RNG order) */
int bchmc_chain_get_momenta(bchmc_handle *h, double *p);
/* p ~ N(0, M) on the device: counter-based Philox4x32-10, a pure function of (seed, attempt, cell).  Statistical
 * stand-in for draw_momenta (HMC_momenta.cc:42-94), whose serial GSL stream it does not reproduce. */
int bchmc_chain_draw_momenta(bchmc_handle *h, uint64_t seed, uint64_t attempt);
/* Hamiltonian_EoM + delta_Hamiltonian from the resident (q, p); terms as in bchmc_delta_hamiltonian.
 * The chain carries gradient_psi and -log L of its state from one attempt to the next: the last force evaluation of an
 * accepted trajectory (HMC.cc:349) is at the point where the next one starts, and a rejected attempt restarts from the
 * same point, so the evaluation of HMC.cc:279 is skipped from the second attempt on (same numbers to round-off;
 * bchmc_upload and bchmc_chain_set_state drop the carried values).  BCHMC_NO_FORCE_CARRY=1 re-evaluates every time. */
int bchmc_chain_attempt(bchmc_handle *h, double eps, uint64_t neps, double *dH, double terms[6], uint64_t *steps_done);
int bchmc_chain_get_proposal(bchmc_handle *h, double *q1, double *p1);
int bchmc_chain_accept(bchmc_handle *h, int accepted);         /* accepted: q := proposal (HMC.cc:497-498) */
/* measure_spectrum (barlib/src/field_statistics.cpp:20-90; callers barcoderunner.cc:328,532 -> dump_ps_it): binned
 * power spectrum of a field, n_bin bins of width |k|_max / n_bin.  `signal` = N host doubles, or NULL for the
 * resident chain state (no transform and no field transfer: its R2C is what the chain keeps).  kmode / power:
 * n_bin doubles each; empty bins stay 0 like upstream. */
int bchmc_measure_spectrum(bchmc_handle *h, const double *signal, uint64_t n_bin, double *kmode, double *power);
int bchmc_philox_kat(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]); /* known-answer hook for tests */

/* Diagnostic (tests, logs): how the particle-mesh path is currently set up.  out = { tile-sorted path in use, one-pass
 * binning in use, record slots per tile in use, record slots per tile allocated, long trajectories poll the slot words,
 * staged density flush available, unrolled 81-cell kernels in use, ALPT planes pipeline available }. */
int bchmc_tile_info(bchmc_handle *h, int32_t out[8]);

/* ---- measurement hooks (bench.py): per-kernel-class HIP-event timing on the engine's stream ---- */
enum {
  BCHMC_K_FFT_C2R = 0, BCHMC_K_FFT_R2C, BCHMC_K_KSPACE_DRIFT_ZA, BCHMC_K_SCATTER, BCHMC_K_MEAN_PARTIAL,
  BCHMC_K_GATHER, BCHMC_K_KSPACE_FORCE_KICK, BCHMC_K_SORT, BCHMC_K_OTHER, BCHMC_K_COUNT
};
int bchmc_profile(bchmc_handle *h, int enable);                 /* 1: record events around every launch */
int bchmc_profile_read(bchmc_handle *h, double ms[BCHMC_K_COUNT], uint64_t launches[BCHMC_K_COUNT]); /* and reset */
const char *bchmc_kernel_name(int kernel_class);

/* ---- cross-chain step-size statistics -------------------------------------------------------
 * Chains are independent (one per GPU); the only exchange on the path is this record, one per finished attempt,
 * so that every chain's acceptance / epsilon tables (acc_flag_N_a, epsilon_N_a: struct_main.h:172-173, written by
 * update_epsilon_acc_rate_tables, time_step.cpp:187-203, read by update_eps_fac, :151-185) fill world_size times
 * faster.  A chain that never calls bchmc_eps_exchange behaves exactly like the single-chain reference. */
typedef struct bchmc_eps_record {
  double epsilon;
  int32_t accepted;
  int32_t neps;
} bchmc_eps_record;

/* Records one rank contributes per exchange.  The exchange happens ONCE PER SAMPLE (a fixed point every rank
 * reaches the same number of times), never per attempt: the attempt loop ends at a data-dependent iteration, so a
 * per-attempt collective would pair up records of different samples and deadlock the rank with more rejections.
 * A sample with more attempts than this sends the rest with the next exchange(s). */
#define BCHMC_EPS_BATCH 32
#define BCHMC_UNIQUE_ID_BYTES 128 /* sizeof(ncclUniqueId) */

typedef struct bchmc_comm bchmc_comm;
/* Transport used by a custom communicator: all-gather `bytes_per_rank` bytes of host memory from every rank into
 * `recv` (world * bytes_per_rank, rank order).  Returns 0 on success.  (MPI_Allgather in an MPI-launched barcode;
 * an in-process stub in the CPU tests.) */
typedef int (*bchmc_allgather_fn)(void *ctx, const void *send, void *recv, size_t bytes_per_rank);

/* RCCL transport: rank 0 calls bchmc_comm_unique_id and hands the 128 bytes to the other ranks (a file, an
 * environment variable, MPI_Bcast: the shim's bootstrap helper uses a file), then every rank
 * calls bchmc_comm_create: ncclCommInitRank on `device`, a side stream and a 2 x world x 520-byte staging buffer.
 * librccl is loaded on first use (dlopen), so single-chain runs do not depend on it.  On failure *out still holds an
 * object (bchmc_comm_last_error tells why); release it with bchmc_comm_destroy like a working one. */
int bchmc_comm_unique_id(unsigned char id[BCHMC_UNIQUE_ID_BYTES]);
int bchmc_comm_create(const unsigned char id[BCHMC_UNIQUE_ID_BYTES], int rank, int world, int device, bchmc_comm **out);
int bchmc_comm_create_custom(bchmc_allgather_fn fn, void *ctx, int rank, int world, bchmc_comm **out);
void bchmc_comm_destroy(bchmc_comm *c);
const char *bchmc_comm_last_error(const bchmc_comm *c);
/* One exchange (ncclAllGather of 520 bytes per rank on the side stream): queue `n_mine` (>= 0) records of this rank,
 * send the oldest <= BCHMC_EPS_BATCH queued ones, and return every rank's contribution in rank order, the own one
 * included: all[0 .. *n_all - 1], rank_of[i] = contributing rank of all[i] (rank_of may be NULL).  `cap` = capacity of
 * `all`, at least world * BCHMC_EPS_BATCH (checked before anything is sent).  Every rank of the communicator must call
 * it the same number of times.  Failure semantics: a call that returns non-zero has queued nothing of `mine` on this
 * rank -- retry with the same records, or drop them; after a TRANSPORT failure (the collective itself broke) the
 * communicator is no longer usable: the peers may or may not have completed the all-gather. */
int bchmc_eps_exchange(bchmc_comm *c, const bchmc_eps_record *mine, int n_mine, bchmc_eps_record *all, int *rank_of,
                       int cap, int *n_all);
int bchmc_comm_pending(const bchmc_comm *c); /* own records still queued for a later exchange */
int bchmc_comm_world(const bchmc_comm *c);   /* ranks of the communicator (sizes the caller's `all` / `rank_of`) */
int bchmc_comm_rank(const bchmc_comm *c);
const char *bchmc_comm_transport(const bchmc_comm *c); /* "rccl", "custom" or "none" (one rank, nothing to move) */

#ifdef __cplusplus
}
#endif
#endif /* BCHMC_H */
