// bchmc_shim.hpp -- C++ host side above the C ABI: the reference's own function names and argument meaning for
// the leapfrog path, on a view of the reference's HAMIL_DATA / HAMIL_NUMERICAL.
//
// The reference (barlib) cannot be compiled in this image (FFTW3 / GSL headers absent), so this layer is built
// against `HamilView` -- a plain struct holding exactly the members of HAMIL_DATA / HAMIL_NUMERICAL that the path
// reads or writes (barlib/include/struct_hamil.h:51-222), under the reference's member names -- instead of against
// struct_hamil.h itself.  Inside barlib the same function bodies compile against the real structs;
// here they are compiled, linked against libbarcode_hip.so and tested (tests/test_gpu_shim.py).
//
//   reference                                              this header
//   void Hamiltonian_EoM(HAMIL_DATA*, real_prec* signali, real_prec* momentai, real_prec* signalf,
//                        real_prec* momentaf, gsl_rng*, DATA*)        HMC.cc:251-369   -> bchmc_shim::Hamiltonian_EoM
//   real_prec delta_Hamiltonian(HAMIL_DATA*, ...)                     HMC.cc:209-248   -> bchmc_shim::delta_Hamiltonian
//   void gradient_psi(HAMIL_DATA*, real_prec* signal, DATA*)          HMC.cc:146-206   -> bchmc_shim::gradient_psi
//   real_prec kinetic_term(...), real_prec psi(...)                   HMC.cc:64-143    -> bchmc_shim::kinetic_term, psi
//   void measure_spectrum(...)                      field_statistics.cpp:20-90         -> bchmc_shim::measure_spectrum
// Errors are std::runtime_error, like the reference's (single catch in main.cc:195-197).
#ifndef BCHMC_SHIM_HPP
#define BCHMC_SHIM_HPP
#include <cstdint>
#include <stdexcept>
#include <string>

#include "bchmc.h"

namespace bchmc_shim {

using real_prec = double;  // DOUBLE_PREC build (define_opt.h:46-50)
using ULONG = unsigned long;

// HAMIL_NUMERICAL members used by the path (struct_hamil.h:51-144), reference names.
struct HamilNumericalView {
  unsigned N1 = 0;  // = N2 = N3 (init_par.cc:116-118)
  ULONG N = 0;
  real_prec L1 = 0, min1 = 0, min2 = 0, min3 = 0, xobs = 0, yobs = 0, zobs = 0;
  bool planepar = true, periodic = true;
  int mk = 3, calc_h = 2, mass_type = 1;
  bool correct_delta = true, div_dH_by_N = false;
  real_prec particle_kernel_h = 0, kth = 4.;
  real_prec grad_psi_prior_factor = 1, grad_psi_likeli_factor = 1, deltaQ_factor = 1;
  // per-attempt state (struct_hamil.h:86-105)
  real_prec N_eps_fac = 0, eps_fac = 0, epsilon = 0;
  ULONG Neps = 0;
  real_prec dH = 0, dK = 0, dE = 0, dprior = 0, dlikeli = 0, psi_prior = 0, psi_likeli = 0;
  real_prec psi_prior_i = 0, psi_prior_f = 0, psi_likeli_i = 0, psi_likeli_f = 0, H_kin_i = 0, H_kin_f = 0;
  // sample bookkeeping read by the step-size schemes (struct_hamil.h:106-112): set by the caller per sample
  ULONG iGibbs = 1;      // sample number; scheme 3's fast initial phase runs while iGibbs == 1 (time_step.cpp:141)
  ULONG rejections = 0;  // rejected attempts of the current sample (HMC.cc:500-501)
  bool accepted = false; // outcome of the last attempt (HMC.cc:503-504)
};

// HAMIL_DATA members used by the path (struct_hamil.h:146-222), reference names.
struct HamilView {
  HamilNumericalView *numerical = nullptr;
  int likelihood = 1;  // which plugin functions set_likelihood_functions bound (init_par.cc:534-559)
  int sfmodel = 1;
  bool rsd_model = false;
  real_prec rho_c = 1, delta_min = -0.999, biasP = 1, biasE = 1, ascale = 1, D1 = 1, D2 = 0, OM = 0, OL = 0;
  // inputs (caller-owned, N doubles each; mass_r / mass_f / noise may be null when the configuration never reads them)
  const real_prec *signal_PS = nullptr, *mass_f = nullptr, *mass_r = nullptr, *nobs = nullptr, *noise = nullptr,
                  *window = nullptr;
  // outputs the reference leaves in HAMIL_DATA (caller-owned, may be null)
  real_prec *gradpsi = nullptr, *deltaX = nullptr, *posx = nullptr, *posy = nullptr, *posz = nullptr;
  int device = 0;     // HIP device of this chain
  void *engine = nullptr;  // owned by the shim: created on first use, released by release()
  // step-size adaptation (DATA::numerical's tables and keys, struct_main.h:95-101,162-173): null = eps_fac stays
  // (eps_fac_update_type 0).  Caller-owned (eps_adapt_create / eps_adapt_destroy).
  struct EpsAdapt *eps = nullptr;
  // cross-chain pooling of the step-size statistics (bchmc.h): null = single chain, the reference's behaviour
  bchmc_comm *comm = nullptr;
  int comm_rank = 0;
  // generation of the input arrays as uploaded: bump with inputs_changed() where HamiltonianMC recomputes or
  // re-reads the mass (HMC.cc:400-423); engine_for() uploads only when it differs from what the engine holds
  unsigned long inputs_generation = 0, uploaded_generation = 0;
  int deterministic = 0;  // bchmc_config.deterministic: bitwise repeatable mass assignment (fixed point)
  // the same for mass_f / mass_r alone (mass_changed()): what HamiltonianMC rewrites every sample.  Unlike the other
  // inputs the mass does not enter gradient_psi or -log L, so the resident chain keeps its carried gradient.
  unsigned long mass_generation = 0, mass_uploaded_generation = 0;
  // Hamiltonian_EoM runs bchmc_leapfrog_dh, which yields the six energy terms of its four arrays with the trajectory;
  // delta_Hamiltonian may answer from them when it is the NEXT call on this view, about the same four arrays, with no
  // inputs_changed() / mass_changed() in between -- HamiltonianMC's order, HMC.cc:455-459.  The C ABI itself never
  // reuses anything.  reuse_eom_energies: 0 = delta_Hamiltonian always evaluates; 1 = reuse on array identity (the
  // caller's contract: the four arrays are not written between the two calls); 2 = identity and a 64-bit hash of the
  // full contents of all four arrays, taken in Hamiltonian_EoM and checked in delta_Hamiltonian (reads 4 N doubles
  // twice on the host: for callers that cannot make that promise).
  int reuse_eom_energies = 1;
  struct EomEnergies {
    bool valid = false;
    const real_prec *ptr[4] = {nullptr, nullptr, nullptr, nullptr};  // signali, momentai, signalf, momentaf
    std::uint64_t hash[4] = {0, 0, 0, 0};                            // mode 2 only
    double terms[6] = {0, 0, 0, 0, 0, 0};
    double dH = 0;
    unsigned long inputs_generation = 0, mass_generation = 0;
  } eom;
};

// Stand-in for gsl_rng_uniform(seed): called exactly where the reference calls it, in its order.
using uniform_fn = double (*)(void *state);

struct Attempt {
  ULONG steps_done = 0;  // < Neps iff "Leap-frogging ... stopped at %lu/%lu, momentum too high" (HMC.cc:360-364)
};

// HMC.cc:251-369.  Draws Neps then epsilon from `uniform` (260-261), clips epsilon at 2 (263-264), integrates,
// increments *count_attempts (368).  Message printing (ncurses) stays with the caller.
Attempt Hamiltonian_EoM(HamilView *hd, const real_prec *signali, const real_prec *momentai, real_prec *signalf,
                        real_prec *momentaf, uniform_fn uniform, void *rng_state, ULONG *count_attempts);
// HMC.cc:209-248: returns dH and fills the numerical->* bookkeeping (139-140, 218-245); hd->deltaX <- psi(signalf)'s.
real_prec delta_Hamiltonian(HamilView *hd, const real_prec *signali, const real_prec *momentai, const real_prec *signalf,
                            const real_prec *momentaf);
// HMC.cc:146-206: hd->gradpsi <- gradient; hd->deltaX / pos* <- this evaluation's.
void gradient_psi(HamilView *hd, const real_prec *signal);
real_prec kinetic_term(HamilView *hd, const real_prec *momenta);  // HMC.cc:64-121
real_prec psi(HamilView *hd, const real_prec *signal);            // HMC.cc:124-143 (stores psi_prior, psi_likeli)
// field_statistics.cpp:20-90
void measure_spectrum(HamilView *hd, const real_prec *signal, real_prec *kmode, real_prec *power, ULONG N_bin);
// ---- step-size adaptation (barlib/src/hmc/leapfrog/time_step.cpp, include/hmc/leapfrog/time_step.hpp) ------------
// The keys of data/input.par:59-87 and the two tables they act on.  Host-only, O(N_a) work per call.
struct EpsAdapt;
struct EpsAdaptConfig {
  int eps_fac_update_type = 3;   // 0 none, 1 power mean every s_eps_total, 2 acceptance rate, 3 = 2 + fast initial phase
  unsigned N_a_eps_update = 100;
  real_prec acc_min = 0.6, acc_max = 0.7;
  int eps_down_smooth = 5;
  real_prec eps_up_fac = 1.;
  real_prec eps_fac_target = 0., eps_fac_power = 2.;
  ULONG s_eps_total = 10;
};
EpsAdapt *eps_adapt_create(const EpsAdaptConfig &cfg);
void eps_adapt_destroy(EpsAdapt *e);
// update_eps_fac (time_step.cpp:151-185): called before every trajectory (HMC.cc:453); may change n->eps_fac.
// Returns the message the reference prints to its ncurses window ("" when nothing was adjusted).
std::string update_eps_fac(HamilView *hd);
// update_epsilon_acc_rate_tables (time_step.cpp:187-203): one finished attempt (n->accepted, n->epsilon) into the
// tables.  The table index is (records - 1) % N_a with records = this chain's attempts plus pooled ones; for a single
// chain that is the reference's (count_attempts - 1) % N_a.
void update_epsilon_acc_rate_tables(HamilView *hd);
// Pooled records of OTHER chains (after bchmc_eps_exchange): same tables, same index rule.
void eps_adapt_append(EpsAdapt *e, bool accepted, real_prec epsilon);
ULONG eps_adapt_records(const EpsAdapt *e);
real_prec eps_adapt_acceptance_rate(const EpsAdapt *e);  // bool_mean(acc_flag_N_a), time_step.cpp:24-28

// ---- HamiltonianMC on the device-resident chain (HMC.cc:431-511) -------------------------------
// The row of performance_log.txt (HMC.cc:40-60) of one attempt.
struct AttemptLog {
  bool accepted = false;
  real_prec epsilon = 0;
  ULONG Neps = 0, steps_done = 0;
  real_prec dH = 0, dK = 0, dE = 0, dprior = 0, dlikeli = 0;
  real_prec psi_prior_i = 0, psi_prior_f = 0, psi_likeli_i = 0, psi_likeli_f = 0, H_kin_i = 0, H_kin_f = 0;
};
// Host-side momentum draw (e.g. the caller's draw_momenta on its gsl_rng, HMC_momenta.cc:42-94): fills N doubles.
using momenta_fn = void (*)(void *state, real_prec *momenta, ULONG N);
void chain_set_state(HamilView *hd, const real_prec *x);  // hd->x -> HBM, once per sample
void chain_get_state(HamilView *hd, real_prec *x);
// One sample: repeat { momenta; (Neps, epsilon) from `uniform` in the reference's order (HMC.cc:260-261); trajectory;
// dH; Metropolis test, statement for statement HMC.cc:462-486 (a uniform is consumed only when p_acceptance < 1) }
// until accepted or `itmax` attempts.  Momenta: `momenta` if given, else the engine's counter-based device draw
// (seed, attempt index).  Fills log[0 .. return value - 1]; *count_attempts advances like HMC.cc:368.
// `log` may be null; otherwise it holds `log_cap` rows and attempts beyond that are not logged (the return value still
// counts them).  Per attempt, like the reference: update_eps_fac before the trajectory (HMC.cc:453), rejections++ on a
// reject (500-501), n->accepted, update_epsilon_acc_rate_tables (506-507).  After the loop, when hd->comm is set, ONE
// bchmc_eps_exchange with this sample's records; the other chains' records are appended to the tables.
ULONG HamiltonianMC(HamilView *hd, uniform_fn uniform, void *rng_state, uint64_t seed, ULONG itmax, ULONG *count_attempts,
                    AttemptLog *log, ULONG log_cap, momenta_fn momenta, void *momenta_state);

// The four engine calls HamiltonianMC makes, as a table: the default binds the C ABI (bchmc_chain_*); the CPU tests
// bind a scripted stand-in so that the loop's bookkeeping is testable without a GPU.
struct ChainOps {
  int (*draw_momenta)(void *engine, uint64_t seed, uint64_t attempt);
  int (*set_momenta)(void *engine, const real_prec *p);
  int (*attempt)(void *engine, double eps, uint64_t neps, double *dH, double terms[6], uint64_t *steps_done);
  int (*accept)(void *engine, int accepted);
};
ULONG HamiltonianMC_ops(HamilView *hd, const ChainOps &ops, void *engine, uniform_fn uniform, void *rng_state,
                        uint64_t seed, ULONG itmax, ULONG *count_attempts, AttemptLog *log, ULONG log_cap,
                        momenta_fn momenta, void *momenta_state);

// Rank bootstrap for the RCCL transport when no launcher hands the unique id around (barcode/main.cc is a plain
// process per GPU): the 128 bytes travel through files next to `path` (<path>, <path>.want.<r>, <path>.ack.<r>).
// Every file carries the nonce of the process that wrote it and the id file echoes the nonces of the ranks it is for,
// so files left behind by an earlier run at the same path are never taken for this run's (hmc_hip_shim.cc); each rank
// removes its own files once the communicator exists.  Ranks of ONE run share `path`; two runs at the same time need
// different paths.  Sets hd->comm; throws after timeout_s seconds.
void comm_bootstrap_file(HamilView *hd, const char *path, int rank, int world, double timeout_s);
// the file protocol on its own (host only): rank 0 passes the id in, the other ranks receive it
void bootstrap_exchange_id(const char *path, int rank, int world, double timeout_s, unsigned char id[BCHMC_UNIQUE_ID_BYTES]);
void bootstrap_cleanup(const char *path, int rank);
void comm_release(HamilView *hd);

// hd's input arrays changed (HamiltonianMC recomputes the mass every sample, HMC.cc:400-423): upload them again
void inputs_changed(HamilView *hd);
// only hd->mass_f / hd->mass_r changed (HMC.cc:400-423): upload those two again
void mass_changed(HamilView *hd);
void release(HamilView *hd);

}  // namespace bchmc_shim

// C-callable hooks over the functions above (exceptions -> return code + message) so that the test-suite can
// drive the compiled C++ layer through ctypes.  0 = ok, 1 = std::runtime_error (message copied to err).
extern "C" {
int bchmc_shim_Hamiltonian_EoM(bchmc_shim::HamilView *hd, const double *signali, const double *momentai, double *signalf,
                               double *momentaf, bchmc_shim::uniform_fn uniform, void *rng_state,
                               unsigned long *count_attempts, unsigned long *steps_done, char *err, size_t errlen);
int bchmc_shim_delta_Hamiltonian(bchmc_shim::HamilView *hd, const double *signali, const double *momentai,
                                 const double *signalf, const double *momentaf, double *dH, char *err, size_t errlen);
int bchmc_shim_gradient_psi(bchmc_shim::HamilView *hd, const double *signal, char *err, size_t errlen);
int bchmc_shim_measure_spectrum(bchmc_shim::HamilView *hd, const double *signal, double *kmode, double *power,
                                unsigned long N_bin, char *err, size_t errlen);
int bchmc_shim_chain_set_state(bchmc_shim::HamilView *hd, const double *x, char *err, size_t errlen);
int bchmc_shim_chain_get_state(bchmc_shim::HamilView *hd, double *x, char *err, size_t errlen);
int bchmc_shim_HamiltonianMC(bchmc_shim::HamilView *hd, bchmc_shim::uniform_fn uniform, void *rng_state, uint64_t seed,
                             unsigned long itmax, unsigned long *count_attempts, bchmc_shim::AttemptLog *log,
                             unsigned long log_cap, unsigned long *n_attempts, char *err, size_t errlen);
/* the same loop on a scripted engine: attempt k returns dH = script_dH[k] (CPU tests of the bookkeeping) */
int bchmc_shim_HamiltonianMC_scripted(bchmc_shim::HamilView *hd, const double *script_dH, unsigned long n_script,
                                      bchmc_shim::uniform_fn uniform, void *rng_state, unsigned long itmax,
                                      unsigned long *count_attempts, bchmc_shim::AttemptLog *log, unsigned long log_cap,
                                      unsigned long *n_attempts, char *err, size_t errlen);
int bchmc_shim_kinetic_term(bchmc_shim::HamilView *hd, const double *momenta, double *out, char *err, size_t errlen);
int bchmc_shim_psi(bchmc_shim::HamilView *hd, const double *signal, double *out, char *err, size_t errlen);
bchmc_shim::EpsAdapt *bchmc_shim_eps_create(int update_type, unsigned N_a, double acc_min, double acc_max, int down_smooth,
                                            double up_fac, double target, double power, unsigned long s_eps_total);
void bchmc_shim_eps_destroy(bchmc_shim::EpsAdapt *e);
void bchmc_shim_eps_append(bchmc_shim::EpsAdapt *e, int accepted, double epsilon);
unsigned long bchmc_shim_eps_records(const bchmc_shim::EpsAdapt *e);
double bchmc_shim_eps_acceptance_rate(const bchmc_shim::EpsAdapt *e);
int bchmc_shim_update_eps_fac(bchmc_shim::HamilView *hd, char *msg, size_t msglen, char *err, size_t errlen);
int bchmc_shim_update_tables(bchmc_shim::HamilView *hd, char *err, size_t errlen);
int bchmc_shim_comm_bootstrap_file(bchmc_shim::HamilView *hd, const char *path, int rank, int world, double timeout_s,
                                   char *err, size_t errlen);
int bchmc_shim_bootstrap_exchange_id(const char *path, int rank, int world, double timeout_s, unsigned char *id, char *err,
                                     size_t errlen);
void bchmc_shim_bootstrap_cleanup(const char *path, int rank);
int bchmc_shim_comm_attach(bchmc_shim::HamilView *hd, bchmc_comm *comm, int rank); /* tests: a custom-transport communicator */
void bchmc_shim_comm_release(bchmc_shim::HamilView *hd);
void bchmc_shim_inputs_changed(bchmc_shim::HamilView *hd);
void bchmc_shim_mass_changed(bchmc_shim::HamilView *hd);
size_t bchmc_shim_sizeof_attempt_log(void);
void bchmc_shim_release(bchmc_shim::HamilView *hd);
size_t bchmc_shim_sizeof_view(void);
size_t bchmc_shim_sizeof_numerical(void);
}
#endif
