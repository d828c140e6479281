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
ULONG HamiltonianMC(HamilView *hd, uniform_fn uniform, void *rng_state, uint64_t seed, ULONG itmax, ULONG *count_attempts,
                    AttemptLog *log, momenta_fn momenta, void *momenta_state);

// hd's input arrays changed (HamiltonianMC recomputes the mass every sample, HMC.cc:400-423): upload them again
void inputs_changed(HamilView *hd);
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
                             unsigned long *n_attempts, char *err, size_t errlen);
size_t bchmc_shim_sizeof_attempt_log(void);
void bchmc_shim_release(bchmc_shim::HamilView *hd);
size_t bchmc_shim_sizeof_view(void);
size_t bchmc_shim_sizeof_numerical(void);
}
#endif
