// hmc_hip_shim.cc -- the reference's leapfrog-path functions implemented on the C ABI (include/bchmc_shim.hpp).
// Pure host C++11 (g++), no HIP: everything numerical happens in libbarcode_hip.so.
#include "bchmc_shim.hpp"

#include <cmath>
#include <cstdio>
#include <cstring>

namespace bchmc_shim {
namespace {

[[noreturn]] void fail(bchmc_handle *h, int rc, const char *where) {
  std::string msg = std::string("In ") + where + ": " + bchmc_strerror(rc);
  const char *detail = h ? bchmc_last_error(h) : nullptr;
  if (detail && detail[0]) msg += std::string(" (") + detail + ")";
  throw std::runtime_error(msg);
}

void upload_inputs(HamilView *hd, bchmc_handle *h) {
  const HamilNumericalView *n = hd->numerical;
  struct { bchmc_field f; const real_prec *p; } arr[] = {
      {BCHMC_F_SIGNAL_PS, hd->signal_PS}, {BCHMC_F_MASS_F, hd->mass_f}, {BCHMC_F_MASS_R, hd->mass_r},
      {BCHMC_F_NOBS, hd->nobs},           {BCHMC_F_NOISE, hd->noise},   {BCHMC_F_WINDOW, hd->window}};
  for (auto &a : arr)
    if (a.p) {
      const int rc = bchmc_upload(h, a.f, a.p, n->N);
      if (rc) fail(h, rc, "bchmc_upload");
    }
}

bchmc_handle *engine_for(HamilView *hd) {
  if (!hd || !hd->numerical) throw std::runtime_error("In hmc_hip_shim: HAMIL_DATA without numerical");
  if (hd->engine) return static_cast<bchmc_handle *>(hd->engine);
  const HamilNumericalView *n = hd->numerical;
  bchmc_config c;
  std::memset(&c, 0, sizeof c);
  c.abi_version = BCHMC_ABI_VERSION;
  c.Nx = n->N1;
  c.L = n->L1;
  c.min1 = n->min1; c.min2 = n->min2; c.min3 = n->min3;
  c.xobs = n->xobs; c.yobs = n->yobs; c.zobs = n->zobs;
  c.planepar = n->planepar; c.periodic = n->periodic;
  c.mk = n->mk; c.calc_h = n->calc_h;
  c.likelihood = hd->likelihood;
  c.sfmodel = hd->sfmodel; c.rsd_model = hd->rsd_model;
  c.mass_type = n->mass_type;
  c.correct_delta = n->correct_delta; c.div_dH_by_N = n->div_dH_by_N;
  c.particle_kernel_h = n->particle_kernel_h;
  c.grad_psi_prior_factor = n->grad_psi_prior_factor;
  c.grad_psi_likeli_factor = n->grad_psi_likeli_factor;
  c.deltaQ_factor = n->deltaQ_factor;
  c.rho_c = hd->rho_c; c.delta_min = hd->delta_min; c.biasP = hd->biasP; c.biasE = hd->biasE;
  c.ascale = hd->ascale; c.D1 = hd->D1; c.D2 = hd->D2; c.OM = hd->OM; c.OL = hd->OL;
  c.kth = n->kth;
  c.precision = 0;
  c.device = hd->device;
  bchmc_handle *h = nullptr;
  const int rc = bchmc_create(&c, &h);
  if (rc) {
    std::string msg = std::string("In bchmc_create: ") + bchmc_strerror(rc);
    if (h && bchmc_last_error(h)[0]) msg += std::string(" (") + bchmc_last_error(h) + ")";
    if (h) bchmc_destroy(h);
    throw std::runtime_error(msg);
  }
  hd->engine = h;
  upload_inputs(hd, h);
  return h;
}

// hd->deltaX / pos*: state of the last force or energy evaluation (dump_deltas reads them, barcoderunner.cc:527)
void fetch_eval_state(HamilView *hd, bchmc_handle *h) {
  if (hd->likelihood == 3) return;  // GRF likelihood: no forward model
  const ULONG N = hd->numerical->N;
  struct { bchmc_field f; real_prec *p; } arr[] = {
      {BCHMC_F_DELTAX, hd->deltaX}, {BCHMC_F_POSX, hd->posx}, {BCHMC_F_POSY, hd->posy}, {BCHMC_F_POSZ, hd->posz}};
  for (auto &a : arr)
    if (a.p) {
      const int rc = bchmc_fetch(h, a.f, a.p, N);
      if (rc) fail(h, rc, "bchmc_fetch");
    }
}

}  // namespace

Attempt Hamiltonian_EoM(HamilView *hd, const real_prec *signali, const real_prec *momentai, real_prec *signalf,
                        real_prec *momentaf, uniform_fn uniform, void *rng_state, ULONG *count_attempts) {
  bchmc_handle *h = engine_for(hd);
  HamilNumericalView *n = hd->numerical;
  // identical RNG consumption and order as HMC.cc:260-264
  n->Neps = static_cast<ULONG>(n->N_eps_fac * uniform(rng_state)) + 1;
  n->epsilon = static_cast<real_prec>(n->eps_fac * uniform(rng_state));
  if (n->epsilon > 2.) n->epsilon = 2.;
  uint64_t done = 0;
  const int rc = bchmc_leapfrog(h, signali, momentai, signalf, momentaf, n->epsilon, n->Neps, &done);
  if (rc) fail(h, rc, "Hamiltonian_EoM");
  if (count_attempts) ++*count_attempts;  // HMC.cc:368
  Attempt a;
  a.steps_done = static_cast<ULONG>(done);
  return a;
}

real_prec delta_Hamiltonian(HamilView *hd, const real_prec *signali, const real_prec *momentai, const real_prec *signalf,
                            const real_prec *momentaf) {
  bchmc_handle *h = engine_for(hd);
  HamilNumericalView *n = hd->numerical;
  double dH = 0., t[6];
  const int rc = bchmc_delta_hamiltonian(h, signali, momentai, signalf, momentaf, &dH, t);
  if (rc) fail(h, rc, "delta_Hamiltonian");
  n->H_kin_i = t[0]; n->psi_prior_i = t[1]; n->psi_likeli_i = t[2];  // HMC.cc:218-245
  n->H_kin_f = t[3]; n->psi_prior_f = t[4]; n->psi_likeli_f = t[5];
  n->psi_prior = t[4]; n->psi_likeli = t[5];                         // psi(signalf) is evaluated last (225)
  n->dprior = t[4] - t[1];
  n->dlikeli = t[5] - t[2];
  n->dK = t[3] - t[0];
  n->dE = (t[4] + t[5]) - (t[1] + t[2]);
  n->dH = dH;
  fetch_eval_state(hd, h);
  return dH;
}

void gradient_psi(HamilView *hd, const real_prec *signal) {
  bchmc_handle *h = engine_for(hd);
  if (!hd->gradpsi) throw std::runtime_error("In gradient_psi: hd->gradpsi is not allocated");
  const int rc = bchmc_gradient(h, signal, hd->gradpsi);
  if (rc) fail(h, rc, "gradient_psi");
  fetch_eval_state(hd, h);
}

real_prec kinetic_term(HamilView *hd, const real_prec *momenta) {
  bchmc_handle *h = engine_for(hd);
  // bchmc_energies evaluates the three terms together; the kinetic one does not depend on the signal
  std::string zeros(hd->numerical->N * sizeof(real_prec), '\0');
  double e[3];
  const int rc = bchmc_energies(h, reinterpret_cast<const real_prec *>(zeros.data()), momenta, e);
  if (rc) fail(h, rc, "kinetic_term");
  return e[0];
}

real_prec psi(HamilView *hd, const real_prec *signal) {
  bchmc_handle *h = engine_for(hd);
  std::string zeros(hd->numerical->N * sizeof(real_prec), '\0');
  double e[3];
  const int rc = bchmc_energies(h, signal, reinterpret_cast<const real_prec *>(zeros.data()), e);
  if (rc) fail(h, rc, "psi");
  hd->numerical->psi_prior = e[1];  // HMC.cc:139-140
  hd->numerical->psi_likeli = e[2];
  fetch_eval_state(hd, h);
  return e[1] + e[2];
}

void measure_spectrum(HamilView *hd, const real_prec *signal, real_prec *kmode, real_prec *power, ULONG N_bin) {
  bchmc_handle *h = engine_for(hd);
  const int rc = bchmc_measure_spectrum(h, signal, N_bin, kmode, power);
  if (rc) fail(h, rc, "measure_spectrum");
}

void chain_set_state(HamilView *hd, const real_prec *x) {
  bchmc_handle *h = engine_for(hd);
  const int rc = bchmc_chain_set_state(h, x);
  if (rc) fail(h, rc, "chain_set_state");
}

void chain_get_state(HamilView *hd, real_prec *x) {
  bchmc_handle *h = engine_for(hd);
  const int rc = bchmc_chain_get_state(h, x);
  if (rc) fail(h, rc, "chain_get_state");
}

ULONG HamiltonianMC(HamilView *hd, uniform_fn uniform, void *rng_state, uint64_t seed, ULONG itmax, ULONG *count_attempts,
                    AttemptLog *log, momenta_fn momenta, void *momenta_state) {
  bchmc_handle *h = engine_for(hd);
  HamilNumericalView *n = hd->numerical;
  std::string host_p;
  if (momenta) host_p.resize(n->N * sizeof(real_prec));
  ULONG it = 0;
  for (; it < itmax;) {
    const ULONG attempt = count_attempts ? *count_attempts : it;
    int rc;
    if (momenta) {  // HMC.cc:445-447 with the caller's generator
      real_prec *p = reinterpret_cast<real_prec *>(&host_p[0]);
      momenta(momenta_state, p, n->N);
      rc = bchmc_chain_set_momenta(h, p);
    } else {
      rc = bchmc_chain_draw_momenta(h, seed, attempt);
    }
    if (rc) fail(h, rc, "draw_momenta");
    // HMC.cc:260-264
    n->Neps = static_cast<ULONG>(n->N_eps_fac * uniform(rng_state)) + 1;
    n->epsilon = static_cast<real_prec>(n->eps_fac * uniform(rng_state));
    if (n->epsilon > 2.) n->epsilon = 2.;
    double dH = 0., t[6];
    uint64_t done = 0;
    rc = bchmc_chain_attempt(h, n->epsilon, n->Neps, &dH, t, &done);
    if (rc) fail(h, rc, "Hamiltonian_EoM");
    if (count_attempts) ++*count_attempts;  // HMC.cc:368
    n->H_kin_i = t[0]; n->psi_prior_i = t[1]; n->psi_likeli_i = t[2];
    n->H_kin_f = t[3]; n->psi_prior_f = t[4]; n->psi_likeli_f = t[5];
    n->psi_prior = t[4]; n->psi_likeli = t[5];
    n->dprior = t[4] - t[1];
    n->dlikeli = t[5] - t[2];
    n->dK = t[3] - t[0];
    n->dE = n->dprior + n->dlikeli;
    n->dH = dH;
    // HMC.cc:462-486
    real_prec p_acceptance = 1.;
    if (dH < 0.)
      p_acceptance = 1.;
    else if (std::exp(-dH) < 1.)
      p_acceptance = std::exp(-dH);
    bool accepted;
    if (p_acceptance >= 1.)
      accepted = true;
    else
      accepted = uniform(rng_state) < p_acceptance;
    rc = bchmc_chain_accept(h, accepted ? 1 : 0);
    if (rc) fail(h, rc, "chain_accept");
    AttemptLog &r = log[it];
    r.accepted = accepted;
    r.epsilon = n->epsilon;
    r.Neps = n->Neps;
    r.steps_done = static_cast<ULONG>(done);
    r.dH = dH; r.dK = n->dK; r.dE = n->dE; r.dprior = n->dprior; r.dlikeli = n->dlikeli;
    r.psi_prior_i = t[1]; r.psi_prior_f = t[4]; r.psi_likeli_i = t[2]; r.psi_likeli_f = t[5];
    r.H_kin_i = t[0]; r.H_kin_f = t[3];
    ++it;
    if (accepted) break;
  }
  return it;
}

void inputs_changed(HamilView *hd) {
  if (hd && hd->engine) upload_inputs(hd, static_cast<bchmc_handle *>(hd->engine));
}

void release(HamilView *hd) {
  if (hd && hd->engine) {
    bchmc_destroy(static_cast<bchmc_handle *>(hd->engine));
    hd->engine = nullptr;
  }
}

}  // namespace bchmc_shim

// ---- C-callable hooks ----------------------------------------------------------------------------------
namespace {
template <typename F>
int guarded(char *err, size_t errlen, F &&f) {
  try {
    f();
    if (err && errlen) err[0] = '\0';
    return 0;
  } catch (const std::runtime_error &e) {
    if (err && errlen) std::snprintf(err, errlen, "%s", e.what());
    return 1;
  }
}
}  // namespace

extern "C" {
int bchmc_shim_Hamiltonian_EoM(bchmc_shim::HamilView *hd, const double *signali, const double *momentai, double *signalf,
                               double *momentaf, bchmc_shim::uniform_fn uniform, void *rng_state,
                               unsigned long *count_attempts, unsigned long *steps_done, char *err, size_t errlen) {
  return guarded(err, errlen, [&] {
    const bchmc_shim::Attempt a =
        bchmc_shim::Hamiltonian_EoM(hd, signali, momentai, signalf, momentaf, uniform, rng_state, count_attempts);
    if (steps_done) *steps_done = a.steps_done;
  });
}
int bchmc_shim_delta_Hamiltonian(bchmc_shim::HamilView *hd, const double *signali, const double *momentai,
                                 const double *signalf, const double *momentaf, double *dH, char *err, size_t errlen) {
  return guarded(err, errlen, [&] { *dH = bchmc_shim::delta_Hamiltonian(hd, signali, momentai, signalf, momentaf); });
}
int bchmc_shim_gradient_psi(bchmc_shim::HamilView *hd, const double *signal, char *err, size_t errlen) {
  return guarded(err, errlen, [&] { bchmc_shim::gradient_psi(hd, signal); });
}
int bchmc_shim_measure_spectrum(bchmc_shim::HamilView *hd, const double *signal, double *kmode, double *power,
                                unsigned long N_bin, char *err, size_t errlen) {
  return guarded(err, errlen, [&] { bchmc_shim::measure_spectrum(hd, signal, kmode, power, N_bin); });
}
int bchmc_shim_chain_set_state(bchmc_shim::HamilView *hd, const double *x, char *err, size_t errlen) {
  return guarded(err, errlen, [&] { bchmc_shim::chain_set_state(hd, x); });
}
int bchmc_shim_chain_get_state(bchmc_shim::HamilView *hd, double *x, char *err, size_t errlen) {
  return guarded(err, errlen, [&] { bchmc_shim::chain_get_state(hd, x); });
}
int bchmc_shim_HamiltonianMC(bchmc_shim::HamilView *hd, bchmc_shim::uniform_fn uniform, void *rng_state, uint64_t seed,
                             unsigned long itmax, unsigned long *count_attempts, bchmc_shim::AttemptLog *log,
                             unsigned long *n_attempts, char *err, size_t errlen) {
  return guarded(err, errlen, [&] {
    *n_attempts = bchmc_shim::HamiltonianMC(hd, uniform, rng_state, seed, itmax, count_attempts, log, nullptr, nullptr);
  });
}
size_t bchmc_shim_sizeof_attempt_log(void) { return sizeof(bchmc_shim::AttemptLog); }
void bchmc_shim_release(bchmc_shim::HamilView *hd) { bchmc_shim::release(hd); }
size_t bchmc_shim_sizeof_view(void) { return sizeof(bchmc_shim::HamilView); }
size_t bchmc_shim_sizeof_numerical(void) { return sizeof(bchmc_shim::HamilNumericalView); }
}
