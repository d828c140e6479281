// hmc_hip_shim.cc -- the reference's leapfrog-path functions implemented on the C ABI (include/bchmc_shim.hpp).
// Pure host C++11 (g++), no HIP: everything numerical happens in libbarcode_hip.so.
#include "bchmc_shim.hpp"

#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <thread>
#include <vector>

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

// 64-bit content hash of a host array (reuse_eom_energies == 2): four interleaved multiply-xorshift lanes per thread,
// several threads; every word enters, so a single changed element changes the hash (up to 2^-64 collisions).
std::uint64_t content_hash(const real_prec *a, ULONG n) {
  const unsigned nt = std::max(1u, std::min(8u, std::thread::hardware_concurrency() / 2));
  std::vector<std::uint64_t> part(nt, 0);
  auto work = [&](unsigned t) {
    const ULONG lo = n * t / nt, hi = n * (t + 1) / nt;
    std::uint64_t l[4] = {0x9E3779B97F4A7C15ull, 0xC2B2AE3D27D4EB4Full, 0x165667B19E3779F9ull, 0x27D4EB2F165667C5ull};
    ULONG i = lo;
    auto mix = [](std::uint64_t h, std::uint64_t w) {
      h = (h ^ w) * 0xFF51AFD7ED558CCDull;
      return h ^ (h >> 29);
    };
    for (; i + 4 <= hi; i += 4) {
      std::uint64_t w[4];
      std::memcpy(w, a + i, sizeof w);
      for (int k = 0; k < 4; k++) l[k] = mix(l[k], w[k]);
    }
    for (; i < hi; i++) {
      std::uint64_t w;
      std::memcpy(&w, a + i, sizeof w);
      l[0] = mix(l[0], w);
    }
    part[t] = mix(mix(mix(l[0], l[1]), l[2]), l[3]);
  };
  std::vector<std::thread> th;
  for (unsigned t = 1; t < nt; t++) th.emplace_back(work, t);
  work(0);
  for (auto &x : th) x.join();
  std::uint64_t h = n;
  for (unsigned t = 0; t < nt; t++) h = (h ^ part[t]) * 0xC4CEB9FE1A85EC53ull + t;
  return h;
}

// keep_eom: only delta_Hamiltonian may find the energies of the preceding Hamiltonian_EoM; any other call in between
// ends their validity (it may leave another forward model in the engine, whose deltaX / pos* would be fetched).
bchmc_handle *engine_for(HamilView *hd, bool keep_eom = false) {
  if (!hd || !hd->numerical) throw std::runtime_error("In hmc_hip_shim: HAMIL_DATA without numerical");
  if (!keep_eom) hd->eom.valid = false;
  if (hd->engine) {
    bchmc_handle *h = static_cast<bchmc_handle *>(hd->engine);
    if (hd->uploaded_generation != hd->inputs_generation) {  // inputs_changed() since the last upload
      upload_inputs(hd, h);
      hd->uploaded_generation = hd->inputs_generation;
      hd->mass_uploaded_generation = hd->mass_generation;
    } else if (hd->mass_uploaded_generation != hd->mass_generation) {  // mass_changed(): the two mass arrays only
      const HamilNumericalView *n = hd->numerical;
      struct { bchmc_field f; const real_prec *p; } arr[] = {{BCHMC_F_MASS_F, hd->mass_f}, {BCHMC_F_MASS_R, hd->mass_r}};
      for (auto &a : arr)
        if (a.p) {
          const int rc = bchmc_upload(h, a.f, a.p, n->N);
          if (rc) fail(h, rc, "bchmc_upload");
        }
      hd->mass_uploaded_generation = hd->mass_generation;
    }
    return h;
  }
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
  c.deterministic = hd->deterministic;
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
  hd->uploaded_generation = hd->inputs_generation;
  hd->mass_uploaded_generation = hd->mass_generation;
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
  if (hd->reuse_eom_energies) {
    // trajectory and the six energy terms of its four arrays in one pass, kept for the delta_Hamiltonian that follows
    HamilView::EomEnergies &e = hd->eom;
    const int rc = bchmc_leapfrog_dh(h, signali, momentai, signalf, momentaf, n->epsilon, n->Neps, &done, &e.dH, e.terms);
    if (rc) fail(h, rc, "Hamiltonian_EoM");
    const real_prec *ptr[4] = {signali, momentai, signalf, momentaf};
    for (int i = 0; i < 4; i++) {
      e.ptr[i] = ptr[i];
      e.hash[i] = hd->reuse_eom_energies >= 2 ? content_hash(ptr[i], n->N) : 0;
    }
    e.inputs_generation = hd->inputs_generation;
    e.mass_generation = hd->mass_generation;
    // in place (signalf == signali): delta_Hamiltonian will be asked about arrays that no longer hold the start state
    e.valid = signali != signalf && momentai != momentaf && signali != momentaf && momentai != signalf;
  } else {
    const int rc = bchmc_leapfrog(h, signali, momentai, signalf, momentaf, n->epsilon, n->Neps, &done);
    if (rc) fail(h, rc, "Hamiltonian_EoM");
  }
  if (count_attempts) ++*count_attempts;  // HMC.cc:368
  Attempt a;
  a.steps_done = static_cast<ULONG>(done);
  return a;
}

real_prec delta_Hamiltonian(HamilView *hd, const real_prec *signali, const real_prec *momentai, const real_prec *signalf,
                            const real_prec *momentaf) {
  bchmc_handle *h = engine_for(hd, /*keep_eom=*/true);
  HamilNumericalView *n = hd->numerical;
  double dH = 0., t[6];
  HamilView::EomEnergies &e = hd->eom;
  const real_prec *ptr[4] = {signali, momentai, signalf, momentaf};
  bool reuse = hd->reuse_eom_energies && e.valid && e.inputs_generation == hd->inputs_generation &&
               e.mass_generation == hd->mass_generation;
  for (int i = 0; reuse && i < 4; i++) reuse = ptr[i] == e.ptr[i];
  if (reuse && hd->reuse_eom_energies >= 2)
    for (int i = 0; reuse && i < 4; i++) reuse = content_hash(ptr[i], n->N) == e.hash[i];
  e.valid = false;  // one use: a second delta_Hamiltonian evaluates
  if (reuse) {
    dH = e.dH;
    std::memcpy(t, e.terms, sizeof t);
  } else {
    const int rc = bchmc_delta_hamiltonian(h, signali, momentai, signalf, momentaf, &dH, t);
    if (rc) fail(h, rc, "delta_Hamiltonian");
  }
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
  double k = 0.;
  const int rc = bchmc_kinetic_term(h, momenta, &k);  // one R2C + one Parseval sum; needs mass_f / mass_r only
  if (rc) fail(h, rc, "kinetic_term");
  return k;
}

real_prec psi(HamilView *hd, const real_prec *signal) {
  bchmc_handle *h = engine_for(hd);
  double e[2];
  const int rc = bchmc_psi(h, signal, e);
  if (rc) fail(h, rc, "psi");
  hd->numerical->psi_prior = e[0];  // HMC.cc:139-140
  hd->numerical->psi_likeli = e[1];
  fetch_eval_state(hd, h);
  return e[0] + e[1];
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

namespace {
int op_draw(void *e, uint64_t seed, uint64_t attempt) { return bchmc_chain_draw_momenta(static_cast<bchmc_handle *>(e), seed, attempt); }
int op_setp(void *e, const real_prec *p) { return bchmc_chain_set_momenta(static_cast<bchmc_handle *>(e), p); }
int op_attempt(void *e, double eps, uint64_t neps, double *dH, double terms[6], uint64_t *done) {
  return bchmc_chain_attempt(static_cast<bchmc_handle *>(e), eps, neps, dH, terms, done);
}
int op_accept(void *e, int a) { return bchmc_chain_accept(static_cast<bchmc_handle *>(e), a); }
}  // namespace

ULONG HamiltonianMC(HamilView *hd, uniform_fn uniform, void *rng_state, uint64_t seed, ULONG itmax, ULONG *count_attempts,
                    AttemptLog *log, ULONG log_cap, momenta_fn momenta, void *momenta_state) {
  bchmc_handle *h = engine_for(hd);
  const ChainOps ops = {op_draw, op_setp, op_attempt, op_accept};
  try {
    return HamiltonianMC_ops(hd, ops, h, uniform, rng_state, seed, itmax, count_attempts, log, log_cap, momenta,
                             momenta_state);
  } catch (const std::runtime_error &e) {
    const char *detail = bchmc_last_error(h);
    if (detail && detail[0]) throw std::runtime_error(std::string(e.what()) + " (" + detail + ")");
    throw;
  }
}

ULONG HamiltonianMC_ops(HamilView *hd, const ChainOps &ops, void *engine, uniform_fn uniform, void *rng_state,
                        uint64_t seed, ULONG itmax, ULONG *count_attempts, AttemptLog *log, ULONG log_cap,
                        momenta_fn momenta, void *momenta_state) {
  if (!hd || !hd->numerical) throw std::runtime_error("In HamiltonianMC: HAMIL_DATA without numerical");
  HamilNumericalView *n = hd->numerical;
  auto chk = [](int rc, const char *where) {
    if (rc) throw std::runtime_error(std::string("In ") + where + ": " + bchmc_strerror(rc));
  };
  std::string host_p;
  if (momenta) host_p.resize(n->N * sizeof(real_prec));
  std::vector<bchmc_eps_record> mine;  // this sample's records, exchanged once after the loop
  // the fixed point every chain reaches once per sample: pool the step-size statistics.  Also reached when
  // the attempt loop throws (an engine error): the other ranks are waiting in their all-gather, so this rank takes part
  // with what it has and reports its own error afterwards instead of leaving them blocked until the RCCL timeout.
  auto exchange = [&]() {
    if (!hd->comm) return;
    const int cap = std::max(1, bchmc_comm_world(hd->comm)) * BCHMC_EPS_BATCH;
    std::vector<bchmc_eps_record> all(static_cast<size_t>(cap));
    std::vector<int> rank_of(static_cast<size_t>(cap));
    int n_all = 0;
    const int rc = bchmc_eps_exchange(hd->comm, mine.data(), static_cast<int>(mine.size()), all.data(), rank_of.data(),
                                      cap, &n_all);
    if (rc) throw std::runtime_error(std::string("In bchmc_eps_exchange: ") + bchmc_comm_last_error(hd->comm));
    if (hd->eps)
      for (int i = 0; i < n_all; i++)
        if (rank_of[static_cast<size_t>(i)] != hd->comm_rank)  // own attempts are in the tables already
          eps_adapt_append(hd->eps, all[static_cast<size_t>(i)].accepted != 0, all[static_cast<size_t>(i)].epsilon);
  };
  ULONG it = 0;
  try {
    while (it < itmax) {  // HMC.cc:431
      const ULONG attempt = count_attempts ? *count_attempts : it;
      if (momenta) {  // HMC.cc:445-449 with the caller's generator
        real_prec *p = reinterpret_cast<real_prec *>(&host_p[0]);
        momenta(momenta_state, p, n->N);
        chk(ops.set_momenta(engine, p), "draw_momenta");
      } else {
        chk(ops.draw_momenta(engine, seed, attempt), "draw_momenta");
      }
      update_eps_fac(hd);  // HMC.cc:453
      // HMC.cc:260-264
      n->Neps = static_cast<ULONG>(n->N_eps_fac * uniform(rng_state)) + 1;
      n->epsilon = static_cast<real_prec>(n->eps_fac * uniform(rng_state));
      if (n->epsilon > 2.) n->epsilon = 2.;
      double dH = 0., t[6] = {0., 0., 0., 0., 0., 0.};
      uint64_t done = 0;
      chk(ops.attempt(engine, n->epsilon, n->Neps, &dH, t, &done), "Hamiltonian_EoM");
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
      chk(ops.accept(engine, accepted ? 1 : 0), "chain_accept");  // HMC.cc:497-498
      if (!accepted) n->rejections++;                              // HMC.cc:500-501
      n->accepted = accepted;                                      // HMC.cc:503-504
      if (log && it < log_cap) {                                   // write_to_performance_log's row, HMC.cc:506
        AttemptLog &r = log[it];
        r.accepted = accepted;
        r.epsilon = n->epsilon;
        r.Neps = n->Neps;
        r.steps_done = static_cast<ULONG>(done);
        r.dH = dH; r.dK = n->dK; r.dE = n->dE; r.dprior = n->dprior; r.dlikeli = n->dlikeli;
        r.psi_prior_i = t[1]; r.psi_prior_f = t[4]; r.psi_likeli_i = t[2]; r.psi_likeli_f = t[5];
        r.H_kin_i = t[0]; r.H_kin_f = t[3];
      }
      update_epsilon_acc_rate_tables(hd);  // HMC.cc:507
      if (hd->comm) {
        bchmc_eps_record rec;
        rec.epsilon = n->epsilon;
        rec.accepted = accepted ? 1 : 0;
        rec.neps = static_cast<int32_t>(n->Neps);
        mine.push_back(rec);
      }
      ++it;
      if (accepted) break;
    }
  } catch (...) {
    try {
      exchange();
    } catch (const std::runtime_error &) {  // the first error is the one to report
    }
    throw;
  }
  exchange();
  return it;
}

namespace {
// ---- file bootstrap of the ncclUniqueId ----------------------------------------------------------------------------
// A fixed path may hold the files of an earlier run (a reader that accepts any 128-byte file picks up last
// run's id and blocks in ncclCommInitRank on a dead address).  So nothing is accepted on its mere existence: every
// file carries the 64-bit nonce of the process that wrote it, and the id file echoes the nonces of the ranks it is
// meant for.
//   rank r > 0: writes <path>.want.<r> = { magic, nonce_r }, polls <path> until it holds { magic, world, nonce_0,
//               id, nonce[1..world-1] } with nonce[r] == nonce_r, then writes <path>.ack.<r> = { magic, nonce_r, nonce_0 };
//   rank 0:     removes a stale <path>, polls the want files, writes <path> (tmp + rename) with the nonces it read,
//               and keeps polling: a want file that CHANGES (a stale one replaced by the live rank's) makes it rewrite
//               <path>; it is done when every ack echoes (nonce_r, nonce_0) of the current <path>.
// Every rank removes its own files when the communicator exists (bootstrap_cleanup); files of a crashed run are
// harmless because their nonces match nothing.
const std::uint64_t kBootMagic = 0x62636d63626f6f74ull;  // "bcmcboot"

std::uint64_t boot_nonce() {
  std::random_device rd;
  std::uint64_t v = (static_cast<std::uint64_t>(rd()) << 32) ^ rd();
  v ^= static_cast<std::uint64_t>(::getpid()) * 0x9E3779B97F4A7C15ull;
  v ^= static_cast<std::uint64_t>(std::chrono::system_clock::now().time_since_epoch().count());
  return v ? v : 1;
}

bool read_words(const std::string &path, std::vector<std::uint64_t> &w, size_t n_words) {
  FILE *f = std::fopen(path.c_str(), "rb");
  if (!f) return false;
  w.assign(n_words, 0);
  const size_t got = std::fread(w.data(), sizeof(std::uint64_t), n_words, f);
  const bool more = std::fgetc(f) != EOF;
  std::fclose(f);
  return got == n_words && !more && w[0] == kBootMagic;
}

void write_words(const std::string &path, const std::vector<std::uint64_t> &w) {
  const std::string tmp = path + ".tmp." + std::to_string(::getpid());
  FILE *f = std::fopen(tmp.c_str(), "wb");
  if (!f || std::fwrite(w.data(), sizeof(std::uint64_t), w.size(), f) != w.size() || std::fclose(f) != 0)
    throw std::runtime_error("In comm_bootstrap_file: cannot write " + tmp);
  if (std::rename(tmp.c_str(), path.c_str()) != 0)  // atomic: readers never see a partial file
    throw std::runtime_error("In comm_bootstrap_file: cannot rename to " + path);
}
}  // namespace

void bootstrap_exchange_id(const char *path_c, int rank, int world, double timeout_s,
                           unsigned char id[BCHMC_UNIQUE_ID_BYTES]) {
  if (!path_c || !id || world < 1 || rank < 0 || rank >= world)
    throw std::runtime_error("In comm_bootstrap_file: bad argument");
  const std::string path(path_c);
  constexpr size_t kIdWords = BCHMC_UNIQUE_ID_BYTES / sizeof(std::uint64_t);
  const size_t id_file_words = 3 + kIdWords + static_cast<size_t>(world);  // magic, world, nonce_0, id, nonce[world]
  const auto t0 = std::chrono::steady_clock::now();
  auto expired = [&] { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s; };
  auto nap = [] { std::this_thread::sleep_for(std::chrono::milliseconds(10)); };
  auto want = [&](int r) { return path + ".want." + std::to_string(r); };
  auto ack = [&](int r) { return path + ".ack." + std::to_string(r); };
  const std::uint64_t mine = boot_nonce();
  std::vector<std::uint64_t> w;
  if (rank == 0) {
    std::remove(path.c_str());  // whatever an earlier run left here is not ours
    std::vector<std::uint64_t> seen(static_cast<size_t>(world), 0), written;
    for (;;) {
      bool all_want = true, changed = false;
      for (int r = 1; r < world; r++) {
        if (read_words(want(r), w, 2) && w[1] != 0) {
          if (w[1] != seen[static_cast<size_t>(r)]) {
            seen[static_cast<size_t>(r)] = w[1];
            changed = true;
          }
        } else if (seen[static_cast<size_t>(r)] == 0) {
          all_want = false;
        }
      }
      if (all_want && (changed || written.empty())) {
        written.assign(id_file_words, 0);
        written[0] = kBootMagic;
        written[1] = static_cast<std::uint64_t>(world);
        written[2] = mine;
        std::memcpy(&written[3], id, BCHMC_UNIQUE_ID_BYTES);
        for (int r = 1; r < world; r++) written[3 + kIdWords + static_cast<size_t>(r)] = seen[static_cast<size_t>(r)];
        write_words(path, written);
      }
      if (!written.empty()) {
        bool all_ack = true;
        for (int r = 1; r < world && all_ack; r++)
          all_ack = read_words(ack(r), w, 3) && w[1] == seen[static_cast<size_t>(r)] && w[2] == mine;
        if (all_ack) return;
      }
      if (expired()) throw std::runtime_error("In comm_bootstrap_file: timed out waiting for the other ranks at " + path);
      nap();
    }
  }
  std::remove(ack(rank).c_str());
  write_words(want(rank), {kBootMagic, mine});
  for (;;) {
    if (read_words(path, w, id_file_words) && w[1] == static_cast<std::uint64_t>(world) &&
        w[3 + kIdWords + static_cast<size_t>(rank)] == mine) {
      std::memcpy(id, &w[3], BCHMC_UNIQUE_ID_BYTES);
      write_words(ack(rank), {kBootMagic, mine, w[2]});
      return;
    }
    if (expired()) throw std::runtime_error("In comm_bootstrap_file: timed out waiting for " + path);
    nap();
  }
}

void bootstrap_cleanup(const char *path_c, int rank) {
  if (!path_c) return;
  const std::string path(path_c);
  if (rank == 0) {
    std::remove(path.c_str());
  } else {
    std::remove((path + ".want." + std::to_string(rank)).c_str());
    std::remove((path + ".ack." + std::to_string(rank)).c_str());
  }
}

void comm_bootstrap_file(HamilView *hd, const char *path, int rank, int world, double timeout_s) {
  if (!hd || !path) throw std::runtime_error("In comm_bootstrap_file: bad argument");
  unsigned char id[BCHMC_UNIQUE_ID_BYTES];
  std::memset(id, 0, sizeof id);
  if (rank == 0) {
    const int rc = bchmc_comm_unique_id(id);
    if (rc) throw std::runtime_error(std::string("In bchmc_comm_unique_id: ") + bchmc_strerror(rc));
  }
  bootstrap_exchange_id(path, rank, world, timeout_s, id);
  bchmc_comm *c = nullptr;
  const int rc = bchmc_comm_create(id, rank, world, hd->device, &c);  // ncclCommInitRank: returns once every rank joined
  bootstrap_cleanup(path, rank);  // every rank has read what it needed (rank 0 saw all acks before it got here)
  if (rc) {
    std::string msg = std::string("In bchmc_comm_create: ") + bchmc_strerror(rc);
    if (c && bchmc_comm_last_error(c)[0]) msg += std::string(" (") + bchmc_comm_last_error(c) + ")";
    if (c) bchmc_comm_destroy(c);
    throw std::runtime_error(msg);
  }
  hd->comm = c;
  hd->comm_rank = rank;
}

void comm_release(HamilView *hd) {
  if (hd && hd->comm) {
    bchmc_comm_destroy(hd->comm);
    hd->comm = nullptr;
  }
}

void inputs_changed(HamilView *hd) {
  if (!hd) return;
  hd->inputs_generation++;  // the next engine_for() uploads the arrays again (once, not per call)
  hd->eom.valid = false;
}

void mass_changed(HamilView *hd) {
  if (!hd) return;
  hd->mass_generation++;
  hd->eom.valid = false;
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
                             unsigned long log_cap, unsigned long *n_attempts, char *err, size_t errlen) {
  return guarded(err, errlen, [&] {
    *n_attempts = bchmc_shim::HamiltonianMC(hd, uniform, rng_state, seed, itmax, count_attempts, log, log_cap, nullptr,
                                            nullptr);
  });
}

namespace {
// scripted engine for the CPU tests of the loop's bookkeeping: attempt k returns dH = dH[k], no device involved
struct Script {
  const double *dH;
  unsigned long n, k;
};
int sc_draw(void *, uint64_t, uint64_t) { return 0; }
int sc_setp(void *, const double *) { return 0; }
int sc_attempt(void *e, double, uint64_t neps, double *dH, double terms[6], uint64_t *done) {
  Script *s = static_cast<Script *>(e);
  if (s->k >= s->n) return BCHMC_ERR_STATE;
  *dH = s->dH[s->k++];
  for (int i = 0; i < 6; i++) terms[i] = 0.;
  terms[3] = *dH;  // the whole difference in the kinetic term
  if (done) *done = neps;
  return 0;
}
int sc_accept(void *, int) { return 0; }
}  // namespace

int bchmc_shim_HamiltonianMC_scripted(bchmc_shim::HamilView *hd, const double *script_dH, unsigned long n_script,
                                      bchmc_shim::uniform_fn uniform, void *rng_state, unsigned long itmax,
                                      unsigned long *count_attempts, bchmc_shim::AttemptLog *log, unsigned long log_cap,
                                      unsigned long *n_attempts, char *err, size_t errlen) {
  return guarded(err, errlen, [&] {
    Script sc = {script_dH, n_script, 0};
    const bchmc_shim::ChainOps ops = {sc_draw, sc_setp, sc_attempt, sc_accept};
    *n_attempts = bchmc_shim::HamiltonianMC_ops(hd, ops, &sc, uniform, rng_state, 0, itmax, count_attempts, log, log_cap,
                                                nullptr, nullptr);
  });
}
int bchmc_shim_kinetic_term(bchmc_shim::HamilView *hd, const double *momenta, double *out, char *err, size_t errlen) {
  return guarded(err, errlen, [&] { *out = bchmc_shim::kinetic_term(hd, momenta); });
}
int bchmc_shim_psi(bchmc_shim::HamilView *hd, const double *signal, double *out, char *err, size_t errlen) {
  return guarded(err, errlen, [&] { *out = bchmc_shim::psi(hd, signal); });
}
bchmc_shim::EpsAdapt *bchmc_shim_eps_create(int update_type, unsigned N_a, double acc_min, double acc_max, int down_smooth,
                                            double up_fac, double target, double power, unsigned long s_eps_total) {
  bchmc_shim::EpsAdaptConfig c;
  c.eps_fac_update_type = update_type;
  c.N_a_eps_update = N_a;
  c.acc_min = acc_min;
  c.acc_max = acc_max;
  c.eps_down_smooth = down_smooth;
  c.eps_up_fac = up_fac;
  c.eps_fac_target = target;
  c.eps_fac_power = power;
  c.s_eps_total = s_eps_total;
  try {
    return bchmc_shim::eps_adapt_create(c);
  } catch (const std::runtime_error &) {
    return nullptr;
  }
}
void bchmc_shim_eps_destroy(bchmc_shim::EpsAdapt *e) { bchmc_shim::eps_adapt_destroy(e); }
void bchmc_shim_eps_append(bchmc_shim::EpsAdapt *e, int accepted, double epsilon) {
  bchmc_shim::eps_adapt_append(e, accepted != 0, epsilon);
}
unsigned long bchmc_shim_eps_records(const bchmc_shim::EpsAdapt *e) { return bchmc_shim::eps_adapt_records(e); }
double bchmc_shim_eps_acceptance_rate(const bchmc_shim::EpsAdapt *e) { return bchmc_shim::eps_adapt_acceptance_rate(e); }
int bchmc_shim_update_eps_fac(bchmc_shim::HamilView *hd, char *msg, size_t msglen, char *err, size_t errlen) {
  return guarded(err, errlen, [&] {
    const std::string m = bchmc_shim::update_eps_fac(hd);
    if (msg && msglen) std::snprintf(msg, msglen, "%s", m.c_str());
  });
}
int bchmc_shim_update_tables(bchmc_shim::HamilView *hd, char *err, size_t errlen) {
  return guarded(err, errlen, [&] { bchmc_shim::update_epsilon_acc_rate_tables(hd); });
}
int bchmc_shim_comm_bootstrap_file(bchmc_shim::HamilView *hd, const char *path, int rank, int world, double timeout_s,
                                   char *err, size_t errlen) {
  return guarded(err, errlen, [&] { bchmc_shim::comm_bootstrap_file(hd, path, rank, world, timeout_s); });
}
int bchmc_shim_bootstrap_exchange_id(const char *path, int rank, int world, double timeout_s, unsigned char *id, char *err,
                                     size_t errlen) {
  return guarded(err, errlen, [&] { bchmc_shim::bootstrap_exchange_id(path, rank, world, timeout_s, id); });
}
void bchmc_shim_bootstrap_cleanup(const char *path, int rank) { bchmc_shim::bootstrap_cleanup(path, rank); }
int bchmc_shim_comm_attach(bchmc_shim::HamilView *hd, bchmc_comm *comm, int rank) {
  if (!hd) return 1;
  hd->comm = comm;
  hd->comm_rank = rank;
  return 0;
}
void bchmc_shim_comm_release(bchmc_shim::HamilView *hd) { bchmc_shim::comm_release(hd); }
void bchmc_shim_inputs_changed(bchmc_shim::HamilView *hd) { bchmc_shim::inputs_changed(hd); }
void bchmc_shim_mass_changed(bchmc_shim::HamilView *hd) { bchmc_shim::mass_changed(hd); }
size_t bchmc_shim_sizeof_attempt_log(void) { return sizeof(bchmc_shim::AttemptLog); }
void bchmc_shim_release(bchmc_shim::HamilView *hd) { bchmc_shim::release(hd); }
size_t bchmc_shim_sizeof_view(void) { return sizeof(bchmc_shim::HamilView); }
size_t bchmc_shim_sizeof_numerical(void) { return sizeof(bchmc_shim::HamilNumericalView); }
}
