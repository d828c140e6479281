// eps_adapt.cc -- the reference's step-size adaptation (barlib/src/hmc/leapfrog/time_step.cpp:24-203 and the
// templates of barlib/include/hmc/leapfrog/time_step.hpp:23-75) on the shim's HAMIL_DATA view.  Host-only C++11.
//
// One deliberate difference, inert for a single chain: the "every N_a attempts" trigger (time_step.cpp:115-116,
// 168-169) is evaluated as "the record count crossed a multiple of N_a since the previous call" instead of
// "count % N_a == 0".  With one record per call the two are the same statement; with pooled records of other chains
// (bchmc_eps_exchange) the count advances by several between calls and equality would fire only at lcm(world, N_a).
#include <algorithm>
#include <cmath>
#include <numeric>
#include <sstream>
#include <vector>

#include "bchmc_shim.hpp"

namespace bchmc_shim {

struct EpsAdapt {
  EpsAdaptConfig cfg;
  std::vector<bool> acc_flag_N_a;      // struct_main.h:172
  std::vector<real_prec> epsilon_N_a;  // struct_main.h:173
  ULONG records = 0;                   // entries written so far (own attempts + pooled): the reference's count_attempts
  ULONG checked = 0;                   // `records` at the previous update_eps_fac call
};

EpsAdapt *eps_adapt_create(const EpsAdaptConfig &cfg) {
  if (cfg.N_a_eps_update == 0) throw std::runtime_error("In eps_adapt_create: N_a_eps_update must be positive");
  EpsAdapt *e = new EpsAdapt();
  e->cfg = cfg;
  e->acc_flag_N_a.assign(cfg.N_a_eps_update, false);
  e->epsilon_N_a.assign(cfg.N_a_eps_update, 0.);
  return e;
}

void eps_adapt_destroy(EpsAdapt *e) { delete e; }
ULONG eps_adapt_records(const EpsAdapt *e) { return e->records; }

namespace {

real_prec bool_mean(const std::vector<bool> &input) {  // time_step.cpp:24-28
  real_prec result = static_cast<real_prec>(std::count(input.begin(), input.end(), true));
  return result / static_cast<real_prec>(input.size());
}

real_prec power_mean(real_prec x, real_prec y, real_prec p) {  // math_funcs.cc:36-44
  if (p == 0.) return std::sqrt(x * y);
  return std::pow((std::pow(x, p) + std::pow(y, p)) / 2., 1. / p);
}

// time_step.hpp:23-49.  std::sort there is not stable, so the order of equal epsilons is unspecified upstream;
// a stable sort picks one of the admissible orders (and the same one as barcode_amd/time_step.py).
std::vector<real_prec> acc_sorted_by_epsilon(const EpsAdapt &e) {
  std::vector<size_t> idx(e.epsilon_N_a.size());
  std::iota(idx.begin(), idx.end(), size_t(0));
  std::stable_sort(idx.begin(), idx.end(), [&](size_t a, size_t b) { return e.epsilon_N_a[a] < e.epsilon_N_a[b]; });
  std::vector<real_prec> out;
  out.reserve(idx.size());
  for (size_t i : idx) out.push_back(e.acc_flag_N_a[i] ? 1. : 0.);
  return out;
}

// time_step.cpp:40-104
std::string downwards(HamilNumericalView *n, const EpsAdapt &e) {
  const EpsAdaptConfig &c = e.cfg;
  const real_prec alpha_N_a = bool_mean(e.acc_flag_N_a);
  const real_prec acc_target = (c.acc_max + c.acc_min) / 2.;
  std::string message = "\nadjusted eps_fac downwards to %f";
  std::vector<real_prec> a_sort = acc_sorted_by_epsilon(e);
  // cumulative moving average (time_step.hpp:51-61)
  std::partial_sum(a_sort.begin(), a_sort.end(), a_sort.begin());
  for (size_t i = 1; i < a_sort.size(); ++i) a_sort[i] /= static_cast<real_prec>(i + 1);
  // boxcar smoothing, clipped at the ends (time_step.hpp:63-75)
  std::vector<real_prec> a_sm(a_sort.size());
  const long sz = static_cast<long>(a_sort.size());
  for (long i = 0; i < sz; ++i) {
    const long lo = std::max(i - c.eps_down_smooth, 0L), hi = std::min(i + c.eps_down_smooth + 1, sz);
    a_sm[static_cast<size_t>(i)] = std::accumulate(a_sort.begin() + lo, a_sort.begin() + hi, 0.0) /
                                   static_cast<real_prec>(hi - lo);
  }
  const auto ix_max = std::max_element(a_sm.begin(), a_sm.end());
  if (*ix_max > acc_target) {
    const auto ix_target = std::find_if(ix_max, a_sm.end(), [&](real_prec v) { return v < acc_target; });
    if (ix_target == a_sm.end()) {
      std::stringstream ss;
      ss << "\neps_fac stays at %f (special: alpha_N_a=" << alpha_N_a << ", a_t=" << acc_target
         << ", a_sm_max=" << *ix_max << ")";
      message = ss.str();
    } else {
      std::vector<real_prec> eps_sort(e.epsilon_N_a);
      std::sort(eps_sort.begin(), eps_sort.end());
      n->eps_fac = eps_sort[static_cast<size_t>(ix_target - a_sm.begin())];
    }
  } else if (alpha_N_a == 0.) {
    n->eps_fac = *std::min_element(e.epsilon_N_a.begin(), e.epsilon_N_a.end());
  } else {
    n->eps_fac /= 3.;
  }
  if (n->eps_fac == 0.)
    throw std::runtime_error("In update_eps_fac_acceptance_rate_downwards: epsilon became zero, shouldn't happen!");
  return message;
}

// "count_attempts % every == 0 && count_attempts > 0" (time_step.cpp:115-116, 168-169), as a crossing test
bool crossed(ULONG before, ULONG now, ULONG every) { return every > 0 && now / every != before / every; }

// time_step.cpp:106-135
std::string acceptance_rate(HamilNumericalView *n, EpsAdapt &e, bool due) {
  if (!due) return "";
  const EpsAdaptConfig &c = e.cfg;
  const real_prec alpha_N_a = bool_mean(e.acc_flag_N_a);
  if (alpha_N_a < c.acc_min) return downwards(n, e);
  if (alpha_N_a > c.acc_max) {
    const real_prec acc_target = (c.acc_max + c.acc_min) / 2.;
    n->eps_fac *= c.eps_up_fac * (alpha_N_a / acc_target);
    return "\nadjusted eps_fac upwards to %f";
  }
  return "\nnot adjusting eps_fac, stays at %f";
}

}  // namespace

real_prec eps_adapt_acceptance_rate(const EpsAdapt *e) { return bool_mean(e->acc_flag_N_a); }

std::string update_eps_fac(HamilView *hd) {
  if (!hd || !hd->numerical) throw std::runtime_error("In update_eps_fac: HAMIL_DATA without numerical");
  EpsAdapt *e = hd->eps;
  if (!e) return "";
  HamilNumericalView *n = hd->numerical;
  const ULONG before = e->checked;
  e->checked = e->records;
  std::string message;
  switch (e->cfg.eps_fac_update_type) {  // time_step.cpp:154-184
    case 0: break;
    case 1:
      if (crossed(before, e->records, e->cfg.s_eps_total)) {
        n->eps_fac = power_mean(n->eps_fac, e->cfg.eps_fac_target, e->cfg.eps_fac_power);
        message = "  updating eps_fac to %f";
      }
      break;
    case 2: message = acceptance_rate(n, *e, crossed(before, e->records, e->cfg.N_a_eps_update)); break;
    case 3:
      if (n->iGibbs == 1 && n->rejections > 0) {  // fast initial phase, time_step.cpp:137-149
        n->eps_fac /= 2.;
        message = "\nadjusted eps_fac downwards to %f";
      } else {
        message = acceptance_rate(n, *e, crossed(before, e->records, e->cfg.N_a_eps_update));
      }
      break;
    default: break;  // the reference's switch has no default either
  }
  return message;
}

void eps_adapt_append(EpsAdapt *e, bool accepted, real_prec epsilon) {
  e->records++;
  const size_t ix = static_cast<size_t>((e->records - 1) % e->cfg.N_a_eps_update);  // time_step.cpp:192-193
  e->acc_flag_N_a[ix] = accepted;
  e->epsilon_N_a[ix] = epsilon;
}

void update_epsilon_acc_rate_tables(HamilView *hd) {
  if (!hd || !hd->numerical) throw std::runtime_error("In update_epsilon_acc_rate_tables: HAMIL_DATA without numerical");
  if (hd->eps) eps_adapt_append(hd->eps, hd->numerical->accepted, hd->numerical->epsilon);
}

}  // namespace bchmc_shim
