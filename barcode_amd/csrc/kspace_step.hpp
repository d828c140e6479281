// kspace_step.hpp -- Conversions, spectrum multipliers, first-step kick + drift + Zel'dovich kernel.
// Part of the bchmc engine's kernel set; include through kernels.hpp (definition order matters).
#pragma once
#include "common.hpp"

namespace bchmc {

// ------------------------------------------------------------------------------------------------------
// Precision conversion for the C ABI (host arrays are always double, like the reference's default build).
// ------------------------------------------------------------------------------------------------------
template <typename A, typename B>
__global__ void k_convert(long long n, const A *__restrict__ in, B *__restrict__ out) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    out[i] = (B)in[i];
}

// ------------------------------------------------------------------------------------------------------
// Spectrum multipliers.  convolveInvCorrFuncWithSignal (HMC_help.cc:41-58) multiplies FFT[x] by
// normFS / C(k) (0 where C <= 0) with C read from a FULL n^3 grid at index k + n*(j + n*i), k <= n/2.
// We precompute that factor once per upload on the half-complex layout (always double: it is a k-space weight).
// ------------------------------------------------------------------------------------------------------
__global__ void k_prepare_mult(Geo g, const double *__restrict__ corr, double *__restrict__ mult, double normFS) {
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < g.Nhp;
       idx += (long long)gridDim.x * blockDim.x) {
    const int k = (int)(idx % g.nhp);
    if (k >= g.nh) continue;   // row padding stays zero
    const long long ij = idx / g.nhp;
    const double c = corr[k + (long long)g.n * ij];
    mult[idx] = (c > 0.0) ? normFS / c : 0.;
  }
}

// Device-side trajectory control: the runaway-momentum guard of HMC.cc:360-364 without a host round trip.
struct StepCtl {
  int *stop;                        // set once the guard fired; later kernels leave (q, p) untouched
  unsigned long long *steps_done;   // initialised to neps by the host
  const double *guard_prev;         // sum over k of hw * Re p^(k) after the previous step (= N * p[0]); may be null
  double guard_limit;               // 1e50 * N
  unsigned long long step_index;    // number of completed steps if the guard fires now
};

__global__ void k_init_ctl(int *stop, unsigned long long *steps_done, unsigned long long neps) {
  *stop = 0;
  *steps_done = neps;
}

// ------------------------------------------------------------------------------------------------------
// First half kick + drift + Zel'dovich displacement kernel, all diagonal in k:
//   p^ -= eps/2 * g^                          HMC.cc:293-294
//   q^ += eps * (wM * p^ [+ extra])           HMC.cc:298-339 via HMC_help.cc:41-58 (extra = R2C[p/mass_r])
//   Psi^_j = (k_j/k^2) * (Im phi^, -Re phi^)  EqSolvers.cc:208-268 with phi = -D1*deltaQ*q (Lag2Eul.cc:88)
// c_za = -D1 * deltaQ_factor / N folds in the 1/N of the following C2R (fftwrapper.cc:99-101).
// Psi^ is zero for k^2 <= 1e-14 and on every Nyquist plane.
// ------------------------------------------------------------------------------------------------------
template <typename T, bool DRIFT>
__global__ void __launch_bounds__(256)
k_kick_drift_za(Geo g, C2<T> *__restrict__ qk, C2<T> *__restrict__ pk, const C2<T> *__restrict__ gk,
                const double *__restrict__ wM, const C2<T> *__restrict__ extra, C2<T> *__restrict__ Ck,
                double half_eps, double eps, double c_za, StepCtl ctl) {
  if (DRIFT) {
    if (*ctl.stop) return;
    if (ctl.guard_prev && fabs(*ctl.guard_prev) > ctl.guard_limit) {
      if (blockIdx.x == 0 && threadIdx.x == 0) {
        *ctl.steps_done = ctl.step_index;
        __threadfence();
        *ctl.stop = 1;
      }
      return;  // NB: *stop is only read by LATER kernels, every thread of this one takes this branch
    }
  }
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < g.Nhp;
       idx += (long long)gridDim.x * blockDim.x) {
    const int k = (int)(idx % g.nhp);
    double2 q = ld2<T>(qk, idx);
    if (DRIFT) {
      double2 p = ld2<T>(pk, idx);
      const double2 gg = ld2<T>(gk, idx);
      p.x -= half_eps * gg.x;
      p.y -= half_eps * gg.y;
      st2<T>(pk, idx, p.x, p.y);
      double2 v = make_double2(0., 0.);
      if (wM) {
        const double w = wM[idx];
        v.x = w * p.x;
        v.y = w * p.y;
      }
      if (extra) {
        const double2 e = ld2<T>(extra, idx);
        v.x += e.x;
        v.y += e.y;
      }
      q.x += eps * v.x;
      q.y += eps * v.y;
      st2<T>(qk, idx, q.x, q.y);
    }
    const long long ij = idx / g.nhp;
    const int j = (int)(ij % g.n), i = (int)(ij / g.n);
    const double kx = kval(i, g.n, g.kfac), ky = kval(j, g.n, g.kfac), kz = kval(k, g.n, g.kfac);
    const double ksq = kx * kx + ky * ky + kz * kz;
    double2 ox = make_double2(0., 0.), oy = ox, oz = ox;
    const bool nyq = (i == g.n / 2) || (j == g.n / 2) || (k == g.n / 2);
    if (ksq > 1.e-14 && !nyq) {
      const double fac = 1. / ksq;
      const double pr = c_za * q.x, pi = c_za * q.y;
      const double fx = fac * kx, fy = fac * ky, fz = fac * kz;
      ox = make_double2(fx * pi, fx * -pr);
      oy = make_double2(fy * pi, fy * -pr);
      oz = make_double2(fz * pi, fz * -pr);
    }
    st2<T>(Ck, idx, ox.x, ox.y);
    st2<T>(Ck, idx + g.Nhp, oy.x, oy.y);
    st2<T>(Ck, idx + 2 * g.Nhp, oz.x, oz.y);
  }
}

}  // namespace bchmc
