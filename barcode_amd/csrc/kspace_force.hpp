// kspace_force.hpp -- Force assembly, fused step boundary, rollback, Parseval energies, element-wise helpers, GRF likelihood.
// Part of the bchmc engine's kernel set; include through kernels.hpp (definition order matters).
#pragma once
#include "common.hpp"

namespace bchmc {

// ------------------------------------------------------------------------------------------------------
// Force assembly in k-space + second half kick + guard sum.
//   h^ = sum_j (k_j/k^2) (Im V^_j, -Re V^_j), Nyquist planes and k = 0 -> 0   gradient.cpp:167-210
//   g^ = a * wS * q^ + b * h^                                                  HMC.cc:170-173,205; HMC_models.cc:458-470
//   p^ -= c * g^                                                               HMC.cc:351-352
// like_mode 0: h^ from the three V^ (calc_h 0/2/3); 1: h^ = Ck[0] as is (calc_h 1, GRF); 2: no likelihood term.
// The guard slot receives sum_k hw_k Re p^_k = N * p[0] (HMC.cc:360).
// ------------------------------------------------------------------------------------------------------
// g^ of one k-space element (shared by k_assemble and the fused step kernel)
template <typename T>
__device__ __forceinline__ double2 assemble_g(const Geo &g, const C2<T> *__restrict__ Ck, const double2 q,
                                              const double *__restrict__ wS, long long idx, int k, double a, double b,
                                              int like_mode) {
  double2 hk = make_double2(0., 0.);
  if (like_mode == 0) {
    const long long ij = idx / g.nhp;
    const int j = (int)(ij % g.n), i = (int)(ij / g.n);
    const bool nyq = (i == g.n / 2) || (j == g.n / 2) || (k == g.n / 2);
    const double kx = kval(i, g.n, g.kfac), ky = kval(j, g.n, g.kfac), kz = kval(k, g.n, g.kfac);
    const double kmod = kx * kx + ky * ky + kz * kz;
    if (kmod > 0 && !nyq) {
      const double f = 1 / kmod;
      const double2 vx = ld2<T>(Ck, idx), vy = ld2<T>(Ck, idx + g.Nhp), vz = ld2<T>(Ck, idx + 2 * g.Nhp);
      const double fx = kx * f, fy = ky * f, fz = kz * f;
      hk.x = fx * vx.y + fy * vy.y + fz * vz.y;
      hk.y = -(fx * vx.x) - fy * vy.x - fz * vz.x;
    }
  } else if (like_mode == 1) {
    hk = ld2<T>(Ck, idx);
  }
  double2 gg = make_double2(b * hk.x, b * hk.y);
  if (a != 0.) {
    const double w = a * wS[idx];
    gg.x += w * q.x;
    gg.y += w * q.y;
  }
  return gg;
}

template <typename T, bool KICK>
__global__ void __launch_bounds__(256)
k_assemble(Geo g, const C2<T> *__restrict__ Ck, const C2<T> *__restrict__ qk, const double *__restrict__ wS,
           C2<T> *__restrict__ gk, C2<T> *__restrict__ pk, double a, double b, int like_mode, double c_kick,
           double *guard_slot, const int *stop) {
  __shared__ double red[4];
  if (KICK && *stop) return;
  double gsum = 0.;
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < g.Nhp;
       idx += (long long)gridDim.x * blockDim.x) {
    const int k = (int)(idx % g.nhp);
    const double2 q = (a != 0.) ? ld2<T>(qk, idx) : make_double2(0., 0.);
    const double2 gg = assemble_g<T>(g, Ck, q, wS, idx, k, a, b, like_mode);
    st2<T>(gk, idx, gg.x, gg.y);
    if (KICK) {
      double2 p = ld2<T>(pk, idx);
      p.x -= c_kick * gg.x;
      p.y -= c_kick * gg.y;
      st2<T>(pk, idx, p.x, p.y);
      const double hw = (k == 0 || ((g.n & 1) == 0 && k == g.n / 2)) ? 1. : 2.;
      if (k < g.nh) gsum += hw * p.x;
    }
  }
  if (KICK) {
    gsum = block_sum(gsum, red);
    if (threadIdx.x == 0) atomic_add_r(guard_slot, gsum);
  }
}

// ------------------------------------------------------------------------------------------------------
// Fused interior step boundary: the k_assemble<KICK> of step s followed by the k_kick_drift_za of step s + 1 in
// one pass (HMC.cc:351-352, then 290-291, 300-337 of the next iteration):
//   g^ = a wS q^ + b h^;  p_end = p - (eps/2) g^  [guard sum of step s];  p' = p_end - (eps/2) g^;
//   q' = q + eps wM p';  Psi^' from q'.
// (q', p') go to the other buffer of a ping-pong pair: if the guard of step s turns out to have tripped, the next
// kernel of this kind stops the trajectory and k_rollback rebuilds the end-of-step-s state from the two buffers.
// V^ is read from and Psi^' written to the same Ck elements by the same thread.
// ------------------------------------------------------------------------------------------------------
// LAST = true is the boundary after the final step: only the half kick (p_out = p_end, which may alias p_in), g^
// stored to gk (hd->gradpsi), q untouched.
template <typename T, bool LAST>
__global__ void __launch_bounds__(256)
k_step_boundary(Geo g, C2<T> *Ck, const C2<T> *q_in, const C2<T> *p_in, C2<T> *q_out, C2<T> *p_out,
                C2<T> *__restrict__ gk, const double *__restrict__ wS, const double *__restrict__ wM, double a,
                double b, int like_mode, double half_eps, double eps, double c_za, double *guard_slot, StepCtl ctl) {
  __shared__ double red[4];
  if (*ctl.stop) return;
  if (ctl.guard_prev && fabs(*ctl.guard_prev) > ctl.guard_limit) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      *ctl.steps_done = ctl.step_index;
      __threadfence();
      *ctl.stop = 1;
    }
    return;  // NB: *stop is only read by LATER kernels, every thread of this one takes this branch
  }
  double gsum = 0.;
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < g.Nhp;
       idx += (long long)gridDim.x * blockDim.x) {
    const int k = (int)(idx % g.nhp);
    double2 q = ld2<T>(q_in, idx);
    const double2 gg = assemble_g<T>(g, Ck, q, wS, idx, k, a, b, like_mode);
    double2 p = ld2<T>(p_in, idx);
    // The unfused kernels store p_end and g^ (rounded to T) between the two half kicks: same roundings here, so
    // that fused and unfused trajectories are the same numbers (bit-identical for T = double).
    C2<T> pe, gs;
    pe.x = (T)(p.x - half_eps * gg.x);
    pe.y = (T)(p.y - half_eps * gg.y);
    gs.x = (T)gg.x;
    gs.y = (T)gg.y;
    const double hw = (k == 0 || ((g.n & 1) == 0 && k == g.n / 2)) ? 1. : 2.;
    if (k < g.nh) gsum += hw * (double)pe.x;
    if (LAST) {
      p_out[idx] = pe;
      gk[idx] = gs;
      continue;
    }
    p.x = (double)pe.x - half_eps * (double)gs.x;
    p.y = (double)pe.y - half_eps * (double)gs.y;
    st2<T>(p_out, idx, p.x, p.y);
    if (wM) {
      const double w = wM[idx];
      q.x += eps * (w * p.x);
      q.y += eps * (w * p.y);
    }
    st2<T>(q_out, idx, q.x, q.y);
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
  gsum = block_sum(gsum, red);
  if (threadIdx.x == 0) atomic_add_r(guard_slot, gsum);
}

// After a trajectory of fused steps: if the guard stopped it at step s (= *steps_done, s >= 1), the state the
// reference would return is (q_s, p_s_end) = (q of the buffer step-boundary s - 1 read, mean of the momenta it
// read and wrote: p_read - (eps/2) g and p_written + (eps/2) g are the same number).  buf[i] are the ping-pong
// pairs; boundary j reads pair j % 2; the result goes to (q_dst, p_dst), which may alias either pair.
template <typename T>
__global__ void k_rollback(long long n, const int *__restrict__ stop, const unsigned long long *__restrict__ steps_done,
                           const C2<T> *q0, const C2<T> *p0, const C2<T> *q1, const C2<T> *p1, C2<T> *q_dst,
                           C2<T> *p_dst) {
  if (!*stop) return;
  const unsigned long long s = *steps_done;  // boundary s detected the trip; boundary s - 1 wrote the overshoot
  const bool read_is_0 = ((s - 1) & 1) == 0;
  const C2<T> *qr = read_is_0 ? q0 : q1, *pr = read_is_0 ? p0 : p1, *pw = read_is_0 ? p1 : p0;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const double2 a = ld2<T>(pr, i), b = ld2<T>(pw, i), q = ld2<T>(qr, i);
    st2<T>(p_dst, i, 0.5 * a.x + 0.5 * b.x, 0.5 * a.y + 0.5 * b.y);
    st2<T>(q_dst, i, q.x, q.y);
  }
}

// sum_k hw_k * w_k * |x^_k|^2 per block: Parseval form of sum_x x * IFFT[w * FFT x]
// (kinetic_term HMC.cc:101-115, prior_gaussian_log_prior gaussian.cpp:24-32).
template <typename T>
__global__ void __launch_bounds__(256)
k_parseval(Geo g, const C2<T> *__restrict__ xk, const double *__restrict__ w, double *__restrict__ partials) {
  __shared__ double red[4];
  double s = 0.;
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < g.Nhp;
       idx += (long long)gridDim.x * blockDim.x) {
    const int k = (int)(idx % g.nhp);
    const double hw = (k == 0 || ((g.n & 1) == 0 && k == g.n / 2)) ? 1. : 2.;
    const double2 x = ld2<T>(xk, idx);
    if (k < g.nh) s += hw * w[idx] * (x.x * x.x + x.y * x.y);
  }
  s = block_sum(s, red);
  if (threadIdx.x == 0) partials[blockIdx.x] = s;
}

template <typename T>
__global__ void k_scale_c(long long n, const C2<T> *__restrict__ in, C2<T> *__restrict__ out, double s) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const double2 v = ld2<T>(in, i);
    st2<T>(out, i, v.x * s, v.y * s);
  }
}

template <typename T>
__global__ void k_add_r(long long n, const T *__restrict__ a, const T *__restrict__ b, T *__restrict__ out) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    out[i] = a[i] + b[i];
}

// Real-space mass term: t = p / mass_r (0 where mass_r <= 0), HMC.cc:317-327.
template <typename T>
__global__ void k_div_mass_r(long long n, const T *__restrict__ p, const T *__restrict__ mass_r, T *__restrict__ out) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const double m = mass_r[i];
    out[i] = (m > 0.0) ? (T)((double)p[i] * (1. / m)) : T(0);
  }
}

// sum 0.5 * p * (p / mass_r): real-space part of kinetic_term (HMC.cc:88-110)
template <typename T>
__global__ void __launch_bounds__(256)
k_kin_rs(long long n, const T *__restrict__ p, const T *__restrict__ mass_r, double *__restrict__ partials) {
  __shared__ double red[4];
  double s = 0.;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const double m = mass_r[i], pv = p[i];
    const double invM = (m > 0.0) ? 1. / m : 0.;
    s += 0.5 * pv * (invM * pv);
  }
  s = block_sum(s, red);
  if (threadIdx.x == 0) partials[blockIdx.x] = s;
}

// GRF likelihood (gaussian_random_field.cpp:25-52): force (q - nobs)/sigma^2 and energy, window-masked.
template <typename T>
__global__ void k_grf_grad(long long n, const T *__restrict__ q, const T *__restrict__ nobs, const T *__restrict__ noise,
                           const T *__restrict__ window, T *__restrict__ out) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const double s = noise[i];
    out[i] = ((double)window[i] > 0.) ? (T)(((double)q[i] - (double)nobs[i]) / (s * s)) : T(0);
  }
}

template <typename T>
__global__ void __launch_bounds__(256)
k_grf_loglike(long long n, const T *__restrict__ q, const T *__restrict__ nobs, const T *__restrict__ noise,
              const T *__restrict__ window, double *__restrict__ partials) {
  __shared__ double red[4];
  double s = 0.;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    if ((double)window[i] > 0.) {
      const double t = ((double)q[i] - (double)nobs[i]) / (double)noise[i];
      s += 0.5 * (t * t);
    }
  s = block_sum(s, red);
  if (threadIdx.x == 0) partials[blockIdx.x] = s;
}

}  // namespace bchmc
