// kernels.hpp -- HIP kernels of the bchmc engine (gfx950 / CDNA4, wave64).
//
// State lives in Fourier space: qk = R2C[q], pk = R2C[p] (unnormalised forward transforms, half-complex
// n x n x (n/2+1), z fastest).  Everything that is diagonal in k (prior force S^-1 q, drift M^-1 p, the
// Zel'dovich displacement kernel, the inverse-Laplacian-divergence of V, kicks) is done pointwise on those
// arrays; only the particle-mesh part (displace, SPH scatter, likelihood partials, SPH-gradient gather)
// runs in real space.  Reference lines restated by each kernel are cited at the kernel.
//
// Every kernel is a template on T, the STORAGE type of the field arrays (double = reference DOUBLE_PREC,
// float = BASELINE config 5).  k-space arithmetic, k-vectors, reductions and the per-cell likelihood are always
// done in double; the particle-mesh kernels (positions, spline evaluations, LDS accumulation) compute in T.
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

// ------------------------------------------------------------------------------------------------------
// Particle positions: disp_part (disp_part.cc:55-126) + plane-parallel RSD (rsd.cc:28-68, Lag2Eul.cc:378-401)
// ------------------------------------------------------------------------------------------------------
struct PosPar {
  double d, L;
  double cpecvel, v_norm;  // c_pecvel(a) and 1/Hub/a
  int rsd, periodic;
};

// Compiled without FMA contraction: every kernel that calls this gets bit-identical positions (the sorted
// path derives a particle's tile in one kernel and its LDS-local home cell in another), and the operation
// sequence is the reference's (multiply, add, add, fmod) as its x86-64 build executes it.
template <typename T>
__device__ __forceinline__ void particle_pos(const PosPar &pp, int i, int j, int k, T psx, T psy, T psz, T &x, T &y,
                                             T &z) {
#pragma clang fp contract(off)
  const T d = (T)pp.d, L = (T)pp.L;
  x = d * (T)i + T(0.5) * d + psx;
  y = d * (T)j + T(0.5) * d + psy;
  z = d * (T)k + T(0.5) * d + psz;
  if (pp.periodic) {
    x = pacman(x, L);
    y = pacman(y, L);
    z = pacman(z, L);
  }
  if (pp.rsd) {
    const T vz = (T)pp.cpecvel * psz;
    z = z + vz * (T)pp.v_norm;
    if (pp.periodic) z = pacman(z, L);
  }
}

template <typename T>
__device__ __forceinline__ bool pos_ok(const Geo &g, T x, T y, T z) {
  // false for non-finite positions (blown-up trajectory): those must never be used as indices
  const T L = (T)g.L;
  return x >= T(0) && x <= L && y >= T(0) && y <= L && z >= T(0) && z <= L;
}

template <typename T>
__global__ void k_positions(Geo g, PosPar pp, const T *__restrict__ psi, T *__restrict__ out, int comp) {
  for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < g.N;
       p += (long long)gridDim.x * blockDim.x) {
    const int k = (int)(p % g.n);
    const long long ij = p / g.n;
    const int j = (int)(ij % g.n), i = (int)(ij / g.n);
    T x, y, z;
    particle_pos<T>(pp, i, j, k, psi[p], psi[p + g.N], psi[p + 2 * g.N], x, y, z);
    out[p] = comp == 0 ? x : (comp == 1 ? y : z);
  }
}

// ------------------------------------------------------------------------------------------------------
// SPH mass assignment (getDensity_SPH, massFunctions.cc:392-495; kernel W_4 at 366-384).
// ------------------------------------------------------------------------------------------------------
struct SphPar {
  double h, h_inv, w_norm;  // kernel scale, its inverse, 1/pi/h^3
  double r2_lim;            // 4 h^2 (1 + 1e-12): beyond this r/h <= 2 cannot hold
  double min1, min2, min3;
  int reach;
};

// SPH_kernel_3D (massFunctions.cc:366-384), reference form
__device__ __forceinline__ double sph_w(double q, double w_norm) {
  if (q <= 1.) return w_norm * (1 - 3. / 2 * q * q + 3. / 4 * q * q * q);
  const double t = 2. - q;
  return w_norm * (1. / 4 * (t * t * t));
}

// W_4 (massFunctions.cc:366-384), branch-free, valid for 0 <= q <= 2, with the normalisation folded into the
// coefficients (w = w_norm).
template <typename T>
__device__ __forceinline__ T sph_w_folded(T q, T w) {
  const T inner = r_fma(q * q, r_fma(T(0.75) * w, q, T(-1.5) * w), w);  // w (1 - 3/2 q^2 + 3/4 q^3)
  const T t = T(2) - q;
  const T outer = (T(0.25) * w * t) * (t * t);
  return (q <= T(1)) ? inner : outer;
}

// dW_4/dq / q in h units times `norm` (grad_SPH_kernel_3D_h_units, SPH_kernel.cpp:148-208), branch-free, folded
// coefficients; q_sq in [0, 4]: q_sq + tiny instead of max(q_sq, tiny) (identical unless q_sq < 1e-264).
template <typename T>
__device__ __forceinline__ T sph_grad_folded(T q_sq, T norm) {
  const T rq = fast_rsqrt(q_sq + tiny_pos<T>());
  const T q = q_sq * rq;
  const T inner = r_fma(T(2.25) * norm, q, T(-3) * norm);
  const T qm2 = q - T(2);
  const T outer = ((qm2 * qm2) * (T(-0.75) * norm)) * rq;
  return (q_sq > T(1)) ? outer : inner;
}

template <typename T>
__device__ __forceinline__ bool in_domain(const Geo &g, const SphPar &sp, T x, T y, T z) {
  // massFunctions.cc:426
  const T L = (T)g.L, m1 = (T)sp.min1, m2 = (T)sp.min2, m3 = (T)sp.min3;
  return (x >= m1 && x < m1 + L) && (y >= m2 && y < m2 + L) && (z >= m3 && z < m3 + L);
}

// Direct version: one thread per particle, global atomics (fallback when no tile shape divides the grid).
// Visits the (2*reach+1)^3 cube like the reference and keeps its `r/h <= 2` decision, but rejects
// columns/cells on squared distance before paying for sqrt and the atomic.
template <typename T>
__global__ void __launch_bounds__(256)
k_scatter_sph(Geo g, PosPar pp, SphPar sp, const T *__restrict__ psi, T *__restrict__ rho) {
  const long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (p >= g.N) return;
  const int k = (int)(p % g.n);
  const long long ij = p / g.n;
  const int j = (int)(ij % g.n), i = (int)(ij / g.n);
  T xt, yt, zt;
  particle_pos<T>(pp, i, j, k, psi[p], psi[p + g.N], psi[p + 2 * g.N], xt, yt, zt);
  if (!in_domain(g, sp, xt, yt, zt)) return;
  const double x = xt, y = yt, z = zt;
  const int n = g.n;
  const double d = g.d;
  const long long ix = (long long)(x / d), iy = (long long)(y / d), iz = (long long)(z / d);
  const double ccx = ((double)ix + 0.5) * d, ccy = ((double)iy + 0.5) * d, ccz = ((double)iz + 0.5) * d;
  const int R = sp.reach;
  for (int i1 = -R; i1 <= R; ++i1) {
    const double dx = x - (ccx + (double)i1 * d);
    const double dx2 = dx * dx;
    if (dx2 > sp.r2_lim) continue;
    const long long kx = (ix + i1 + (long long)n * 4) % n;
    for (int i2 = -R; i2 <= R; ++i2) {
      const double dy = y - (ccy + (double)i2 * d);
      const double r2ab = dx2 + dy * dy;
      if (r2ab > sp.r2_lim) continue;
      const long long ky = (iy + i2 + (long long)n * 4) % n;
      T *row = rho + (long long)n * (ky + (long long)n * kx);
      for (int i3 = -R; i3 <= R; ++i3) {
        const double dz = z - (ccz + (double)i3 * d);
        const double r2 = r2ab + dz * dz;
        if (r2 > sp.r2_lim) continue;
        const double r = sqrt(r2);
        const double q = r / sp.h;
        if (q <= 2.) {
          const long long kz = (iz + i3 + (long long)n * 4) % n;
          atomic_add_r(row + kz, (T)sph_w(q, sp.w_norm));
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------
// Reductions (always double)
// ------------------------------------------------------------------------------------------------------
constexpr int kRedBlocks = 1024;  // fixed partial count -> deterministic two-stage sums

template <typename T>
__global__ void __launch_bounds__(256) k_sum(const T *__restrict__ a, long long n, double *__restrict__ partials) {
  __shared__ double red[4];
  double s = 0.;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    s += (double)a[i];
  s = block_sum(s, red);
  if (threadIdx.x == 0) partials[blockIdx.x] = s;
}

// Sum of the kRedBlocks partials, identical in every block that calls it (deterministic order).
__device__ __forceinline__ double sum_partials(const double *__restrict__ partials, double *red) {
  double s = 0.;
  for (int i = threadIdx.x; i < kRedBlocks; i += blockDim.x) s += partials[i];
  s = block_sum(s, red);
  __shared__ double bc;
  if (threadIdx.x == 0) bc = s;
  __syncthreads();
  return bc;
}

// ------------------------------------------------------------------------------------------------------
// overdens (massFunctions.cc:30-47) fused with the per-cell likelihood partial
// (gaussian_independent.cpp:24-42, poissonian.cpp:19-34, lognormal_independent.cpp:40-55).
// ------------------------------------------------------------------------------------------------------
struct LikePar {
  double rho_c, biasP, biasE, delta_min;
  int likelihood;
  int bias_is_identity;  // biasE == 1: pow(x, 1) == x and pow(x, 0) == 1 exactly, skip the pow calls
};

__device__ __forceinline__ double pow_bias(double x, const LikePar &lp) {
  return lp.bias_is_identity ? x : pow(x, lp.biasE);
}

template <typename T>
__global__ void __launch_bounds__(256)
k_partial_like(Geo g, LikePar lp, const T *__restrict__ rho, const double *__restrict__ rho_partials,
               const T *__restrict__ nobs, const T *__restrict__ noise, const T *__restrict__ window,
               T *__restrict__ plike) {
  __shared__ double red[4];
  const double nmean = sum_partials(rho_partials, red) / (double)g.N;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < g.N;
       i += (long long)gridDim.x * blockDim.x) {
    const double dX = (double)rho[i] / nmean - 1.;
    const double w = window[i];
    double out = 0.;
    if (lp.likelihood == 1) {
      const double Lambda = w * lp.rho_c * pow_bias(1. + lp.biasP * dX, lp);
      if ((w > 0.) && (Lambda > 0.0)) {
        const double s = noise[i];
        out = ((double)nobs[i] - Lambda) / (s * s);
      }
    } else if (lp.likelihood == 0) {
      const double dens = 1. + lp.biasP * dX;
      if ((w > 0.0) && (dens > 0.0)) {
        const double Lambda = w * lp.rho_c * pow_bias(dens, lp);
        const double dpow = lp.bias_is_identity ? 1. : pow(dens, lp.biasE - 1);
        out = (1 - (double)nobs[i] / Lambda) * lp.rho_c * lp.biasE * lp.biasP * dpow;
      }
    } else {  // 2: log-normal
      if (w > 0.) {
        const double Lambda = log(lp.rho_c * pow_bias(1. + lp.biasP * dX, lp));
        const double s = noise[i];
        out = ((double)nobs[i] - Lambda) / (s * s);
      }
    }
    plike[i] = (T)out;
  }
}

// -log L per cell summed per block (gaussian_independent.cpp:82-89, poissonian.cpp:62-71,
// lognormal_independent.cpp:111-121); the host adds the kRedBlocks partials.
template <typename T>
__global__ void __launch_bounds__(256)
k_loglike(Geo g, LikePar lp, const T *__restrict__ rho, const double *__restrict__ rho_partials,
          const T *__restrict__ nobs, const T *__restrict__ noise, const T *__restrict__ window,
          double *__restrict__ out_partials) {
  __shared__ double red[4];
  const double nmean = sum_partials(rho_partials, red) / (double)g.N;
  double acc = 0.;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < g.N;
       i += (long long)gridDim.x * blockDim.x) {
    const double dX = (double)rho[i] / nmean - 1.;
    const double w = window[i];
    if (lp.likelihood == 1) {
      const double Lambda = w * lp.rho_c * pow_bias(1. + lp.biasP * dX, lp);
      if ((w > 0.) && (Lambda > 0.0)) {
        const double t = (Lambda - (double)nobs[i]) / (double)noise[i];
        acc += 0.5 * (t * t);
      }
    } else if (lp.likelihood == 0) {
      const double Lambda = w * lp.rho_c * pow_bias(1. + lp.biasP * dX, lp);
      if ((w > 0.) && (Lambda > 0.0)) acc += Lambda - (double)nobs[i] * log(Lambda);
    } else {
      double dc = dX;
      if (dc < lp.delta_min) dc = lp.delta_min;
      const double Lambda = log(lp.rho_c * (1. + dc));
      if (w > 0.) {
        const double resid = Lambda - (double)nobs[i];
        const double s = noise[i];
        acc += 0.5 * resid * resid / (s * s);
      }
    }
  }
  acc = block_sum(acc, red);
  if (threadIdx.x == 0) out_partials[blockIdx.x] = acc;
}

template <typename T>
__global__ void __launch_bounds__(256)
k_overdens(Geo g, const T *__restrict__ rho, const double *__restrict__ rho_partials, T *__restrict__ out) {
  __shared__ double red[4];
  const double nmean = sum_partials(rho_partials, red) / (double)g.N;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < g.N;
       i += (long long)gridDim.x * blockDim.x)
    out[i] = (T)((double)rho[i] / nmean - 1.);
}

// ------------------------------------------------------------------------------------------------------
// SPH-kernel adjoint gather V(q) (likelihood_calc_V_SPH, HMC_models.cc:200-303; inner loop 77-128;
// gradient of W_4 in h units, SPH_kernel.cpp:148-208).  Pure gather over the stencil hull
// (SPH_kernel.cpp:110-139): `ncol` (i, j) columns with an inclusive k-range each.
// ------------------------------------------------------------------------------------------------------
struct HullPar {
  const int4 *cols;  // {i, j, k_begin, k_last}
  int ncol;
  double h_inv, d_h;       // 1/h, d/h
  double norm;             // 1 / (pi h^4)
  double normalize;        // rho_c * V / N
  double f1;               // fgrow(a), applied to V_z under RSD (HMC_models.cc:295-300)
};

// Direct version (fallback): one thread per particle, part_like read from global memory with periodic wrap.
template <typename T>
__global__ void __launch_bounds__(256)
k_gather_sph(Geo g, PosPar pp, HullPar hp, const T *__restrict__ psi, const T *__restrict__ plike, T *__restrict__ V) {
  extern __shared__ int4 s_cols_direct[];
  for (int m = threadIdx.x; m < hp.ncol; m += blockDim.x) s_cols_direct[m] = hp.cols[m];
  __syncthreads();
  const long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (p >= g.N) return;
  const int n = g.n;
  const int k = (int)(p % n);
  const long long ij = p / n;
  const int j = (int)(ij % n), i = (int)(ij / n);
  T xt, yt, zt;
  particle_pos<T>(pp, i, j, k, psi[p], psi[p + g.N], psi[p + 2 * g.N], xt, yt, zt);
  // A non-finite position (blown-up trajectory) must not index out of bounds: such a particle gets V = 0.
  if (!pos_ok(g, xt, yt, zt)) {
    V[p] = T(0);
    V[p + g.N] = T(0);
    V[p + 2 * g.N] = T(0);
    return;
  }
  const double px = xt, py = yt, pz = zt;
  const int ix = (int)(px / g.d), iy = (int)(py / g.d), iz = (int)(pz / g.d);
  const double d_h = hp.d_h;
  const double dpcx = px * hp.h_inv - ((double)ix + 0.5) * d_h;
  const double dpcy = py * hp.h_inv - ((double)iy + 0.5) * d_h;
  const double dpcz = pz * hp.h_inv - ((double)iz + 0.5) * d_h;
  double ox = 0., oy = 0., oz = 0.;
  for (int m = 0; m < hp.ncol; ++m) {
    const int4 c = s_cols_direct[m];
    const double xh = dpcx - (double)c.x * d_h;
    const double yh = dpcy - (double)c.y * d_h;
    const double r2ab = xh * xh + yh * yh;
    if (r2ab > 4.) continue;  // q_sq > 4 -> zero gradient for the whole column
    const int kx = (ix + c.x + 4 * n) % n, ky = (iy + c.y + 4 * n) % n;
    const T *row = plike + (long long)n * (ky + (long long)n * kx);
    double zh = dpcz - (double)c.z * d_h;
    for (int i3 = c.z; i3 <= c.w; ++i3) {
      const double q_sq = r2ab + zh * zh;
      if (q_sq <= 4.) {
        const double q = sqrt(q_sq);
        double partial;
        if (q_sq > 1.) {
          const double qm2 = q - 2.;
          partial = -0.75 * qm2 * qm2 * hp.norm / q;
        } else {
          partial = (2.25 * q - 3.) * hp.norm;
        }
        const int kz = (iz + i3 + 4 * n) % n;
        const double common = (double)row[kz] * partial;
        ox += common * xh;
        oy += common * yh;
        oz += common * zh;
      }
      zh -= d_h;
    }
  }
  ox *= hp.normalize;
  oy *= hp.normalize;
  oz *= hp.normalize;
  if (pp.rsd) oz += hp.f1 * oz;
  V[p] = (T)ox;
  V[p + g.N] = (T)oy;
  V[p + 2 * g.N] = (T)oz;
}

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

// ------------------------------------------------------------------------------------------------------
// NGP / CIC / TSC mass assignment (forward model only: getDensity_NGP massFunctions.cc:49-98,
// getDensity_CIC :100-164 with getCICcells/getCICweights interpolate_grid.cpp:27-79, getDensity_TSC :167-364).
// One thread per particle, 1 / 8 / 27 global atomics.  Index and weight formulas are the reference's,
// including the cell-centred CIC shift (x - d/2 wrapped) and TSC's inclusive `<= min + L` domain test.
// ------------------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256)
k_scatter_low_order(Geo g, PosPar pp, SphPar sp, int mk, const T *__restrict__ psi, T *__restrict__ rho) {
  const long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (p >= g.N) return;
  const int n = g.n;
  const int k = (int)(p % n);
  const long long ij = p / n;
  const int j = (int)(ij % n), i = (int)(ij / n);
  T xt, yt, zt;
  particle_pos<T>(pp, i, j, k, psi[p], psi[p + g.N], psi[p + 2 * g.N], xt, yt, zt);
  const double x = xt, y = yt, z = zt;
  const double d = g.d, L = g.L;
  if (mk == 2) {
    if (!((x >= sp.min1 && x <= sp.min1 + L) && (y >= sp.min2 && y <= sp.min2 + L) && (z >= sp.min3 && z <= sp.min3 + L)))
      return;
  } else {
    if (!((x >= sp.min1 && x < sp.min1 + L) && (y >= sp.min2 && y < sp.min2 + L) && (z >= sp.min3 && z < sp.min3 + L)))
      return;
  }
#define RHO_AT(a, b, c) (rho + (c) + (long long)n * ((b) + (long long)n * (a)))
  if (mk == 0) {
    const unsigned ci = (unsigned)floor((x - sp.min1) / d) % n, cj = (unsigned)floor((y - sp.min2) / d) % n,
                   ck = (unsigned)floor((z - sp.min3) / d) % n;
    atomic_add_r(RHO_AT(ci, cj, ck), T(1));
  } else if (mk == 1) {
    double q[3] = {x - 0.5 * d, y - 0.5 * d, z - 0.5 * d};
    long long c1[3], c2[3];
    double dx[3], tx[3];
#pragma unroll
    for (int a = 0; a < 3; a++) {
      q[a] = pacman(q[a], L);
      c1[a] = (long long)(q[a] / d);
      c1[a] = (c1[a] + n) % n;
      c2[a] = (c1[a] + 1) % n;
      dx[a] = q[a] / d - (double)c1[a];
      tx[a] = 1. - dx[a];
    }
    const double mass = 1.;
    atomic_add_r(RHO_AT(c1[0], c1[1], c1[2]), (T)(mass * tx[0] * tx[1] * tx[2]));
    atomic_add_r(RHO_AT(c2[0], c1[1], c1[2]), (T)(mass * dx[0] * tx[1] * tx[2]));
    atomic_add_r(RHO_AT(c1[0], c2[1], c1[2]), (T)(mass * tx[0] * dx[1] * tx[2]));
    atomic_add_r(RHO_AT(c1[0], c1[1], c2[2]), (T)(mass * tx[0] * tx[1] * dx[2]));
    atomic_add_r(RHO_AT(c2[0], c2[1], c1[2]), (T)(mass * dx[0] * dx[1] * tx[2]));
    atomic_add_r(RHO_AT(c2[0], c1[1], c2[2]), (T)(mass * dx[0] * tx[1] * dx[2]));
    atomic_add_r(RHO_AT(c1[0], c2[1], c2[2]), (T)(mass * tx[0] * dx[1] * dx[2]));
    atomic_add_r(RHO_AT(c2[0], c2[1], c2[2]), (T)(mass * dx[0] * dx[1] * dx[2]));
  } else {
    const double pos[3] = {(x - sp.min1) / d, (y - sp.min2) / d, (z - sp.min3) / d};
    unsigned c[3][3];
    double w[3][3];
#pragma unroll
    for (int a = 0; a < 3; a++) {
      const unsigned ci = (unsigned)floor(pos[a]) % (unsigned)n;
      c[a][1] = ci;
      c[a][2] = (ci + 1) % (unsigned)n;
      c[a][0] = (ci - 1 + (unsigned)n) % (unsigned)n;
      const double dd = pos[a] - ((double)ci + 0.5);
      w[a][1] = 0.75 - dd * dd;
      w[a][2] = 0.5 * (0.5 + dd) * (0.5 + dd);
      w[a][0] = 0.5 * (0.5 - dd) * (0.5 - dd);
    }
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
      for (int b = 0; b < 3; b++)
#pragma unroll
        for (int e = 0; e < 3; e++) atomic_add_r(RHO_AT(c[0][a], c[1][b], c[2][e]), (T)(1. * w[0][a] * w[1][b] * w[2][e]));
  }
#undef RHO_AT
}

// ------------------------------------------------------------------------------------------------------
// calc_h = 3: V from a Fourier-space convolution with the SPH kernel and TSC interpolation to the particles
// (likelihood_calc_V_SPH_fourier_TSC, HMC_models_testing.cpp:54-188; interpolate_TSC, interpolate_grid.cpp:134-202).
// k_conv_kernel: conv^_j = i h k_j W^(k) part_like^ / N for j = x, y, z (the 1/N folds the following C2R).
// W^(k) = norm (3 + cos 2k - k sin k + cos k (k sin k - 4)) / k^6 cancels catastrophically at small k (relative
// conditioning ~1e-16 / k^6): one ulp of difference in sin/cos moves it by 1e-9.  It only depends on |k|, so it is
// tabulated once per handle on the HOST with the C library (bchmc.hip: build_conv_table), which keeps the engine
// on the same values as a CPU build of the reference and keeps sin/cos out of the step loop.
// ------------------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256)
k_conv_kernel(Geo g, const C2<T> *__restrict__ pl, const double *__restrict__ F, C2<T> *__restrict__ Ck, double hh,
              double inv_n) {
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < g.Nhp;
       idx += (long long)gridDim.x * blockDim.x) {
    const int k = (int)(idx % g.nhp);
    const long long ij = idx / g.nhp;
    const int j = (int)(ij % g.n), i = (int)(ij / g.n);
    const double kx = kval(i, g.n, g.kfac), ky = kval(j, g.n, g.kfac), kz = kval(k, g.n, g.kfac);
    const double f = F[idx];
    const double2 v = ld2<T>(pl, idx);
    // re = h * k_j * -Im(pl) * F, im = h * k_j * Re(pl) * F  (HMC_models_testing.cpp:117-130), then / N
    st2<T>(Ck, idx, hh * kx * -v.y * f * inv_n, hh * kx * v.x * f * inv_n);
    st2<T>(Ck, idx + g.Nhp, hh * ky * -v.y * f * inv_n, hh * ky * v.x * f * inv_n);
    st2<T>(Ck, idx + 2 * g.Nhp, hh * kz * -v.y * f * inv_n, hh * kz * v.x * f * inv_n);
  }
}

// TSC interpolation of the three convolved fields to every particle.  Bug-for-bug with the reference:
// the upper weights of x and y are computed from dz (interpolate_grid.cpp:166-168).
template <typename T>
__global__ void __launch_bounds__(256)
k_interp_tsc(Geo g, PosPar pp, double f1, const T *__restrict__ psi, const T *__restrict__ conv, T *__restrict__ V) {
  const long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (p >= g.N) return;
  const int n = g.n;
  const int k = (int)(p % n);
  const long long ij = p / n;
  const int j = (int)(ij % n), i = (int)(ij / n);
  T xt, yt, zt;
  particle_pos<T>(pp, i, j, k, psi[p], psi[p + g.N], psi[p + 2 * g.N], xt, yt, zt);
  if (!pos_ok(g, xt, yt, zt)) {
    V[p] = T(0);
    V[p + g.N] = T(0);
    V[p + 2 * g.N] = T(0);
    return;
  }
  const double x = xt, y = yt, z = zt;
  const double xk = x / g.d, yk = y / g.d, zk = z / g.d;
  const unsigned cx = (unsigned)xk, cy = (unsigned)yk, cz = (unsigned)zk;
  const double dx = xk - ((double)cx + 0.5), dy = yk - ((double)cy + 0.5), dz = zk - ((double)cz + 0.5);
  double wx[3], wy[3], wz[3];
  wx[1] = 0.75 - dx * dx;
  wy[1] = 0.75 - dy * dy;
  wz[1] = 0.75 - dz * dz;
  wx[0] = 0.5 * ((1.5 - fabs(dx + 1)) * (1.5 - fabs(dx + 1)));
  wy[0] = 0.5 * ((1.5 - fabs(dy + 1)) * (1.5 - fabs(dy + 1)));
  wz[0] = 0.5 * ((1.5 - fabs(dz + 1)) * (1.5 - fabs(dz + 1)));
  wx[2] = wy[2] = wz[2] = 0.5 * ((1.5 - fabs(dz - 1)) * (1.5 - fabs(dz - 1)));
  const unsigned un = (unsigned)n;
  const unsigned ixx[3] = {(cx % un + un - 1) % un, cx % un, (cx + 1) % un};
  const unsigned ixy[3] = {(cy % un + un - 1) % un, cy % un, (cy + 1) % un};
  const unsigned ixz[3] = {(cz % un + un - 1) % un, cz % un, (cz + 1) % un};
  double o0 = 0., o1 = 0., o2 = 0.;
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b)
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const long long f = ((long long)ixx[a] * n + ixy[b]) * n + ixz[c];
        const double w = wx[a] * wy[b] * wz[c];
        o0 += w * (double)conv[f];
        o1 += w * (double)conv[f + g.N];
        o2 += w * (double)conv[f + 2 * g.N];
      }
  if (pp.rsd) o2 += f1 * o2;
  V[p] = (T)o0;
  V[p + g.N] = (T)o1;
  V[p + 2 * g.N] = (T)o2;
}

// ------------------------------------------------------------------------------------------------------
// calc_h = 0 (likelihood_calc_h, HMC_models_testing.cpp:25-50; labelled WRONG upstream but selectable):
// V_j = part_like * d f(delta_x)/dx_j with the gradient taken spectrally for the Gaussian likelihood (gradfft,
// gradient.cpp:22-78) and by 4th-order central differences otherwise (gradfindif, gradient.cpp:81-154).
// ------------------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256)
k_gradfft_mult(Geo g, const C2<T> *__restrict__ fk, C2<T> *__restrict__ Ck, double inv_n) {
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < g.Nhp;
       idx += (long long)gridDim.x * blockDim.x) {
    const int k = (int)(idx % g.nhp);
    const long long ij = idx / g.nhp;
    const int j = (int)(ij % g.n), i = (int)(ij / g.n);
    const bool nyq = (i == g.n / 2) || (j == g.n / 2) || (k == g.n / 2);
    const double2 v = ld2<T>(fk, idx);
    const double kk[3] = {kval(i, g.n, g.kfac), kval(j, g.n, g.kfac), kval(k, g.n, g.kfac)};
#pragma unroll
    for (int c = 0; c < 3; c++) {
      if (nyq)
        st2<T>(Ck, idx + c * g.Nhp, 0., 0.);
      else
        st2<T>(Ck, idx + c * g.Nhp, -kk[c] * v.y * inv_n, kk[c] * v.x * inv_n);
    }
  }
}

template <typename T>
__global__ void k_mul3(long long n, const T *__restrict__ a, const T *__restrict__ b3, T *__restrict__ out3) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const T v = a[i];
    out3[i] = v * b3[i];
    out3[i + n] = v * b3[i + n];
    out3[i + 2 * n] = v * b3[i + 2 * n];
  }
}

template <typename T>
__global__ void __launch_bounds__(256)
k_findif_mul(Geo g, LikePar lp, const T *__restrict__ dX, const T *__restrict__ plike, T *__restrict__ V) {
  const int n = g.n;
  const double fac = n / (2. * g.L);
  for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < g.N; p += (long long)gridDim.x * blockDim.x) {
    const int c[3] = {(int)(p / ((long long)n * n)), (int)((p / n) % n), (int)(p % n)};
    const long long stride[3] = {(long long)n * n, n, 1};
    const double pl = plike[p];
#pragma unroll
    for (int a = 0; a < 3; a++) {
      double f[4];
      const int off[4] = {-1, 1, -2, 2};  // l, r, ll, rr
#pragma unroll
      for (int m = 0; m < 4; m++) {
        const int ca = (c[a] + off[m] + n) % n;
        double v = dX[p + (ca - c[a]) * stride[a]];
        if (lp.likelihood == 2) {  // lognormal_likelihood_f_delta_x_i_calc, lognormal_independent.cpp:57-64
          if (v < lp.delta_min) v = lp.delta_min;
          v = log(lp.rho_c * (1. + v));
        }
        f[m] = v;
      }
      V[p + a * g.N] = (T)(pl * -(fac * ((4.0 / 3) * (f[0] - f[1]) - (1.0 / 6) * (f[2] - f[3]))));
    }
  }
}

// ------------------------------------------------------------------------------------------------------
// Device momentum draw (statistical stand-in for draw_momenta, HMC_momenta.cc:42-94, which
// consumes the host GSL stream serially).  Counter-based Philox4x32-10 (Salmon et al. 2011, Random123 constants):
// value i of attempt a is a pure function of (seed, a, stream, i), so the draw is reproducible and order-free.
// p = IFFT[ white^ / sqrt(wM) ] has covariance M for mass_f (K = 1/2 p^T M^-1 p averages N/2); the real-space
// part adds sqrt(mass_r) * white (HMC_momenta.cc:76-94).
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint4 philox4x32_10(uint4 c, uint2 k) {
  constexpr unsigned M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int r = 0; r < 10; r++) {
    const unsigned hi0 = __umulhi(M0, c.x), lo0 = M0 * c.x;
    const unsigned hi1 = __umulhi(M1, c.z), lo1 = M1 * c.z;
    c = make_uint4(hi1 ^ c.y ^ k.x, lo1, hi0 ^ c.w ^ k.y, lo0);
    k.x += W0;
    k.y += W1;
  }
  return c;
}

__global__ void k_philox_kat(uint4 ctr, uint2 key, uint4 *out) { *out = philox4x32_10(ctr, key); }

// out[2i], out[2i+1] = two independent N(0,1) (Box-Muller on two 53-bit uniforms), optionally times sqrt(var[.]).
template <typename T>
__global__ void __launch_bounds__(256)
k_white_noise(long long n, uint2 key, unsigned attempt, unsigned stream, const T *__restrict__ var, T *__restrict__ out) {
  const long long pairs = (n + 1) / 2;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < pairs; i += (long long)gridDim.x * blockDim.x) {
    const uint4 r = philox4x32_10(make_uint4((unsigned)i, (unsigned)(i >> 32), attempt, stream), key);
    const double u1 = ((double)((((unsigned long long)r.x << 32) | r.y) >> 11) + 0.5) * (1.0 / 9007199254740992.0);
    const double u2 = ((double)((((unsigned long long)r.z << 32) | r.w) >> 11) + 0.5) * (1.0 / 9007199254740992.0);
    const double rad = sqrt(-2.0 * log(u1));
    double sn, cs;
    sincos(6.283185307179586476925 * u2, &sn, &cs);
    double g0 = rad * cs, g1 = rad * sn;
    if (var) {
      const double v0 = var[2 * i];
      g0 *= v0 > 0. ? sqrt(v0) : 0.;
      if (2 * i + 1 < n) {
        const double v1 = var[2 * i + 1];
        g1 *= v1 > 0. ? sqrt(v1) : 0.;
      }
    }
    out[2 * i] = (T)g0;
    if (2 * i + 1 < n) out[2 * i + 1] = (T)g1;
  }
}

// pk = [pk +] wk / sqrt(wM)   (0 where wM <= 0, i.e. where the mass is not positive)
template <typename T>
__global__ void k_color_momenta(long long nh, const C2<T> *__restrict__ wk, const double *__restrict__ wM,
                                C2<T> *__restrict__ pk, int accumulate) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < nh; i += (long long)gridDim.x * blockDim.x) {
    double2 v = ld2<T>(wk, i);
    if (wM) {
      const double w = wM[i];
      const double a = w > 0. ? 1. / sqrt(w) : 0.;
      v.x *= a;
      v.y *= a;
    }
    if (accumulate) {
      const double2 o = ld2<T>(pk, i);
      v.x += o.x;
      v.y += o.y;
    }
    st2<T>(pk, i, v.x, v.y);
  }
}

// ======================================================================================================
// Tile-sorted particle-mesh path (the fast path for masskernel 3 when the tile shape divides the grid).
//
// Zel'dovich displacements at the BASELINE resolution are many cells long (rms 3-10 cells at 256^3 in a
// 200 Mpc/h box), so a Lagrangian brick of particles does NOT stay inside an LDS-sized Eulerian tile.  We
// therefore bin the particles by the Eulerian tile of their home cell every force evaluation (counting
// sort: block-aggregated atomics, one scan, one reorder pass), and then
//   * scatter: one workgroup per (tile, chunk of <= `chunk` particles) accumulates W into an LDS copy of the
//     tile plus a halo of `R` cells with LDS float atomics and flushes its non-zero cells to HBM once
//     (a few coalesced global atomics per cell instead of ~34 scattered ones per particle);
//   * gather: the same work items stage part_like (tile + halo) in LDS and each particle reads its 81
//     stencil cells from there.
// The (particle, cell) pair set is identical to the direct kernels above, which stay as the fallback; the
// spline evaluations use fast_rsqrt (<= 2 ulp) instead of IEEE sqrt + divide.
// ======================================================================================================
struct TilePar {
  int tx, ty, tz;     // tile shape in cells (z fastest)
  int ntx, nty, ntz;  // tiles per axis
  int ntiles;
  int R;              // halo = farthest stencil offset
  int lx, ly, lz;     // LDS tile shape = t + 2R
  int chunk;          // max particles per work item
  int cap;            // record slots reserved per tile by the one-pass binning (k_bin<DIRECT>)
};

constexpr int kSortFlagNoScatter = 1 << 30;  // record flag: particle fails getDensity_SPH's domain test

__device__ __forceinline__ int tile_of(const TilePar &tp, int n, long long ix, long long iy, long long iz) {
  const int cx = (int)(ix % n), cy = (int)(iy % n), cz = (int)(iz % n);
  return (cz / tp.tz) + tp.ntz * ((cy / tp.ty) + tp.nty * (cx / tp.tx));
}

// Home cell of a position: (ULONG)(xp/d1), massFunctions.cc:434-436.
template <typename T>
__device__ __forceinline__ long long home_cell(T x, T d) {
  return (long long)(x / d);
}

// The same cell for the sorted path (positions there are in [0, L], so the result is in [0, n]) without the IEEE
// divide: x * (1/d) is within a few ulp of x / d, so truncating it gives the reference's cell unless the quotient
// is that close to an integer; only then is the division itself evaluated.  Deterministic in (x, d): the binning
// pass and the scatter/gather passes always agree.
template <typename T> struct HomeCell {
  T d, inv_d, thr;
  int n;
};
template <typename T>
__device__ __forceinline__ HomeCell<T> make_home(const Geo &g) {
  HomeCell<T> hc;
  hc.d = (T)g.d;
  hc.inv_d = T(1) / hc.d;
  hc.thr = (T)g.n * (sizeof(T) == 8 ? T(1e-15) : T(5e-7));  // >= 4 ulp of the largest quotient
  hc.n = g.n;
  return hc;
}
template <typename T>
__device__ __forceinline__ int home_cell_i(const HomeCell<T> &hc, T x) {
  const T f = x * hc.inv_d;
  if (__builtin_expect(fabs(f - rint(f)) < hc.thr, 0)) return (int)(x / hc.d);
  return (int)f;
}
__device__ __forceinline__ int wrap_cell(int c, int n) { return c >= n ? c - n : c; }
__device__ __forceinline__ int tile_of_wrapped(const TilePar &tp, int cx, int cy, int cz) {
  return (cz / tp.tz) + tp.ntz * ((cy / tp.ty) + tp.nty * (cx / tp.tx));
}

// Workgroup -> particles.  When 16 divides n a workgroup takes a 4 x 4 x 16 brick of the Lagrangian lattice
// (neighbours in space: few distinct tiles per workgroup, 128-byte rows of psi); otherwise 256 consecutive ones.
__device__ __forceinline__ long long brick_particle(const Geo &g, int b, int &i, int &j, int &k) {
  const int n = g.n, tid = threadIdx.x;
  if ((n & 15) == 0) {
    const int nbz = n >> 4, nby = n >> 2;
    const int bk = b % nbz, bj = (b / nbz) % nby, bi = b / (nbz * nby);
    i = bi * 4 + (tid >> 6);
    j = bj * 4 + ((tid >> 4) & 3);
    k = bk * 16 + (tid & 15);
    return k + (long long)n * (j + (long long)n * i);
  }
  const long long p = b * (long long)blockDim.x + tid;
  k = (int)(p % n);
  const long long ij = p / n;
  j = (int)(ij % n);
  i = (int)(ij / n);
  return p;
}

// Binning.  The workgroup first counts its particles per tile in an LDS hash table, then reserves one contiguous
// rank range per distinct tile with a single global atomic (a handful per workgroup instead of one returning
// atomic per particle on ~n^3/2048 hot counters).  Particles with a non-finite position are left out; the gather
// gives them V = 0.
//   DIRECT = true  (one-pass sort): every tile owns `tp.cap` record slots, the particle's record (position, index |
//                  flag) goes straight to slot tile * cap + rank.  A rank >= cap raises *ovf and the record is dropped:
//                  the two-pass kernels below then redo the sort from scratch (they return at once otherwise).
//   DIRECT = false (two-pass fallback, pass 1): tile id and arrival rank of every particle to tile_rank.
template <typename T, bool DIRECT>
__global__ void __launch_bounds__(256)
k_bin(Geo g, PosPar pp, SphPar sp, TilePar tp, int nbricks, const T *__restrict__ psi, int *__restrict__ cnt,
      int *__restrict__ ovf,
      int2 *__restrict__ tile_rank, T *__restrict__ sx, T *__restrict__ sy, T *__restrict__ sz, int *__restrict__ sidx,
      T *__restrict__ V) {
  constexpr int kSlots = 512;
  __shared__ int hkey[kSlots], hcnt[kSlots], hbase[kSlots];
  if (!DIRECT && !*ovf) return;
  // DIRECT: one brick per workgroup; fallback: a small grid strides over the bricks (it usually returns above)
  for (int brick = blockIdx.x; brick < nbricks; brick += gridDim.x) {
    for (int s = threadIdx.x; s < kSlots; s += blockDim.x) {
      hkey[s] = 0;
      hcnt[s] = 0;
    }
    __syncthreads();
    int i, j, k;
    const long long p = brick_particle(g, brick, i, j, k);
    const bool live = p < g.N;
    int t = -1, slot = 0, local = 0, flag = 0;
    T x = T(0), y = T(0), z = T(0);
    if (live) {
      particle_pos<T>(pp, i, j, k, psi[p], psi[p + g.N], psi[p + 2 * g.N], x, y, z);
      if (pos_ok(g, x, y, z)) {
        const HomeCell<T> hc = make_home<T>(g);
        t = tile_of_wrapped(tp, wrap_cell(home_cell_i(hc, x), g.n), wrap_cell(home_cell_i(hc, y), g.n),
                            wrap_cell(home_cell_i(hc, z), g.n));
        flag = in_domain(g, sp, x, y, z) ? 0 : kSortFlagNoScatter;
        slot = (int)(((unsigned)t * 2654435761u) >> 23) & (kSlots - 1);
        for (;;) {
          const int old = atomicCAS(&hkey[slot], 0, t + 1);
          if (old == 0 || old == t + 1) break;
          slot = (slot + 1) & (kSlots - 1);
        }
        local = atomicAdd(&hcnt[slot], 1);
      } else {
        V[p] = T(0);
        V[p + g.N] = T(0);
        V[p + 2 * g.N] = T(0);
      }
    }
    __syncthreads();
    for (int s = threadIdx.x; s < kSlots; s += blockDim.x)
      if (hkey[s]) hbase[s] = atomicAdd(&cnt[hkey[s] - 1], hcnt[s]);
    __syncthreads();
    if (live && DIRECT) {
      if (t >= 0) {
        const int rank = hbase[slot] + local;
        if (rank >= tp.cap) {
          ovf[0] = 1;  // benign race: every writer stores 1
          ovf[1] = 1;  // sticky copy: the host enlarges the slots before the next trajectory
        } else {
          const long long dst = (long long)t * tp.cap + rank;
          sx[dst] = x;
          sy[dst] = y;
          sz[dst] = z;
          sidx[dst] = (int)p | flag;
        }
      }
    } else if (live) {
      tile_rank[p] = (t < 0) ? make_int2(-1, 0) : make_int2(t, (hbase[slot] + local) | flag);
    }
    __syncthreads();  // the hash table is reused by the next brick
  }
}

// One workgroup: record range [off, tend) of every tile -- fixed slots after a successful one-pass binning, an
// exclusive scan of the fallback's counts otherwise -- and the exclusive scan of the per-tile chunk counts
// (-> work-item offsets, ntiles + 1 entries).
__device__ __forceinline__ int block_exclusive_scan_1024(int v, int *buf) {
  const int tid = threadIdx.x;
  buf[tid] = v;
  __syncthreads();
  for (int s = 1; s < 1024; s <<= 1) {
    const int u = tid >= s ? buf[tid - s] : 0;
    __syncthreads();
    buf[tid] += u;
    __syncthreads();
  }
  const int incl = buf[tid];
  __syncthreads();
  return incl - v;
}

__global__ void __launch_bounds__(1024)
k_scan_tiles(TilePar tp, const int *__restrict__ cnt_direct, const int *__restrict__ cnt_fallback,
             const int *__restrict__ ovf, int *__restrict__ off, int *__restrict__ tend, int *__restrict__ woff) {
  __shared__ int buf[1024];
  const bool direct = !*ovf;
  const int *cnt = direct ? cnt_direct : cnt_fallback;
  const int T = tp.ntiles, tid = threadIdx.x;
  const int per = (T + 1023) / 1024;
  const int lo = min(tid * per, T), hi = min(lo + per, T);
  int a = 0, b = 0;
  for (int t = lo; t < hi; t++) {
    a += cnt[t];
    b += (cnt[t] + tp.chunk - 1) / tp.chunk;
  }
  int ea = block_exclusive_scan_1024(a, buf);
  int eb = block_exclusive_scan_1024(b, buf);
  for (int t = lo; t < hi; t++) {
    const int o = direct ? t * tp.cap : ea;  // one-pass layout: fixed slots per tile; fallback: packed
    off[t] = o;
    tend[t] = o + cnt[t];
    woff[t] = eb;
    ea += cnt[t];
    eb += (cnt[t] + tp.chunk - 1) / tp.chunk;
  }
  if (tid == 1023) woff[T] = eb;
}

// Fallback pass 3: write each particle's record (position, original index | flag) to its sorted slot.
template <typename T>
__global__ void __launch_bounds__(256)
k_reorder(Geo g, PosPar pp, int nbricks, const T *__restrict__ psi, const int2 *__restrict__ tile_rank,
          const int *__restrict__ off, const int *__restrict__ ovf, T *__restrict__ sx, T *__restrict__ sy,
          T *__restrict__ sz, int *__restrict__ sidx) {
  if (!*ovf) return;  // the one-pass binning succeeded
  for (int brick = blockIdx.x; brick < nbricks; brick += gridDim.x) {
    int i, j, k;
    const long long p = brick_particle(g, brick, i, j, k);
    if (p >= g.N) continue;
    const int2 tr = tile_rank[p];
    if (tr.x < 0) continue;
    T x, y, z;
    particle_pos<T>(pp, i, j, k, psi[p], psi[p + g.N], psi[p + 2 * g.N], x, y, z);
    const int slot = off[tr.x] + (tr.y & ~kSortFlagNoScatter);
    sx[slot] = x;
    sy[slot] = y;
    sz[slot] = z;
    sidx[slot] = (int)p | (tr.y & kSortFlagNoScatter);
  }
}

// Work item -> (tile, particle range).  Returns false when this workgroup has nothing to do.
__device__ __forceinline__ bool tile_work(const TilePar &tp, const int *__restrict__ off, const int *__restrict__ tend,
                                          const int *__restrict__ woff, int &tile, int &p_begin, int &p_end) {
  __shared__ int s_tile, s_b, s_e;
  if (threadIdx.x == 0) {
    const int w = blockIdx.x;
    int t = -1, b = 0, e = 0;
    if (w < woff[tp.ntiles]) {
      int lo = 0, hi = tp.ntiles;  // last t with woff[t] <= w
      while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (woff[mid] <= w) lo = mid; else hi = mid;
      }
      t = lo;
      b = off[t] + (w - woff[t]) * tp.chunk;
      e = min(b + tp.chunk, tend[t]);
    }
    s_tile = t;
    s_b = b;
    s_e = e;
  }
  __syncthreads();
  tile = s_tile;
  p_begin = s_b;
  p_end = s_e;
  return tile >= 0;
}

// Order one work item's records by the sub-cell position of the particle (in place; the gather reads the same
// order): `bits` binary digits of the fractional cell coordinate per axis, octant digits most significant.  The 64
// lanes of a wave then share most of the stencil cells that can pass the `r/h <= 2` test, and a wave pays for
// every candidate ANY of its lanes needs (81 unsorted, ~51 with octants, fewer with 4 x 4 x 4 bins).  Pure
// reordering: results do not depend on it.  Requires chunk == 256 * 8 and blockDim.x == 256.
template <typename T>
__device__ __forceinline__ void subsort_subcell(int bits, int pb, int pe, T inv_d, T *sx, T *sy, T *sz, int *sidx) {
  constexpr int kPer = 8;  // tp.chunk == 256 * kPer
  __shared__ int hist[64], base[64];
  const int nb = 1 << (3 * bits);
  if (threadIdx.x < 64) hist[threadIdx.x] = 0;
  __syncthreads();
  T rx[kPer], ry[kPer], rz[kPer];
  int id[kPer], key[kPer], rank[kPer];
  const T scale = (T)(1 << bits);
#pragma unroll
  for (int m = 0; m < kPer; m++) {
    const int s = pb + (int)threadIdx.x + 256 * m;
    if (s < pe) {
      rx[m] = sx[s];
      ry[m] = sy[s];
      rz[m] = sz[s];
      id[m] = sidx[s];
      const T fx = rx[m] * inv_d, fy = ry[m] * inv_d, fz = rz[m] * inv_d;  // ordering only
      const int ux = min((int)((fx - r_floor(fx)) * scale), (1 << bits) - 1);
      const int uy = min((int)((fy - r_floor(fy)) * scale), (1 << bits) - 1);
      const int uz = min((int)((fz - r_floor(fz)) * scale), (1 << bits) - 1);
      int kk = 0;
      for (int b = bits - 1; b >= 0; b--) kk = (kk << 3) | (((ux >> b) & 1) << 2) | (((uy >> b) & 1) << 1) | ((uz >> b) & 1);
      key[m] = kk;
      rank[m] = atomicAdd(&hist[kk], 1);
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int acc = 0;
    for (int b = 0; b < nb; b++) {
      base[b] = acc;
      acc += hist[b];
    }
  }
  __syncthreads();
#pragma unroll
  for (int m = 0; m < kPer; m++) {
    const int s = pb + (int)threadIdx.x + 256 * m;
    if (s < pe) {
      const int dst = pb + base[key[m]] + rank[m];
      sx[dst] = rx[m];
      sy[dst] = ry[m];
      sz[dst] = rz[m];
      sidx[dst] = id[m];
    }
  }
  __threadfence_block();
}

// getDensity_SPH on sorted particles: LDS accumulation per (tile, chunk), one flush.
template <typename T>
__global__ void __launch_bounds__(256)
k_scatter_tile(Geo g, SphPar sp, TilePar tp, const int4 *__restrict__ cols, int ncol, int reorder, T *sx, T *sy, T *sz,
               int *sidx, const int *__restrict__ off, const int *__restrict__ tend, const int *__restrict__ woff,
               T *__restrict__ rho) {
  extern __shared__ __attribute__((aligned(16))) unsigned char s_raw_scatter[];
  double *s_tile_acc = reinterpret_cast<double *>(s_raw_scatter);  // accumulate in double also for float fields
  int tile, pb, pe;
  if (!tile_work(tp, off, tend, woff, tile, pb, pe)) return;
  const int ncell = tp.lx * tp.ly * tp.lz;
  int4 *s_cols = reinterpret_cast<int4 *>(s_raw_scatter + (((size_t)ncell * sizeof(double) + 15) & ~(size_t)15));
  for (int m = threadIdx.x; m < ncol; m += blockDim.x) s_cols[m] = cols[m];
  for (int c = threadIdx.x; c < ncell; c += blockDim.x) s_tile_acc[c] = 0.;
  const T d = (T)g.d;
  const HomeCell<T> hc = make_home<T>(g);
  if (reorder) subsort_subcell<T>(reorder, pb, pe, hc.inv_d, sx, sy, sz, sidx);
  __syncthreads();
  const int tzi = tile % tp.ntz, tyi = (tile / tp.ntz) % tp.nty, txi = tile / (tp.ntz * tp.nty);
  const int ox = txi * tp.tx - tp.R, oy = tyi * tp.ty - tp.R, oz = tzi * tp.tz - tp.R;  // global cell of LDS (0,0,0)
  const int n = g.n, R = sp.reach;
  const T r2_lim = (T)sp.r2_lim, h_inv = (T)sp.h_inv, w_norm = (T)sp.w_norm;
  for (int s = pb + threadIdx.x; s < pe; s += blockDim.x) {
    if (sidx[s] & kSortFlagNoScatter) continue;
    const T x = sx[s], y = sy[s], z = sz[s];
    const int ix = home_cell_i(hc, x), iy = home_cell_i(hc, y), iz = home_cell_i(hc, z);
    const T ccx = ((T)ix + T(0.5)) * d, ccy = ((T)iy + T(0.5)) * d, ccz = ((T)iz + T(0.5)) * d;
    const int hx = wrap_cell(ix, n) - ox, hy = wrap_cell(iy, n) - oy, hz = wrap_cell(iz, n) - oz;  // home cell in LDS coords
    if ((unsigned)(hx - tp.R) >= (unsigned)tp.tx || (unsigned)(hy - tp.R) >= (unsigned)tp.ty ||
        (unsigned)(hz - tp.R) >= (unsigned)tp.tz)
      continue;  // cannot happen (binning and this kernel see the same stored position); keeps LDS indexing safe
    if (ncol > 0) {
      // Exact hull (host-verified: no cell outside it can satisfy r/h <= 2): 81 candidates instead of 343.
      for (int m = 0; m < ncol; ++m) {
        const int4 c = s_cols[m];
        const T dx = x - (ccx + (T)c.x * d);
        const T dy = y - (ccy + (T)c.y * d);
        const T r2ab = dx * dx + dy * dy;
        if (r2ab > r2_lim) continue;
        double *row = s_tile_acc + tp.lz * ((hy + c.y) + tp.ly * (hx + c.x)) + hz;
        for (int i3 = c.z; i3 <= c.w; ++i3) {
          const T dz = z - (ccz + (T)i3 * d);
          const T r2 = r2ab + dz * dz;
          if (r2 <= r2_lim) {
            const T q = (r2 * fast_rsqrt(r2 + tiny_pos<T>())) * h_inv;
            if (q <= T(2)) atomic_add_r(row + i3, (double)sph_w_folded<T>(q, w_norm));
          }
        }
      }
    } else {
      for (int i1 = -R; i1 <= R; ++i1) {
        const T dx = x - (ccx + (T)i1 * d);
        const T dx2 = dx * dx;
        if (dx2 > r2_lim) continue;
        for (int i2 = -R; i2 <= R; ++i2) {
          const T dy = y - (ccy + (T)i2 * d);
          const T r2ab = dx2 + dy * dy;
          if (r2ab > r2_lim) continue;
          double *row = s_tile_acc + tp.lz * ((hy + i2) + tp.ly * (hx + i1)) + hz;
          for (int i3 = -R; i3 <= R; ++i3) {
            const T dz = z - (ccz + (T)i3 * d);
            const T r2 = r2ab + dz * dz;
            if (r2 > r2_lim) continue;
            const T q = (r2 * fast_rsqrt(r2 + tiny_pos<T>())) * h_inv;
            if (q <= T(2)) atomic_add_r(row + i3, (double)sph_w_folded<T>(q, w_norm));
          }
        }
      }
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < ncell; c += blockDim.x) {
    const double v = s_tile_acc[c];
    if (v != 0.) {
      const int cz = c % tp.lz, cy = (c / tp.lz) % tp.ly, cx = c / (tp.lz * tp.ly);
      const int gx = (ox + cx + n) % n, gy = (oy + cy + n) % n, gz = (oz + cz + n) % n;
      atomic_add_r(rho + gz + (long long)n * (gy + (long long)n * gx), (T)v);
    }
  }
}

// likelihood_calc_V_SPH on sorted particles: part_like tile + halo staged in LDS.
template <typename T>
__global__ void __launch_bounds__(256)
k_gather_tile(Geo g, HullPar hp, TilePar tp, int rsd, const T *__restrict__ sx, const T *__restrict__ sy,
              const T *__restrict__ sz, const int *__restrict__ sidx, const int *__restrict__ off,
              const int *__restrict__ tend, const int *__restrict__ woff, const T *__restrict__ plike,
              T *__restrict__ V) {
  extern __shared__ __attribute__((aligned(16))) unsigned char s_raw_gather[];
  T *s_tile_pl = reinterpret_cast<T *>(s_raw_gather);
  int tile, pb, pe;
  if (!tile_work(tp, off, tend, woff, tile, pb, pe)) return;
  const int ncell = tp.lx * tp.ly * tp.lz;
  int4 *s_cols = reinterpret_cast<int4 *>(s_raw_gather + (((size_t)ncell * sizeof(T) + 15) & ~(size_t)15));
  for (int m = threadIdx.x; m < hp.ncol; m += blockDim.x) s_cols[m] = hp.cols[m];
  const int n = g.n;
  const int tzi = tile % tp.ntz, tyi = (tile / tp.ntz) % tp.nty, txi = tile / (tp.ntz * tp.nty);
  const int ox = txi * tp.tx - tp.R, oy = tyi * tp.ty - tp.R, oz = tzi * tp.tz - tp.R;
  for (int c = threadIdx.x; c < ncell; c += blockDim.x) {
    const int cz = c % tp.lz, cy = (c / tp.lz) % tp.ly, cx = c / (tp.lz * tp.ly);
    const int gx = (ox + cx + n) % n, gy = (oy + cy + n) % n, gz = (oz + cz + n) % n;
    s_tile_pl[c] = plike[gz + (long long)n * (gy + (long long)n * gx)];
  }
  __syncthreads();
  const T d_h = (T)hp.d_h, h_inv = (T)hp.h_inv, norm = (T)hp.norm;
  const HomeCell<T> hc = make_home<T>(g);
  for (int s = pb + threadIdx.x; s < pe; s += blockDim.x) {
    const T px = sx[s], py = sy[s], pz = sz[s];
    const int ix = home_cell_i(hc, px), iy = home_cell_i(hc, py), iz = home_cell_i(hc, pz);
    const T dpcx = px * h_inv - ((T)ix + T(0.5)) * d_h;
    const T dpcy = py * h_inv - ((T)iy + T(0.5)) * d_h;
    const T dpcz = pz * h_inv - ((T)iz + T(0.5)) * d_h;
    const int hx = wrap_cell(ix, n) - ox, hy = wrap_cell(iy, n) - oy, hz = wrap_cell(iz, n) - oz;
    T vx = T(0), vy = T(0), vz = T(0);
    const bool home_ok = (unsigned)(hx - tp.R) < (unsigned)tp.tx && (unsigned)(hy - tp.R) < (unsigned)tp.ty &&
                         (unsigned)(hz - tp.R) < (unsigned)tp.tz;  // always true; keeps LDS indexing safe
    for (int m = 0; home_ok && m < hp.ncol; ++m) {
      const int4 c = s_cols[m];
      const T xh = dpcx - (T)c.x * d_h;
      const T yh = dpcy - (T)c.y * d_h;
      const T r2ab = xh * xh + yh * yh;
      if (r2ab > T(4)) continue;
      const T *row = s_tile_pl + tp.lz * ((hy + c.y) + tp.ly * (hx + c.x)) + hz;
      T zh = dpcz - (T)c.z * d_h;
      for (int i3 = c.z; i3 <= c.w; ++i3) {
        const T q_sq = r2ab + zh * zh;
        if (q_sq <= T(4)) {
          const T common = row[i3] * sph_grad_folded<T>(q_sq, norm);
          vx += common * xh;
          vy += common * yh;
          vz += common * zh;
        }
        zh -= d_h;
      }
    }
    const T normalize = (T)hp.normalize;
    vx *= normalize;
    vy *= normalize;
    vz *= normalize;
    if (rsd) vz += (T)hp.f1 * vz;
    const long long p = sidx[s] & ~kSortFlagNoScatter;
    V[p] = vx;
    V[p + g.N] = vy;
    V[p + 2 * g.N] = vz;
  }
}

// ------------------------------------------------------------------------------------------------------
// Specialisations for the standard stencil (h = d: the 81-cell hull of SPH_kernel_3D_cells_hull_1,
// SPH_kernel.cpp:110-139) on 8 x 8 x 16 tiles with a 2-cell halo.  The hull and the LDS tile shape are compile-time
// constants, so the column/cell loops unroll completely: the squared axis offsets are computed once per particle
// (r^2 = X[a] + Y[b] + Z[c], one add per candidate instead of convert + fma + subtract + fma), every LDS access
// has an immediate offset, and a rejected candidate costs add + compare + branch.  Same (particle, cell) pairs
// and the same kernel evaluations as the generic kernels above; r^2 differs from theirs by rounding only.
// ------------------------------------------------------------------------------------------------------
// z half-width of hull column (a - 2, b - 2): -1 = not in the hull
__host__ __device__ constexpr int hull81_zw(int a, int b) {
  const int i1 = a < 2 ? 2 - a : a - 2, i2 = b < 2 ? 2 - b : b - 2;
  return (i1 == 2 && i2 == 2) ? -1 : ((i1 == 2 || i2 == 2) ? 1 : 2);
}

template <typename T, int LY, int LZ>
__global__ void __launch_bounds__(256)
k_scatter_tile81(Geo g, SphPar sp, TilePar tp, int reorder, T *sx, T *sy, T *sz, int *sidx,
                 const int *__restrict__ off, const int *__restrict__ tend, const int *__restrict__ woff,
                 T *__restrict__ rho) {
  extern __shared__ __attribute__((aligned(16))) unsigned char s_raw_scatter81[];
  double *s_tile_acc = reinterpret_cast<double *>(s_raw_scatter81);
  int tile, pb, pe;
  if (!tile_work(tp, off, tend, woff, tile, pb, pe)) return;
  const int ncell = tp.lx * LY * LZ;
  for (int c = threadIdx.x; c < ncell; c += blockDim.x) s_tile_acc[c] = 0.;
  const T d = (T)g.d;
  const HomeCell<T> hc = make_home<T>(g);
  if (reorder) subsort_subcell<T>(reorder, pb, pe, hc.inv_d, sx, sy, sz, sidx);
  __syncthreads();
  const int tzi = tile % tp.ntz, tyi = (tile / tp.ntz) % tp.nty, txi = tile / (tp.ntz * tp.nty);
  const int ox = txi * tp.tx - 2, oy = tyi * tp.ty - 2, oz = tzi * tp.tz - 2;  // global cell of LDS (0,0,0)
  const int n = g.n;
  const T r2_lim = (T)sp.r2_lim, h_inv = (T)sp.h_inv, w_norm = (T)sp.w_norm;
  for (int s = pb + threadIdx.x; s < pe; s += blockDim.x) {
    if (sidx[s] & kSortFlagNoScatter) continue;
    const T x = sx[s], y = sy[s], z = sz[s];
    const int ix = home_cell_i(hc, x), iy = home_cell_i(hc, y), iz = home_cell_i(hc, z);
    const T ccx = ((T)ix + T(0.5)) * d, ccy = ((T)iy + T(0.5)) * d, ccz = ((T)iz + T(0.5)) * d;
    const int hx = wrap_cell(ix, n) - ox, hy = wrap_cell(iy, n) - oy, hz = wrap_cell(iz, n) - oz;  // home cell in LDS coords
    if ((unsigned)(hx - 2) >= (unsigned)tp.tx || (unsigned)(hy - 2) >= (unsigned)tp.ty ||
        (unsigned)(hz - 2) >= (unsigned)tp.tz)
      continue;  // cannot happen (binning and this kernel see the same stored position); keeps LDS indexing safe
    T X[5], Y[5], Z[5];
#pragma unroll
    for (int a = 0; a < 5; a++) {
      const T dx = x - (ccx + (T)(a - 2) * d), dy = y - (ccy + (T)(a - 2) * d), dz = z - (ccz + (T)(a - 2) * d);
      X[a] = dx * dx;
      Y[a] = dy * dy;
      Z[a] = dz * dz;
    }
    double *corner = s_tile_acc + LZ * ((hy - 2) + LY * (hx - 2)) + (hz - 2);
#pragma unroll
    for (int a = 0; a < 5; a++) {
#pragma unroll
      for (int b = 0; b < 5; b++) {
        const int zw = hull81_zw(a, b);  // folds after unrolling
        if (zw < 0) continue;
        const T r2ab = X[a] + Y[b];
        if (r2ab > r2_lim) continue;
        double *row = corner + LZ * (b + LY * a);
#pragma unroll
        for (int c = 0; c < 5; c++) {
          if (c < 2 - zw || c > 2 + zw) continue;
          const T r2 = r2ab + Z[c];
          if (r2 <= r2_lim) {
            const T q = (r2 * fast_rsqrt(r2 + tiny_pos<T>())) * h_inv;
            if (q <= T(2)) atomic_add_r(row + c, (double)sph_w_folded<T>(q, w_norm));
          }
        }
      }
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < ncell; c += blockDim.x) {
    const double v = s_tile_acc[c];
    if (v != 0.) {
      const int cz = c % LZ, cy = (c / LZ) % LY, cx = c / (LZ * LY);
      const int gx = (ox + cx + n) % n, gy = (oy + cy + n) % n, gz = (oz + cz + n) % n;
      atomic_add_r(rho + gz + (long long)n * (gy + (long long)n * gx), (T)v);
    }
  }
}

template <typename T, int LY, int LZ>
__global__ void __launch_bounds__(256)
k_gather_tile81(Geo g, HullPar hp, TilePar tp, int rsd, const T *__restrict__ sx, const T *__restrict__ sy,
                const T *__restrict__ sz, const int *__restrict__ sidx, const int *__restrict__ off,
                const int *__restrict__ tend, const int *__restrict__ woff, const T *__restrict__ plike,
                T *__restrict__ V) {
  extern __shared__ __attribute__((aligned(16))) unsigned char s_raw_gather81[];
  T *s_tile_pl = reinterpret_cast<T *>(s_raw_gather81);
  int tile, pb, pe;
  if (!tile_work(tp, off, tend, woff, tile, pb, pe)) return;
  const int ncell = tp.lx * LY * LZ;
  const int n = g.n;
  const int tzi = tile % tp.ntz, tyi = (tile / tp.ntz) % tp.nty, txi = tile / (tp.ntz * tp.nty);
  const int ox = txi * tp.tx - 2, oy = tyi * tp.ty - 2, oz = tzi * tp.tz - 2;
  for (int c = threadIdx.x; c < ncell; c += blockDim.x) {
    const int cz = c % LZ, cy = (c / LZ) % LY, cx = c / (LZ * LY);
    const int gx = (ox + cx + n) % n, gy = (oy + cy + n) % n, gz = (oz + cz + n) % n;
    s_tile_pl[c] = plike[gz + (long long)n * (gy + (long long)n * gx)];
  }
  __syncthreads();
  const T d_h = (T)hp.d_h, h_inv = (T)hp.h_inv, norm = (T)hp.norm;
  const HomeCell<T> hc = make_home<T>(g);
  for (int s = pb + threadIdx.x; s < pe; s += blockDim.x) {
    const T px = sx[s], py = sy[s], pz = sz[s];
    const int ix = home_cell_i(hc, px), iy = home_cell_i(hc, py), iz = home_cell_i(hc, pz);
    const T dpcx = px * h_inv - ((T)ix + T(0.5)) * d_h;
    const T dpcy = py * h_inv - ((T)iy + T(0.5)) * d_h;
    const T dpcz = pz * h_inv - ((T)iz + T(0.5)) * d_h;
    const int hx = wrap_cell(ix, n) - ox, hy = wrap_cell(iy, n) - oy, hz = wrap_cell(iz, n) - oz;
    T vx = T(0), vy = T(0), vz = T(0);
    const bool home_ok = (unsigned)(hx - 2) < (unsigned)tp.tx && (unsigned)(hy - 2) < (unsigned)tp.ty &&
                         (unsigned)(hz - 2) < (unsigned)tp.tz;  // always true; keeps LDS indexing safe
    if (home_ok) {
      T xh[5], yh[5], zh[5], X[5], Y[5], Z[5];
#pragma unroll
      for (int a = 0; a < 5; a++) {
        xh[a] = dpcx - (T)(a - 2) * d_h;
        yh[a] = dpcy - (T)(a - 2) * d_h;
        zh[a] = dpcz - (T)(a - 2) * d_h;
        X[a] = xh[a] * xh[a];
        Y[a] = yh[a] * yh[a];
        Z[a] = zh[a] * zh[a];
      }
      const T *corner = s_tile_pl + LZ * ((hy - 2) + LY * (hx - 2)) + (hz - 2);
#pragma unroll
      for (int a = 0; a < 5; a++) {
#pragma unroll
        for (int b = 0; b < 5; b++) {
          const int zw = hull81_zw(a, b);  // folds after unrolling
          if (zw < 0) continue;
          const T r2ab = X[a] + Y[b];
          if (r2ab > T(4)) continue;
          const T *row = corner + LZ * (b + LY * a);
#pragma unroll
          for (int c = 0; c < 5; c++) {
            if (c < 2 - zw || c > 2 + zw) continue;
            const T q_sq = r2ab + Z[c];
            if (q_sq <= T(4)) {
              const T common = row[c] * sph_grad_folded<T>(q_sq, norm);
              vx += common * xh[a];
              vy += common * yh[b];
              vz += common * zh[c];
            }
          }
        }
      }
    }
    const T normalize = (T)hp.normalize;
    vx *= normalize;
    vy *= normalize;
    vz *= normalize;
    if (rsd) vz += (T)hp.f1 * vz;
    const long long p = sidx[s] & ~kSortFlagNoScatter;
    V[p] = vx;
    V[p + g.N] = vy;
    V[p + 2 * g.N] = vz;
  }
}

// ======================================================================================================
// ALPT displacement (Lag2Eul_non_zeldovich, Lag2Eul.cc:160-267; used when sfmodel != 1 and rsd_model is off).
//   delta(1) = dq q;  Phi = IFFT[-delta^(1)/k^2];  delta(2) from 4th-order finite differences of Phi (GFINDIFF);
//   A = D1 delta(1) - D2 delta(2);  B = -3 (sqrt(1 - 2/3 D1 delta(1)) - 1) (or 3);
//   Psi^_j = (k_j/k^2)(Im, -Re)[K A^ + (1 - K) B^], K = exp(-k^2 kth^2/2) / wtot;  cell-boundary average.
// Everything after the two R2Cs is linear in k-space, so the reference's 12 transforms per evaluation (3 convcomp
// + 6 theta2velcomp + ...) collapse into one k-space pass and the usual batched C2R.
// NB: the reference feeds +D1 delta (minus the divergence) to the velocity kernel here but -D1 delta in the
// Zel'dovich routine (Lag2Eul.cc:88), so its ALPT displacement has the opposite sign; reproduced as is.
// ======================================================================================================
// q^ -> (delta(1)^, Phi^) scaled for the following unnormalised C2Rs.  EqSolvers.cc:29-64.
template <typename T>
__global__ void __launch_bounds__(256)
k_alpt_poisson(Geo g, const C2<T> *__restrict__ qk, C2<T> *__restrict__ d1k, C2<T> *__restrict__ phik, double scale) {
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < g.Nhp;
       idx += (long long)gridDim.x * blockDim.x) {
    const int k = (int)(idx % g.nhp);
    const long long ij = idx / g.nhp;
    const int j = (int)(ij % g.n), i = (int)(ij / g.n);
    const double kx = kval(i, g.n, g.kfac), ky = kval(j, g.n, g.kfac), kz = kval(k, g.n, g.kfac);
    const double kmod2 = kx * kx + ky * ky + kz * kz;
    const double2 q = ld2<T>(qk, idx);
    const double fackern = (kmod2 > 0.) ? -1. / kmod2 : 0.;
    st2<T>(d1k, idx, scale * q.x, scale * q.y);
    st2<T>(phik, idx, fackern * (scale * q.x), fackern * (scale * q.y));
  }
}

// gradfindif (gradient.cpp:81-154) along one axis at cell (i, j, k); `stride` = element stride of that axis.
template <typename T>
__device__ __forceinline__ double findif_axis(const T *__restrict__ a, long long base, int c, int n, long long stride,
                                              double fac) {
  const int l = c > 0 ? c - 1 : n - 1, r = c + 1 < n ? c + 1 : 0;
  const int ll = c > 1 ? c - 2 : c - 2 + n, rr = c + 2 < n ? c + 2 : c + 2 - n;
  const long long o = base - (long long)c * stride;
  return -(fac * ((4.0 / 3) * ((double)a[o + l * stride] - (double)a[o + r * stride]) -
                  (1.0 / 6) * ((double)a[o + ll * stride] - (double)a[o + rr * stride])));
}

// First derivatives of Phi: g3[c] = d Phi / d x_c  (the `dummy` arrays of calc_m2v_mem, EqSolvers.cc:403-412)
template <typename T>
__global__ void __launch_bounds__(256) k_alpt_grad(Geo g, const T *__restrict__ phi, T *__restrict__ g3) {
  const double fac = g.n / (2. * g.L);
  const long long n = g.n;
  for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < g.N; p += (long long)gridDim.x * blockDim.x) {
    const int k = (int)(p % n);
    const long long ij = p / n;
    const int j = (int)(ij % n), i = (int)(ij / n);
    g3[p] = (T)findif_axis<T>(phi, p, i, g.n, n * n, fac);
    g3[p + g.N] = (T)findif_axis<T>(phi, p, j, g.n, n, fac);
    g3[p + 2 * g.N] = (T)findif_axis<T>(phi, p, k, g.n, 1, fac);
  }
}

// delta(2) (EqSolvers.cc:415-421) and the two divergence sources (Lag2Eul.cc:199-226).  d1 holds delta(1) on
// entry and the spherical-collapse source on exit; a2 receives D1 delta(1) - D2 delta(2).
template <typename T>
__global__ void __launch_bounds__(256)
k_alpt_sources(Geo g, const T *__restrict__ g3, T *__restrict__ d1, T *__restrict__ a2, double D1, double D2) {
  const double fac = g.n / (2. * g.L);
  const long long n = g.n;
  const T *gx = g3, *gy = g3 + g.N, *gz = g3 + 2 * g.N;
  for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < g.N; p += (long long)gridDim.x * blockDim.x) {
    const int k = (int)(p % n);
    const long long ij = p / n;
    const int j = (int)(ij % n), i = (int)(ij / n);
    const double xx = findif_axis<T>(gx, p, i, g.n, n * n, fac), xy = findif_axis<T>(gx, p, j, g.n, n, fac),
                 xz = findif_axis<T>(gx, p, k, g.n, 1, fac);
    const double yy = findif_axis<T>(gy, p, j, g.n, n, fac), yz = findif_axis<T>(gy, p, k, g.n, 1, fac);
    const double zz = findif_axis<T>(gz, p, k, g.n, 1, fac);
    const double m2v = xx * yy - xy * xy + xx * zz - xz * xz + yy * zz - yz * yz;
    const double dl = (double)d1[p];
    a2[p] = (T)(D1 * dl - D2 * m2v);
    const double psilin = -D1 * dl;
    double psisc;
    if (1. + 2. / 3. * psilin > 0.)
      psisc = 3. * (sqrt(1. + 2. / 3. * psilin) - 1.);
    else
      psisc = -3.;
    d1[p] = (T)(-psisc);
  }
}

// Gaussian split kernel on the half-complex grid (kernelcomp, convolution.cpp:224-324, filtertype 1)
template <typename T>
__global__ void k_alpt_kernel_table(Geo g, C2<T> *__restrict__ out, double smol) {
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < g.Nhp;
       idx += (long long)gridDim.x * blockDim.x) {
    const int k = (int)(idx % g.nhp);
    const long long ij = idx / g.nhp;
    const int j = (int)(ij % g.n), i = (int)(ij / g.n);
    const double kx = kval(i, g.n, g.kfac), ky = kval(j, g.n, g.kfac), kz = kval(k, g.n, g.kfac);
    st2<T>(out, idx, (k < g.nh) ? exp(-(kx * kx + ky * ky + kz * kz) * smol * smol / 2.) : 0., 0.);
  }
}

// Psi^_j = (k_j/k^2)(Im, -Re)[K A^ + (1 - K) B^] / N, Nyquist planes and k^2 <= 1e-14 -> 0
// (theta2velcomp EqSolvers.cc:280-368 + convcomp convolution.cpp:327-377, combined).  A^ = Ck[0], B^ = Ck[1] on entry.
template <typename T>
__global__ void __launch_bounds__(256) k_alpt_mix(Geo g, C2<T> *Ck, double smol, double inv_wtot, double inv_n) {
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < g.Nhp;
       idx += (long long)gridDim.x * blockDim.x) {
    const int k = (int)(idx % g.nhp);
    const long long ij = idx / g.nhp;
    const int j = (int)(ij % g.n), i = (int)(ij / g.n);
    const double kx = kval(i, g.n, g.kfac), ky = kval(j, g.n, g.kfac), kz = kval(k, g.n, g.kfac);
    const double ksq = kx * kx + ky * ky + kz * kz;
    const double2 A = ld2<T>(Ck, idx), B = ld2<T>(Ck, idx + g.Nhp);
    double2 ox = make_double2(0., 0.), oy = ox, oz = ox;
    const bool nyq = (i == g.n / 2) || (j == g.n / 2) || (k == g.n / 2);
    if (ksq > 1.e-14 && !nyq) {
      const double K = exp(-ksq * smol * smol / 2.) * inv_wtot;
      // K o Psi^2LPT + Psi^SC - K o Psi^SC, in the reference's order of operations (Lag2Eul.cc:240-250)
      const double mr = (K * A.x + B.x) - K * B.x, mi = (K * A.y + B.y) - K * B.y;
      const double fac = inv_n / ksq;
      const double fx = fac * kx, fy = fac * ky, fz = fac * kz;
      ox = make_double2(fx * mi, fx * -mr);
      oy = make_double2(fy * mi, fy * -mr);
      oz = make_double2(fz * mi, fz * -mr);
    }
    st2<T>(Ck, idx, ox.x, ox.y);
    st2<T>(Ck, idx + g.Nhp, oy.x, oy.y);
    st2<T>(Ck, idx + 2 * g.Nhp, oz.x, oz.y);
  }
}

// cellboundcomp (massFunctions.cc:588-658): out[l] = (in[l] + in[l - (1,1,1)]) / 2, periodic; 3 components
template <typename T>
__global__ void __launch_bounds__(256) k_alpt_cellbound(Geo g, const T *__restrict__ in3, T *__restrict__ out3) {
  const long long n = g.n;
  for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < g.N; p += (long long)gridDim.x * blockDim.x) {
    const int k = (int)(p % n);
    const long long ij = p / n;
    const int j = (int)(ij % n), i = (int)(ij / n);
    const int im = i > 0 ? i - 1 : g.n - 1, jm = j > 0 ? j - 1 : g.n - 1, km = k > 0 ? k - 1 : g.n - 1;
    const long long m = km + n * (jm + n * (long long)im);
#pragma unroll
    for (int c = 0; c < 3; c++) out3[p + c * g.N] = (T)(0.5 * ((double)in3[m + c * g.N] + (double)in3[p + c * g.N]));
  }
}

// ------------------------------------------------------------------------------------------------------
// measure_spectrum (field_statistics.cpp:20-90) on a half-complex transform: per bin sum of |k|, of |F|^2 and the
// mode count, every mode weighted by the number of full-grid modes it stands for (itself + its conjugate partner).
// bins = [3][n_bin] doubles (ksum, psum, count); LDS histogram per workgroup, one flush.
// Compiled without FMA contraction so that |k| and the bin index are the reference's (x86-64, no FMA) numbers.
// ------------------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256)
k_spectrum(Geo g, const C2<T> *__restrict__ xk, int n_bin, double dk, double *__restrict__ bins) {
#pragma clang fp contract(off)
  extern __shared__ double s_bins[];
  for (int b = threadIdx.x; b < 3 * n_bin; b += blockDim.x) s_bins[b] = 0.;
  __syncthreads();
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < g.Nhp;
       idx += (long long)gridDim.x * blockDim.x) {
    const int k = (int)(idx % g.nhp);
    if (k >= g.nh) continue;
    const long long ij = idx / g.nhp;
    const int j = (int)(ij % g.n), i = (int)(ij / g.n);
    const double kx = kval(i, g.n, g.kfac), ky = kval(j, g.n, g.kfac), kz = kval(k, g.n, g.kfac);
    const double ktot = sqrt(kx * kx + ky * ky + kz * kz);
    const unsigned long long nbin = (unsigned long long)(ktot / dk);
    if (nbin < (unsigned long long)n_bin) {
      const double hw = (k == 0 || ((g.n & 1) == 0 && k == g.n / 2)) ? 1. : 2.;
      const double2 x = ld2<T>(xk, idx);
      atomic_add_r(&s_bins[nbin], hw * ktot);
      atomic_add_r(&s_bins[n_bin + nbin], hw * (x.x * x.x + x.y * x.y));
      atomic_add_r(&s_bins[2 * n_bin + nbin], hw);
    }
  }
  __syncthreads();
  for (int b = threadIdx.x; b < 3 * n_bin; b += blockDim.x)
    if (s_bins[b] != 0.) atomic_add_r(&bins[b], s_bins[b]);
}

// ======================================================================================================
// Step boundary with the x passes of both transforms fused in ("planes" mode).
//
// A 3-D real transform is the batched 2-D transform of the (y, z) planes followed by complex FFTs along x, and the
// step boundary is element-wise in k-space.  So between the gather and the next scatter the pipeline can be
//   rocFFT 2-D R2C over the 3 n planes of V  ->  THIS kernel: [x-FFT of the three V^ columns, the boundary
//   arithmetic of k_step_boundary, inverse x-FFT of the three Psi^ columns]  ->  rocFFT 2-D C2R over the planes,
// which saves two of the six strided passes over the 3-component arrays (measured: 2-D batched transforms
// 0.36 + 0.33 ms against 0.58 + 0.55 ms for the 3-D ones at 256^3 fp64, scripts/fft2d_bench.hip).
//
// One workgroup owns the x-columns of KB = 128 B / sizeof(complex) adjacent k at one j: n x KB elements,
// staged in LDS, four transforms per workgroup (V_x and ky V_y + kz V_z forward; kx B and B inverse, see below); radix-4 decimation-in-time FFTs in place (bit-reversed fill, twiddles from a table).
// FFT arithmetic in T (like rocFFT's plan precision), boundary arithmetic in double (like every k-space kernel).
// Requires n a power of two with n == PER * NT / KB, and nhp a multiple of KB.
// ======================================================================================================
template <typename T>
__device__ __forceinline__ C2<T> cmul(const C2<T> a, const C2<T> b) {
  C2<T> r;
  r.x = a.x * b.x - a.y * b.y;
  r.y = a.x * b.y + a.y * b.x;
  return r;
}

// In-place decimation-in-time FFT of KB interleaved columns: s[i * KB + c], bit-reversed input order on entry,
// natural order on exit.  Two radix-2 stages are fused into one radix-4 pass (4 LDS reads + 4 writes per 4 points
// per two stages); an odd log2 n gets one plain radix-2 stage first.  tw[r] = exp(-2 pi i r / n), r < n / 2.
template <typename T>
__device__ __forceinline__ void xfft_inplace(C2<T> *__restrict__ s, const C2<T> *__restrict__ tw, int n, int log2n,
                                             int KB, bool inverse) {
  int st = 1;
  if (log2n & 1) {  // stage 1: half = 1, twiddle 1
    const int nb = (n >> 1) * KB;
    for (int b = threadIdx.x; b < nb; b += blockDim.x) {
      const int c = b % KB, i0 = (b / KB) << 1;
      const C2<T> a = s[i0 * KB + c], x = s[(i0 + 1) * KB + c];
      C2<T> o0, o1;
      o0.x = a.x + x.x; o0.y = a.y + x.y;
      o1.x = a.x - x.x; o1.y = a.y - x.y;
      s[i0 * KB + c] = o0;
      s[(i0 + 1) * KB + c] = o1;
    }
    __syncthreads();
    st = 2;
  }
  const int nq = (n >> 2) * KB;
  for (; st < log2n; st += 2) {  // stages st and st + 1
    const int half = 1 << (st - 1);
    const int t1 = n >> st, t2 = n >> (st + 1);  // twiddle strides of the two stages
    for (int b = threadIdx.x; b < nq; b += blockDim.x) {
      const int c = b % KB, bf = b / KB;
      const int r = bf & (half - 1), grp = bf >> (st - 1);
      const int j = (grp << (st + 1)) + r;
      C2<T> w1 = tw[r * t1], w2 = tw[r * t2];
      if (inverse) {
        w1.y = -w1.y;
        w2.y = -w2.y;
      }
      const C2<T> e0 = s[j * KB + c], e1 = s[(j + half) * KB + c], e2 = s[(j + 2 * half) * KB + c],
                  e3 = s[(j + 3 * half) * KB + c];
      const C2<T> m1 = cmul<T>(w1, e1), m3 = cmul<T>(w1, e3);
      C2<T> a0, a1, a2, a3;
      a0.x = e0.x + m1.x; a0.y = e0.y + m1.y;
      a1.x = e0.x - m1.x; a1.y = e0.y - m1.y;
      a2.x = e2.x + m3.x; a2.y = e2.y + m3.y;
      a3.x = e2.x - m3.x; a3.y = e2.y - m3.y;
      const C2<T> n2 = cmul<T>(w2, a2), n3 = cmul<T>(w2, a3);
      // second-stage twiddle of the odd pair is w2 * exp(-+ i pi / 2): multiply by -i (forward) / +i (inverse)
      C2<T> r3;
      if (inverse) {
        r3.x = -n3.y; r3.y = n3.x;
      } else {
        r3.x = n3.y; r3.y = -n3.x;
      }
      C2<T> o0, o1, o2, o3;
      o0.x = a0.x + n2.x; o0.y = a0.y + n2.y;
      o2.x = a0.x - n2.x; o2.y = a0.y - n2.y;
      o1.x = a1.x + r3.x; o1.y = a1.y + r3.y;
      o3.x = a1.x - r3.x; o3.y = a1.y - r3.y;
      s[j * KB + c] = o0;
      s[(j + half) * KB + c] = o1;
      s[(j + 2 * half) * KB + c] = o2;
      s[(j + 3 * half) * KB + c] = o3;
    }
    __syncthreads();
  }
}

template <typename T, int NT, int PER>
__global__ void __launch_bounds__(NT, 4)
k_step_boundary_x(Geo g, int log2n, const C2<T> *__restrict__ twiddle, C2<T> *Ck, const C2<T> *q_in, const C2<T> *p_in,
                  C2<T> *q_out, C2<T> *p_out, const double *__restrict__ wS, const double *__restrict__ wM, double a,
                  double b, double half_eps, double eps, double c_za, double *guard_slot, StepCtl ctl) {
  constexpr int KB = 128 / (int)sizeof(C2<T>);
  constexpr int kMaxPer = PER;  // elements of one component per thread: n == PER * NT / KB (checked by the host)
  extern __shared__ __attribute__((aligned(16))) unsigned char s_raw_x[];
  __shared__ double red[NT / 64];
  if (*ctl.stop) return;
  if (ctl.guard_prev && fabs(*ctl.guard_prev) > ctl.guard_limit) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      *ctl.steps_done = ctl.step_index;
      __threadfence();
      *ctl.stop = 1;
    }
    return;
  }
  const int n = g.n;
  C2<T> *s = reinterpret_cast<C2<T> *>(s_raw_x);  // n * KB
  C2<T> *tw = s + (size_t)n * KB;                 // n / 2
  for (int t = threadIdx.x; t < n / 2; t += blockDim.x) tw[t] = twiddle[t];
  const int ntk = g.nhp / KB;
  const int j = blockIdx.x / ntk, k0 = (blockIdx.x % ntk) * KB;
  const int c = threadIdx.x % KB, irow = threadIdx.x / KB;
  constexpr int rows = NT / KB, per = PER;
  const int k = k0 + c;
  const long long plane = (long long)g.n * g.nhp;  // elements between consecutive i
  const long long col = k + (long long)g.nhp * j;  // element (0, j, k)
  const double ky = kval(j, g.n, g.kfac), kz = kval(k, g.n, g.kfac);
  const int shift = 32 - log2n;
  double2 hk[kMaxPer];
  // ---- forward x passes.  ky and kz are constants of a column, so V^_y and V^_z go through ONE transform as
  // ky V_y + kz V_z:  h^ = (1/k^2) [ kx (Im V^_x, -Re V^_x) + (Im W^, -Re W^) ],  W = ky V_y + kz V_z ----
  for (int pass = 0; pass < 2; pass++) {
    __syncthreads();
    for (int m = 0; m < per; m++) {
      const int i = irow + rows * m;
      const long long e = col + plane * i;
      C2<T> v;
      if (pass == 0) {
        v = Ck[e];
      } else {
        const C2<T> vy = Ck[e + g.Nhp], vz = Ck[e + 2 * g.Nhp];
        v.x = (T)(ky * (double)vy.x + kz * (double)vz.x);
        v.y = (T)(ky * (double)vy.y + kz * (double)vz.y);
      }
      s[(int)(__brev((unsigned)i) >> shift) * KB + c] = v;
    }
    __syncthreads();
    xfft_inplace<T>(s, tw, n, log2n, KB, false);
#pragma unroll
    for (int m = 0; m < kMaxPer; m++) {
      const int i = irow + rows * m;
      const C2<T> v = s[i * KB + c];
      if (pass == 0) {
        const double kx = kval(i, g.n, g.kfac);
        hk[m] = make_double2(kx * (double)v.y, -(kx * (double)v.x));
      } else {
        hk[m].x += (double)v.y;
        hk[m].y -= (double)v.x;
      }
    }
  }
  // ---- boundary arithmetic (as k_step_boundary); the first inverse transform's input is filled on the way ----
  // Psi^_j = k_j B with B = (1/k^2)(Im phi^, -Re phi^), phi^ = c_za q^': kx B is transformed on its own, B once for
  // both the y and the z component (ky, kz are constants of the column again).
  double gsum = 0.;
  __syncthreads();
#pragma unroll
  for (int m = 0; m < kMaxPer; m++) {
    const int i = irow + rows * m;
    const long long idx = col + plane * i;
    const double kx = kval(i, g.n, g.kfac);
    const double ksq = kx * kx + ky * ky + kz * kz;
    const bool nyq = (i == g.n / 2) || (j == g.n / 2) || (k == g.n / 2);
    double2 q = ld2<T>(q_in, idx);
    double2 gg = make_double2(0., 0.);
    if (ksq > 0 && !nyq) {
      const double f = b * (1 / ksq);
      gg = make_double2(f * hk[m].x, f * hk[m].y);
    }
    if (a != 0.) {
      const double w = a * wS[idx];
      gg.x += w * q.x;
      gg.y += w * q.y;
    }
    double2 p = ld2<T>(p_in, idx);
    C2<T> pe, gs;
    pe.x = (T)(p.x - half_eps * gg.x);
    pe.y = (T)(p.y - half_eps * gg.y);
    gs.x = (T)gg.x;
    gs.y = (T)gg.y;
    const double hw = (k == 0 || ((g.n & 1) == 0 && k == g.n / 2)) ? 1. : 2.;
    if (k < g.nh) gsum += hw * (double)pe.x;
    p.x = (double)pe.x - half_eps * (double)gs.x;
    p.y = (double)pe.y - half_eps * (double)gs.y;
    st2<T>(p_out, idx, p.x, p.y);
    if (wM) {
      const double w = wM[idx];
      q.x += eps * (w * p.x);
      q.y += eps * (w * p.y);
    }
    st2<T>(q_out, idx, q.x, q.y);
    C2<T> o;
    o.x = T(0);
    o.y = T(0);
    if (ksq > 1.e-14 && !nyq) {
      const double f = (1. / ksq) * kx;
      o.x = (T)(f * (c_za * q.y));
      o.y = (T)(f * -(c_za * q.x));
    }
    s[(int)(__brev((unsigned)i) >> shift) * KB + c] = o;
  }
  __syncthreads();
  xfft_inplace<T>(s, tw, n, log2n, KB, true);
  for (int m = 0; m < per; m++) {
    const int i = irow + rows * m;
    Ck[col + plane * i] = s[i * KB + c];
  }
  __syncthreads();
  for (int m = 0; m < per; m++) {
    const int i = irow + rows * m;
    const double kx = kval(i, g.n, g.kfac);
    const double ksq = kx * kx + ky * ky + kz * kz;
    const bool nyq = (i == g.n / 2) || (j == g.n / 2) || (k == g.n / 2);
    C2<T> o;
    o.x = T(0);
    o.y = T(0);
    if (ksq > 1.e-14 && !nyq) {
      // q' as stored (each thread re-reads its own stores; for T = float this is the rounded value)
      const double2 qn = ld2<T>(q_out, col + plane * i);
      const double f = 1. / ksq;
      o.x = (T)(f * (c_za * qn.y));
      o.y = (T)(f * -(c_za * qn.x));
    }
    s[(int)(__brev((unsigned)i) >> shift) * KB + c] = o;
  }
  __syncthreads();
  xfft_inplace<T>(s, tw, n, log2n, KB, true);
  for (int m = 0; m < per; m++) {
    const int i = irow + rows * m;
    const C2<T> v = s[i * KB + c];
    C2<T> oy, oz;
    oy.x = (T)(ky * (double)v.x);
    oy.y = (T)(ky * (double)v.y);
    oz.x = (T)(kz * (double)v.x);
    oz.y = (T)(kz * (double)v.y);
    Ck[col + plane * i + g.Nhp] = oy;
    Ck[col + plane * i + 2 * g.Nhp] = oz;
  }
  gsum = block_sum(gsum, red);
  if (threadIdx.x == 0) atomic_add_r(guard_slot, gsum);
}

}  // namespace bchmc
