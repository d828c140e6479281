// forward_model.hpp -- Particle positions, SPH mass assignment (direct kernels), mean density, likelihood partials, direct SPH-gradient gather.
// Part of the bchmc engine's kernel set; include through kernels.hpp (definition order matters).
#pragma once
#include "common.hpp"

namespace bchmc {

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
template <typename T, bool FIX>
__global__ void __launch_bounds__(256)
k_scatter_sph(Geo g, PosPar pp, SphPar sp, const T *__restrict__ psi, typename Cell<FIX, T>::type *__restrict__ rho,
              double fix_scale) {
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
      auto *row = rho + (long long)n * (ky + (long long)n * kx);
      for (int i3 = -R; i3 <= R; ++i3) {
        const double dz = z - (ccz + (double)i3 * d);
        const double r2 = r2ab + dz * dz;
        if (r2 > sp.r2_lim) continue;
        const double r = sqrt(r2);
        const double q = r / sp.h;
        if (q <= 2.) {
          const long long kz = (iz + i3 + (long long)n * 4) % n;
          cell_add(row + kz, sph_w(q, sp.w_norm), fix_scale);
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

// Deterministic mode: fixed-point density -> field array, and the kRedBlocks partial sums of the converted values
// in a fixed order (grid = kRedBlocks workgroups, fixed cell -> workgroup map, tree reductions).
template <typename T>
__global__ void __launch_bounds__(256)
k_fix_to_rho(long long N, const long long *__restrict__ fix, double inv_scale, T *__restrict__ rho,
             double *__restrict__ partials, int *__restrict__ saturated, long long sat_limit) {
  __shared__ double red[4];
  double acc = 0.;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < N; i += (long long)gridDim.x * blockDim.x) {
    const long long f = fix[i];
    // every contribution is >= 0, so a negative cell has wrapped, and one beyond 2^62 (2^16 maximal contributions) is
    // within a factor two of doing so: the host turns the flag into BCHMC_ERR_STATE at its next read-back instead of
    // handing out a plausible-looking density
    if (f > sat_limit || f < 0) *saturated = 1;  // sat_limit = 2^62; benign race: every writer stores 1
    const T v = (T)((double)f * inv_scale);
    rho[i] = v;
    acc += (double)v;
  }
  acc = block_sum(acc, red);
  if (threadIdx.x == 0) partials[blockIdx.x] = acc;
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

// One cell of partial_f_delta_x_log_like (gaussian_independent.cpp:24-42, poissonian.cpp:19-34,
// lognormal_independent.cpp:40-55) from its operands: w = window, nv = nobs, sv = noise (unused by the Poissonian).
__device__ __forceinline__ double partial_like_value(const LikePar &lp, double dX, double w, double nv, double sv) {
  double out = 0.;
  if (lp.likelihood == 1) {
    const double Lambda = w * lp.rho_c * pow_bias(1. + lp.biasP * dX, lp);
    if ((w > 0.) && (Lambda > 0.0)) out = (nv - Lambda) / (sv * sv);
  } else if (lp.likelihood == 0) {
    const double dens = 1. + lp.biasP * dX;
    if ((w > 0.0) && (dens > 0.0)) {
      const double Lambda = w * lp.rho_c * pow_bias(dens, lp);
      const double dpow = lp.bias_is_identity ? 1. : pow(dens, lp.biasE - 1);
      out = (1 - nv / Lambda) * lp.rho_c * lp.biasE * lp.biasP * dpow;
    }
  } else {  // 2: log-normal
    if (w > 0.) {
      const double Lambda = log(lp.rho_c * pow_bias(1. + lp.biasP * dX, lp));
      out = (nv - Lambda) / (sv * sv);
    }
  }
  return out;
}

// The three data operands of a cell, all requested before any of them is used (streaming hints: read once per step).
template <typename T>
__device__ __forceinline__ void like_operands(const LikePar &lp, long long i, const T *__restrict__ nobs,
                                              const T *__restrict__ noise, const T *__restrict__ window, double &w,
                                              double &nv, double &sv) {
  w = (double)stream_load<BCHMC_NT_LIKE != 0>(window + i);
  nv = (double)stream_load<BCHMC_NT_LIKE != 0>(nobs + i);
  sv = lp.likelihood != 0 ? (double)stream_load<BCHMC_NT_LIKE != 0>(noise + i) : 1.;
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
    double w, nv, sv;
    like_operands<T>(lp, i, nobs, noise, window, w, nv, sv);
    const double dX = (double)rho[i] / nmean - 1.;
    plike[i] = (T)partial_like_value(lp, dX, w, nv, sv);
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

}  // namespace bchmc
