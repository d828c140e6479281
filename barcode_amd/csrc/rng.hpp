// rng.hpp -- Philox4x32-10 momentum draw.
// Part of the bchmc engine's kernel set; include through kernels.hpp (definition order matters).
#pragma once
#include "common.hpp"

namespace bchmc {

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

}  // namespace bchmc
