// spectrum.hpp -- measure_spectrum.
// Part of the bchmc engine's kernel set; include through kernels.hpp (definition order matters).
#pragma once
#include "common.hpp"

namespace bchmc {

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

}  // namespace bchmc
