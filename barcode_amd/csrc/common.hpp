// common.hpp -- shared device helpers for the bchmc engine (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bchmc {

constexpr int kWave = 64;  // CDNA wavefront width

// Geometry and scalars every kernel needs, passed by value (fits in SGPRs).
struct Geo {
  int n;         // cells per axis
  int nh;        // n/2 + 1 (half-complex fastest axis)
  long long N;   // n^3
  long long Nh;  // n^2 * nh
  double L;      // box side
  double d;      // cell size
  double kfac;   // 2*pi/L
};

// calc_ki, scale_space.cpp:41-51
__device__ __forceinline__ double kval(int i, int n, double kfac) {
  return (i <= n / 2) ? kfac * (double)i : -kfac * (double)(n - i);
}

// pacman_coordinate, pacman.cpp:20-28
__device__ __forceinline__ double pacman(double x, double L) {
  if (x < 0.) {
    x = fmod(x, L);
    x += L;
  }
  if (x >= L) x = fmod(x, L);
  return x;
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) v += __shfl_down(v, off, kWave);
  return v;
}

// Block-wide sum; result valid in thread 0.  `red` must hold blockDim.x/64 doubles.
__device__ __forceinline__ double block_sum(double v, double *red) {
  v = wave_sum(v);
  const int lane = threadIdx.x & (kWave - 1), w = threadIdx.x / kWave;
  if (lane == 0) red[w] = v;
  __syncthreads();
  double r = 0.;
  if (w == 0) {
    const int nw = (blockDim.x + kWave - 1) / kWave;
    r = (lane < nw) ? red[lane] : 0.;
    r = wave_sum(r);
  }
  __syncthreads();
  return r;
}

// fp64 hardware atomic add (global_atomic_add_f64, no CAS loop); order-dependent in the last bits.
__device__ __forceinline__ void atomic_add_f64(double *p, double v) { unsafeAtomicAdd(p, v); }

}  // namespace bchmc
