// common.hpp -- shared device helpers for the bchmc engine (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bchmc {

constexpr int kWave = 64;  // CDNA wavefront width

// Storage precision of the field arrays: T = double (reference DOUBLE_PREC) or float (BASELINE config 5).
template <typename T> struct Vec2;
template <> struct Vec2<double> { using type = double2; };
template <> struct Vec2<float> { using type = float2; };
template <typename T> using C2 = typename Vec2<T>::type;

template <typename T>
__device__ __forceinline__ double2 ld2(const C2<T> *p, long long i) {
  const C2<T> v = p[i];
  return make_double2((double)v.x, (double)v.y);
}
template <typename T>
__device__ __forceinline__ void st2(C2<T> *p, long long i, double x, double y) {
  C2<T> v;
  v.x = (T)x;
  v.y = (T)y;
  p[i] = v;
}

// Geometry and scalars every kernel needs, passed by value (fits in SGPRs).
struct Geo {
  int n;         // cells per axis
  int nh;        // n/2 + 1 (half-complex fastest axis, logical row length)
  int nhp;       // row stride of the half-complex arrays in complex elements: nh padded so that a row is a whole
                 // number of 128-byte lines (rocFFT's strided passes run 15-30 % faster on such rows)
  long long N;   // n^3
  long long Nh;  // n^2 * nh  (logical number of half-complex elements)
  long long Nhp; // n^2 * nhp (allocated elements per half-complex array; element (i, j, k) is at k + nhp * (j + n i)).
                 // The padding k >= nh holds zeros and is processed like data by the element-wise k-space kernels
                 // (whole 128-byte lines are written: partial-line stores cost ~10 % there); reductions skip it.
  double L;      // box side
  double d;      // cell size
  double kfac;   // 2*pi/L
};

// calc_ki, scale_space.cpp:41-51
__device__ __forceinline__ double kval(int i, int n, double kfac) {
  return (i <= n / 2) ? kfac * (double)i : -kfac * (double)(n - i);
}

__device__ __forceinline__ double r_fmod(double a, double b) { return fmod(a, b); }
__device__ __forceinline__ float r_fmod(float a, float b) { return fmodf(a, b); }
__device__ __forceinline__ double r_floor(double a) { return floor(a); }
__device__ __forceinline__ float r_floor(float a) { return floorf(a); }
__device__ __forceinline__ double r_max(double a, double b) { return fmax(a, b); }
__device__ __forceinline__ float r_max(float a, float b) { return fmaxf(a, b); }
__device__ __forceinline__ double r_fma(double a, double b, double c) { return fma(a, b, c); }
__device__ __forceinline__ float r_fma(float a, float b, float c) { return fmaf(a, b, c); }

// pacman_coordinate, pacman.cpp:20-28: if (x < 0) { x = fmod(x, L); x += L; }  if (x >= L) x = fmod(x, L);
// Same results without the fmod expansion on the paths that occur: for -L < x < 0 fmod(x, L) is x itself, and for
// L <= x < 2L it is x - L, which the subtraction gives exactly (Sterbenz); farther out (a blown-up trajectory) the
// library function runs.  The order of the two tests is the reference's (x + L may round to L and is then folded to 0).
template <typename T>
__device__ __forceinline__ T pacman(T x, T L) {
  if (x < T(0)) {
    if (__builtin_expect(!(x > -L), 0)) x = r_fmod(x, L);
    x += L;
  }
  if (x >= L) {
    if (__builtin_expect(!(x < L + L), 0)) x = r_fmod(x, L);
    else x = x - L;
  }
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

// Hardware float atomics (global_atomic_add_f64 / _f32, ds_add_f64 / _f32; no CAS loop); order-dependent
// in the last bits.
__device__ __forceinline__ void atomic_add_r(double *p, double v) { unsafeAtomicAdd(p, v); }
__device__ __forceinline__ void atomic_add_r(float *p, float v) { unsafeAtomicAdd(p, v); }

// Deterministic mode (bchmc_config.deterministic): mass assignment accumulates in 64-bit FIXED POINT -- integer adds
// are associative, so the density no longer depends on the order in which the atomics land (the reference announces
// exactly this run-to-run noise for its OpenMP build, barcode/main.cc:86-90).  A contribution v becomes
// llrint(v * scale) with scale = 2^46 / (largest possible contribution): quantisation 7e-15 of that value per
// contribution, head-room for 2^17 maximal contributions per cell.  `cell_add` is the one accumulate primitive of all
// mass-assignment kernels: hardware float atomics in the default mode, integer atomics here.
__device__ __forceinline__ void cell_add(double *p, double v, double) { unsafeAtomicAdd(p, v); }
__device__ __forceinline__ void cell_add(float *p, double v, double) { unsafeAtomicAdd(p, (float)v); }
__device__ __forceinline__ void cell_add(long long *p, double v, double scale) {
  atomicAdd(reinterpret_cast<unsigned long long *>(p), (unsigned long long)__double2ll_rn(v * scale));
}
template <bool FIX, typename T> struct Cell { using type = T; };
template <typename T> struct Cell<true, T> { using type = long long; };

// 1/sqrt(x) to ~1 ulp: hardware rsq seed + Newton steps.  Replaces the IEEE sqrt + divide pair of the
// reference's kernel evaluations (about 35 fp64 instructions with range scaling and fix-ups); results differ
// from the correctly rounded ones by <= 2 ulp, far inside the stated tolerance.  x must be positive and normal.
__device__ __forceinline__ double fast_rsqrt(double x) {
  double y = __builtin_amdgcn_rsq(x);
  const double hx = 0.5 * x;
  double e = fma(-hx * y, y, 0.5);
  y = fma(y, e, y);
  e = fma(-hx * y, y, 0.5);
  y = fma(y, e, y);
  return y;
}
__device__ __forceinline__ float fast_rsqrt(float x) {
  float y = __builtin_amdgcn_rsqf(x);
  const float e = fmaf(-0.5f * x * y, y, 0.5f);
  return fmaf(y, e, y);
}

// q = sqrt(x) together with rq = 1/sqrt(x), x > 0 and normal: hardware rsq seed (relative error e ~ 5e-8, measured
// with scripts/rsq_accuracy.hip) + ONE third-order step  y1 = y (1 + r/2 + 3 r^2/8),  r = 1 - x y^2  (error
// (5/16)(2e)^3 ~ 3e-22: the result is the correctly rounded one up to the roundings of the last fma, <= 1 ulp).
// Five fp64 instructions after the rsq for q (six with rq) instead of the nine of two Newton steps + x*y + scaling
// (scripts/valu_rates.hip: v_rsq_f64 costs 3.8 plain fp64 instructions, so the arithmetic around it is what counts).
__device__ __forceinline__ double sqrt_rsq(double x, double &rq) {
  const double y = __builtin_amdgcn_rsq(x);
  const double t = x * y;
  const double r = fma(-t, y, 1.0);
  const double s = r * fma(r, 0.375, 0.5);
  rq = fma(y, s, y);
  return fma(t, s, t);
}
__device__ __forceinline__ float sqrt_rsq(float x, float &rq) {
  rq = fast_rsqrt(x);
  return x * rq;
}

// Streaming access hints (the `nt` bit: the line is not kept in L2 after this access).  For arrays that are read or
// written once per step and are larger than the caches: the outputs and inputs of k_step_boundary_x (BCHMC_BX_NT),
// the data arrays k_partial_like reads (BCHMC_NT_LIKE).  Measured and NOT used where partial lines must merge in L2:
// the record stores of the binning (0.345 -> 0.43 ms) and the gather's V stores (0.97 -> 1.22 ms).
#ifndef BCHMC_NT_LIKE
#define BCHMC_NT_LIKE 1
#endif
template <bool ON, typename U>
__device__ __forceinline__ U stream_load(const U *p) {
  return ON ? __builtin_nontemporal_load(p) : *p;
}
template <bool ON, typename U>
__device__ __forceinline__ void stream_store(U *p, U v) {
  if (ON) __builtin_nontemporal_store(v, p);
  else *p = v;
}

template <typename T> __device__ __forceinline__ T tiny_pos();
template <> __device__ __forceinline__ double tiny_pos<double>() { return 1e-280; }
template <> __device__ __forceinline__ float tiny_pos<float>() { return 1e-30f; }

}  // namespace bchmc
