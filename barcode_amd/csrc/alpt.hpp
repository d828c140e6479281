// alpt.hpp -- ALPT displacement (Lag2Eul_non_zeldovich).
// Part of the bchmc engine's kernel set; include through kernels.hpp (definition order matters).
#pragma once
#include "common.hpp"

#include <algorithm>

namespace bchmc {

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

// Workgroup -> z-rows for the two stencil passes.  Workgroups are dealt round-robin over the 8 XCDs (blocks b and b + 8
// share one), each XCD has its own L2, and the 4th-order stencils reach two rows (j) and two planes (i) either way: with
// consecutive rows on consecutive workgroups every XCD fetched every row of the array for its y neighbours.  Here the
// row slots a workgroup takes (b, b + gridDim, ...; gridDim a multiple of 8) all lie in ONE slab of n / 8 consecutive j
// -- the same slab for all workgroups of an XCD -- so the y and x neighbours of a row are rows of the same XCD, but for
// the two rows at each slab edge.  Pure scheduling: any mapping gives the same numbers.  (Measured: the two passes
// 0.70 -> 0.61 ms at 256^3.  Staging the rows of a group of 4 output rows in LDS instead -- 6 and 9 coalesced row loads
// per output row, neighbours read from LDS -- was measured too and ran 1.10 ms: 72 KB of LDS per workgroup leave two
// workgroups per CU and nothing to overlap their load phases with; profiles/r03_ab_alpt_rows.txt.)
__device__ __forceinline__ void stencil_row(int slot, int n, int &i, int &j) {
  if ((n & 7) == 0) {
    const int slab = n >> 3, x = slot & 7, loc = slot >> 3;
    j = x * slab + loc % slab;
    i = loc / slab;
  } else {
    j = slot % n;
    i = slot / n;
  }
}
inline int stencil_grid(int n) {  // workgroups of the stencil passes: a multiple of 8, at most one per row
  const long long rows = (long long)n * n;
  return (int)std::max<long long>(std::min<long long>(rows, 4096) / 8 * 8, 8);
}

// First derivatives of Phi: g3[c] = d Phi / d x_c  (the `dummy` arrays of calc_m2v_mem, EqSolvers.cc:403-412)
template <typename T>
__global__ void __launch_bounds__(256) k_alpt_grad(Geo g, const T *__restrict__ phi, T *__restrict__ g3) {
  const double fac = g.n / (2. * g.L);
  const long long n = g.n;
  for (int slot = blockIdx.x; slot < g.n * g.n; slot += gridDim.x) {
    int i, j;
    stencil_row(slot, g.n, i, j);
    for (int k = threadIdx.x; k < g.n; k += blockDim.x) {
      const long long p = k + n * (j + n * (long long)i);
      g3[p] = (T)findif_axis<T>(phi, p, i, g.n, n * n, fac);
      g3[p + g.N] = (T)findif_axis<T>(phi, p, j, g.n, n, fac);
      g3[p + 2 * g.N] = (T)findif_axis<T>(phi, p, k, g.n, 1, fac);
    }
  }
}

// delta(2) (EqSolvers.cc:415-421) and the two divergence sources (Lag2Eul.cc:199-226).  d1 holds delta(1); a_out
// receives D1 delta(1) - D2 delta(2), b_out the spherical-collapse source.  Element-wise in (d1, a_out, b_out): either
// output may be the d1 array itself.
template <typename T>
__global__ void __launch_bounds__(256)
k_alpt_sources(Geo g, const T *__restrict__ g3, const T *d1, T *a_out, T *b_out, double D1, double D2) {
  const double fac = g.n / (2. * g.L);
  const long long n = g.n;
  const T *gx = g3, *gy = g3 + g.N, *gz = g3 + 2 * g.N;
  for (int slot = blockIdx.x; slot < g.n * g.n; slot += gridDim.x) {
    int i, j;
    stencil_row(slot, g.n, i, j);
    for (int k = threadIdx.x; k < g.n; k += blockDim.x) {
    const long long p = k + n * (j + n * (long long)i);
    const double xx = findif_axis<T>(gx, p, i, g.n, n * n, fac), xy = findif_axis<T>(gx, p, j, g.n, n, fac),
                 xz = findif_axis<T>(gx, p, k, g.n, 1, fac);
    const double yy = findif_axis<T>(gy, p, j, g.n, n, fac), yz = findif_axis<T>(gy, p, k, g.n, 1, fac);
    const double zz = findif_axis<T>(gz, p, k, g.n, 1, fac);
    const double m2v = xx * yy - xy * xy + xx * zz - xz * xz + yy * zz - yz * yz;
    const double dl = (double)d1[p];
    a_out[p] = (T)(D1 * dl - D2 * m2v);
    const double psilin = -D1 * dl;
    double psisc;
    if (1. + 2. / 3. * psilin > 0.)
      psisc = 3. * (sqrt(1. + 2. / 3. * psilin) - 1.);
    else
      psisc = -3.;
    b_out[p] = (T)(-psisc);
    }
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

// Psi^_j = P (k_j/k^2)(Im, -Re)[K A^ + (1 - K) B^] / N, Nyquist planes and k^2 <= 1e-14 -> 0
// (theta2velcomp EqSolvers.cc:280-368 + convcomp convolution.cpp:327-377, combined).  A^ = Ck[0], B^ = Ck[1] on entry.
// P = (1 + exp(-2 pi i (i + j + k) / n)) / 2 is cellboundcomp (massFunctions.cc:588-658: out[l] = (in[l] +
// in[l - (1,1,1)]) / 2, periodic) by the shift theorem -- exact on a periodic grid, so the averaging pass over the three
// real-space components is this factor (same numbers to round-off).
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
      double sn, cs;
      sincospi(-2. * (double)((i + j + k) % g.n) / (double)g.n, &sn, &cs);
      const double pr = 0.5 * (1. + cs), pi = 0.5 * sn;
      const double ex = fac * mi, ey = fac * -mr;
      const double er = ex * pr - ey * pi, ei = ex * pi + ey * pr;
      ox = make_double2(kx * er, kx * ei);
      oy = make_double2(ky * er, ky * ei);
      oz = make_double2(kz * er, kz * ei);
    }
    st2<T>(Ck, idx, ox.x, ox.y);
    st2<T>(Ck, idx + g.Nhp, oy.x, oy.y);
    st2<T>(Ck, idx + 2 * g.Nhp, oz.x, oz.y);
  }
}

}  // namespace bchmc
