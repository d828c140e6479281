// variants.hpp -- NGP / CIC / TSC mass assignment and the calc_h 0 / 3 likelihood-force variants.
// Part of the bchmc engine's kernel set; include through kernels.hpp (definition order matters).
#pragma once
#include "common.hpp"

namespace bchmc {

// ------------------------------------------------------------------------------------------------------
// NGP / CIC / TSC mass assignment (forward model only: getDensity_NGP massFunctions.cc:49-98,
// getDensity_CIC :100-164 with getCICcells/getCICweights interpolate_grid.cpp:27-79, getDensity_TSC :167-364).
// One thread per particle, 1 / 8 / 27 global atomics.  Index and weight formulas are the reference's,
// including the cell-centred CIC shift (x - d/2 wrapped) and TSC's inclusive `<= min + L` domain test.
// ------------------------------------------------------------------------------------------------------
template <typename T, bool FIX>
__global__ void __launch_bounds__(256)
k_scatter_low_order(Geo g, PosPar pp, SphPar sp, int mk, const T *__restrict__ psi,
                    typename Cell<FIX, T>::type *__restrict__ rho, double fix_scale) {
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
    cell_add(RHO_AT(ci, cj, ck), 1., fix_scale);
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
    cell_add(RHO_AT(c1[0], c1[1], c1[2]), (double)(T)(mass * tx[0] * tx[1] * tx[2]), fix_scale);
    cell_add(RHO_AT(c2[0], c1[1], c1[2]), (double)(T)(mass * dx[0] * tx[1] * tx[2]), fix_scale);
    cell_add(RHO_AT(c1[0], c2[1], c1[2]), (double)(T)(mass * tx[0] * dx[1] * tx[2]), fix_scale);
    cell_add(RHO_AT(c1[0], c1[1], c2[2]), (double)(T)(mass * tx[0] * tx[1] * dx[2]), fix_scale);
    cell_add(RHO_AT(c2[0], c2[1], c1[2]), (double)(T)(mass * dx[0] * dx[1] * tx[2]), fix_scale);
    cell_add(RHO_AT(c2[0], c1[1], c2[2]), (double)(T)(mass * dx[0] * tx[1] * dx[2]), fix_scale);
    cell_add(RHO_AT(c1[0], c2[1], c2[2]), (double)(T)(mass * tx[0] * dx[1] * dx[2]), fix_scale);
    cell_add(RHO_AT(c2[0], c2[1], c2[2]), (double)(T)(mass * dx[0] * dx[1] * dx[2]), fix_scale);
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
        for (int e = 0; e < 3; e++) cell_add(RHO_AT(c[0][a], c[1][b], c[2][e]), (double)(T)(1. * w[0][a] * w[1][b] * w[2][e]), fix_scale);
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

}  // namespace bchmc
