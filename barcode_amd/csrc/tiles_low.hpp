// tiles_low.hpp -- NGP / CIC / TSC mass assignment and the TSC interpolation of calc_h = 3 on the tile-sorted records.
// Part of the bchmc engine's kernel set; include through kernels.hpp (after tiles.hpp).
#pragma once
#include "tiles.hpp"

namespace bchmc {

// ======================================================================================================
// The low-order mass kernels (getDensity_NGP massFunctions.cc:49-98, getDensity_CIC :100-164 with getCICcells /
// getCICweights interpolate_grid.cpp:27-79, getDensity_TSC :167-364) and interpolate_TSC (interpolate_grid.cpp:134-202)
// touch 1 / 8 / 27 cells around the particle's home cell.  variants.hpp holds their one-thread-per-Lagrangian-particle
// form (global atomics / global gathers at the Eulerian position: the formulation that ran the SPH scatter at 20.8 ms in
// round 1).  Here they run on the (tile, octant) records the binning pass produces anyway: one workgroup per (tile,
// chunk) accumulates into an LDS image of the tile plus a ONE-cell halo (10 x 10 x 18 cells for 8 x 8 x 16 tiles) and
// flushes it once, resp. stages the three convolved fields of the tile in LDS and interpolates from there.
// Cell indices and weights are the reference's expressions, evaluated in double like variants.hpp does (IEEE divide,
// floor, modulo; CIC's cell-centred shift x - d/2; TSC's weights from the distance to the home cell centre; the `dz`
// for `dx, dy` slip of interpolate_TSC); only WHERE the sum is formed differs.  Requires xllc = yllc = zllc = 0 (the
// binning keys on floor(x / d), the reference's cells on floor((x - min) / d)); otherwise the direct kernels run.
// ======================================================================================================
constexpr int kLowHalo = 1;

// LDS image coordinate of global cell c (0 <= c < n) along an axis whose image starts at global cell o (may be -1)
// and is l cells long; -1 if the cell is not in the image (cannot happen for the 27 cells around the home cell of a
// particle binned into this tile; the callers then fall back to a global access, so a wrong guess costs time only).
__device__ __forceinline__ int low_local(int c, int o, int l, int n) {
  int r = c - o;
  if (r >= n) r -= n;
  if (r < 0) r += n;
  return r < l ? r : -1;
}

template <typename T, bool FIX>
__global__ void __launch_bounds__(256)
k_scatter_tile_low(Geo g, TilePar tp, int mk, const RecQuad *__restrict__ srec, const long long *__restrict__ off,
                   const long long *__restrict__ tend, const int *__restrict__ woff, const int4 *__restrict__ oct,
                   const int *__restrict__ seg_in, typename Cell<FIX, T>::type *__restrict__ rho,
                   double *__restrict__ rho_part, int *__restrict__ cnt_zero, int ncnt_zero, double fix_scale) {
  extern __shared__ __attribute__((aligned(16))) unsigned char s_raw_low[];
  using Acc = typename Cell<FIX, double>::type;
  Acc *img = reinterpret_cast<Acc *>(s_raw_low);
  // clear the binning counters for the next force evaluation (as k_scatter_tile81 does): nine per workgroup
  if (threadIdx.x < kOct + 1) {
    const int i = (kOct + 1) * (int)blockIdx.x + (int)threadIdx.x;
    if (i < ncnt_zero) cnt_zero[i] = 0;
  }
  int tile, pb, pe;
  long long rec0;
  OctMap om;
  if (!tile_work((int)blockIdx.x, tp, off, tend, woff, oct, seg_in, tile, rec0, pb, pe, om)) return;
  srec += rec0 * rec_quads<T>();
  const int lx = tp.tx + 2 * kLowHalo, ly = tp.ty + 2 * kLowHalo, lz = tp.tz + 2 * kLowHalo;
  const int ncell = lx * ly * lz;
  for (int c = threadIdx.x; c < ncell; c += blockDim.x) img[c] = Acc(0);
  __syncthreads();
  const int n = g.n;
  const int tzi = tile % tp.ntz, tyi = (tile / tp.ntz) % tp.nty, txi = tile / (tp.ntz * tp.nty);
  const int ox = txi * tp.tx - kLowHalo, oy = tyi * tp.ty - kLowHalo, oz = tzi * tp.tz - kLowHalo;
  const double d = g.d, L = g.L;
  // one contribution: into the LDS image, or -- should a cell ever lie outside it -- straight into the global array
  auto add = [&](long long cx, long long cy, long long cz, double v) {
    const int ax = low_local((int)cx, ox, lx, n), ay = low_local((int)cy, oy, ly, n), az = low_local((int)cz, oz, lz, n);
    if (ax >= 0 && ay >= 0 && az >= 0) cell_add(img + az + lz * (ay + ly * ax), v, fix_scale);
    else cell_add(rho + cz + (long long)n * (cy + (long long)n * cx), v, fix_scale);
  };
  for (int s = pb + threadIdx.x; s < pe; s += blockDim.x) {
    T xt, yt, zt;
    int rid;
    rec_load<T>(srec, om.slot(s), xt, yt, zt, rid);
    if (rid & kSortFlagNoScatter) continue;  // fails the domain test (massFunctions.cc:60, 118, 193 with min = 0)
    const double x = xt, y = yt, z = zt;
    if (mk == 0) {
      const unsigned ci = (unsigned)floor(x / d) % n, cj = (unsigned)floor(y / d) % n, ck = (unsigned)floor(z / d) % n;
      add(ci, cj, ck, 1.);
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
      add(c1[0], c1[1], c1[2], (double)(T)(mass * tx[0] * tx[1] * tx[2]));
      add(c2[0], c1[1], c1[2], (double)(T)(mass * dx[0] * tx[1] * tx[2]));
      add(c1[0], c2[1], c1[2], (double)(T)(mass * tx[0] * dx[1] * tx[2]));
      add(c1[0], c1[1], c2[2], (double)(T)(mass * tx[0] * tx[1] * dx[2]));
      add(c2[0], c2[1], c1[2], (double)(T)(mass * dx[0] * dx[1] * tx[2]));
      add(c2[0], c1[1], c2[2], (double)(T)(mass * dx[0] * tx[1] * dx[2]));
      add(c1[0], c2[1], c2[2], (double)(T)(mass * tx[0] * dx[1] * dx[2]));
      add(c2[0], c2[1], c2[2], (double)(T)(mass * dx[0] * dx[1] * dx[2]));
    } else {
      const double pos[3] = {x / d, y / d, z / d};
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
          for (int e = 0; e < 3; e++) add(c[0][a], c[1][b], c[2][e], (double)(T)(1. * w[0][a] * w[1][b] * w[2][e]));
    }
  }
  __syncthreads();
  double flushed = 0.;  // what this work item adds to rho: the mean density needs no pass over rho
  for (int c = threadIdx.x; c < ncell; c += blockDim.x) {
    const Acc v = img[c];
    if (v != Acc(0)) {
      const int cz = c % lz, cy = (c / lz) % ly, cx = c / (lz * ly);
      const int gx = (ox + cx + n) % n, gy = (oy + cy + n) % n, gz = (oz + cz + n) % n;
      flushed += flush_cell(rho + gz + (long long)n * (gy + (long long)n * gx), v);
    }
  }
  if (!FIX) {
    __shared__ double s_red_low[4];
    flushed = block_sum(flushed, s_red_low);
    if (threadIdx.x == 0 && flushed != 0.) atomic_add_r(rho_part + (blockIdx.x & (kRedBlocks - 1)), flushed);
  }
}

// interpolate_TSC of the three convolved fields (calc_h = 3; k_interp_tsc in variants.hpp is the direct form): the
// tile's 10 x 10 x 18 cells of each field staged in LDS, 27 x 3 LDS reads per particle instead of global gathers.
template <typename T>
__global__ void __launch_bounds__(256)
k_interp_tsc_tile(Geo g, TilePar tp, int rsd, double f1, const RecQuad *__restrict__ srec,
                  const long long *__restrict__ off, const long long *__restrict__ tend, const int *__restrict__ woff,
                  const int4 *__restrict__ oct, const int *__restrict__ seg_in, const T *__restrict__ conv,
                  T *__restrict__ V) {
  extern __shared__ __attribute__((aligned(16))) unsigned char s_raw_tsc[];
  T *img = reinterpret_cast<T *>(s_raw_tsc);  // three images, component-major
  int tile, pb, pe;
  long long rec0;
  OctMap om;
  if (!tile_work((int)blockIdx.x, tp, off, tend, woff, oct, seg_in, tile, rec0, pb, pe, om)) return;
  srec += rec0 * rec_quads<T>();
  const int lx = tp.tx + 2 * kLowHalo, ly = tp.ty + 2 * kLowHalo, lz = tp.tz + 2 * kLowHalo;
  const int ncell = lx * ly * lz;
  const int n = g.n;
  const int tzi = tile % tp.ntz, tyi = (tile / tp.ntz) % tp.nty, txi = tile / (tp.ntz * tp.nty);
  const int ox = txi * tp.tx - kLowHalo, oy = tyi * tp.ty - kLowHalo, oz = tzi * tp.tz - kLowHalo;
  for (int c = threadIdx.x; c < ncell; c += blockDim.x) {
    const int cz = c % lz, cy = (c / lz) % ly, cx = c / (lz * ly);
    const int gx = (ox + cx + n) % n, gy = (oy + cy + n) % n, gz = (oz + cz + n) % n;
    const long long f = gz + (long long)n * (gy + (long long)n * gx);
    img[c] = conv[f];
    img[c + ncell] = conv[f + g.N];
    img[c + 2 * ncell] = conv[f + 2 * g.N];
  }
  __syncthreads();
  for (int s = pb + threadIdx.x; s < pe; s += blockDim.x) {
    T xt, yt, zt;
    int rid;
    rec_load<T>(srec, om.slot(s), xt, yt, zt, rid);
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
    wx[2] = wy[2] = wz[2] = 0.5 * ((1.5 - fabs(dz - 1)) * (1.5 - fabs(dz - 1)));  // interpolate_grid.cpp:166-168
    const unsigned un = (unsigned)n;
    const unsigned ixx[3] = {(cx % un + un - 1) % un, cx % un, (cx + 1) % un};
    const unsigned ixy[3] = {(cy % un + un - 1) % un, cy % un, (cy + 1) % un};
    const unsigned ixz[3] = {(cz % un + un - 1) % un, cz % un, (cz + 1) % un};
    int ax[3], ay[3], az[3];
    bool inside = true;
#pragma unroll
    for (int a = 0; a < 3; a++) {
      ax[a] = low_local((int)ixx[a], ox, lx, n);
      ay[a] = low_local((int)ixy[a], oy, ly, n);
      az[a] = low_local((int)ixz[a], oz, lz, n);
      inside = inside && ax[a] >= 0 && ay[a] >= 0 && az[a] >= 0;
    }
    double o0 = 0., o1 = 0., o2 = 0.;
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int b = 0; b < 3; ++b)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const double w = wx[a] * wy[b] * wz[c];
          double v0, v1, v2;
          if (inside) {
            const int l = az[c] + lz * (ay[b] + ly * ax[a]);
            v0 = (double)img[l];
            v1 = (double)img[l + ncell];
            v2 = (double)img[l + 2 * ncell];
          } else {
            const long long f = ((long long)ixx[a] * n + ixy[b]) * n + ixz[c];
            v0 = (double)conv[f];
            v1 = (double)conv[f + g.N];
            v2 = (double)conv[f + 2 * g.N];
          }
          o0 += w * v0;
          o1 += w * v1;
          o2 += w * v2;
        }
    if (rsd) o2 += f1 * o2;
    const long long p = rid & ~kSortFlagNoScatter;
    V[p] = (T)o0;
    V[p + g.N] = (T)o1;
    V[p + 2 * g.N] = (T)o2;
  }
}

}  // namespace bchmc
