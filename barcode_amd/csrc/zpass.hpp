// zpass.hpp -- The last two passes of the planes-mode inverse transform done by the engine itself, so that the z pass
// can end in the binning instead of in HBM: k_ypass (inverse complex FFTs along y, in place) and k_zbin_direct (inverse
// real FFTs along z of the three displacement components, particle positions, one-pass binning).
// Reference work these two kernels cover: the last two of the three passes of each of theta2vel's inverse transforms
// (EqSolvers.cc:274-276 -> fftC2Rplanned, fftwrapper.cc:88-102; the x pass is in k_step_boundary_x), and everything
// k_bin_direct covers (disp_part.cc:55-126, pacman.cpp:20-28, rsd.cc:28-68 + the tile binning, which has no counterpart
// upstream).  Same numbers as the rocFFT path to transform round-off (tests/test_gpu_large.py::*z_pass_inside_the_binning*).
// Part of the bchmc engine's kernel set; include through kernels.hpp (after step_boundary_x.hpp and tiles.hpp).
#pragma once
#include "common.hpp"

#ifndef BCHMC_YPASS_NT
#define BCHMC_YPASS_NT 0  // streaming hints on k_ypass (1 loads and stores, 2 loads, 3 stores).  Alone, scripts/zpass_bench.hip,
                          // they are worth 0.161 -> 0.142 ms on some boxes and nothing on others; in the engine the kernel that
                          // follows re-reads what this one wrote, and without hints the y + z pair is 0.012 ms faster (C2R class
                          // 0.189 / 0.177 / 0.176 ms with 1 / 2 / 0, same box, 3 alternations)
#endif

namespace bchmc {

// ======================================================================================================
// k_ypass: in-place inverse complex FFT along y of `ncomp` planes-space arrays (element (i, j, k) at k + nhp (j + n i),
// components Nhp apart).  One workgroup owns the y-columns of KB = 128 B / sizeof(complex) adjacent k of one i plane
// of one component: n x KB elements in LDS, bit-reversed fill, the radix-4 passes of xfft_inplace, natural-order
// store -- rocFFT's column pass (`sbcc`) of the 2-D plan, written here because rocFFT cannot be asked for the y pass
// alone (its batch is one-dimensional; the y columns are batched over i AND k).  Unnormalised, like rocFFT.
// Requires n a power of two with n == PER * NT / KB and nhp a multiple of KB.
// ======================================================================================================
// INV = false: the forward transform (the y pass of the planes-mode R2C after rocFFT's row pass, used where rocFFT's own
// column kernel is the slower one: 512^3).
template <typename T, int NT, int PER, int HINT = 0, bool INV = true>  // streaming hints: 1 loads and stores, 2 loads only, 3 stores only
__global__ void __launch_bounds__(NT)
k_ypass(Geo g, int log2n, const C2<T> *__restrict__ twiddle, C2<T> *c) {
  constexpr int KB = 128 / (int)sizeof(C2<T>);
  extern __shared__ __attribute__((aligned(16))) unsigned char s_raw_y[];
  const int n = g.n;
  C2<T> *s = reinterpret_cast<C2<T> *>(s_raw_y);  // n * KB
  C2<T> *tw = s + (size_t)n * KB;                 // n / 2
  for (int t = threadIdx.x; t < n / 2; t += blockDim.x) tw[t] = twiddle[t];
  const int ntk = g.nhp / KB;
  const int per_comp = n * ntk;
  const int comp = (int)blockIdx.x / per_comp, b = (int)blockIdx.x % per_comp;
  const int i = b / ntk, k0 = (b % ntk) * KB;
  const int col = threadIdx.x % KB, jrow = threadIdx.x / KB;
  constexpr int rows = NT / KB;
  C2<T> *base = c + (long long)comp * g.Nhp + (long long)g.nhp * g.n * i + k0 + col;  // element (i, 0, k0 + col)
  const int shift = 32 - log2n;
  C2<T> v[PER];
#pragma unroll
  for (int m = 0; m < PER; m++)
    v[m] = (HINT == 1 || HINT == 2) ? bx_load(base + (long long)g.nhp * (jrow + rows * m))
                                    : base[(long long)g.nhp * (jrow + rows * m)];
#pragma unroll
  for (int m = 0; m < PER; m++) s[(int)(__brev((unsigned)(jrow + rows * m)) >> shift) * KB + col] = v[m];
  __syncthreads();
  xfft_inplace<T>(s, tw, n, log2n, KB, INV);
#pragma unroll
  for (int m = 0; m < PER; m++) {
    const int j = jrow + rows * m;
    if (HINT == 1 || HINT == 3) bx_store(base + (long long)g.nhp * j, s[j * KB + col]);
    else base[(long long)g.nhp * j] = s[j * KB + col];
  }
}


// ======================================================================================================
// k_zbin_direct: the z pass of the inverse transform of the three displacement components fused into the one-pass
// binning (k_bin_direct, tiles.hpp): Psi never goes to HBM between them (3 R written by rocFFT's row pass + 3 R read by
// the binning, 0.8 GB per step at 256^3 fp64).
//
// One workgroup of n threads takes the four z rows (i0 + f, j0 + e), f, e in {0, 1}, of the 2 x 2 x n column of the
// Lagrangian lattice: 4 n particles.  Two real rows go through ONE complex transform: with half-complex spectra A, B
// of the rows (i0 + f, j0) and (i0 + f, j0 + 1), Z[k] = A[k] + i B[k] for k <= n/2 and Z[n - k] = conj(A[k]) + i conj(B[k])
// is the spectrum of a + i b, so the inverse complex transform returns row a in its real and row b in its imaginary
// part (unnormalised, like rocFFT's C2R; the imaginary parts of the k = 0 and k = n/2 inputs do not enter, as in a
// C2R).  The 2 pairs x 3 components are six interleaved columns of one LDS tile and one xfft_inplace call (four
// barriers).  Thread t then owns the four particles at k = t and does what k_bin_direct does with them.
//
// Counters.  A z column crosses all n / tz tiles along z, so a workgroup meets about four times as many (tile, octant)
// counters per particle as a brick-shaped one, and the returning global atomics on them -- many columns hit the same
// tiles at the same time -- cost more than everything else in the kernel (scripts/zpass_bench.hip: 0.44 ms with one
// atomic per counter, 0.30 ms without any).  Two neighbouring counters (octant segments 2 q and 2 q + 1 of a tile: two
// adjacent ints) are therefore reserved with ONE 64-bit atomic add of both counts (the low word cannot carry into the
// high one: a counter stays far below 2^32), in LDS and in HBM: 0.31 ms.
//
// PSI_ONLY: the kernel for the rare step in which the binning overflowed -- returns at once unless *ovf is set, else
// transforms again and stores the displacements where rocFFT's row pass would have, for the two-pass fallback sort.
// psi_out != nullptr (binning variant): the same store on the way (force evaluations whose positions are fetched).
// Requires n == NZ (one lattice site along z per thread), n in {128, 256, 512}.
// ======================================================================================================
#ifndef BCHMC_ZBIN_WAVES  // 3 / 4 / 5 waves per SIMD measured alike for fp64 at 256^3 (0.350 / 0.353 / 0.344 ms); a 512-thread
#define BCHMC_ZBIN_WAVES ((sizeof(T) == 8 && NZ < 512) ? 3 : 4)  // workgroup needs 4 for two workgroups per CU
#endif
// NZ = n = threads per workgroup (one lattice site along z per thread): 128, 256 or 512
template <typename T, int NZ, bool PSI_ONLY = false>
__global__ void __launch_bounds__(NZ) __attribute__((amdgpu_waves_per_eu(BCHMC_ZBIN_WAVES, BCHMC_ZBIN_WAVES)))
k_zbin_direct(Geo g, PosPar pp, SphPar sp, TilePar tp, int log2n, const C2<T> *__restrict__ twiddle,
              const C2<T> *__restrict__ ck, int *__restrict__ cnt, int *__restrict__ ovf, RecQuad *__restrict__ srec,
              T *__restrict__ V, double *__restrict__ zero_part, T *__restrict__ rho_zero,
              long long *__restrict__ fix_zero, T *__restrict__ psi_out) {
  constexpr int kSlots = 4 * NZ;  // = particles per workgroup: more distinct counters cannot occur, the probing terminates
  constexpr int kHashShift = 32 - (NZ == 128 ? 9 : (NZ == 256 ? 10 : 11));
  static_assert(NZ == 128 || NZ == 256 || NZ == 512, "k_zbin_direct: one thread per lattice site along z");
  constexpr int KF = 6;         // interleaved transforms: (component, row pair)
  // One LDS area, used twice: the transform tile + twiddles ((6 n + n / 2) complex), then -- after every thread has
  // taken its displacements out of it -- the hash table of the binning (20 KB).  zbin_lds() is its size.
  extern __shared__ __attribute__((aligned(16))) unsigned char s_raw_z[];
  unsigned long long *hcnt = reinterpret_cast<unsigned long long *>(s_raw_z);  // two 32-bit fields: a pair's counters
  unsigned long long *hbase = hcnt + kSlots;
  int *hkey = reinterpret_cast<int *>(hbase + kSlots);
  if (PSI_ONLY && !*ovf) return;
  const int n = g.n, tid = (int)threadIdx.x;
  C2<T> *s = reinterpret_cast<C2<T> *>(s_raw_z);  // n * KF
  C2<T> *tw = s + (size_t)n * KF;                 // n / 2
  if (!PSI_ONLY && zero_part && blockIdx.x == 0)
    for (int i = tid; i < kRedBlocks; i += NZ) zero_part[i] = 0.;
  for (int t = tid; t < n / 2; t += NZ) tw[t] = twiddle[t];
  const int nb = n >> 1;
  const int j0 = 2 * ((int)blockIdx.x % nb), i0 = 2 * ((int)blockIdx.x / nb);
  const int shift = 32 - log2n;
  {
    // fill: wave w takes pair f = w & 1 and the wavenumbers kk = 64 (w >> 1) + lane: the n / 64 waves cover kk < n / 2
    const int w = tid >> 6, f = w & 1;
    const long long row = (long long)g.nhp * (j0 + (long long)n * (i0 + f));  // element (i0 + f, j0, 0)
    {
      const int kk = ((w >> 1) << 6) + (tid & 63);
      C2<T> a[3], b[3], an[3], bn[3];
#pragma unroll
      for (int c = 0; c < 3; c++) {
        a[c] = ck[(long long)c * g.Nhp + row + kk];
        b[c] = ck[(long long)c * g.Nhp + row + g.nhp + kk];
        if (kk == 0) {
          an[c] = ck[(long long)c * g.Nhp + row + n / 2];
          bn[c] = ck[(long long)c * g.Nhp + row + g.nhp + n / 2];
        }
      }
#pragma unroll
      for (int c = 0; c < 3; c++) {
        const int colf = 2 * c + f;
        C2<T> lo, hi;
        if (kk == 0) {
          lo.x = a[c].x; lo.y = b[c].x;      // Z[0]
          hi.x = an[c].x; hi.y = bn[c].x;    // Z[n / 2]
          s[(int)(__brev(0u) >> shift) * KF + colf] = lo;
          s[(int)(__brev((unsigned)(n / 2)) >> shift) * KF + colf] = hi;
        } else {
          lo.x = a[c].x - b[c].y; lo.y = a[c].y + b[c].x;   // A + i B
          hi.x = a[c].x + b[c].y; hi.y = b[c].x - a[c].y;   // conj(A) + i conj(B)
          s[(int)(__brev((unsigned)kk) >> shift) * KF + colf] = lo;
          s[(int)(__brev((unsigned)(n - kk)) >> shift) * KF + colf] = hi;
        }
      }
    }
  }
  __syncthreads();
  xfft_inplace<T>(s, tw, n, log2n, KF, true);
  // ---- thread t: the particles (i0 + f, j0 + e, k = t) ----
  const HomeCell<T> hc = make_home<T>(g);
  long long p[4];
  int key[4], slot[4], local[4], flag[4];
  T x[4], y[4], z[4];
#pragma unroll
  for (int f = 0; f < 2; f++) {
    const C2<T> vx = s[tid * KF + f], vy = s[tid * KF + 2 + f], vz = s[tid * KF + 4 + f];
    x[2 * f] = vx.x; x[2 * f + 1] = vx.y;
    y[2 * f] = vy.x; y[2 * f + 1] = vy.y;
    z[2 * f] = vz.x; z[2 * f + 1] = vz.y;
  }
  if (!PSI_ONLY) {
    __syncthreads();  // the tile is free: it becomes the hash table
    for (int t = tid; t < kSlots; t += NZ) {
      hkey[t] = 0;
      hcnt[t] = 0ull;
    }
    __syncthreads();
  }
#pragma unroll
  for (int m = 0; m < 4; m++) {
    const int li = i0 + (m >> 1), lj = j0 + (m & 1);
    p[m] = tid + (long long)n * (lj + (long long)n * li);
    if (PSI_ONLY || psi_out) {
      psi_out[p[m]] = x[m];
      psi_out[p[m] + g.N] = y[m];
      psi_out[p[m] + 2 * g.N] = z[m];
    }
    if (PSI_ONLY) continue;
    if (rho_zero) rho_zero[p[m]] = T(0);
    if (fix_zero) fix_zero[p[m]] = 0;
    key[m] = -1;
    slot[m] = local[m] = flag[m] = 0;
    particle_pos<T>(pp, li, lj, tid, x[m], y[m], z[m], x[m], y[m], z[m]);
    if (pos_ok(g, x[m], y[m], z[m])) {
      const int t = tile_of_wrapped(tp, wrap_cell(home_cell_i(hc, x[m]), g.n), wrap_cell(home_cell_i(hc, y[m]), g.n),
                                    wrap_cell(home_cell_i(hc, z[m]), g.n));
      flag[m] = in_domain(g, sp, x[m], y[m], z[m]) ? 0 : kSortFlagNoScatter;
      key[m] = t * kOct + subcell_octant<T>(x[m], y[m], z[m], hc.inv_d);
      const int pk = key[m] >> 1;  // the pair of counters (2 pk, 2 pk + 1)
      int sl = (int)(((unsigned)pk * 2654435761u) >> kHashShift) & (kSlots - 1);
      for (;;) {
        const int old = atomicCAS(&hkey[sl], 0, pk + 1);
        if (old == 0 || old == pk + 1) break;
        sl = (sl + 1) & (kSlots - 1);
      }
      slot[m] = sl;
      const int sh = (key[m] & 1) << 5;
      local[m] = (int)((atomicAdd(&hcnt[sl], 1ull << sh) >> sh) & 0xffffffffull);
    } else {
      V[p[m]] = T(0);
      V[p[m] + g.N] = T(0);
      V[p[m] + 2 * g.N] = T(0);
    }
  }
  if (PSI_ONLY) return;
  __syncthreads();
  {
    constexpr int kPer = kSlots / NZ;
    int hk[kPer];
    unsigned long long hb[kPer];
#pragma unroll
    for (int u = 0; u < kPer; u++) {
      const int sl = tid + u * NZ;
      hk[u] = hkey[sl];
      hb[u] = 0ull;
      if (hk[u])
        hb[u] = atomicAdd(reinterpret_cast<unsigned long long *>(cnt) + (hk[u] - 1), hcnt[sl]);
    }
#pragma unroll
    for (int u = 0; u < kPer; u++)
      if (hk[u]) hbase[tid + u * NZ] = hb[u];
  }
  __syncthreads();
  const int seg = tp.cap / kOct;
#pragma unroll
  for (int m = 0; m < 4; m++) {
    if (key[m] < 0) continue;
    const int sh = (key[m] & 1) << 5;
    const int rank = (int)((hbase[slot[m]] >> sh) & 0xffffffffull) + local[m];
    if (rank >= seg) {
      ovf[0] = 1;
      ovf[1] = seg;
    } else {
      const int t = key[m] / kOct;
      const long long dst = (long long)t * tp.cap + (long long)(key[m] - t * kOct) * seg + rank;
      rec_store<T>(srec, dst, x[m], y[m], z[m], (int)p[m] | flag[m]);
    }
  }
}

// ======================================================================================================
// k_zr2c: the row pass of the planes-mode R2C (unnormalised forward real transforms along z of the three components
// of V), the mirror image of k_zbin_direct's transform part -- used where rocFFT's 2-D R2C is the slower pair
// (512^3: its length-512 column kernel runs at 2.3 TB/s; k_zr2c + k_ypass<forward> at 5+).  One workgroup of n
// threads per 2 x 2 x n column: thread t packs the rows (i0 + f, j0) and (i0 + f, j0 + 1) at z = t as a + i b into the
// six interleaved columns of the LDS tile, one radix-4 call, and the half-complex spectra come out of
//   A[k] = (Z[k] + conj(Z[n - k])) / 2,   B[k] = (Z[k] - conj(Z[n - k])) / (2 i),   k = 0 .. n / 2.
// Output layout: element (i, j, k) at k + nhp (j + n i) of each component, like rocFFT's row pass inside the 2-D plan;
// the row padding k > n / 2 is left as it is (columns of the later passes are independent, reductions skip it).
// Replaces the z part of fftR2Cplanned (fftwrapper.cc:104-119) for HMC_models.cc:342-349's three transforms of V.
// ======================================================================================================
template <typename T, int NZ>
__global__ void __launch_bounds__(NZ)
k_zr2c(Geo g, int log2n, const C2<T> *__restrict__ twiddle, const T *__restrict__ V, C2<T> *__restrict__ ck) {
  constexpr int KF = 6;
  extern __shared__ __attribute__((aligned(16))) unsigned char s_raw_zr[];
  const int n = g.n, tid = (int)threadIdx.x;
  C2<T> *s = reinterpret_cast<C2<T> *>(s_raw_zr);  // n * KF
  C2<T> *tw = s + (size_t)n * KF;                 // n / 2
  for (int t = tid; t < n / 2; t += NZ) tw[t] = twiddle[t];
  const int nb = n >> 1;
  const int j0 = 2 * ((int)blockIdx.x % nb), i0 = 2 * ((int)blockIdx.x / nb);
  const int shift = 32 - log2n;
  {
    const int r = (int)(__brev((unsigned)tid) >> shift) * KF;
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
      for (int f = 0; f < 2; f++) {
        const long long p = tid + (long long)n * (j0 + (long long)n * (i0 + f));  // lattice site (i0 + f, j0, t)
        C2<T> v;
        v.x = V[(long long)c * g.N + p];
        v.y = V[(long long)c * g.N + p + n];  // (i0 + f, j0 + 1, t)
        s[r + 2 * c + f] = v;
      }
  }
  __syncthreads();
  xfft_inplace<T>(s, tw, n, log2n, KF, false);
  // wave w: pair f = w & 1, wavenumbers kk = 64 (w >> 1) + lane < n / 2; the threads with kk == 0 also store k = n / 2
  const int w = tid >> 6, f = w & 1, kk = ((w >> 1) << 6) + (tid & 63);
  const long long row = (long long)g.nhp * (j0 + (long long)n * (i0 + f));
#pragma unroll
  for (int c = 0; c < 3; c++) {
    const int colf = 2 * c + f;
    const C2<T> zk = s[kk * KF + colf], zm = s[((n - kk) & (n - 1)) * KF + colf];
    C2<T> a, b;
    a.x = T(0.5) * (zk.x + zm.x);
    a.y = T(0.5) * (zk.y - zm.y);
    b.x = T(0.5) * (zk.y + zm.y);
    b.y = T(-0.5) * (zk.x - zm.x);
    ck[(long long)c * g.Nhp + row + kk] = a;
    ck[(long long)c * g.Nhp + row + g.nhp + kk] = b;
    if (kk == 0) {
      const C2<T> zh = s[(n / 2) * KF + colf];
      C2<T> ah, bh;
      ah.x = zh.x; ah.y = T(0);
      bh.x = zh.y; bh.y = T(0);
      ck[(long long)c * g.Nhp + row + n / 2] = ah;
      ck[(long long)c * g.Nhp + row + g.nhp + n / 2] = bh;
    }
  }
}

// dynamic LDS of k_zbin_direct
template <typename T> inline size_t zbin_lds(int n) {
  const size_t tile = ((size_t)n * 6 + n / 2) * sizeof(C2<T>), hash = (size_t)4 * n * 20;
  return tile > hash ? tile : hash;
}

}  // namespace bchmc
