// tiles.hpp -- Tile-sorted particle-mesh path: binning, scan, LDS scatter / gather kernels.
// Part of the bchmc engine's kernel set; include through kernels.hpp (definition order matters).
#pragma once
#include "common.hpp"

namespace bchmc {

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
  int cap;            // record slots reserved per tile by the one-pass binning (k_bin<DIRECT>), a multiple of 8:
                      // the tile's slots are eight segments of cap / 8, one per sub-cell octant of the particle
};

constexpr int kOct = 8;  // sub-cell octants: (x, y, z) in the upper half of the home cell -> bits 2, 1, 0

// Sub-cell octant of a position: the 64 lanes of a wave should share most of the stencil cells that can pass the
// `r/h <= 2` test, and a wave pays for every candidate ANY of its lanes needs (81 unsorted, ~51 within one octant).
// Ordering only: results do not depend on it.
template <typename T>
__device__ __forceinline__ int subcell_octant(T x, T y, T z, T inv_d) {
  const T fx = x * inv_d, fy = y * inv_d, fz = z * inv_d;
  const int b = ((fx - r_floor(fx) >= T(0.5)) ? 4 : 0) | ((fy - r_floor(fy) >= T(0.5)) ? 2 : 0) |
                ((fz - r_floor(fz) >= T(0.5)) ? 1 : 0);
  // position of b in the 3-bit Gray sequence: consecutive segments are octants that differ along ONE axis, so a wave
  // that straddles a segment boundary (segment lengths are not multiples of 64) pays for the union of two adjacent
  // octants, not of two opposite ones
  return b ^ (b >> 1) ^ (b >> 2);
}

constexpr int kSortFlagNoScatter = 1 << 30;  // record flag: particle fails getDensity_SPH's domain test

// Tile-sorted particle records, array of structures: { x, y, z, original index | flag } = 4 * sizeof(T) bytes (the index
// sits in the low 32 bits of the fourth element).  One record is one 32-byte (fp64) or 16-byte (fp32) store in the
// binning pass -- with four separate arrays the same record was four scattered 8- / 4-byte stores, and the octant-ordered
// slots made the runs of neighbouring particles too short for the L2 to merge them -- and two (one) 16-byte loads in
// the scatter / gather kernels.
using RecQuad = uint4;  // 16 bytes
template <typename T> __device__ __forceinline__ void rec_store(RecQuad *base, long long slot, T x, T y, T z, int idx);
template <> __device__ __forceinline__ void rec_store<double>(RecQuad *base, long long slot, double x, double y, double z,
                                                              int idx) {
  double2 *b = reinterpret_cast<double2 *>(base) + 2 * slot;
  b[0] = make_double2(x, y);
  b[1] = make_double2(z, __longlong_as_double((long long)(unsigned)idx));
}
template <> __device__ __forceinline__ void rec_store<float>(RecQuad *base, long long slot, float x, float y, float z,
                                                             int idx) {
  reinterpret_cast<float4 *>(base)[slot] = make_float4(x, y, z, __int_as_float(idx));
}
template <typename T> __device__ __forceinline__ void rec_load(const RecQuad *base, long long slot, T &x, T &y, T &z, int &idx);
template <> __device__ __forceinline__ void rec_load<double>(const RecQuad *base, long long slot, double &x, double &y,
                                                             double &z, int &idx) {
  const double2 *b = reinterpret_cast<const double2 *>(base) + 2 * slot;
  const double2 a = b[0], c = b[1];
  x = a.x;
  y = a.y;
  z = c.x;
  idx = (int)__double_as_longlong(c.y);
}
template <> __device__ __forceinline__ void rec_load<float>(const RecQuad *base, long long slot, float &x, float &y, float &z,
                                                            int &idx) {
  const float4 v = reinterpret_cast<const float4 *>(base)[slot];
  x = v.x;
  y = v.y;
  z = v.z;
  idx = __float_as_int(v.w);
}
template <typename T> constexpr int rec_quads() { return sizeof(T) == 8 ? 2 : 1; }  // 16-byte units per record

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
// tile sides are powers of two (4, 8 or 16: bchmc_create): shifts, not three integer divisions per particle
__device__ __forceinline__ int tile_of_wrapped(const TilePar &tp, int cx, int cy, int cz) {
  const int sx = __builtin_ctz(tp.tx), sy = __builtin_ctz(tp.ty), sz = __builtin_ctz(tp.tz);
  return (cz >> sz) + tp.ntz * ((cy >> sy) + tp.nty * (cx >> sx));
}

// Workgroup -> particles.  When 16 divides n a workgroup takes a 4 x 4 x 16 brick of the Lagrangian lattice
// (neighbours in space: few distinct tiles per workgroup, 128-byte rows of psi); otherwise 256 consecutive ones.
__device__ __forceinline__ long long brick_particle(const Geo &g, int b, int tid, int &i, int &j, int &k) {
  const int n = g.n;  // tid: 0..255 within the brick
  if ((n & 15) == 0) {
    const int nbz = n >> 4, nby = n >> 2;
    const int bk = b % nbz, bj = (b / nbz) % nby, bi = b / (nbz * nby);
    i = bi * 4 + (tid >> 6);
    j = bj * 4 + ((tid >> 4) & 3);
    k = bk * 16 + (tid & 15);
    return k + (long long)n * (j + (long long)n * i);
  }
  const long long p = b * 256ll + tid;
  k = (int)(p % n);
  const long long ij = p / n;
  j = (int)(ij % n);
  i = (int)(ij / n);
  return p;
}
__device__ __forceinline__ long long brick_particle(const Geo &g, int b, int &i, int &j, int &k) {
  return brick_particle(g, b, (int)threadIdx.x, i, j, k);  // 256-thread workgroups
}

// Binning.  The workgroup first counts its particles per counter (tile, or tile and sub-cell octant) in an LDS hash
// table, then reserves one contiguous rank range per distinct counter with a single global atomic (a handful per
// workgroup instead of one returning atomic per particle on hot counters).  Particles with a non-finite position are
// left out; the gather gives them V = 0.
//
// k_bin_direct (one-pass sort): every (tile, sub-cell octant) owns `tp.cap / 8` record slots, the particle's record
//   (position, index | flag) goes straight to slot tile * cap + octant * cap / 8 + rank: the records of a tile come out
//   ordered by octant, which is the order the scatter / gather kernels want (no sorting prologue in them).  A rank
//   >= cap / 8 raises *ovf and the record is dropped: the two-pass kernels below then redo the sort from scratch (they
//   return at once otherwise).  A workgroup bins kBinPer bricks (an 8 x 8 x 16 block of the Lagrangian lattice) through
//   one hash table: eight counters per tile mean eight times the global atomics per brick, four bricks per table bring
//   them back to about twice the per-tile number (k_bin_direct with one brick per table: 0.32 ms at 256^3, the per-tile
//   binning it replaced: 0.25 ms).
// k_bin (two-pass fallback, pass 1): tile id and arrival rank of every particle to tile_rank.
#ifndef BCHMC_BIN_PER
#define BCHMC_BIN_PER 4
#endif
#ifndef BCHMC_BIN_THREADS
#define BCHMC_BIN_THREADS 256  // kBinPer bricks per hash table are handled by BCHMC_BIN_THREADS / 256 thread groups
#endif
constexpr int kBinPer = BCHMC_BIN_PER;
constexpr int kBinGroups = BCHMC_BIN_THREADS / 256;   // bricks processed side by side
constexpr int kBinLoop = kBinPer / kBinGroups;        // bricks per thread

// brick m of super-brick sb: a PI x PJ group in (i, j) when the bricks are 4 x 4 x 16 (kBinPer = 1, 2 or 4), else
// kBinPer consecutive ones
__device__ __forceinline__ int super_brick(const Geo &g, int sb, int m) {
  static_assert(kBinPer == 1 || kBinPer == 2 || kBinPer == 4, "kBinPer");
  constexpr int PJ = kBinPer >= 2 ? 2 : 1, PI = kBinPer >= 4 ? 2 : 1;
  const int n = g.n;
  if ((n & 15) == 0) {
    const int nbz = n >> 4, nby = n >> 2, nbys = nby / PJ;
    const int sbk = sb % nbz, sbj = (sb / nbz) % nbys, sbi = sb / (nbz * nbys);
    return sbk + nbz * ((PJ * sbj + (m % PJ)) + nby * (PI * sbi + (m / PJ)));
  }
  return sb * kBinPer + m;  // may run past the last brick: brick_particle then returns p >= N
}

// fp64: three workgroups per CU (12 waves) -- measured 6 % faster than the four the register count allows and 15 % faster
// than two (scripts/bin_bench.hip, dynamic-LDS sweep): fewer concurrent record streams merge better in the L2.  fp32
// (16-byte records): four are faster than three (0.228 against 0.252-0.263 ms at 256^3).
#ifndef BCHMC_BIN_WAVES
#define BCHMC_BIN_WAVES (sizeof(T) == 8 ? 3 : 4)
#endif
template <typename T>
__global__ void __launch_bounds__(BCHMC_BIN_THREADS) __attribute__((amdgpu_waves_per_eu(BCHMC_BIN_WAVES, BCHMC_BIN_WAVES)))
k_bin_direct(Geo g, PosPar pp, SphPar sp, TilePar tp, int nsuper, const T *__restrict__ psi, int *__restrict__ cnt,
             int *__restrict__ ovf, RecQuad *__restrict__ srec, T *__restrict__ V, double *__restrict__ zero_part,
             T *__restrict__ rho_zero, long long *__restrict__ fix_zero) {
  constexpr int kSlots = 2048;  // > kBinPer * 256 distinct counters can never occur: the probing always terminates
  __shared__ int hkey[kSlots], hcnt[kSlots], hbase[kSlots];
  // the scatter that follows accumulates sum(rho) into these partials: cleared here instead of by a fill launch
  if (zero_part && blockIdx.x == 0)
    for (int i = threadIdx.x; i < kRedBlocks; i += blockDim.x) zero_part[i] = 0.;
  const HomeCell<T> hc = make_home<T>(g);
  for (int sb = blockIdx.x; sb < nsuper; sb += gridDim.x) {
    for (int s = threadIdx.x; s < kSlots; s += blockDim.x) {
      hkey[s] = 0;
      hcnt[s] = 0;
    }
    __syncthreads();
    long long p[kBinLoop];
    int key[kBinLoop], slot[kBinLoop], local[kBinLoop], flag[kBinLoop];
    int li[kBinLoop], lj[kBinLoop], lk[kBinLoop];
    T x[kBinLoop], y[kBinLoop], z[kBinLoop];
    const int tid = (int)threadIdx.x & 255, grp = (int)threadIdx.x >> 8;
    // all displacement loads of the workgroup's bricks are issued before the first one is used: one memory round trip
    // per super-brick instead of one per brick (the position arithmetic between them is ~500 instructions per brick)
    // lattice coordinates of the super-brick once per workgroup (scalar divisions), its bricks by offsets: flattening
    // each brick's number and dividing it apart again cost ~280 vector instructions per brick
    constexpr int PJ = kBinPer >= 2 ? 2 : 1, PI = kBinPer >= 4 ? 2 : 1;
    const bool lattice = (g.n & 15) == 0;
    int sbi = 0, sbj = 0, sbk = 0;
    if (lattice) {
      const int nbz = g.n >> 4, nbys = (g.n >> 2) / PJ;
      sbk = sb % nbz;
      sbj = (sb / nbz) % nbys;
      sbi = sb / (nbz * nbys);
    }
#pragma unroll
    for (int m = 0; m < kBinLoop; m++) {
      const int mm = m * kBinGroups + grp;
      if (lattice) {
        li[m] = (PI * sbi + mm / PJ) * 4 + (tid >> 6);
        lj[m] = (PJ * sbj + mm % PJ) * 4 + ((tid >> 4) & 3);
        lk[m] = sbk * 16 + (tid & 15);
        p[m] = lk[m] + (long long)g.n * (lj[m] + (long long)g.n * li[m]);
      } else {
        p[m] = brick_particle(g, super_brick(g, sb, mm), tid, li[m], lj[m], lk[m]);
      }
      x[m] = y[m] = z[m] = T(0);
      if (p[m] >= g.N) continue;
      x[m] = psi[p[m]];
      y[m] = psi[p[m] + g.N];
      z[m] = psi[p[m] + 2 * g.N];
      // the bricks visit every lattice index once: clear the density the scatter accumulates into (no fill launch)
      if (rho_zero) rho_zero[p[m]] = T(0);
      if (fix_zero) fix_zero[p[m]] = 0;
    }
#pragma unroll
    for (int m = 0; m < kBinLoop; m++) {
      key[m] = -1;
      slot[m] = local[m] = flag[m] = 0;
      if (p[m] >= g.N) continue;
      particle_pos<T>(pp, li[m], lj[m], lk[m], x[m], y[m], z[m], x[m], y[m], z[m]);
      if (pos_ok(g, x[m], y[m], z[m])) {
        const int t = tile_of_wrapped(tp, wrap_cell(home_cell_i(hc, x[m]), g.n), wrap_cell(home_cell_i(hc, y[m]), g.n),
                                      wrap_cell(home_cell_i(hc, z[m]), g.n));
        flag[m] = in_domain(g, sp, x[m], y[m], z[m]) ? 0 : kSortFlagNoScatter;
        key[m] = t * kOct + subcell_octant<T>(x[m], y[m], z[m], hc.inv_d);
        int sl = (int)(((unsigned)key[m] * 2654435761u) >> 21) & (kSlots - 1);
        for (;;) {
          const int old = atomicCAS(&hkey[sl], 0, key[m] + 1);
          if (old == 0 || old == key[m] + 1) break;
          sl = (sl + 1) & (kSlots - 1);
        }
        slot[m] = sl;
        local[m] = atomicAdd(&hcnt[sl], 1);
      } else {
        V[p[m]] = T(0);
        V[p[m] + g.N] = T(0);
        V[p[m] + 2 * g.N] = T(0);
      }
    }
    __syncthreads();
    {
      // the returning atomics of a thread's occupied counters are all in flight before the first result is used
      constexpr int kPer = kSlots / BCHMC_BIN_THREADS;
      int hk[kPer], hb[kPer];
#pragma unroll
      for (int u = 0; u < kPer; u++) {
        const int s = (int)threadIdx.x + u * BCHMC_BIN_THREADS;
        hk[u] = hkey[s];
        hb[u] = 0;
        if (hk[u]) hb[u] = atomicAdd(&cnt[hk[u] - 1], hcnt[s]);
      }
#pragma unroll
      for (int u = 0; u < kPer; u++)
        if (hk[u]) hbase[(int)threadIdx.x + u * BCHMC_BIN_THREADS] = hb[u];
    }
    __syncthreads();
    const int seg = tp.cap / kOct;
#pragma unroll
    for (int m = 0; m < kBinLoop; m++) {
      if (key[m] < 0) continue;
      const int rank = hbase[slot[m]] + local[m];
      if (rank >= seg) {
        ovf[0] = 1;    // benign race: every writer stores 1
        ovf[1] = seg;  // sticky copy for the host, stamped with the segment size that was too small (every writer of
                       // a launch stores the same value; a stamp below the host's current size is a stale one)
      } else {
        const int t = key[m] / kOct;
        const long long dst = (long long)t * tp.cap + (long long)(key[m] - t * kOct) * seg + rank;
        rec_store<T>(srec, dst, x[m], y[m], z[m], (int)p[m] | flag[m]);
      }
    }
    __syncthreads();  // the hash table is reused by the next super-brick
  }
}

template <typename T>
__global__ void __launch_bounds__(256)
k_bin(Geo g, PosPar pp, SphPar sp, TilePar tp, int nbricks, const T *__restrict__ psi, int *__restrict__ cnt,
      const int *__restrict__ ovf, int2 *__restrict__ tile_rank, T *__restrict__ V) {
  constexpr int kSlots = 512;
  __shared__ int hkey[kSlots], hcnt[kSlots], hbase[kSlots];
  if (!*ovf) return;  // the one-pass binning succeeded
  // a small grid strides over the bricks (it usually returns above)
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
    if (live) {
      T x, y, z;
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
    if (live) tile_rank[p] = (t < 0) ? make_int2(-1, 0) : make_int2(t, (hbase[slot] + local) | flag);
    __syncthreads();  // the hash table is reused by the next brick
  }
}

// One workgroup: record range [off, tend) of every tile -- fixed slots after a successful one-pass binning, an
// exclusive scan of the fallback's counts otherwise -- and the exclusive scan of the per-tile chunk counts
// (-> work-item offsets, ntiles + 1 entries).
// Exclusive scans of two values per thread over a 1024-thread workgroup: wave shuffles, one LDS hop for the 16
// wave totals, two barriers.  wtot: 16 int2 of LDS.
__device__ __forceinline__ int2 block_exclusive_scan2_1024(int a, int b, int2 *wtot) {
  const int lane = threadIdx.x & (kWave - 1), w = threadIdx.x / kWave;
  int ia = a, ib = b;
#pragma unroll
  for (int off = 1; off < kWave; off <<= 1) {
    const int ta = __shfl_up(ia, off, kWave), tb = __shfl_up(ib, off, kWave);
    if (lane >= off) {
      ia += ta;
      ib += tb;
    }
  }
  if (lane == kWave - 1) wtot[w] = make_int2(ia, ib);
  __syncthreads();
  if (w == 0) {
    int2 v = lane < 16 ? wtot[lane] : make_int2(0, 0);
    const int2 own = v;
#pragma unroll
    for (int off = 1; off < 16; off <<= 1) {
      const int ta = __shfl_up(v.x, off, kWave), tb = __shfl_up(v.y, off, kWave);
      if (lane >= off) {
        v.x += ta;
        v.y += tb;
      }
    }
    if (lane < 16) wtot[lane] = make_int2(v.x - own.x, v.y - own.y);
  }
  __syncthreads();
  const int2 base = wtot[w];
  return make_int2(base.x + ia - a, base.y + ib - b);
}

// cnt_direct holds 8 counters per tile (one per octant); `oct` receives, per tile, the exclusive prefix of those eight
// (logical record index where each octant's segment starts) -- all zero after a fallback sort (records contiguous).
__global__ void __launch_bounds__(1024)
k_scan_tiles(TilePar tp, const int *__restrict__ cnt_direct, const int *__restrict__ cnt_fallback,
             const int *__restrict__ ovf, long long *__restrict__ off, long long *__restrict__ tend,
             int *__restrict__ woff, int4 *__restrict__ oct, int *__restrict__ seg_out, int *__restrict__ max_out) {
  // One tile per thread, ceil(ntiles / 1024) workgroups.  A workgroup first sums the counts of all tiles before its
  // own range (coalesced reads, at most 4 * ntiles bytes), then scans its 1024 tiles: every load and store is
  // coalesced, which a single workgroup striding over all tiles was not (33 us for 16384 tiles).
  __shared__ int2 wtot[16];
  __shared__ int2 s_base;
  __shared__ int s_pop_max;
  const bool direct = !*ovf;
  const int T = tp.ntiles, tid = threadIdx.x;
  if (tid == 0) s_pop_max = 0;  // read after the barriers of the scans below
  int pop_max = 0;
  // record layout of this sort for the tile kernels (they must not read *ovf: the scatter clears it for the next
  // force evaluation while the gather still needs the layout): slots per octant segment, 0 = contiguous records
  if (blockIdx.x == 0 && tid == 0) *seg_out = direct ? tp.cap / kOct : 0;
  const int4 *cnt8 = reinterpret_cast<const int4 *>(cnt_direct);
  auto count_of = [&](int t) {
    if (!direct) return cnt_fallback[t];
    const int4 a = cnt8[2 * t], b = cnt8[2 * t + 1];
    return (a.x + a.y) + (a.z + a.w) + (b.x + b.y) + (b.z + b.w);
  };
  // chunk is a power of two unless overridden for experiments: shift instead of a runtime division per tile
  const int sh = (tp.chunk & (tp.chunk - 1)) == 0 ? __ffs(tp.chunk) - 1 : -1;
  auto items = [&](int c) { return sh >= 0 ? (c + tp.chunk - 1) >> sh : (c + tp.chunk - 1) / tp.chunk; };
  const int first = blockIdx.x * 1024;
  int pa = 0, pb = 0;
  for (int t = tid; t < first; t += 1024) {
    const int c = count_of(t);
    pa += c;
    pb += items(c);
  }
  // workgroup sum of (pa, pb): inclusive scan value of the last thread
  const int2 pex = block_exclusive_scan2_1024(pa, pb, wtot);
  if (tid == 1023) s_base = make_int2(pex.x + pa, pex.y + pb);
  __syncthreads();
  const int2 base = s_base;
  const int t = first + tid;
  const int c = t < T ? count_of(t) : 0;
  const int2 ex = block_exclusive_scan2_1024(c, items(c), wtot);
  const int ea = base.x + ex.x, eb = base.y + ex.y;
  if (t < T) {
    const long long o = direct ? (long long)t * tp.cap : (long long)ea;  // one-pass: fixed slots per tile; fallback: packed
    off[t] = o;
    tend[t] = o + c;
    woff[t] = eb;
    if (t == T - 1) woff[T] = eb + items(c);
    int4 lo = make_int4(0, 0, 0, 0), hi = make_int4(0, 0, 0, 0);
    {
      // largest (tile, octant) population of this binning -- the counters keep counting past a full segment, so the
      // figure is exact also when it overflowed: what the host sizes the segments from (1.5x, at its next read-back)
      const int4 a = cnt8[2 * t], b = cnt8[2 * t + 1];
      pop_max = max(max(max(a.x, a.y), max(a.z, a.w)), max(max(b.x, b.y), max(b.z, b.w)));
    }
    if (direct) {
      const int4 a = cnt8[2 * t], b = cnt8[2 * t + 1];
      lo = make_int4(0, a.x, a.x + a.y, a.x + a.y + a.z);
      const int h0 = lo.w + a.w;
      hi = make_int4(h0, h0 + b.x, h0 + b.x + b.y, h0 + b.x + b.y + b.z);
    }
    oct[2 * t] = lo;
    oct[2 * t + 1] = hi;
  }
  // one atomic per workgroup on the host-read word, not one per tile
#pragma unroll
  for (int o = kWave / 2; o > 0; o >>= 1) pop_max = max(pop_max, __shfl_down(pop_max, o, kWave));
  if ((tid & (kWave - 1)) == 0 && pop_max > 0) atomicMax(&s_pop_max, pop_max);
  __syncthreads();
  if (tid == 0 && s_pop_max > 0) atomicMax(max_out, s_pop_max);
}

// Fallback pass 3: write each particle's record (position, original index | flag) to its sorted slot.
template <typename T>
__global__ void __launch_bounds__(256)
k_reorder(Geo g, PosPar pp, int nbricks, const T *__restrict__ psi, const int2 *__restrict__ tile_rank,
          const long long *__restrict__ off, const int *__restrict__ ovf, RecQuad *__restrict__ srec) {
  if (!*ovf) return;  // the one-pass binning succeeded
  for (int brick = blockIdx.x; brick < nbricks; brick += gridDim.x) {
    int i, j, k;
    const long long p = brick_particle(g, brick, i, j, k);
    if (p >= g.N) continue;
    const int2 tr = tile_rank[p];
    if (tr.x < 0) continue;
    T x, y, z;
    particle_pos<T>(pp, i, j, k, psi[p], psi[p + g.N], psi[p + 2 * g.N], x, y, z);
    const long long slot = off[tr.x] + (tr.y & ~kSortFlagNoScatter);
    rec_store<T>(srec, slot, x, y, z, (int)p | (tr.y & kSortFlagNoScatter));
  }
}

// Logical record index of a tile -> slot relative to the tile's first slot.  After the one-pass binning the records
// sit in eight octant segments of `seg` slots; P[o] = logical index of the first record of octant o.  After a
// fallback sort they are contiguous (seg = 0: identity).
struct OctMap {
  int seg;
  int P[kOct];
  __device__ __forceinline__ int slot(int s) const {
    if (seg == 0) return s;
    int o = 0, p0 = 0;
#pragma unroll
    for (int m = 1; m < kOct; m++)
      if (s >= P[m]) {
        o = m;
        p0 = P[m];
      }
    return o * seg + (s - p0);
  }
};

// Work item -> (tile, particle range).  Returns false when this workgroup has nothing to do.
__device__ __forceinline__ bool tile_work(int w, const TilePar &tp, const long long *__restrict__ off,
                                          const long long *__restrict__ tend, const int *__restrict__ woff,
                                          const int4 *__restrict__ oct, const int *__restrict__ seg_in, int &tile,
                                          long long &base, int &p_begin, int &p_end, OctMap &om) {
  // base = first record slot of the tile (64-bit: ntiles * cap may exceed 2^31), [p_begin, p_end) relative to it
  __shared__ int s_tile, s_b, s_e;
  __shared__ long long s_base;
  __shared__ int s_oct[kOct + 1];
  if (threadIdx.x == 0) {
    int t = -1, b = 0, e = 0;
    long long o = 0;
    if (w < woff[tp.ntiles]) {
      int lo = 0, hi = tp.ntiles;  // last t with woff[t] <= w
      while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (woff[mid] <= w) lo = mid; else hi = mid;
      }
      t = lo;
      o = off[t];
      b = (w - woff[t]) * tp.chunk;
      e = min(b + tp.chunk, (int)(tend[t] - o));
    }
    s_tile = t;
    s_b = b;
    s_e = e;
    s_base = o;
    s_oct[kOct] = *seg_in;  // written by k_scan_tiles for this sort
    if (t >= 0) {
      const int4 lo = oct[2 * t], hi = oct[2 * t + 1];
      s_oct[0] = lo.x; s_oct[1] = lo.y; s_oct[2] = lo.z; s_oct[3] = lo.w;
      s_oct[4] = hi.x; s_oct[5] = hi.y; s_oct[6] = hi.z; s_oct[7] = hi.w;
    }
  }
  __syncthreads();
  tile = s_tile;
  p_begin = s_b;
  p_end = s_e;
  base = s_base;
  // uniform over the workgroup: keep the map in scalar registers
  om.seg = __builtin_amdgcn_readfirstlane(s_oct[kOct]);
#pragma unroll
  for (int m = 0; m < kOct; m++) om.P[m] = __builtin_amdgcn_readfirstlane(s_oct[m]);
  return tile >= 0;
}

// Order one work item's records by the sub-cell position of the particle (in place; the gather reads the same
// order): `bits` binary digits of the fractional cell coordinate per axis, octant digits most significant.  The 64
// lanes of a wave then share most of the stencil cells that can pass the `r/h <= 2` test, and a wave pays for
// every candidate ANY of its lanes needs (81 unsorted, ~51 with octants, fewer with 4 x 4 x 4 bins).  Pure
// reordering: results do not depend on it.  Requires chunk <= 256 * 8 and blockDim.x == 256.
template <typename T>
__device__ __forceinline__ void subsort_subcell(int bits, int pb, int pe, T inv_d, RecQuad *srec) {
  constexpr int kPer = 8;  // tp.chunk <= 256 * kPer
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
      rec_load<T>(srec, s, rx[m], ry[m], rz[m], id[m]);
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
      rec_store<T>(srec, dst, rx[m], ry[m], rz[m], id[m]);
    }
  }
  __threadfence_block();
}

// After a FALLBACK sort the records of a tile are contiguous but in arrival order: this pass orders every work item's
// records by sub-cell bin for the scatter and the gather (the one-pass binning delivers octant order by construction, so
// the kernel returns at once then; a separate kernel so that the scatter does not carry its 70 record registers).
template <typename T>
__global__ void __launch_bounds__(256)
k_subsort(Geo g, TilePar tp, int bits, RecQuad *srec_all, const long long *__restrict__ off,
          const long long *__restrict__ tend, const int *__restrict__ woff, const int4 *__restrict__ oct,
          const int *__restrict__ seg_in) {
  if (*seg_in != 0 || bits <= 0) return;
  const int nitems = woff[tp.ntiles];
  const HomeCell<T> hc = make_home<T>(g);
  for (int w = blockIdx.x; w < nitems; w += gridDim.x) {
    int tile, pb, pe;
    long long rec0;
    OctMap om;
    if (tile_work(w, tp, off, tend, woff, oct, seg_in, tile, rec0, pb, pe, om))
      subsort_subcell<T>(bits, pb, pe, hc.inv_d, srec_all + rec0 * rec_quads<T>());
    __syncthreads();  // tile_work's and the sort's LDS are reused by the next work item
  }
}

// One LDS cell of a work item into the global density; returns what the default mode adds to its running sum(rho).
template <typename T>
__device__ __forceinline__ double flush_cell(T *dst, double v) {
  atomic_add_r(dst, (T)v);
  return (double)(T)v;
}
__device__ __forceinline__ double flush_cell(long long *dst, long long v) {
  atomicAdd(reinterpret_cast<unsigned long long *>(dst), (unsigned long long)v);
  return 0.;
}

// getDensity_SPH on sorted particles: LDS accumulation per (tile, chunk), one flush.
template <typename T, bool FIX>
__global__ void __launch_bounds__(256)
k_scatter_tile(Geo g, SphPar sp, TilePar tp, const int4 *__restrict__ cols, int ncol, const RecQuad *__restrict__ srec,
               const long long *__restrict__ off, const long long *__restrict__ tend, const int *__restrict__ woff,
               const int4 *__restrict__ oct, const int *__restrict__ seg_in,
               typename Cell<FIX, T>::type *__restrict__ rho, double *__restrict__ rho_part, double fix_scale) {
  extern __shared__ __attribute__((aligned(16))) unsigned char s_raw_scatter[];
  // accumulate in double also for float fields (fixed point in deterministic mode)
  using Acc = typename Cell<FIX, double>::type;
  Acc *s_tile_acc = reinterpret_cast<Acc *>(s_raw_scatter);
  int tile, pb, pe;
  long long rec0;
  OctMap om;
  if (!tile_work((int)blockIdx.x, tp, off, tend, woff, oct, seg_in, tile, rec0, pb, pe, om)) return;
  srec += rec0 * rec_quads<T>();  // this tile's record slots; pb, pe are relative to them
  const int ncell = tp.lx * tp.ly * tp.lz;
  int4 *s_cols = reinterpret_cast<int4 *>(s_raw_scatter + (((size_t)ncell * sizeof(double) + 15) & ~(size_t)15));
  for (int m = threadIdx.x; m < ncol; m += blockDim.x) s_cols[m] = cols[m];
  for (int c = threadIdx.x; c < ncell; c += blockDim.x) s_tile_acc[c] = Acc(0);
  const T d = (T)g.d;
  const HomeCell<T> hc = make_home<T>(g);
  __syncthreads();
  const int tzi = tile % tp.ntz, tyi = (tile / tp.ntz) % tp.nty, txi = tile / (tp.ntz * tp.nty);
  const int ox = txi * tp.tx - tp.R, oy = tyi * tp.ty - tp.R, oz = tzi * tp.tz - tp.R;  // global cell of LDS (0,0,0)
  const int n = g.n, R = sp.reach;
  const T r2_lim = (T)sp.r2_lim, h_inv = (T)sp.h_inv, w_norm = (T)sp.w_norm;
  for (int s = pb + threadIdx.x; s < pe; s += blockDim.x) {
    T x, y, z;
    int rid;
    rec_load<T>(srec, om.slot(s), x, y, z, rid);
    if (rid & kSortFlagNoScatter) continue;
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
        Acc *row = s_tile_acc + tp.lz * ((hy + c.y) + tp.ly * (hx + c.x)) + hz;
        for (int i3 = c.z; i3 <= c.w; ++i3) {
          const T dz = z - (ccz + (T)i3 * d);
          const T r2 = r2ab + dz * dz;
          if (r2 <= r2_lim) {
            const T q = (r2 * fast_rsqrt(r2 + tiny_pos<T>())) * h_inv;
            if (q <= T(2)) cell_add(row + i3, (double)sph_w_folded<T>(q, w_norm), fix_scale);
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
          Acc *row = s_tile_acc + tp.lz * ((hy + i2) + tp.ly * (hx + i1)) + hz;
          for (int i3 = -R; i3 <= R; ++i3) {
            const T dz = z - (ccz + (T)i3 * d);
            const T r2 = r2ab + dz * dz;
            if (r2 > r2_lim) continue;
            const T q = (r2 * fast_rsqrt(r2 + tiny_pos<T>())) * h_inv;
            if (q <= T(2)) cell_add(row + i3, (double)sph_w_folded<T>(q, w_norm), fix_scale);
          }
        }
      }
    }
  }
  __syncthreads();
  double flushed = 0.;  // sum of everything this work item adds to rho: the mean density needs no pass over rho
  for (int c = threadIdx.x; c < ncell; c += blockDim.x) {
    const Acc v = s_tile_acc[c];
    if (v != Acc(0)) {
      const int cz = c % tp.lz, cy = (c / tp.lz) % tp.ly, cx = c / (tp.lz * tp.ly);
      const int gx = (ox + cx + n) % n, gy = (oy + cy + n) % n, gz = (oz + cz + n) % n;
      flushed += flush_cell(rho + gz + (long long)n * (gy + (long long)n * gx), v);
    }
  }
  if (!FIX) {  // deterministic mode: k_fix_to_rho sums the converted field in a fixed order instead
    __shared__ double s_red_flush[4];
    flushed = block_sum(flushed, s_red_flush);
    if (threadIdx.x == 0 && flushed != 0.) atomic_add_r(rho_part + (blockIdx.x & (kRedBlocks - 1)), flushed);
  }
}

// likelihood_calc_V_SPH on sorted particles: part_like tile + halo staged in LDS.
template <typename T>
__global__ void __launch_bounds__(256)
k_gather_tile(Geo g, HullPar hp, TilePar tp, int rsd, const RecQuad *__restrict__ srec,
              const long long *__restrict__ off,
              const long long *__restrict__ tend, const int *__restrict__ woff, const int4 *__restrict__ oct,
              const int *__restrict__ seg_in, const T *__restrict__ plike, T *__restrict__ V) {
  extern __shared__ __attribute__((aligned(16))) unsigned char s_raw_gather[];
  T *s_tile_pl = reinterpret_cast<T *>(s_raw_gather);
  int tile, pb, pe;
  long long rec0;
  OctMap om;
  if (!tile_work((int)blockIdx.x, tp, off, tend, woff, oct, seg_in, tile, rec0, pb, pe, om)) return;
  srec += rec0 * rec_quads<T>();  // this tile's record slots; pb, pe are relative to them
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
    T px, py, pz;
    int rid;
    rec_load<T>(srec, om.slot(s), px, py, pz, rid);
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
    const long long p = rid & ~kSortFlagNoScatter;
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

// Register budgets (waves per SIMD the compiler must leave room for), A/B'd at 256^3 fp64 on one box
// (gpurun_out/ab*.json, r02): scatter 4 / 5 / 6 waves -> 1.042 / 0.997 / 0.983 ms per launch (6 waves = 80 VGPRs spills
// a few of the sub-cell sort's record registers, and still wins: the kernel is stall-bound at 71 % VALU busy);
// gather 5 / 6 -> 0.966 / 1.12 ms (its spills land in the candidate loop).
#ifndef BCHMC_SCATTER_WAVES
#define BCHMC_SCATTER_WAVES 6
#endif
#ifndef BCHMC_GATHER_LEAN
#define BCHMC_GATHER_LEAN 1
#endif
#ifndef BCHMC_SCATTER_NOCLAMP
#define BCHMC_SCATTER_NOCLAMP 0
#endif
#ifndef BCHMC_GATHER_WAVES
#define BCHMC_GATHER_WAVES 5
#endif
// STAGE: the work item does not add its LDS image to the global density with atomics but writes the whole image --
// 12 x 12 x 20 doubles, halo included -- with plain coalesced stores to slot blockIdx.x of a staging area;
// k_stage_combine81 below sums, per cell, the <= 8 images that cover it.  rho then needs no clearing, and the stores
// overlap with the arithmetic of the other workgroups on the CU where the 2880 fp64 atomics per work item did not
// (bound measured first, profiles/r03_ab_levers.txt: scatter 0.94 -> 0.81 ms per launch in the HIP-event profile).
template <typename T, int LY, int LZ, bool FIX, bool STAGE = false>
__global__ void __launch_bounds__(256, BCHMC_SCATTER_WAVES)
k_scatter_tile81(Geo g, SphPar sp, TilePar tp, const RecQuad *__restrict__ srec,
                 const long long *__restrict__ off, const long long *__restrict__ tend, const int *__restrict__ woff,
                 const int4 *__restrict__ oct, const int *__restrict__ seg_in,
                 typename Cell<FIX, T>::type *__restrict__ rho, double *__restrict__ rho_part, int *__restrict__ cnt_zero,
                 int ncnt_zero, double fix_scale, double *__restrict__ stage = nullptr,
                 const unsigned short *__restrict__ stage_inv = nullptr) {
  extern __shared__ __attribute__((aligned(16))) unsigned char s_raw_scatter81[];
  using Acc = typename Cell<FIX, double>::type;
  Acc *s_tile_acc = reinterpret_cast<Acc *>(s_raw_scatter81);
  // The binning counters (8 + 1 per tile, and the overflow flag) have been consumed by k_scan_tiles / k_reorder:
  // clear them for the next force evaluation's k_bin here, nine per workgroup, instead of with a fill launch
  // (grid >= ntiles + 1).
  if (threadIdx.x < kOct + 1) {
    const int i = (kOct + 1) * (int)blockIdx.x + (int)threadIdx.x;
    if (i < ncnt_zero) cnt_zero[i] = 0;
  }
  int tile, pb, pe;
  long long rec0;
  OctMap om;
  if (!tile_work((int)blockIdx.x, tp, off, tend, woff, oct, seg_in, tile, rec0, pb, pe, om)) return;
  srec += rec0 * rec_quads<T>();  // this tile's record slots; pb, pe are relative to them
  const int ncell = tp.lx * LY * LZ;
  for (int c = threadIdx.x; c < ncell; c += blockDim.x) s_tile_acc[c] = Acc(0);
  const T d = (T)g.d;
  const HomeCell<T> hc = make_home<T>(g);
  __syncthreads();
  const int tzi = tile % tp.ntz, tyi = (tile / tp.ntz) % tp.nty, txi = tile / (tp.ntz * tp.nty);
  const int ox = txi * tp.tx - 2, oy = tyi * tp.ty - 2, oz = tzi * tp.tz - 2;  // global cell of LDS (0,0,0)
  const int n = g.n;
  // Distances in units of h: r^2/h^2 = X[a] + Y[b] + Z[c] IS q^2 (no scaling per candidate), the test r/h <= 2 is
  // q^2 <= 4 (1 + 1e-12) followed by the clamp below, W's inner branch takes q^2 from the sum it already has.
  const T h_inv = (T)sp.h_inv, w_norm = (T)sp.w_norm, d_h = d * h_inv;
  const T q2_lim = (T)(sp.r2_lim * sp.h_inv * sp.h_inv);
  const T c34w = T(0.75) * w_norm, c32w = T(-1.5) * w_norm, c14w = T(0.25) * w_norm;
  for (int s = pb + threadIdx.x; s < pe; s += blockDim.x) {
    T x, y, z;
    int rid;
    rec_load<T>(srec, om.slot(s), x, y, z, rid);
    if (rid & kSortFlagNoScatter) continue;
    const int ix = home_cell_i(hc, x), iy = home_cell_i(hc, y), iz = home_cell_i(hc, z);
    const T ccx = ((T)ix + T(0.5)) * d, ccy = ((T)iy + T(0.5)) * d, ccz = ((T)iz + T(0.5)) * d;
    const int hx = wrap_cell(ix, n) - ox, hy = wrap_cell(iy, n) - oy, hz = wrap_cell(iz, n) - oz;  // home cell in LDS coords
    if ((unsigned)(hx - 2) >= (unsigned)tp.tx || (unsigned)(hy - 2) >= (unsigned)tp.ty ||
        (unsigned)(hz - 2) >= (unsigned)tp.tz)
      continue;  // cannot happen (binning and this kernel see the same stored position); keeps LDS indexing safe
    const T ux = (x - ccx) * h_inv, uy = (y - ccy) * h_inv, uz = (z - ccz) * h_inv;  // offset from the home centre
    T X[5], Y[5], Z[5];
#pragma unroll
    for (int a = 0; a < 5; a++) {
      const T dx = ux - (T)(a - 2) * d_h, dy = uy - (T)(a - 2) * d_h, dz = uz - (T)(a - 2) * d_h;
      X[a] = r_fma(dx, dx, tiny_pos<T>());  // keeps q^2 > 0 for a particle exactly on a cell centre (rsq(0) = inf)
      Y[a] = dy * dy;
      Z[a] = dz * dz;
    }
    Acc *corner = s_tile_acc + LZ * ((hy - 2) + LY * (hx - 2)) + (hz - 2);
#pragma unroll
    for (int a = 0; a < 5; a++) {
#pragma unroll
      for (int b = 0; b < 5; b++) {
        const int zw = hull81_zw(a, b);  // folds after unrolling
        if (zw < 0) continue;
        const T q2ab = X[a] + Y[b];
        if (q2ab > q2_lim) continue;
        Acc *row = corner + LZ * (b + LY * a);
#pragma unroll
        for (int c = 0; c < 5; c++) {
          if (c < 2 - zw || c > 2 + zw) continue;
          const T q2 = q2ab + Z[c];
          if (q2 <= q2_lim) {
            T rq;
            const T q = sqrt_rsq(q2, rq);
            // SPH_kernel_3D (massFunctions.cc:366-384): w (1 - 3/2 q^2 + 3/4 q^3) for q <= 1, w/4 (2 - q)^3 up to
            // q = 2; a q that rounding left a hair above 2 contributes exactly 0, like the reference's `r/h <= 2`.
            // Which branch a candidate can take is known when the loops unroll (h = d, offset from the home centre
            // within half a cell): a cell two away along any axis is at q >= 1.5 -- outer branch only; the home cell
            // is at q <= 0.87 -- inner branch only; only the other 26 need both and the select.
            const bool far = (a == 0 || a == 4 || b == 0 || b == 4 || c == 0 || c == 4);
            const bool home = (a == 2 && b == 2 && c == 2);
            T val;
            if (home) {
              val = r_fma(q2, r_fma(c34w, q, c32w), w_norm);
            } else {
#if BCHMC_SCATTER_NOCLAMP
              // no clamp at q = 2: a q that the test above admitted a hair beyond 2 (q2_lim = 4 (1 + 1e-12)) deposits
              // -(q - 2)^3 w / 4 >= -1e-37 w instead of exactly 0 -- below the resolution of every sum it enters, and
              // where it is the only contribution to a cell, a negative density takes the same `Lambda > 0` / `dens > 0`
              // branch of the likelihood partial as the reference's 0 (forward_model.hpp: partial_like_value)
              const T t = T(2) - q;
#else
              const T t = r_max(T(2) - q, T(0));
#endif
              const T outer = (c14w * t) * (t * t);
              val = far ? outer : ((q2 <= T(1)) ? r_fma(q2, r_fma(c34w, q, c32w), w_norm) : outer);
            }
            cell_add(row + c, (double)val, fix_scale);
          }
        }
      }
    }
  }
  __syncthreads();
  double flushed = 0.;  // sum of everything this work item adds to rho: the mean density needs no pass over rho
  if (STAGE && !FIX) {
    // staged image in OWNER-BLOCKED order: the cells that belong to each of the 27 tiles around (and including) this
    // one are contiguous, so that the owner's combine pass reads whole runs (stage_inv: position -> LDS image cell)
    double *st = stage + (size_t)blockIdx.x * (size_t)ncell;
    const double *img = reinterpret_cast<const double *>(s_tile_acc);
    for (int c = threadIdx.x; c < ncell; c += blockDim.x) {
      const double v = img[stage_inv[c]];
      st[c] = v;
      flushed += v;
    }
  } else {
    for (int c = threadIdx.x; c < ncell; c += blockDim.x) {
      const Acc v = s_tile_acc[c];
      if (v != Acc(0)) {
        const int cz = c % LZ, cy = (c / LZ) % LY, cx = c / (LZ * LY);
        const int gx = (ox + cx + n) % n, gy = (oy + cy + n) % n, gz = (oz + cz + n) % n;
        flushed += flush_cell(rho + gz + (long long)n * (gy + (long long)n * gx), v);
      }
    }
  }
  if (!FIX) {  // deterministic mode: k_fix_to_rho sums the converted field in a fixed order instead
    __shared__ double s_red_flush[4];
    flushed = block_sum(flushed, s_red_flush);
    if (threadIdx.x == 0 && flushed != 0.) atomic_add_r(rho_part + (blockIdx.x & (kRedBlocks - 1)), flushed);
  }
}

// Sum of the staged images (k_scatter_tile81<STAGE>) into rho.  Every cell of a 12 x 12 x 20 image belongs to exactly one
// tile: the image's own tile (8 x 8 x 16 cells) or one of its 26 neighbours (the halo); the scatter stores an image
// blocked by owner.  Seen from a tile, its 1024 cells receive 2880 staged cells per work item of the 27 tiles around
// it -- the same 2880 (neighbour, staged position, own cell) triples for every tile, so they are a table (built once
// on the host, ordered by staged position: a neighbour's contribution is ONE contiguous run, 8 KB from the tile's own
// image down to 64 bytes from a corner neighbour, and consecutive threads read consecutive doubles).  One workgroup per tile: every thread owns
// 12 table entries, issues its loads for all of them (one per work item of the neighbour: usually one) before it adds
// anything -- the first version of this kernel walked the <= 8 images per cell in nested loops with one load in flight
// per thread and took 0.46 ms at 256^3; this one is bound by the 0.4 + 0.4 GB it moves -- and accumulates in an LDS
// image of the tile (ds_add_f64: up to 8 entries meet in a cell).
// LIKE: overdens + the per-cell likelihood partial in the same pass (what k_partial_like does from rho), so the
// density makes one trip; rho is still written (energies, fetch and the calc_h variants read it).
constexpr int kStageCells = 12 * 12 * 20;
constexpr int kStagePairs = kStageCells / 2;          // table entries: pairs of z-adjacent cells (16-byte loads)
constexpr int kStagePer = (kStagePairs + 255) / 256;  // table entries per thread
// entry: neighbour (0..26) | first own cell of the pair (0..1022, even) << 5 | staged position (0..2878, even) << 15
__host__ __device__ constexpr unsigned stage_entry(int nb, int own, int pos) {
  return (unsigned)nb | ((unsigned)own << 5) | ((unsigned)pos << 15);
}

// WRHO = false: the density itself is not stored (interior steps of a trajectory: only the likelihood partial is read).
template <typename T, bool LIKE, bool WRHO>
__global__ void __launch_bounds__(256)
k_stage_combine81(Geo g, TilePar tp, LikePar lp, const double *__restrict__ stage, const unsigned *__restrict__ table,
                  const int *__restrict__ woff, T *__restrict__ rho, const double *__restrict__ rho_partials,
                  const T *__restrict__ nobs, const T *__restrict__ noise, const T *__restrict__ window,
                  T *__restrict__ plike) {
  __shared__ int s_first[2][32], s_cnt[2][32], s_maxcnt[2];
  __shared__ __attribute__((aligned(16))) double acc[2][1024];
  __shared__ double red[4];
  double nmean = 1.;
  if (LIKE) nmean = sum_partials(rho_partials, red) / (double)g.N;
  unsigned ent[kStagePer];
#pragma unroll
  for (int m = 0; m < kStagePer; m++) {
    const int e = (int)threadIdx.x + 256 * m;
    ent[m] = e < kStagePairs ? table[e] : 0xffffffffu;
  }
  const int n = g.n, tid = (int)threadIdx.x;
  // work items of the 27 tiles around `tile` (threads 0..26 of wave 0): first item and count, and the largest count
  auto neighbours = [&](int tile, int &first, int &cnt) {
    const int tzi = tile % tp.ntz, tyi = (tile / tp.ntz) % tp.nty, txi = tile / (tp.ntz * tp.nty);
    const int dz = tid % 3 - 1, dy = (tid / 3) % 3 - 1, dx = tid / 9 - 1;
    const int nx = (txi + dx + tp.ntx) % tp.ntx, ny = (tyi + dy + tp.nty) % tp.nty, nz = (tzi + dz + tp.ntz) % tp.ntz;
    const int t = nz + tp.ntz * (ny + tp.nty * nx);
    first = woff[t];
    cnt = woff[t + 1] - first;
  };
  auto publish = [&](int buf, int first, int cnt) {  // wave 0 only
    int mx = tid < 27 ? cnt : 0;
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) mx = max(mx, __shfl_xor(mx, o, 32));
    if (tid < 27) {
      s_first[buf][tid] = first;
      s_cnt[buf][tid] = cnt;
    }
    if (tid == 0) s_maxcnt[buf] = mx;
  };
#pragma unroll
  for (int r = 0; r < 8; r++) (&acc[0][0])[tid + 256 * r] = 0.;
  int tile = blockIdx.x, buf = 0;
  if (tile < tp.ntiles && tid < 32) {
    int f = 0, c = 0;
    if (tid < 27) neighbours(tile, f, c);
    publish(0, f, c);
  }
  __syncthreads();
  // One barrier per tile: the accumulators are double-buffered and self-cleaning (a thread zeroes the four cells it has
  // just read), and the neighbour lists of the NEXT tile are fetched while this one is summed.
  // Consecutive tiles (z fastest) run on consecutive workgroups: the images a sweep touches are neighbours in memory.
  for (; tile < tp.ntiles; tile += gridDim.x, buf ^= 1) {
    const int next = tile + (int)gridDim.x;
    int nf = 0, nc = 0;
    if (next < tp.ntiles && tid < 27) neighbours(next, nf, nc);
    const int tzi = tile % tp.ntz, tyi = (tile / tp.ntz) % tp.nty, txi = tile / (tp.ntz * tp.nty);
    // the likelihood's data operands of this thread's four cells: requested now, used after the images are summed
    double lw[4], ln[4], ls[4];
    if (LIKE) {
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int c = tid + 256 * r;
        const int z = c & 15, y = (c >> 4) & 7, x = c >> 7;
        const long long i = (tzi * 16 + z) + (long long)n * ((tyi * 8 + y) + (long long)n * (txi * 8 + x));
        like_operands<T>(lp, i, nobs, noise, window, lw[r], ln[r], ls[r]);
      }
    }
    const int maxcnt = s_maxcnt[buf];
    for (int w = 0; w < maxcnt; w++) {  // work items of a tile: one unless the tile holds more than `chunk` particles
      double2 v[kStagePer];
#pragma unroll
      for (int m = 0; m < kStagePer; m++) {
        const unsigned e = ent[m];
        const int nb = (int)(e & 31u);
        v[m] = make_double2(0., 0.);
        if (e != 0xffffffffu && w < s_cnt[buf][nb])
          v[m] = *reinterpret_cast<const double2 *>(stage + (size_t)(s_first[buf][nb] + w) * kStageCells + (e >> 15));
      }
#pragma unroll
      for (int m = 0; m < kStagePer; m++) {
        double *a = &acc[buf][(ent[m] >> 5) & 1023u];
        if (v[m].x != 0.) unsafeAtomicAdd(a, v[m].x);
        if (v[m].y != 0.) unsafeAtomicAdd(a + 1, v[m].y);
      }
    }
    if (next < tp.ntiles && tid < 32) publish(buf ^ 1, nf, nc);
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int c = tid + 256 * r;
      const int z = c & 15, y = (c >> 4) & 7, x = c >> 7;
      const long long i = (tzi * 16 + z) + (long long)n * ((tyi * 8 + y) + (long long)n * (txi * 8 + x));
      const T d = (T)acc[buf][c];
      acc[buf][c] = 0.;
      if (WRHO) rho[i] = d;
      if (LIKE) plike[i] = (T)partial_like_value(lp, (double)d / nmean - 1., lw[r], ln[r], ls[r]);
    }
  }
}

// Row stride of the gather's LDS image in cells: LZ + BCHMC_GATHER_LZPAD.  The lanes of a wave are DIFFERENT particles
// in arbitrary home cells of the tile, so their ds_read_b64 addresses LZS (LY hx + hy) + hz are as good as random over
// the 32 eight-byte banks whatever the stride: the 58 % bank-conflict share of the LDS-active cycles (r02 SQ counters)
// is the birthday statistics of 16 random addresses per cycle on 32 banks, not a stride effect -- measured with pad 1
// (profiles/r03_ab_levers.txt): no change.
#ifndef BCHMC_GATHER_LZPAD
#define BCHMC_GATHER_LZPAD 0
#endif
#ifndef BCHMC_GATHER_EXPANDED
#define BCHMC_GATHER_EXPANDED 1
#endif
template <typename T, int LY, int LZ>
__global__ void __launch_bounds__(256, BCHMC_GATHER_WAVES)
k_gather_tile81(Geo g, HullPar hp, TilePar tp, int rsd, const RecQuad *__restrict__ srec,
                const long long *__restrict__ off,
                const long long *__restrict__ tend, const int *__restrict__ woff, const int4 *__restrict__ oct,
                const int *__restrict__ seg_in, const T *__restrict__ plike, T *__restrict__ V) {
  extern __shared__ __attribute__((aligned(16))) unsigned char s_raw_gather81[];
  T *s_tile_pl = reinterpret_cast<T *>(s_raw_gather81);
  int tile, pb, pe;
  long long rec0;
  OctMap om;
  if (!tile_work((int)blockIdx.x, tp, off, tend, woff, oct, seg_in, tile, rec0, pb, pe, om)) return;
  srec += rec0 * rec_quads<T>();  // this tile's record slots; pb, pe are relative to them
  constexpr int LZS = LZ + BCHMC_GATHER_LZPAD;
  const int ncell = tp.lx * LY * LZ;
  const int n = g.n;
  const int tzi = tile % tp.ntz, tyi = (tile / tp.ntz) % tp.nty, txi = tile / (tp.ntz * tp.nty);
  const int ox = txi * tp.tx - 2, oy = tyi * tp.ty - 2, oz = tzi * tp.tz - 2;
  for (int c = threadIdx.x; c < ncell; c += blockDim.x) {
    const int cz = c % LZ, cy = (c / LZ) % LY, cx = c / (LZ * LY);
    const int gx = (ox + cx + n) % n, gy = (oy + cy + n) % n, gz = (oz + cz + n) % n;
    s_tile_pl[cz + LZS * (cy + LY * cx)] = plike[gz + (long long)n * (gy + (long long)n * gx)];
  }
  __syncthreads();
  const T d_h = (T)hp.d_h, h_inv = (T)hp.h_inv, norm = (T)hp.norm;
  const HomeCell<T> hc = make_home<T>(g);
  for (int s = pb + threadIdx.x; s < pe; s += blockDim.x) {
    T px, py, pz;
    int rid;
    rec_load<T>(srec, om.slot(s), px, py, pz, rid);
    const int ix = home_cell_i(hc, px), iy = home_cell_i(hc, py), iz = home_cell_i(hc, pz);
    const T dpcx = px * h_inv - ((T)ix + T(0.5)) * d_h;
    const T dpcy = py * h_inv - ((T)iy + T(0.5)) * d_h;
    const T dpcz = pz * h_inv - ((T)iz + T(0.5)) * d_h;
    const int hx = wrap_cell(ix, n) - ox, hy = wrap_cell(iy, n) - oy, hz = wrap_cell(iz, n) - oz;
    T vx = T(0), vy = T(0), vz = T(0);
    const bool home_ok = (unsigned)(hx - 2) < (unsigned)tp.tx && (unsigned)(hy - 2) < (unsigned)tp.ty &&
                         (unsigned)(hz - 2) < (unsigned)tp.tz;  // always true; keeps LDS indexing safe
    if (home_ok) {
#if BCHMC_GATHER_LEAN
      // offsets only: their squares enter through fused multiply-adds where the sums are formed (an add becomes an
      // fma, no instruction more), which frees the 20 registers of the y and z squares: 6 waves per SIMD without spills
      T yh[5], zh[5];
#pragma unroll
      for (int a = 0; a < 5; a++) {
        yh[a] = dpcy - (T)(a - 2) * d_h;
        zh[a] = dpcz - (T)(a - 2) * d_h;
      }
#else
      T xh[5], yh[5], zh[5], X[5], Y[5], Z[5];
#pragma unroll
      for (int a = 0; a < 5; a++) {
        xh[a] = dpcx - (T)(a - 2) * d_h;
        yh[a] = dpcy - (T)(a - 2) * d_h;
        zh[a] = dpcz - (T)(a - 2) * d_h;
        X[a] = r_fma(xh[a], xh[a], tiny_pos<T>());  // q^2 > 0 also for a particle exactly on a cell centre
        Y[a] = yh[a] * yh[a];
        Z[a] = zh[a] * zh[a];
      }
#endif
      const T c225n = T(2.25) * norm, c3n = T(-3) * norm, c34n = T(-0.75) * norm;
#if BCHMC_GATHER_EXPANDED
      const T c3p = T(3) * norm;  // -4 c34n; 4 c34n is c3n
#endif
      const T *corner = s_tile_pl + LZS * ((hy - 2) + LY * (hx - 2)) + (hz - 2);
#pragma unroll
      for (int a = 0; a < 5; a++) {
#if BCHMC_GATHER_LEAN
        const T xh_a = dpcx - (T)(a - 2) * d_h;
        const T X_a = r_fma(xh_a, xh_a, tiny_pos<T>());  // q^2 > 0 also for a particle exactly on a cell centre
#else
        const T xh_a = xh[a], X_a = X[a];
#endif
#pragma unroll
        for (int b = 0; b < 5; b++) {
          const int zw = hull81_zw(a, b);  // folds after unrolling
          if (zw < 0) continue;
#if BCHMC_GATHER_LEAN
          const T r2ab = r_fma(yh[b], yh[b], X_a);
#else
          const T r2ab = X_a + Y[b];
#endif
          if (r2ab > T(4)) continue;
          const T *row = corner + LZS * (b + LY * a);
          // the column's part_like values first: their LDS latency hides behind the first candidate's arithmetic
          // (read where they are used, every candidate waited ~100 cycles on its own ds_read)
          T pl[5];
#pragma unroll
          for (int c = 0; c < 5; c++)
            if (c >= 2 - zw && c <= 2 + zw) pl[c] = row[c];
#pragma unroll
          for (int c = 0; c < 5; c++) {
            if (c < 2 - zw || c > 2 + zw) continue;
#if BCHMC_GATHER_LEAN
            const T q_sq = r_fma(zh[c], zh[c], r2ab);
#else
            const T q_sq = r2ab + Z[c];
#endif
            if (q_sq <= T(4)) {
              // grad_SPH_kernel_3D_h_units (SPH_kernel.cpp:148-208): dW/dq / q
              T rq;
              const T q = sqrt_rsq(q_sq, rq);
              // the branch is known at unroll time for the cells two away (q >= 1.5: outer) and the home cell
              // (q <= 0.87: inner), see k_scatter_tile81
              const bool far = (a == 0 || a == 4 || b == 0 || b == 4 || c == 0 || c == 4);
              const bool home = (a == 2 && b == 2 && c == 2);
              T gr;
              if (home) {
                gr = r_fma(c225n, q, c3n);
              } else {
#if BCHMC_GATHER_EXPANDED
                // -3/4 (q - 2)^2 / q = -3/4 (q - 4 + 4 / q): two fused multiply-adds from q and 1/q instead of a
                // subtraction and three multiplications.  Cancels near q = 2, where the term itself vanishes: the
                // absolute error stays at a few 1e-16 of the O(1) terms it is summed with.
                const T outer = r_fma(c3n, rq, r_fma(c34n, q, c3p));
#else
                const T qm2 = q - T(2);
                const T outer = ((qm2 * qm2) * c34n) * rq;
#endif
                gr = far ? outer : ((q_sq > T(1)) ? outer : r_fma(c225n, q, c3n));
              }
              const T common = pl[c] * gr;
              vx += common * xh_a;
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
    const long long p = rid & ~kSortFlagNoScatter;
    V[p] = vx;
    V[p + g.N] = vy;
    V[p + 2 * g.N] = vz;
  }
}

}  // namespace bchmc
