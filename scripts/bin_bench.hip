// scripts/bin_bench.hip -- where the one-pass binning's time goes: the product kernel k_bin_direct<double> and copies
// of it with one ingredient removed each (timing only: the copies' outputs are wrong on purpose and go to scratch).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -munsafe-fp-atomics -I barcode_amd/csrc scripts/bin_bench.hip -o scripts/bin_bench
//   scripts/bin_bench [n=256] [rms displacement in cells=5]
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "kernels.hpp"

using namespace bchmc;

#define CK(x)                                                                  \
  do {                                                                         \
    hipError_t e_ = (x);                                                       \
    if (e_ != hipSuccess) {                                                    \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      exit(1);                                                                 \
    }                                                                          \
  } while (0)

// E: 1 record stores at the Lagrangian index (coalesced), 2 no record stores, 3 no rho clearing, 4 no psi loads,
//    5 no LDS hash and no global atomics (rank = thread), 6 = 2 + 3 (no stores at all), 7 = nt record stores
template <int E>
__global__ void __launch_bounds__(256)
k_bin_expt(Geo g, PosPar pp, SphPar sp, TilePar tp, int nsuper, const double *__restrict__ psi, int *__restrict__ cnt,
           int *__restrict__ ovf, RecQuad *__restrict__ srec, double *__restrict__ V, double *__restrict__ rho_zero) {
  using T = double;
  constexpr int kSlots = 2048;
  __shared__ int hkey[kSlots], hcnt[kSlots], hbase[kSlots];
  const HomeCell<T> hc = make_home<T>(g);
  for (int sb = blockIdx.x; sb < nsuper; sb += gridDim.x) {
    for (int s = threadIdx.x; s < kSlots; s += blockDim.x) {
      hkey[s] = 0;
      hcnt[s] = 0;
    }
    __syncthreads();
    long long p[4];
    int key[4], slot[4], local[4], flag[4], li[4], lj[4], lk[4];
    T x[4], y[4], z[4];
    const int tid = (int)threadIdx.x;
    const int nbz = g.n >> 4, nbys = (g.n >> 2) / 2;
    const int sbk = sb % nbz, sbj = (sb / nbz) % nbys, sbi = sb / (nbz * nbys);
#pragma unroll
    for (int m = 0; m < 4; m++) {
      li[m] = (2 * sbi + m / 2) * 4 + (tid >> 6);
      lj[m] = (2 * sbj + m % 2) * 4 + ((tid >> 4) & 3);
      lk[m] = sbk * 16 + (tid & 15);
      p[m] = lk[m] + (long long)g.n * (lj[m] + (long long)g.n * li[m]);
      if (E == 4) {
        x[m] = 1e-3 * lk[m];
        y[m] = 2e-3 * lj[m];
        z[m] = 3e-3 * li[m];
      } else {
        x[m] = psi[p[m]];
        y[m] = psi[p[m] + g.N];
        z[m] = psi[p[m] + 2 * g.N];
      }
      if (E != 3 && E != 6) rho_zero[p[m]] = T(0);
    }
#pragma unroll
    for (int m = 0; m < 4; m++) {
      key[m] = -1;
      slot[m] = local[m] = flag[m] = 0;
      particle_pos<T>(pp, li[m], lj[m], lk[m], x[m], y[m], z[m], x[m], y[m], z[m]);
      if (pos_ok(g, x[m], y[m], z[m])) {
        const int t = tile_of_wrapped(tp, wrap_cell(home_cell_i(hc, x[m]), g.n), wrap_cell(home_cell_i(hc, y[m]), g.n),
                                      wrap_cell(home_cell_i(hc, z[m]), g.n));
        flag[m] = in_domain(g, sp, x[m], y[m], z[m]) ? 0 : kSortFlagNoScatter;
        key[m] = t * kOct + subcell_octant<T>(x[m], y[m], z[m], hc.inv_d);
        if (E != 5) {
          int sl = (int)(((unsigned)key[m] * 2654435761u) >> 21) & (kSlots - 1);
          for (;;) {
            const int old = atomicCAS(&hkey[sl], 0, key[m] + 1);
            if (old == 0 || old == key[m] + 1) break;
            sl = (sl + 1) & (kSlots - 1);
          }
          slot[m] = sl;
          local[m] = atomicAdd(&hcnt[sl], 1);
        }
      } else {
        V[p[m]] = T(0);
      }
    }
    if (E != 5) {
      __syncthreads();
      int hk[8], hb[8];
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const int s = tid + u * 256;
        hk[u] = hkey[s];
        hb[u] = 0;
        if (hk[u]) hb[u] = atomicAdd(&cnt[hk[u] - 1], hcnt[s]);
      }
#pragma unroll
      for (int u = 0; u < 8; u++)
        if (hk[u]) hbase[tid + u * 256] = hb[u];
      __syncthreads();
    }
    const int seg = tp.cap / kOct;
#pragma unroll
    for (int m = 0; m < 4; m++) {
      if (key[m] < 0) continue;
      const int rank = E == 5 ? (tid & 63) : hbase[slot[m]] + local[m];
      if (rank >= seg) {
        ovf[0] = 1;
      } else if (E != 2 && E != 6) {
        const int t = key[m] / kOct;
        long long dst = (long long)t * tp.cap + (long long)(key[m] - t * kOct) * seg + rank;
        if (E == 1) dst = p[m];
        if (E == 7) {
          double2 *b = reinterpret_cast<double2 *>(srec) + 2 * dst;
          __builtin_nontemporal_store(x[m], &b[0].x);
          __builtin_nontemporal_store(y[m], &b[0].y);
          __builtin_nontemporal_store(z[m], &b[1].x);
          __builtin_nontemporal_store(__longlong_as_double((long long)(unsigned)((int)p[m] | flag[m])), &b[1].y);
        } else {
          rec_store<T>(srec, dst, x[m], y[m], z[m], (int)p[m] | flag[m]);
        }
      }
    }
    __syncthreads();
  }
}


// Staged output: records go to LDS in run order, then the workgroup writes them out with consecutive lanes on
// consecutive 16-byte quads of a run.  SLOTS: hash table size.
template <int SLOTS, int NB>
__global__ void __launch_bounds__(256)
k_bin_staged(Geo g, PosPar pp, SphPar sp, TilePar tp, int nsuper, const double *__restrict__ psi, int *__restrict__ cnt,
             int *__restrict__ ovf, RecQuad *__restrict__ srec, double *__restrict__ V, double *__restrict__ rho_zero) {
  using T = double;
  constexpr int kSlots = SLOTS, kShift = 32 - (SLOTS == 2048 ? 11 : SLOTS == 1024 ? 10 : 9);
  __shared__ int hkey[kSlots], hcnt[kSlots], hbase[kSlots], lbase[kSlots];
  __shared__ RecQuad lrec[NB * 256 * 2];
  __shared__ long long ldst[NB * 256];
  __shared__ int ltotal;
  const HomeCell<T> hc = make_home<T>(g);
  for (int sb = blockIdx.x; sb < nsuper; sb += gridDim.x) {
    for (int s = threadIdx.x; s < kSlots; s += blockDim.x) {
      hkey[s] = 0;
      hcnt[s] = 0;
    }
    if (threadIdx.x == 0) ltotal = 0;
    __syncthreads();
    long long p[NB];
    int key[NB], slot[NB], local[NB], flag[NB], li[NB], lj[NB], lk[NB];
    T x[NB], y[NB], z[NB];
    const int tid = (int)threadIdx.x;
    const int nbz = g.n >> 4, nbys = (g.n >> 2) / 2;
    const int sbk = sb % nbz, sbj = (sb / nbz) % nbys, sbi = sb / (nbz * nbys);
#pragma unroll
    for (int m = 0; m < NB; m++) {
      li[m] = (2 * sbi + m / 2) * 4 + (tid >> 6);
      lj[m] = (2 * sbj + m % 2) * 4 + ((tid >> 4) & 3);
      lk[m] = sbk * 16 + (tid & 15);
      p[m] = lk[m] + (long long)g.n * (lj[m] + (long long)g.n * li[m]);
      x[m] = psi[p[m]];
      y[m] = psi[p[m] + g.N];
      z[m] = psi[p[m] + 2 * g.N];
      rho_zero[p[m]] = T(0);
    }
#pragma unroll
    for (int m = 0; m < NB; m++) {
      key[m] = -1;
      slot[m] = local[m] = flag[m] = 0;
      particle_pos<T>(pp, li[m], lj[m], lk[m], x[m], y[m], z[m], x[m], y[m], z[m]);
      if (pos_ok(g, x[m], y[m], z[m])) {
        const int t = tile_of_wrapped(tp, wrap_cell(home_cell_i(hc, x[m]), g.n), wrap_cell(home_cell_i(hc, y[m]), g.n),
                                      wrap_cell(home_cell_i(hc, z[m]), g.n));
        flag[m] = in_domain(g, sp, x[m], y[m], z[m]) ? 0 : kSortFlagNoScatter;
        key[m] = t * kOct + subcell_octant<T>(x[m], y[m], z[m], hc.inv_d);
        int sl = (int)(((unsigned)key[m] * 2654435761u) >> kShift) & (kSlots - 1);
        for (;;) {
          const int old = atomicCAS(&hkey[sl], 0, key[m] + 1);
          if (old == 0 || old == key[m] + 1) break;
          sl = (sl + 1) & (kSlots - 1);
        }
        slot[m] = sl;
        local[m] = atomicAdd(&hcnt[sl], 1);
      } else {
        V[p[m]] = T(0);
      }
    }
    __syncthreads();
    {
      constexpr int kPer = kSlots / 256;
      int hk[kPer], hb[kPer];
#pragma unroll
      for (int u = 0; u < kPer; u++) {
        const int s = tid + u * 256;
        hk[u] = hkey[s];
        hb[u] = 0;
        if (hk[u]) {
          hb[u] = atomicAdd(&cnt[hk[u] - 1], hcnt[s]);
          lbase[s] = atomicAdd(&ltotal, hcnt[s]);
        }
      }
#pragma unroll
      for (int u = 0; u < kPer; u++)
        if (hk[u]) hbase[tid + u * 256] = hb[u];
    }
    __syncthreads();
    const int seg = tp.cap / kOct;
#pragma unroll
    for (int m = 0; m < NB; m++) {
      if (key[m] < 0) continue;
      const int rank = hbase[slot[m]] + local[m], r = lbase[slot[m]] + local[m];
      long long dst = -1;
      if (rank >= seg) {
        ovf[0] = 1;
      } else {
        const int t = key[m] / kOct;
        dst = (long long)t * tp.cap + (long long)(key[m] - t * kOct) * seg + rank;
      }
      ldst[r] = dst;
      rec_store<T>(lrec, r, x[m], y[m], z[m], (int)p[m] | flag[m]);
    }
    __syncthreads();
    const int nq = ltotal * 2;
    for (int qi = tid; qi < nq; qi += 256) {
      const long long d = ldst[qi >> 1];
      if (d >= 0) srec[2 * d + (qi & 1)] = lrec[qi];
    }
    __syncthreads();
  }
}

int main(int argc, char **argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 256;
  const double rms_cells = argc > 2 ? atof(argv[2]) : 5.0;
  Geo g{};
  g.n = n;
  g.nh = n / 2 + 1;
  g.nhp = g.nh;
  g.N = (long long)n * n * n;
  g.L = 200.0 * n / 256;
  g.d = g.L / n;
  g.kfac = 2 * M_PI / g.L;
  PosPar pp{};
  pp.d = g.d;
  pp.L = g.L;
  pp.rsd = 1;
  pp.periodic = 1;
  pp.cpecvel = 0.5;
  pp.v_norm = 1.0;
  SphPar sp{};
  sp.h = g.d;
  sp.h_inv = 1 / sp.h;
  sp.min1 = sp.min2 = sp.min3 = 0;
  sp.reach = 2;
  TilePar tp{};
  tp.tx = tp.ty = 8;
  tp.tz = 16;
  tp.ntx = n / 8;
  tp.nty = n / 8;
  tp.ntz = n / 16;
  tp.ntiles = tp.ntx * tp.nty * tp.ntz;
  tp.R = 2;
  tp.cap = 16384;
  tp.chunk = 2048;
  const long long N = g.N;
  // smooth displacement field: a few long waves (coherent bulk flows, as Zel'dovich displacements are) + small noise
  std::vector<double> psi(3 * N);
  const double amp = rms_cells * g.d;
  srand(1);
  for (int c = 0; c < 3; c++)
    for (int i = 0; i < n; i++)
      for (int j = 0; j < n; j++)
        for (int k = 0; k < n; k++) {
          const double u = 2 * M_PI * i / n, v = 2 * M_PI * j / n, w = 2 * M_PI * k / n;
          double s = sin(u + 0.3 * c) + 0.7 * sin(2 * v + c) + 0.5 * cos(3 * w + 2 * c) + 0.4 * sin(5 * u + 3 * v + w) +
                     0.3 * cos(9 * v - 7 * w + c);
          s += 0.2 * ((rand() & 0xffff) / 65536.0 - 0.5);
          psi[(size_t)c * N + k + (size_t)n * (j + (size_t)n * i)] = amp * s / 1.0;
        }
  double *d_psi, *d_V, *d_rho, *d_zero;
  int *d_cnt, *d_ovf;
  RecQuad *d_rec;
  long long *d_fix = nullptr;
  CK(hipMalloc(&d_psi, 3 * N * 8));
  CK(hipMalloc(&d_V, 3 * N * 8));
  CK(hipMalloc(&d_rho, N * 8));
  CK(hipMalloc(&d_zero, 4096 * 8));
  CK(hipMalloc(&d_cnt, (size_t)tp.ntiles * 8 * 4));
  CK(hipMalloc(&d_ovf, 64));
  CK(hipMalloc(&d_rec, (size_t)tp.ntiles * tp.cap * 32));
  CK(hipMemcpy(d_psi, psi.data(), 3 * N * 8, hipMemcpyHostToDevice));
  const int nsuper = (n / 4) * (n / 4) * (n / 16) / 4;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  auto run = [&](const char *name, auto launch) {
    float best = 1e9, sum = 0;
    const int reps = 12;
    for (int r = 0; r < reps + 2; r++) {
      CK(hipMemsetAsync(d_cnt, 0, (size_t)tp.ntiles * 8 * 4));
      CK(hipMemsetAsync(d_ovf, 0, 64));
      CK(hipEventRecord(e0));
      launch();
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (r >= 2) {
        best = std::min(best, ms);
        sum += ms;
      }
    }
    int ovf[2];
    CK(hipMemcpy(ovf, d_ovf, 8, hipMemcpyDeviceToHost));
    printf("%-58s mean %.4f ms  best %.4f ms  ovf %d\n", name, sum / reps, best, ovf[0]);
    fflush(stdout);
  };
  printf("n %d, rms displacement %.1f cells, nsuper %d, cap %d\n", n, rms_cells, nsuper, tp.cap);
  run("product k_bin_direct<double>", [&] {
    k_bin_direct<double><<<nsuper, BCHMC_BIN_THREADS>>>(g, pp, sp, tp, nsuper, d_psi, d_cnt, d_ovf, d_rec, d_V, d_zero, d_rho,
                                                        d_fix);
  });
#define EX(E, what) \
  run(what, [&] { k_bin_expt<E><<<nsuper, 256>>>(g, pp, sp, tp, nsuper, d_psi, d_cnt, d_ovf, d_rec, d_V, d_rho); })
  EX(0, "copy of the product kernel");
  EX(1, "records stored at the Lagrangian index (coalesced)");
  EX(2, "no record stores");
  EX(3, "no rho clearing");
  EX(6, "no stores at all");
  EX(4, "no psi loads");
  EX(5, "no LDS hash, no global atomics");
  EX(7, "nt record stores");
  run("staged through LDS, 2048 slots", [&] { k_bin_staged<2048, 4><<<nsuper, 256>>>(g, pp, sp, tp, nsuper, d_psi, d_cnt, d_ovf, d_rec, d_V, d_rho); });
  run("staged through LDS, 1024 slots", [&] { k_bin_staged<1024, 4><<<nsuper, 256>>>(g, pp, sp, tp, nsuper, d_psi, d_cnt, d_ovf, d_rec, d_V, d_rho); });
  run("product k_bin_direct<double> again", [&] {
    k_bin_direct<double><<<nsuper, BCHMC_BIN_THREADS>>>(g, pp, sp, tp, nsuper, d_psi, d_cnt, d_ovf, d_rec, d_V, d_zero, d_rho,
                                                        d_fix);
  });
  for (int extra : {0, 8192, 16384, 29000, 56000}) {
    char nm[96];
    snprintf(nm, sizeof nm, "product + %d bytes of dynamic LDS per workgroup", extra);
    run(nm, [&] {
      k_bin_direct<double><<<nsuper, BCHMC_BIN_THREADS, extra>>>(g, pp, sp, tp, nsuper, d_psi, d_cnt, d_ovf, d_rec, d_V, d_zero,
                                                                 d_rho, d_fix);
    });
  }
  for (int grid : {256 * 2, 256 * 3, 256 * 4, 256 * 8}) {
    char nm[96];
    snprintf(nm, sizeof nm, "product, persistent grid of %d workgroups", grid);
    run(nm, [&] {
      k_bin_direct<double><<<grid, BCHMC_BIN_THREADS>>>(g, pp, sp, tp, nsuper, d_psi, d_cnt, d_ovf, d_rec, d_V, d_zero,
                                                                 d_rho, d_fix);
    });
  }
  run("product without zero_part", [&] {
    k_bin_direct<double><<<nsuper, BCHMC_BIN_THREADS>>>(g, pp, sp, tp, nsuper, d_psi, d_cnt, d_ovf, d_rec, d_V, nullptr, d_rho,
                                                        d_fix);
  });
  return 0;
}
