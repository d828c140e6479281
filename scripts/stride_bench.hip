// scripts/stride_bench.hip -- what the memory system gives the access pattern of k_step_boundary_x: a copy kernel with
// the kernel's byte mix (6 half-complex arrays read, 5 written, 256^3 fp64, rows padded to 136 complex) in three
// patterns: contiguous, x-columns (one 128-byte segment per (i) plane: the planes-mode boundary), y-columns (one
// 128-byte segment per row of a plane: rocFFT's column pass).  No arithmetic, no LDS.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 scripts/stride_bench.hip -o scripts/stride_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>

#define CK(x)                                                                  \
  do {                                                                         \
    hipError_t e_ = (x);                                                       \
    if (e_ != hipSuccess) {                                                    \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      exit(1);                                                                 \
    }                                                                          \
  } while (0)

typedef double dv2 __attribute__((ext_vector_type(2)));
constexpr int n = 256, nhp = 136, KB = 8;
constexpr long long Nhp = (long long)n * n * nhp;

struct Arrs {
  const dv2 *in[6];
  dv2 *out[5];
};

template <bool NT>
__device__ __forceinline__ dv2 ld(const dv2 *p) { return NT ? __builtin_nontemporal_load(p) : *p; }
template <bool NT>
__device__ __forceinline__ void st(dv2 *p, dv2 v) {
  if (NT) __builtin_nontemporal_store(v, p);
  else *p = v;
}

// MODE 0: contiguous (workgroup b takes elements [b * 2048, (b + 1) * 2048)); 1: x-columns (j, k-block) x all i;
// 2: y-columns (i, k-block) x all j.  512 threads, 4 elements per thread and array, like the boundary kernel.
template <int MODE, bool NT, int NIN, int NOUT>
__global__ void __launch_bounds__(512) k_copy(Arrs a) {
  const int ntk = nhp / KB;
  const int c = threadIdx.x % KB, row = threadIdx.x / KB;  // 64 rows of 8 columns
  long long e[4];
#pragma unroll
  for (int m = 0; m < 4; m++) {
    const int r = row + 64 * m;
    if (MODE == 0) e[m] = (long long)blockIdx.x * 2048 + threadIdx.x + 512 * m;
    if (MODE == 1) {
      const int j = blockIdx.x / ntk, k = (blockIdx.x % ntk) * KB + c;
      e[m] = k + (long long)nhp * (j + (long long)n * r);
    }
    if (MODE == 2) {
      const int i = blockIdx.x / ntk, k = (blockIdx.x % ntk) * KB + c;
      e[m] = k + (long long)nhp * (r + (long long)n * i);
    }
  }
  dv2 acc[4] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};
#pragma unroll
  for (int s = 0; s < NIN; s++)
#pragma unroll
    for (int m = 0; m < 4; m++) acc[m] += ld<NT>(a.in[s] + e[m]);
#pragma unroll
  for (int s = 0; s < NOUT; s++)
#pragma unroll
    for (int m = 0; m < 4; m++) st<NT>(a.out[s] + e[m], acc[m]);
}

int main() {
  Arrs a;
  for (int s = 0; s < 6; s++) {
    dv2 *p;
    CK(hipMalloc(&p, Nhp * 16));
    CK(hipMemset(p, 0, Nhp * 16));
    a.in[s] = p;
  }
  for (int s = 0; s < 5; s++) CK(hipMalloc(&a.out[s], Nhp * 16));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int grid = n * (nhp / KB);  // 4352 workgroups of 2048 elements = Nhp
  auto run = [&](const char *name, auto launch, double bytes) {
    float best = 1e9f;
    for (int r = 0; r < 12; r++) {
      CK(hipEventRecord(e0));
      launch();
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (r >= 2) best = std::min(best, ms);
    }
    printf("%-44s %.4f ms  %.2f TB/s\n", name, best, bytes / best * 1e-9);
    fflush(stdout);
  };
  const double B11 = 11.0 * Nhp * 16, B2 = 2.0 * Nhp * 16, B6 = 6.0 * Nhp * 16, B5 = 5.0 * Nhp * 16;
  run("contiguous, 6 in + 5 out", [&] { k_copy<0, false, 6, 5><<<grid, 512>>>(a); }, B11);
  run("contiguous, 6 in + 5 out, nt", [&] { k_copy<0, true, 6, 5><<<grid, 512>>>(a); }, B11);
  run("x-columns, 6 in + 5 out", [&] { k_copy<1, false, 6, 5><<<grid, 512>>>(a); }, B11);
  run("x-columns, 6 in + 5 out, nt", [&] { k_copy<1, true, 6, 5><<<grid, 512>>>(a); }, B11);
  run("y-columns, 6 in + 5 out", [&] { k_copy<2, false, 6, 5><<<grid, 512>>>(a); }, B11);
  run("y-columns, 6 in + 5 out, nt", [&] { k_copy<2, true, 6, 5><<<grid, 512>>>(a); }, B11);
  run("x-columns, 6 in only (1 out)", [&] { k_copy<1, true, 6, 1><<<grid, 512>>>(a); }, B6 + B2 / 2);
  run("x-columns, 1 in + 5 out", [&] { k_copy<1, true, 1, 5><<<grid, 512>>>(a); }, B5 + B2 / 2);
  run("contiguous, 6 in only (1 out)", [&] { k_copy<0, true, 6, 1><<<grid, 512>>>(a); }, B6 + B2 / 2);
  run("contiguous, 1 in + 5 out", [&] { k_copy<0, true, 1, 5><<<grid, 512>>>(a); }, B5 + B2 / 2);
  run("x-columns, 1 in + 1 out", [&] { k_copy<1, true, 1, 1><<<grid, 512>>>(a); }, B2);
  run("y-columns, 1 in + 1 out", [&] { k_copy<2, true, 1, 1><<<grid, 512>>>(a); }, B2);
  run("contiguous, 1 in + 1 out", [&] { k_copy<0, true, 1, 1><<<grid, 512>>>(a); }, B2);
  return 0;
}
