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
typedef float fv2 __attribute__((ext_vector_type(2)));

template <typename E> struct Arrs {
  const E *in[6];
  E *out[5];
};
struct Dim {
  int n, nhp;
};

template <bool NT, typename E>
__device__ __forceinline__ E ld(const E *p) { return NT ? __builtin_nontemporal_load(p) : *p; }
template <bool NT, typename E>
__device__ __forceinline__ void st(E *p, E v) {
  if (NT) __builtin_nontemporal_store(v, p);
  else *p = v;
}

// MODE 0: contiguous (workgroup b takes elements [b * 2048, (b + 1) * 2048)); 1: x-columns (j, k-block) x all i;
// 2: y-columns (i, k-block) x all j.  512 threads, 4 elements per thread and array, like the boundary kernel.
// workgroup: NTH threads, PER elements per thread and array, KB = 128 B / sizeof(E) columns: n == PER * NTH / KB rows
template <int MODE, bool NT, int NIN, int NOUT, typename E, int NTH, int PER>
__global__ void __launch_bounds__(NTH) k_copy(Arrs<E> a, Dim d) {
  constexpr int KB = 128 / (int)sizeof(E), ROWS = NTH / KB;
  const int n = d.n, nhp = d.nhp;
  const int ntk = nhp / KB;
  const int c = threadIdx.x % KB, row = threadIdx.x / KB;
  long long e[PER];
#pragma unroll
  for (int m = 0; m < PER; m++) {
    const int r = row + ROWS * m;
    if (MODE == 0) e[m] = (long long)blockIdx.x * (NTH * PER) + threadIdx.x + NTH * m;
    if (MODE == 1) {
      const int j = blockIdx.x / ntk, k = (blockIdx.x % ntk) * KB + c;
      e[m] = k + (long long)nhp * (j + (long long)n * r);
    }
    if (MODE == 2) {
      const int i = blockIdx.x / ntk, k = (blockIdx.x % ntk) * KB + c;
      e[m] = k + (long long)nhp * (r + (long long)n * i);
    }
  }
  E acc[PER];
#pragma unroll
  for (int m = 0; m < PER; m++) acc[m] = E{0, 0};
#pragma unroll
  for (int s = 0; s < NIN; s++)
#pragma unroll
    for (int m = 0; m < PER; m++) acc[m] += ld<NT>(a.in[s] + e[m]);
#pragma unroll
  for (int s = 0; s < NOUT; s++)
#pragma unroll
    for (int m = 0; m < PER; m++) st<NT>(a.out[s] + e[m], acc[m]);
}

template <typename E, int NTH, int PER>
static void suite(const char *title, int n, int nhp) {
  constexpr int KB = 128 / (int)sizeof(E);
  const long long Nhp = (long long)n * n * nhp;
  Arrs<E> a;
  for (int s = 0; s < 6; s++) {
    E *p;
    CK(hipMalloc(&p, Nhp * sizeof(E)));
    CK(hipMemset(p, 0, Nhp * sizeof(E)));
    a.in[s] = p;
  }
  for (int s = 0; s < 5; s++) CK(hipMalloc(&a.out[s], Nhp * sizeof(E)));
  const Dim d{n, nhp};
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int grid = n * (nhp / KB);  // workgroups of n * KB elements
  auto run = [&](const char *name, auto launch, double bytes) {
    float best = 1e9f;
    for (int r = 0; r < 10; r++) {
      CK(hipEventRecord(e0));
      launch();
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (r >= 2) best = std::min(best, ms);
    }
    printf("  %-42s %.4f ms  %.2f TB/s\n", name, best, bytes / best * 1e-9);
    fflush(stdout);
  };
  printf("%s\n", title);
  const double B = (double)Nhp * sizeof(E);
  run("contiguous, 6 in + 5 out, nt", [&] { k_copy<0, true, 6, 5, E, NTH, PER><<<grid, NTH>>>(a, d); }, 11 * B);
  run("x-columns, 6 in + 5 out", [&] { k_copy<1, false, 6, 5, E, NTH, PER><<<grid, NTH>>>(a, d); }, 11 * B);
  run("x-columns, 6 in + 5 out, nt", [&] { k_copy<1, true, 6, 5, E, NTH, PER><<<grid, NTH>>>(a, d); }, 11 * B);
  run("y-columns, 6 in + 5 out, nt", [&] { k_copy<2, true, 6, 5, E, NTH, PER><<<grid, NTH>>>(a, d); }, 11 * B);
  run("x-columns, 6 in only (1 out)", [&] { k_copy<1, true, 6, 1, E, NTH, PER><<<grid, NTH>>>(a, d); }, 7 * B);
  run("x-columns, 1 in + 5 out", [&] { k_copy<1, true, 1, 5, E, NTH, PER><<<grid, NTH>>>(a, d); }, 6 * B);
  run("x-columns, 1 in + 1 out", [&] { k_copy<1, true, 1, 1, E, NTH, PER><<<grid, NTH>>>(a, d); }, 2 * B);
  run("y-columns, 1 in + 1 out", [&] { k_copy<2, true, 1, 1, E, NTH, PER><<<grid, NTH>>>(a, d); }, 2 * B);
  run("contiguous, 1 in + 1 out", [&] { k_copy<0, true, 1, 1, E, NTH, PER><<<grid, NTH>>>(a, d); }, 2 * B);
  for (int s = 0; s < 6; s++) CK(hipFree((void *)a.in[s]));
  for (int s = 0; s < 5; s++) CK(hipFree(a.out[s]));
}

int main() {
  suite<dv2, 512, 4>("256^3 fp64 (rows of 136 complex), 512 threads x 4", 256, 136);
  suite<fv2, 1024, 8>("512^3 fp32 (rows of 272 complex), 1024 threads x 8", 512, 272);
  suite<fv2, 512, 16>("512^3 fp32, 512 threads x 16", 512, 272);
  suite<dv2, 512, 8>("512^3 fp64 (rows of 264 complex), 512 threads x 8", 512, 264);
  return 0;
}
