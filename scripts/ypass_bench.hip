// scripts/ypass_bench.hip -- k_ypass (barcode_amd/csrc/zpass.hpp) at the grids of BASELINE configs 3 and 5 in several
// workgroup shapes, against the bytes it moves (in place: 2 x 3 x Nhp complex).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -munsafe-fp-atomics -I barcode_amd/csrc scripts/ypass_bench.hip -o scripts/ypass_bench
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

template <typename T, int NT, int PER, int HINT>
static void run(const char *name, int n, int log2n) {
  using CT = C2<T>;
  constexpr int KB = 128 / (int)sizeof(CT);
  if (n != PER * NT / KB) {
    printf("%-44s (shape does not fit n = %d)\n", name, n);
    return;
  }
  Geo g{};
  g.n = n;
  g.nh = n / 2 + 1;
  g.nhp = (g.nh + KB - 1) / KB * KB;
  g.Nhp = (long long)n * n * g.nhp;
  CT *d_c, *d_tw;
  CK(hipMalloc(&d_c, 3 * g.Nhp * sizeof(CT)));
  CK(hipMemset(d_c, 0, 3 * g.Nhp * sizeof(CT)));
  std::vector<CT> tw(n / 2);
  for (int r = 0; r < n / 2; r++) {
    tw[r].x = (T)cos(-2 * M_PI * r / n);
    tw[r].y = (T)sin(-2 * M_PI * r / n);
  }
  CK(hipMalloc(&d_tw, n / 2 * sizeof(CT)));
  CK(hipMemcpy(d_tw, tw.data(), n / 2 * sizeof(CT), hipMemcpyHostToDevice));
  const int grid = 3 * n * (g.nhp / KB);
  const size_t lds = ((size_t)n * KB + n / 2) * sizeof(CT);
  auto kern = k_ypass<T, NT, PER, HINT>;
  if (lds > 48 * 1024)
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  float best = 1e9f;
  for (int r = 0; r < 8; r++) {
    CK(hipEventRecord(e0));
    kern<<<grid, NT, lds>>>(g, log2n, d_tw, d_c);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (r >= 2) best = std::min(best, ms);
  }
  CK(hipGetLastError());
  printf("%-44s %.4f ms  %.2f TB/s\n", name, best, 2.0 * 3 * g.Nhp * sizeof(CT) / best * 1e-9);
  fflush(stdout);
  CK(hipFree(d_c));
  CK(hipFree(d_tw));
}

int main() {
  printf("256^3 fp64\n");
  run<double, 512, 4, 0>("  512 threads x 4", 256, 8);
  run<double, 256, 8, 0>("  256 threads x 8", 256, 8);
  printf("512^3 fp32\n");
  run<float, 512, 16, 0>("  512 threads x 16 (engine)", 512, 9);
  run<float, 1024, 8, 0>("  1024 threads x 8", 512, 9);
  run<float, 512, 16, 1>("  512 threads x 16, nt", 512, 9);
  run<float, 1024, 8, 1>("  1024 threads x 8, nt", 512, 9);
  printf("512^3 fp64\n");
  run<double, 512, 8, 0>("  512 threads x 8 (engine)", 512, 9);
  run<double, 1024, 4, 0>("  1024 threads x 4", 512, 9);
  run<double, 512, 8, 1>("  512 threads x 8, nt", 512, 9);
  run<double, 1024, 4, 1>("  1024 threads x 4, nt", 512, 9);
  printf("256^3 fp32\n");
  run<float, 512, 8, 0>("  512 threads x 8 (engine)", 256, 8);
  run<float, 1024, 4, 0>("  1024 threads x 4", 256, 8);
  return 0;
}
