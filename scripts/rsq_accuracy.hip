// How accurate is v_rsq_f64 on gfx950, and how many Newton steps does fast_rsqrt need for <= 2 ulp?
//   hipcc -O2 --offload-arch=gfx950 scripts/rsq_accuracy.hip -o /tmp/rsq && /tmp/rsq
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>

__global__ void k(const double *x, double *y0, double *y1, double *y2, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double v = x[i];
  double y = __builtin_amdgcn_rsq(v);
  y0[i] = y;
  const double hx = 0.5 * v;
  double e = fma(-hx * y, y, 0.5);
  y = fma(y, e, y);
  y1[i] = y;
  e = fma(-hx * y, y, 0.5);
  y = fma(y, e, y);
  y2[i] = y;
}

int main() {
  const int n = 1 << 22;
  std::vector<double> x(n);
  for (int i = 0; i < n; i++) x[i] = 1e-6 * std::pow(1.6e7, (i + 0.37) / n);  // 1e-6 .. 16, log-spaced
  double *dx, *d0, *d1, *d2;
  hipMalloc(&dx, n * 8); hipMalloc(&d0, n * 8); hipMalloc(&d1, n * 8); hipMalloc(&d2, n * 8);
  hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
  k<<<n / 256, 256>>>(dx, d0, d1, d2, n);
  std::vector<double> y0(n), y1(n), y2(n);
  hipMemcpy(y0.data(), d0, n * 8, hipMemcpyDeviceToHost);
  hipMemcpy(y1.data(), d1, n * 8, hipMemcpyDeviceToHost);
  hipMemcpy(y2.data(), d2, n * 8, hipMemcpyDeviceToHost);
  long double m0 = 0, m1 = 0, m2 = 0;
  for (int i = 0; i < n; i++) {
    const long double ex = 1.0L / sqrtl((long double)x[i]);
    m0 = fmaxl(m0, fabsl((y0[i] - ex) / ex));
    m1 = fmaxl(m1, fabsl((y1[i] - ex) / ex));
    m2 = fmaxl(m2, fabsl((y2[i] - ex) / ex));
  }
  printf("max relative error: v_rsq_f64 %.3Le, +1 Newton %.3Le (%.2Lf ulp), +2 Newton %.3Le (%.2Lf ulp)\n", m0, m1,
         m1 / 1.11e-16L, m2, m2 / 1.11e-16L);
  return 0;
}
