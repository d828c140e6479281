// scripts/zpass_bench.hip -- check + bench of the two kernels that replace rocFFT's 2-D C2R in front of the binning
// (barcode_amd/csrc/zpass.hpp): k_ypass against a host DFT on sampled columns; k_zbin_direct's displacements against the
// field its input was made from and its counters against k_bin_direct's; their times, and what the no-op fallback
// launches cost.  (The timing experiments that shaped k_zbin_direct -- without the transform, the record stores, the
// counter atomics, with one counter per two / four octants, decorrelated workgroup order -- are recorded in
// profiles/r03_zpass_bench.txt; their switches are no longer in the kernel.)
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -munsafe-fp-atomics -I barcode_amd/csrc scripts/zpass_bench.hip -o scripts/zpass_bench
#include <cmath>
#include <complex>
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

int main(int argc, char **argv) {
  const int n = 256, nh = n / 2 + 1, nhp = 136, log2n = 8;
  Geo g{};
  g.n = n;
  g.nh = nh;
  g.nhp = nhp;
  g.N = (long long)n * n * n;
  g.Nh = (long long)n * n * nh;
  g.Nhp = (long long)n * n * nhp;
  g.L = 200.0;
  g.d = g.L / n;
  g.kfac = 2 * M_PI / g.L;
  const long long Nhp = g.Nhp;
  std::vector<double2> hin(3 * Nhp);
  srand(3);
  for (auto &v : hin) v = make_double2((rand() & 0xffff) / 65536.0 - 0.5, (rand() & 0xffff) / 65536.0 - 0.5);
  std::vector<double2> tw(n / 2);
  for (int r = 0; r < n / 2; r++) tw[r] = make_double2(cos(-2 * M_PI * r / n), sin(-2 * M_PI * r / n));
  double2 *d_c, *d_tw;
  CK(hipMalloc(&d_c, 3 * Nhp * 16));
  CK(hipMalloc(&d_tw, n / 2 * 16));
  CK(hipMemcpy(d_tw, tw.data(), n / 2 * 16, hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  auto time_it = [&](const char *name, auto launch, double bytes) {
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
    printf("%-50s %.4f ms  %.2f TB/s\n", name, best, bytes / best * 1e-9);
    fflush(stdout);
  };
  // correctness: one launch on fresh data, sampled columns against a host DFT
  CK(hipMemcpy(d_c, hin.data(), 3 * Nhp * 16, hipMemcpyHostToDevice));
  const int grid = 3 * n * (nhp / 8);
  const size_t lds = ((size_t)n * 8 + n / 2) * 16;
  k_ypass<double, 512, 4><<<grid, 512, lds>>>(g, log2n, d_tw, d_c);
  CK(hipGetLastError());
  std::vector<double2> hout(3 * Nhp);
  CK(hipMemcpy(hout.data(), d_c, 3 * Nhp * 16, hipMemcpyDeviceToHost));
  double worst = 0;
  for (int s = 0; s < 12; s++) {
    const int comp = s % 3, i = (s * 37) % n, k = (s * 29) % nh;
    for (int y = 0; y < n; y += 17) {
      std::complex<double> acc = 0;
      for (int j = 0; j < n; j++) {
        const double2 v = hin[(size_t)comp * Nhp + k + (size_t)nhp * (j + (size_t)n * i)];
        acc += std::complex<double>(v.x, v.y) * std::polar(1.0, 2 * M_PI * j * y / n);
      }
      const double2 o = hout[(size_t)comp * Nhp + k + (size_t)nhp * (y + (size_t)n * i)];
      worst = std::max(worst, std::abs(acc - std::complex<double>(o.x, o.y)) / 16.0);
    }
  }
  printf("k_ypass: worst deviation from the host DFT on sampled columns %.3e (values ~ 16)\n", worst);
  time_it("k_ypass<double,512,4> in place, 3 components", [&] { k_ypass<double, 512, 4><<<grid, 512, lds>>>(g, log2n, d_tw, d_c); },
          2.0 * 3 * Nhp * 16);
  time_it("k_ypass<double,256,8> in place, 3 components", [&] { k_ypass<double, 256, 8><<<grid, 256, lds>>>(g, log2n, d_tw, d_c); },
          2.0 * 3 * Nhp * 16);
  time_it("k_ypass<double,512,4,nt>", [&] { k_ypass<double, 512, 4, 1><<<grid, 512, lds>>>(g, log2n, d_tw, d_c); }, 2.0 * 3 * Nhp * 16);
  time_it("k_ypass<double,256,8,nt>", [&] { k_ypass<double, 256, 8, 1><<<grid, 256, lds>>>(g, log2n, d_tw, d_c); }, 2.0 * 3 * Nhp * 16);
  time_it("k_ypass<double,512,4,nt loads>", [&] { k_ypass<double, 512, 4, 2><<<grid, 512, lds>>>(g, log2n, d_tw, d_c); }, 2.0 * 3 * Nhp * 16);
  time_it("k_ypass<double,512,4,nt stores>", [&] { k_ypass<double, 512, 4, 3><<<grid, 512, lds>>>(g, log2n, d_tw, d_c); }, 2.0 * 3 * Nhp * 16);
  time_it("k_ypass<double,1024,2>", [&] { k_ypass<double, 1024, 2><<<grid, 1024, lds>>>(g, log2n, d_tw, d_c); }, 2.0 * 3 * Nhp * 16);
  time_it("k_ypass<double,128,16>", [&] { k_ypass<double, 128, 16><<<grid, 128, lds>>>(g, log2n, d_tw, d_c); }, 2.0 * 3 * Nhp * 16);
  // ---- z pass fused into the binning ----
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
  sp.reach = 2;
  TilePar tp{};
  tp.tx = tp.ty = 8;
  tp.tz = 16;
  tp.ntx = tp.nty = n / 8;
  tp.ntz = n / 16;
  tp.ntiles = tp.ntx * tp.nty * tp.ntz;
  tp.R = 2;
  tp.cap = 16384;
  tp.chunk = 2048;
  const long long N = g.N;
  std::vector<double> psi(3 * N);
  const double amp = 5.0 * g.d;
  srand(1);
  for (int c = 0; c < 3; c++)
    for (int i = 0; i < n; i++)
      for (int j = 0; j < n; j++)
        for (int k = 0; k < n; k++) {
          const double u = 2 * M_PI * i / n, v = 2 * M_PI * j / n, w = 2 * M_PI * k / n;
          double t = sin(u + 0.3 * c) + 0.7 * sin(2 * v + c) + 0.5 * cos(3 * w + 2 * c) + 0.4 * sin(5 * u + 3 * v + w) +
                     0.3 * cos(9 * v - 7 * w + c);
          t += 0.2 * ((rand() & 0xffff) / 65536.0 - 0.5);
          psi[(size_t)c * N + k + (size_t)n * (j + (size_t)n * i)] = amp * t;
        }
  // half-complex spectra of every z row (forward, divided by n: the unnormalised inverse returns psi); host radix-2
  std::vector<double2> hck(3 * Nhp, make_double2(0, 0));
  {
    std::vector<std::complex<double>> buf(n), wt(n / 2);
    for (int r = 0; r < n / 2; r++) wt[r] = std::polar(1.0, -2 * M_PI * r / n);
    for (long long row = 0; row < 3ll * n * n; row++) {
      const double *src = psi.data() + row * n;
      for (int k = 0; k < n; k++) {
        unsigned rk = 0;
        for (int b = 0; b < log2n; b++) rk |= ((k >> b) & 1u) << (log2n - 1 - b);
        buf[rk] = src[k];
      }
      for (int len = 2; len <= n; len <<= 1)
        for (int st = 0; st < n; st += len)
          for (int q = 0; q < len / 2; q++) {
            const std::complex<double> a = buf[st + q], b = buf[st + q + len / 2] * wt[q * (n / len)];
            buf[st + q] = a + b;
            buf[st + q + len / 2] = a - b;
          }
      double2 *dst = hck.data() + row * nhp;  // (comp, i, j) rows are consecutive in both layouts
      for (int k = 0; k < nh; k++) dst[k] = make_double2(buf[k].real() / n, buf[k].imag() / n);
    }
  }
  double *d_psi, *d_psi2, *d_V, *d_rho, *d_zero;
  int *d_cnt, *d_cnt2, *d_ovf;
  RecQuad *d_rec;
  CK(hipMalloc(&d_psi, 3 * N * 8));
  CK(hipMalloc(&d_psi2, 3 * N * 8));
  CK(hipMalloc(&d_V, 3 * N * 8));
  CK(hipMalloc(&d_rho, N * 8));
  CK(hipMalloc(&d_zero, 4096 * 8));
  CK(hipMalloc(&d_cnt, (size_t)tp.ntiles * 8 * 4));
  CK(hipMalloc(&d_cnt2, (size_t)tp.ntiles * 8 * 4));
  CK(hipMalloc(&d_ovf, 64));
  CK(hipMalloc(&d_rec, (size_t)tp.ntiles * tp.cap * 32));
  CK(hipMemcpy(d_psi, psi.data(), 3 * N * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_c, hck.data(), 3 * Nhp * 16, hipMemcpyHostToDevice));
  CK(hipMemset(d_cnt, 0, (size_t)tp.ntiles * 8 * 4));
  CK(hipMemset(d_cnt2, 0, (size_t)tp.ntiles * 8 * 4));
  CK(hipMemset(d_ovf, 0, 64));
  const int nsuper = (n / 4) * (n / 4) * (n / 16) / 4;
  const size_t ldsz = zbin_lds<double>(n);
  k_bin_direct<double><<<nsuper, 256>>>(g, pp, sp, tp, nsuper, d_psi, d_cnt, d_ovf, d_rec, d_V, d_zero, d_rho, nullptr);
  k_zbin_direct<double, 256><<<(n / 2) * (n / 2), 256, ldsz>>>(g, pp, sp, tp, log2n, d_tw, d_c, d_cnt2, d_ovf, d_rec, d_V, d_zero,
                                                         d_rho, nullptr, d_psi2);
  CK(hipGetLastError());
  CK(hipDeviceSynchronize());
  {
    std::vector<double> p2(3 * N);
    std::vector<int> c1((size_t)tp.ntiles * 8), c2((size_t)tp.ntiles * 8);
    CK(hipMemcpy(p2.data(), d_psi2, 3 * N * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(c1.data(), d_cnt, c1.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(c2.data(), d_cnt2, c2.size() * 4, hipMemcpyDeviceToHost));
    double dmax = 0;
    for (size_t q = 0; q < p2.size(); q++) dmax = std::max(dmax, fabs(p2[q] - psi[q]));
    long long t1 = 0, t2 = 0, diff = 0;
    for (size_t q = 0; q < c1.size(); q++) {
      t1 += c1[q];
      t2 += c2[q];
      diff += std::abs(c1[q] - c2[q]);
    }
    printf("k_zbin_direct: max |psi - psi_ref| %.3e (amplitude %.2f); records %lld vs %lld of %lld; counters differ by %lld in total\n",
           dmax, amp, t2, t1, N, diff);
    if (dmax > 1e-11 || t2 != N) return 1;
  }
  auto clear = [&] {
    CK(hipMemsetAsync(d_cnt, 0, (size_t)tp.ntiles * 8 * 4));
    CK(hipMemsetAsync(d_ovf, 0, 64));
  };
  CK(hipMemset(d_ovf, 0, 64));
  time_it("no-op: k_zbin_direct<PSI_ONLY>, 16384 workgroups", [&] {
    k_zbin_direct<double, 256, true><<<(n / 2) * (n / 2), 256, ldsz>>>(g, pp, sp, tp, log2n, d_tw, d_c, d_cnt, d_ovf, nullptr, nullptr,
                                                                    nullptr, nullptr, nullptr, d_psi2);
  }, 0);
  time_it("no-op: k_bin<double> fallback pass, 4096 workgroups", [&] {
    k_bin<double><<<4096, 256>>>(g, pp, sp, tp, (int)(N / 256), d_psi, d_cnt, d_ovf, nullptr, d_V);
  }, 0);
  time_it("no-op: k_bin<double> fallback pass, 1024 workgroups", [&] {
    k_bin<double><<<1024, 256>>>(g, pp, sp, tp, (int)(N / 256), d_psi, d_cnt, d_ovf, nullptr, d_V);
  }, 0);
  time_it("no-op: k_bin<double> fallback pass, 256 workgroups", [&] {
    k_bin<double><<<256, 256>>>(g, pp, sp, tp, (int)(N / 256), d_psi, d_cnt, d_ovf, nullptr, d_V);
  }, 0);
  for (int rep = 0; rep < 2; rep++) {
    time_it("k_bin_direct<double> (reads psi)", [&] {
      clear();
      k_bin_direct<double><<<nsuper, 256>>>(g, pp, sp, tp, nsuper, d_psi, d_cnt, d_ovf, d_rec, d_V, d_zero, d_rho, nullptr);
    }, 0);
    time_it("k_zbin_direct<double> (z pass + binning)", [&] {
      clear();
      k_zbin_direct<double, 256><<<(n / 2) * (n / 2), 256, ldsz>>>(g, pp, sp, tp, log2n, d_tw, d_c, d_cnt, d_ovf, d_rec, d_V,
                                                             d_zero, d_rho, nullptr, nullptr);
    }, 0);
    time_it("k_zbin_direct<double> + psi stored", [&] {
      clear();
      k_zbin_direct<double, 256><<<(n / 2) * (n / 2), 256, ldsz>>>(g, pp, sp, tp, log2n, d_tw, d_c, d_cnt, d_ovf, d_rec, d_V,
                                                             d_zero, d_rho, nullptr, d_psi2);
    }, 0);
  }
  return worst < 1e-12 ? 0 : 1;
}
