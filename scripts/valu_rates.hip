// valu_rates.hip -- issue cost (shader cycles per wave64 instruction per SIMD) of the fp64 vector instructions the
// particle-mesh kernels are made of, measured with s_memtime around unrolled streams of independent instructions.
//   hipcc -O3 --offload-arch=gfx950 -Wno-unused-result scripts/valu_rates.hip -o scripts/valu_rates && scripts/valu_rates
// Each test runs 1, 2 and 4 waves per SIMD on every CU (occupancy changes what one wave can issue per cycle: the
// microarch guide's "vector-instruction ISSUE cost" row is for one wave alone).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int OP>
__global__ void __launch_bounds__(1024) k_rate(double *out, unsigned long long *cyc, double seed, int iters) {
  __shared__ double lds[2048];
  double a0 = seed + threadIdx.x, a1 = a0 + 1., a2 = a0 + 2., a3 = a0 + 3., a4 = a0 + 4., a5 = a0 + 5., a6 = a0 + 6.,
         a7 = a0 + 7.;
  float f0 = (float)a0;
  lds[threadIdx.x] = 0.;
  lds[threadIdx.x + 1024] = 0.;
  __syncthreads();
  const unsigned lds_addr = (unsigned)((threadIdx.x * 37u) & 2047u) * 8u;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it++) {
    if (OP == 0) {  // v_add_f64
      REP8(asm volatile("v_add_f64 %0, %0, %8\n v_add_f64 %1, %1, %8\n v_add_f64 %2, %2, %8\n v_add_f64 %3, %3, %8\n"
                        "v_add_f64 %4, %4, %8\n v_add_f64 %5, %5, %8\n v_add_f64 %6, %6, %8\n v_add_f64 %7, %7, %8\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                        : "v"(seed));)
    } else if (OP == 1) {  // v_fma_f64
      REP8(asm volatile("v_fma_f64 %0, %0, %8, %8\n v_fma_f64 %1, %1, %8, %8\n v_fma_f64 %2, %2, %8, %8\n"
                        "v_fma_f64 %3, %3, %8, %8\n v_fma_f64 %4, %4, %8, %8\n v_fma_f64 %5, %5, %8, %8\n"
                        "v_fma_f64 %6, %6, %8, %8\n v_fma_f64 %7, %7, %8, %8\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                        : "v"(seed));)
    } else if (OP == 2) {  // v_mul_f64
      REP8(asm volatile("v_mul_f64 %0, %0, %8\n v_mul_f64 %1, %1, %8\n v_mul_f64 %2, %2, %8\n v_mul_f64 %3, %3, %8\n"
                        "v_mul_f64 %4, %4, %8\n v_mul_f64 %5, %5, %8\n v_mul_f64 %6, %6, %8\n v_mul_f64 %7, %7, %8\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                        : "v"(seed));)
    } else if (OP == 3) {  // v_rsq_f64
      REP8(asm volatile("v_rsq_f64 %0, %0\n v_rsq_f64 %1, %1\n v_rsq_f64 %2, %2\n v_rsq_f64 %3, %3\n"
                        "v_rsq_f64 %4, %4\n v_rsq_f64 %5, %5\n v_rsq_f64 %6, %6\n v_rsq_f64 %7, %7\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
    } else if (OP == 4) {  // v_cmp_le_f64 (writes vcc)
      REP8(asm volatile("v_cmp_le_f64 vcc, %0, %8\n v_cmp_le_f64 vcc, %1, %8\n v_cmp_le_f64 vcc, %2, %8\n"
                        "v_cmp_le_f64 vcc, %3, %8\n v_cmp_le_f64 vcc, %4, %8\n v_cmp_le_f64 vcc, %5, %8\n"
                        "v_cmp_le_f64 vcc, %6, %8\n v_cmp_le_f64 vcc, %7, %8\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                        : "v"(seed)
                        : "vcc");)
    } else if (OP == 5) {  // v_rsq_f32
      REP64(asm volatile("v_rsq_f32 %0, %0\n" : "+v"(f0));)
    } else if (OP == 6) {  // v_cvt_f32_f64 + v_rsq_f32 + v_cvt_f64_f32 (8 triples)
      REP8(asm volatile("v_cvt_f32_f64 %1, %0\n v_rsq_f32 %1, %1\n v_cvt_f64_f32 %0, %1\n" : "+v"(a0), "+v"(f0));
           asm volatile("v_cvt_f32_f64 %1, %0\n v_rsq_f32 %1, %1\n v_cvt_f64_f32 %0, %1\n" : "+v"(a1), "+v"(f0));
           asm volatile("v_cvt_f32_f64 %1, %0\n v_rsq_f32 %1, %1\n v_cvt_f64_f32 %0, %1\n" : "+v"(a2), "+v"(f0));
           asm volatile("v_cvt_f32_f64 %1, %0\n v_rsq_f32 %1, %1\n v_cvt_f64_f32 %0, %1\n" : "+v"(a3), "+v"(f0));
           asm volatile("v_cvt_f32_f64 %1, %0\n v_rsq_f32 %1, %1\n v_cvt_f64_f32 %0, %1\n" : "+v"(a4), "+v"(f0));
           asm volatile("v_cvt_f32_f64 %1, %0\n v_rsq_f32 %1, %1\n v_cvt_f64_f32 %0, %1\n" : "+v"(a5), "+v"(f0));
           asm volatile("v_cvt_f32_f64 %1, %0\n v_rsq_f32 %1, %1\n v_cvt_f64_f32 %0, %1\n" : "+v"(a6), "+v"(f0));
           asm volatile("v_cvt_f32_f64 %1, %0\n v_rsq_f32 %1, %1\n v_cvt_f64_f32 %0, %1\n" : "+v"(a7), "+v"(f0));)
    } else if (OP == 7) {  // ds_add_f64, pseudo-random cells of a 16 KB tile
      REP8(asm volatile("ds_add_f64 %0, %1\n ds_add_f64 %0, %1 offset:8\n ds_add_f64 %0, %1 offset:16\n"
                        "ds_add_f64 %0, %1 offset:24\n ds_add_f64 %0, %1 offset:32\n ds_add_f64 %0, %1 offset:40\n"
                        "ds_add_f64 %0, %1 offset:48\n ds_add_f64 %0, %1 offset:56\n"
                        :
                        : "v"(lds_addr & 0x3fc0u), "v"(a0)
                        : "memory");)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    } else if (OP == 8) {  // v_cndmask_b32 pair (fp64 select)
      REP64(asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n" : "+v"(f0) : "v"(f0) : "vcc");)
    } else if (OP == 9) {  // v_add_f32 (reference point)
      REP64(asm volatile("v_add_f32 %0, %0, %1\n" : "+v"(f0) : "v"(f0));)
    } else if (OP == 10) {  // v_max_f64
      REP8(asm volatile("v_max_f64 %0, %0, %8\n v_max_f64 %1, %1, %8\n v_max_f64 %2, %2, %8\n v_max_f64 %3, %3, %8\n"
                        "v_max_f64 %4, %4, %8\n v_max_f64 %5, %5, %8\n v_max_f64 %6, %6, %8\n v_max_f64 %7, %7, %8\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                        : "v"(seed));)
    } else if (OP == 11) {  // v_sqrt_f64
      REP8(asm volatile("v_sqrt_f64 %0, %0\n v_sqrt_f64 %1, %1\n v_sqrt_f64 %2, %2\n v_sqrt_f64 %3, %3\n"
                        "v_sqrt_f64 %4, %4\n v_sqrt_f64 %5, %5\n v_sqrt_f64 %6, %6\n v_sqrt_f64 %7, %7\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (double)f0 + lds[threadIdx.x];
}

template <int OP>
void run(const char *name, int per_iter) {
  const int iters = 200, ncu = 256;
  printf("%-44s", name);
  for (int waves_per_simd : {1, 2, 4}) {
    const int threads = 256 * waves_per_simd;  // one workgroup per CU: 4 SIMDs x waves_per_simd waves
    double *out;
    unsigned long long *cyc;
    hipMalloc(&out, sizeof(double) * ncu * threads);
    hipMalloc(&cyc, sizeof(unsigned long long) * ncu * (threads / 64));
    k_rate<OP><<<ncu, threads>>>(out, cyc, 1.000001, 10);  // warm
    k_rate<OP><<<ncu, threads>>>(out, cyc, 1.000001, iters);
    std::vector<unsigned long long> h(ncu * (threads / 64));
    hipMemcpy(h.data(), cyc, h.size() * sizeof(h[0]), hipMemcpyDeviceToHost);
    double sum = 0;
    for (auto v : h) sum += (double)v;
    const double per_wave = sum / h.size() / ((double)iters * per_iter);  // s_memtime ticks per instruction, one wave
    // per-SIMD throughput cost: a SIMD ran waves_per_simd such streams concurrently
    printf("  %dw/SIMD: %6.2f cyc/inst/wave = %5.2f cyc/inst/SIMD", waves_per_simd, per_wave, per_wave / waves_per_simd);
    hipFree(out);
    hipFree(cyc);
  }
  printf("\n");
}

int main() {
  run<9>("v_add_f32", 64);
  run<0>("v_add_f64", 64);
  run<2>("v_mul_f64", 64);
  run<1>("v_fma_f64", 64);
  run<10>("v_max_f64", 64);
  run<4>("v_cmp_le_f64", 64);
  run<8>("v_cndmask_b32", 64);
  run<3>("v_rsq_f64", 64);
  run<11>("v_sqrt_f64", 64);
  run<5>("v_rsq_f32", 64);
  run<6>("cvt_f32_f64 + rsq_f32 + cvt_f64_f32 (triple)", 64);
  run<7>("ds_add_f64 (scattered cells)", 64);
  return 0;
}
