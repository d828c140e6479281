// Micro-benchmark: rocFFT batched 2-D real transforms over the (y, z) planes of an n^3 grid (batch 3 n planes,
// half-complex rows padded to nhp) against the full 3-D batched transform -- the question is whether
// "2-D planes by rocFFT + the x pass fused into the k-space kernel" can beat three rocFFT passes.
//   hipcc -O2 --offload-arch=gfx950 scripts/fft2d_bench.hip -o /tmp/fft2d -lrocfft && /tmp/fft2d 256 [32]
#include <hip/hip_runtime.h>
#include <rocfft/rocfft.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { auto e_ = (x); if (e_ != 0) { printf("error %d at %s:%d\n", (int)e_, __FILE__, __LINE__); exit(1);} } while (0)

static double time_plan(rocfft_plan plan, void *in, void *out, hipStream_t st, rocfft_execution_info info, int reps) {
  void *ib[1] = {in}, *ob[1] = {out};
  for (int i = 0; i < 3; i++) CK(rocfft_execute(plan, ib, ob, info));
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  hipEventRecord(a, st);
  for (int i = 0; i < reps; i++) CK(rocfft_execute(plan, ib, ob, info));
  hipEventRecord(b, st);
  hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms / reps;
}

int main(int argc, char **argv) {
  const size_t n = argc > 1 ? atoi(argv[1]) : 256;
  const bool f32 = argc > 2 && atoi(argv[2]) == 32;
  const size_t esz = f32 ? 4 : 8;
  const size_t nh = n / 2 + 1, per = 128 / (2 * esz), nhp = (nh + per - 1) / per * per;
  CK(rocfft_setup());
  hipStream_t st; hipStreamCreate(&st);
  void *R, *C;
  hipMalloc(&R, 3 * n * n * n * esz);
  hipMalloc(&C, 3 * n * n * nhp * 2 * esz);
  hipMemset(R, 0, 3 * n * n * n * esz);
  hipMemset(C, 0, 3 * n * n * nhp * 2 * esz);
  for (int dims = 2; dims <= 3; dims++)
    for (int dir = 0; dir < 2; dir++) {
      rocfft_plan_description d; CK(rocfft_plan_description_create(&d));
      size_t len[3] = {n, n, n};
      size_t rstr[3] = {1, n, n * n}, cstr[3] = {1, nhp, nhp * n};
      const size_t batch = dims == 2 ? 3 * n : 3;
      const size_t rdist = dims == 2 ? n * n : n * n * n, cdist = dims == 2 ? n * nhp : n * n * nhp;
      if (dir == 0)
        CK(rocfft_plan_description_set_data_layout(d, rocfft_array_type_real, rocfft_array_type_hermitian_interleaved,
                                                   nullptr, nullptr, dims, rstr, rdist, dims, cstr, cdist));
      else
        CK(rocfft_plan_description_set_data_layout(d, rocfft_array_type_hermitian_interleaved, rocfft_array_type_real,
                                                   nullptr, nullptr, dims, cstr, cdist, dims, rstr, rdist));
      rocfft_plan plan;
      CK(rocfft_plan_create(&plan, rocfft_placement_notinplace,
                            dir == 0 ? rocfft_transform_type_real_forward : rocfft_transform_type_real_inverse,
                            f32 ? rocfft_precision_single : rocfft_precision_double, dims, len, batch, d));
      size_t wb = 0; CK(rocfft_plan_get_work_buffer_size(plan, &wb));
      void *work = nullptr; if (wb) hipMalloc(&work, wb);
      rocfft_execution_info info; CK(rocfft_execution_info_create(&info));
      CK(rocfft_execution_info_set_stream(info, st));
      if (wb) CK(rocfft_execution_info_set_work_buffer(info, work, wb));
      const double ms = dir == 0 ? time_plan(plan, R, C, st, info, 20) : time_plan(plan, C, R, st, info, 20);
      printf("%s n=%zu %dD %s x3 components: %.3f ms (work buffer %.1f MB)\n", f32 ? "f32" : "f64", n, dims,
             dir == 0 ? "R2C" : "C2R", ms, wb / 1e6);
      rocfft_plan_destroy(plan); rocfft_execution_info_destroy(info); rocfft_plan_description_destroy(d);
      if (work) hipFree(work);
    }
  rocfft_cleanup();
  return 0;
}
