// Micro-benchmark: rocFFT batched 3-D real transforms (n^3 double, batch 3) with the half-complex array stored
// (a) contiguously (row length n/2+1 complex) and (b) with padded rows / planes.  Build & run on the GPU box:
//   hipcc -O2 --offload-arch=gfx950 scripts/fft_layout_bench.hip -o /tmp/fftb -lrocfft && /tmp/fftb 256
#include <hip/hip_runtime.h>
#include <rocfft/rocfft.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

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
  const size_t nh = n / 2 + 1;
  CK(rocfft_setup());
  hipStream_t st; hipStreamCreate(&st);
  const size_t len[3] = {n, n, n};
  struct Layout { char name[64]; size_t row, plane; };
  std::vector<Layout> layouts;
  for (size_t pr : {0, 1, 2, 3, 4, 5, 7, 8, 11, 15, 16, 31})
    for (size_t pp : {0, 1, 4}) {
      Layout L;
      L.row = nh + pr;
      L.plane = L.row * (n + pp);
      snprintf(L.name, sizeof L.name, "row %zu, plane rows %zu", L.row, n + pp);
      layouts.push_back(L);
    }
  for (auto &L : layouts) {
    const size_t cdist = L.plane * n;
    double *R; double2 *C;
    hipMalloc(&R, 3 * n * n * n * esz);
    hipMalloc(&C, 3 * cdist * 2 * esz);
    hipMemset(R, 0, 3 * n * n * n * esz);
    hipMemset(C, 0, 3 * cdist * 2 * esz);
    for (int dir = 0; dir < 2; dir++) {
      rocfft_plan_description d; CK(rocfft_plan_description_create(&d));
      size_t rstr[3] = {1, n, n * n}, cstr[3] = {1, L.row, L.plane};
      if (dir == 0)
        CK(rocfft_plan_description_set_data_layout(d, rocfft_array_type_real, rocfft_array_type_hermitian_interleaved,
                                                   nullptr, nullptr, 3, rstr, n * n * n, 3, cstr, cdist));
      else
        CK(rocfft_plan_description_set_data_layout(d, rocfft_array_type_hermitian_interleaved, rocfft_array_type_real,
                                                   nullptr, nullptr, 3, cstr, cdist, 3, rstr, n * n * n));
      rocfft_plan plan;
      CK(rocfft_plan_create(&plan, rocfft_placement_notinplace,
                            dir == 0 ? rocfft_transform_type_real_forward : rocfft_transform_type_real_inverse,
                            f32 ? rocfft_precision_single : rocfft_precision_double, 3, len, 3, d));
      size_t wb = 0; CK(rocfft_plan_get_work_buffer_size(plan, &wb));
      void *work = nullptr; if (wb) hipMalloc(&work, wb);
      rocfft_execution_info info; CK(rocfft_execution_info_create(&info));
      CK(rocfft_execution_info_set_stream(info, st));
      if (wb) CK(rocfft_execution_info_set_work_buffer(info, work, wb));
      const double ms = dir == 0 ? time_plan(plan, R, C, st, info, 20) : time_plan(plan, C, R, st, info, 20);
      printf("%s n=%zu %-26s %s batch3: %.3f ms (work buffer %.1f MB)\n", f32 ? "f32" : "f64", n, L.name, dir == 0 ? "R2C" : "C2R", ms, wb / 1e6);
      rocfft_plan_destroy(plan); rocfft_execution_info_destroy(info); rocfft_plan_description_destroy(d);
      if (work) hipFree(work);
    }
    hipFree(R); hipFree(C);
  }
  rocfft_cleanup();
  return 0;
}
