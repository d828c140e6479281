// h2d_bench.hip -- what the host-array entry points (bchmc_leapfrog, bchmc_delta_hamiltonian) can expect from the
// PCIe link for one 256^3 double array (134 MB): pageable hipMemcpy, pinned hipMemcpy, hipHostRegister cost, and a
// chunked pipeline "N-thread memcpy into pinned staging | DMA" in both directions.
//   hipcc -O3 --offload-arch=gfx950 -Wno-unused-result scripts/h2d_bench.hip -o scripts/h2d_bench -lpthread
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

static double now() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

static void par_memcpy(void *dst, const void *src, size_t bytes, int nt) {
  if (nt <= 1 || bytes < (1u << 20)) {
    std::memcpy(dst, src, bytes);
    return;
  }
  std::vector<std::thread> th;
  const size_t per = ((bytes / nt) + 4095) & ~(size_t)4095;
  for (int t = 0; t < nt; t++) {
    const size_t off = per * t;
    if (off >= bytes) break;
    const size_t len = std::min(per, bytes - off);
    th.emplace_back([=] { std::memcpy((char *)dst + off, (const char *)src + off, len); });
  }
  for (auto &t : th) t.join();
}

int main() {
  const size_t bytes = (size_t)256 * 256 * 256 * 8;
  void *dev, *pin;
  hipMalloc(&dev, bytes);
  hipHostMalloc(&pin, bytes);
  char *pageable = (char *)malloc(bytes);
  memset(pageable, 1, bytes);
  memset(pin, 1, bytes);
  hipStream_t s;
  hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  hipDeviceSynchronize();
  for (int rep = 0; rep < 2; rep++) {
    double t0 = now();
    hipMemcpyAsync(dev, pageable, bytes, hipMemcpyHostToDevice, s);
    hipStreamSynchronize(s);
    double t1 = now();
    printf("pageable H2D   %6.2f ms  %5.1f GB/s\n", (t1 - t0) * 1e3, bytes / (t1 - t0) / 1e9);
    t0 = now();
    hipMemcpyAsync(pageable, dev, bytes, hipMemcpyDeviceToHost, s);
    hipStreamSynchronize(s);
    t1 = now();
    printf("pageable D2H   %6.2f ms  %5.1f GB/s\n", (t1 - t0) * 1e3, bytes / (t1 - t0) / 1e9);
    t0 = now();
    hipMemcpyAsync(dev, pin, bytes, hipMemcpyHostToDevice, s);
    hipStreamSynchronize(s);
    t1 = now();
    printf("pinned   H2D   %6.2f ms  %5.1f GB/s\n", (t1 - t0) * 1e3, bytes / (t1 - t0) / 1e9);
    t0 = now();
    hipMemcpyAsync(pin, dev, bytes, hipMemcpyDeviceToHost, s);
    hipStreamSynchronize(s);
    t1 = now();
    printf("pinned   D2H   %6.2f ms  %5.1f GB/s\n", (t1 - t0) * 1e3, bytes / (t1 - t0) / 1e9);
  }
  {
    double t0 = now();
    hipError_t e = hipHostRegister(pageable, bytes, hipHostRegisterDefault);
    double t1 = now();
    printf("hipHostRegister(134 MB): %s, %6.2f ms (%5.1f GB/s)\n", hipGetErrorString(e), (t1 - t0) * 1e3,
           bytes / (t1 - t0) / 1e9);
    if (e == hipSuccess) {
      t0 = now();
      hipMemcpyAsync(dev, pageable, bytes, hipMemcpyHostToDevice, s);
      hipStreamSynchronize(s);
      t1 = now();
      printf("registered H2D %6.2f ms  %5.1f GB/s\n", (t1 - t0) * 1e3, bytes / (t1 - t0) / 1e9);
      t0 = now();
      hipHostUnregister(pageable);
      t1 = now();
      printf("hipHostUnregister: %6.2f ms\n", (t1 - t0) * 1e3);
    }
  }
  for (int nt : {1, 2, 4, 8, 16}) {
    double t0 = now();
    par_memcpy(pin, pageable, bytes, nt);
    double t1 = now();
    printf("memcpy pageable -> pinned, %2d threads: %6.2f ms  %5.1f GB/s\n", nt, (t1 - t0) * 1e3,
           bytes / (t1 - t0) / 1e9);
  }
  // chunked pipeline
  for (size_t chunk : {(size_t)8 << 20, (size_t)16 << 20, (size_t)32 << 20})
    for (int nt : {4, 8, 16}) {
      void *stg[2];
      hipEvent_t ev[2];
      for (int b = 0; b < 2; b++) {
        hipHostMalloc(&stg[b], chunk);
        hipEventCreateWithFlags(&ev[b], hipEventDisableTiming);
      }
      const int nch = (int)((bytes + chunk - 1) / chunk);
      double t0 = now();
      for (int c = 0; c < nch; c++) {
        const int b = c & 1;
        const size_t off = (size_t)c * chunk, len = std::min(chunk, bytes - off);
        if (c >= 2) hipEventSynchronize(ev[b]);
        par_memcpy(stg[b], pageable + off, len, nt);
        hipMemcpyAsync((char *)dev + off, stg[b], len, hipMemcpyHostToDevice, s);
        hipEventRecord(ev[b], s);
      }
      hipStreamSynchronize(s);
      double t1 = now();
      double h2d = t1 - t0;
      t0 = now();
      for (int c = 0; c <= nch; c++) {
        const int b = c & 1;
        if (c < nch) {
          const size_t off = (size_t)c * chunk, len = std::min(chunk, bytes - off);
          hipMemcpyAsync(stg[b], (char *)dev + off, len, hipMemcpyDeviceToHost, s);
          hipEventRecord(ev[b], s);
        }
        if (c >= 1) {
          const size_t off = (size_t)(c - 1) * chunk, len = std::min(chunk, bytes - off);
          hipEventSynchronize(ev[b ^ 1]);
          par_memcpy(pageable + off, stg[b ^ 1], len, nt);
        }
      }
      t1 = now();
      printf("pipeline chunk %2zu MB, %2d threads: H2D %6.2f ms %5.1f GB/s | D2H %6.2f ms %5.1f GB/s\n", chunk >> 20, nt,
             h2d * 1e3, bytes / h2d / 1e9, (t1 - t0) * 1e3, bytes / (t1 - t0) / 1e9);
      for (int b = 0; b < 2; b++) {
        hipHostFree(stg[b]);
        hipEventDestroy(ev[b]);
      }
    }
  return 0;
}
