// microbench_pcie.hip -- what the host-pointer entry points (vcmi_gmmmap_convert & co.) can reach over PCIe:
// pageable hipMemcpy, pinned hipMemcpyAsync, hipHostRegister cost, threaded memcpy into pinned staging, and
// full-duplex H2D + D2H on two streams.  Build: hipcc --offload-arch=gfx950 -O2 -pthread -o microbench_pcie microbench_pcie.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#define CK(x)                                                                       \
  do {                                                                              \
    hipError_t e = (x);                                                             \
    if (e != hipSuccess) {                                                          \
      printf("%s failed: %s\n", #x, hipGetErrorString(e));                          \
      exit(1);                                                                      \
    }                                                                               \
  } while (0)

static double now() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

static void par_memcpy(char *dst, const char *src, size_t n, int nthr) {
  if (nthr <= 1) {
    memcpy(dst, src, n);
    return;
  }
  std::vector<std::thread> th;
  size_t per = (n / nthr + 4095) & ~(size_t)4095;
  for (int i = 0; i < nthr; ++i) {
    size_t lo = (size_t)i * per, hi = lo + per > n ? n : lo + per;
    if (lo >= n) break;
    th.emplace_back([=] { memcpy(dst + lo, src + lo, hi - lo); });
  }
  for (auto &t : th) t.join();
}

int main() {
  const size_t N = (size_t)320 << 20;
  printf("host threads: %u\n", std::thread::hardware_concurrency());
  char *pg_in = (char *)malloc(N), *pg_out = (char *)malloc(N);
  memset(pg_in, 1, N);
  memset(pg_out, 2, N);
  char *d_in, *d_out, *pin_in, *pin_out;
  CK(hipMalloc(&d_in, N));
  CK(hipMalloc(&d_out, N));
  CK(hipHostMalloc(&pin_in, N, hipHostMallocDefault));
  CK(hipHostMalloc(&pin_out, N, hipHostMallocDefault));
  memset(pin_in, 1, N);
  memset(pin_out, 1, N);
  hipStream_t s1, s2;
  CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  double t;
  for (int rep = 0; rep < 2; ++rep) {
    t = now();
    CK(hipMemcpy(d_in, pg_in, N, hipMemcpyHostToDevice));
    printf("pageable H2D 320MB: %.2f ms\n", (now() - t) * 1e3);
    t = now();
    CK(hipMemcpy(pg_out, d_out, N, hipMemcpyDeviceToHost));
    printf("pageable D2H 320MB: %.2f ms\n", (now() - t) * 1e3);
  }
  for (int rep = 0; rep < 2; ++rep) {
    t = now();
    CK(hipMemcpyAsync(d_in, pin_in, N, hipMemcpyHostToDevice, s1));
    CK(hipStreamSynchronize(s1));
    printf("pinned H2D 320MB: %.2f ms\n", (now() - t) * 1e3);
    t = now();
    CK(hipMemcpyAsync(pin_out, d_out, N, hipMemcpyDeviceToHost, s2));
    CK(hipStreamSynchronize(s2));
    printf("pinned D2H 320MB: %.2f ms\n", (now() - t) * 1e3);
    t = now();
    CK(hipMemcpyAsync(d_in, pin_in, N, hipMemcpyHostToDevice, s1));
    CK(hipMemcpyAsync(pin_out, d_out, N, hipMemcpyDeviceToHost, s2));
    CK(hipStreamSynchronize(s1));
    CK(hipStreamSynchronize(s2));
    printf("pinned full duplex 2x320MB: %.2f ms\n", (now() - t) * 1e3);
  }
  for (int rep = 0; rep < 2; ++rep) {
    t = now();
    CK(hipHostRegister(pg_in, N, hipHostRegisterDefault));
    double tr = now() - t;
    t = now();
    CK(hipMemcpyAsync(d_in, pg_in, N, hipMemcpyHostToDevice, s1));
    CK(hipStreamSynchronize(s1));
    double tc = now() - t;
    t = now();
    CK(hipHostUnregister(pg_in));
    printf("hipHostRegister 320MB: %.2f ms, copy %.2f ms, unregister %.2f ms\n", tr * 1e3, tc * 1e3, (now() - t) * 1e3);
  }
  // registration in 16 MB pieces (pipelinable)
  {
    const size_t P = (size_t)16 << 20;
    t = now();
    for (size_t o = 0; o < N; o += P) CK(hipHostRegister(pg_out + o, P, hipHostRegisterDefault));
    double tr = now() - t;
    t = now();
    for (size_t o = 0; o < N; o += P) CK(hipHostUnregister(pg_out + o));
    printf("hipHostRegister 20 x 16MB: %.2f ms, unregister %.2f ms\n", tr * 1e3, (now() - t) * 1e3);
  }
  for (int nthr : {1, 2, 4, 8, 16}) {
    par_memcpy(pin_in, pg_in, N, nthr);
    t = now();
    par_memcpy(pin_in, pg_in, N, nthr);
    double a = now() - t;
    t = now();
    par_memcpy(pg_out, pin_out, N, nthr);
    double b = now() - t;
    printf("memcpy 320MB with %2d threads: pageable->pinned %.2f ms (%.1f GB/s), pinned->pageable %.2f ms (%.1f GB/s)\n", nthr,
           a * 1e3, N / a / 1e9, b * 1e3, N / b / 1e9);
  }
  // chunked pageable hipMemcpyAsync (the runtime's own staging), 8 MB chunks on one stream
  {
    const size_t P = (size_t)8 << 20;
    t = now();
    for (size_t o = 0; o < N; o += P) CK(hipMemcpyAsync(d_in + o, pg_in + o, P, hipMemcpyHostToDevice, s1));
    CK(hipStreamSynchronize(s1));
    printf("pageable hipMemcpyAsync 40 x 8MB: %.2f ms\n", (now() - t) * 1e3);
  }
  return 0;
}
