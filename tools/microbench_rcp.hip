// microbench_rcp.hip -- accuracy of v_rcp_f64 / v_rsq_f64 (hardware estimates) and of the estimate after one and two Newton
// steps, over 2^24 random doubles: how many refinement steps the pivot chain of the trajectory solver needs.
// build: hipcc -O3 --offload-arch=gfx950 tools/microbench_rcp.hip -o tools/microbench_rcp
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdint>
__global__ void k(double *maxerr, uint64_t seed) {
  uint64_t s = seed + 0x9E3779B97F4A7C15ull * (blockIdx.x * blockDim.x + threadIdx.x + 1);
  double e[6] = {0, 0, 0, 0, 0, 0};
  for (int it = 0; it < 256; ++it) {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    // mantissa uniform, exponent in [-20, 20]
    const double mant = 1.0 + (double)(s >> 12) * 0x1p-52;
    const double x = ldexp(mant, (int)((s >> 3) % 41) - 20);
    double w0 = __builtin_amdgcn_rcp(x);
    double w1 = w0 * fma(-x, w0, 2.0);
    double w2 = w1 * fma(-x, w1, 2.0);
    // relative error of w as 1/x: |x w - 1| evaluated exactly by fma
    e[0] = fmax(e[0], fabs(fma(x, w0, -1.0)));
    e[1] = fmax(e[1], fabs(fma(x, w1, -1.0)));
    e[2] = fmax(e[2], fabs(fma(x, w2, -1.0)));
    double r0 = __builtin_amdgcn_rsq(x);
    double h = 0.5 * x * r0;                       // one Newton step of rsqrt: r1 = r0 (1.5 - 0.5 x r0^2)
    double r1 = r0 * fma(-h, r0, 1.5);
    h = 0.5 * x * r1;
    double r2 = r1 * fma(-h, r1, 1.5);
    e[3] = fmax(e[3], fabs(fma(x * r0, r0, -1.0)) * 0.5);
    e[4] = fmax(e[4], fabs(fma(x * r1, r1, -1.0)) * 0.5);
    e[5] = fmax(e[5], fabs(fma(x * r2, r2, -1.0)) * 0.5);
  }
  for (int q = 0; q < 6; ++q) {
    // max over the grid through ordered-int atomics (non-negative doubles order like their bit patterns)
    atomicMax(reinterpret_cast<unsigned long long *>(maxerr + q), (unsigned long long)__double_as_longlong(e[q]));
  }
}
int main() {
  double *d, h[6] = {0};
  hipMalloc(&d, sizeof(h));
  hipMemset(d, 0, sizeof(h));
  hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, 0, d, 12345ull);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char *names[6] = {"v_rcp_f64", "  + 1 Newton step", "  + 2 Newton steps", "v_rsq_f64", "  + 1 Newton step", "  + 2 Newton steps"};
  for (int q = 0; q < 6; ++q) printf("%-20s max relative error %.3e = 2^%.1f\n", names[q], h[q], log2(h[q] > 0 ? h[q] : 1e-300));
  return 0;
}
