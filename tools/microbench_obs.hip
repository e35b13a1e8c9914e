// Floor of the DTW observation pattern on MI355X: per (column, d) one v_add_f64 (SGPR - VGPR), one v_mul_f64, one
// v_add_f64 onto a sequential accumulator; RPL rows per lane (= independent chains); NO memory traffic inside the loop
// (the 40 "sequence" values sit in SGPRs for the whole kernel).  Occupancy is varied through dynamic LDS.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int RPL, bool SGPR>
__global__ void __launch_bounds__(256) k(const double* __restrict__ sv, const double* __restrict__ tv, double* out, int iters) {
  double tm[RPL][40];
  for (int q = 0; q < RPL; ++q)
    for (int d = 0; d < 40; ++d) tm[q][d] = tv[(threadIdx.x + 256 * q) * 40 + d];
  double s[40];
  for (int d = 0; d < 40; ++d) s[d] = SGPR ? sv[d] : sv[d + (threadIdx.x & 1)];
  double acc[RPL];
  for (int q = 0; q < RPL; ++q) acc[q] = 0;
  for (int it = 0; it < iters; ++it) {
    double o[RPL];
#pragma unroll
    for (int q = 0; q < RPL; ++q) o[q] = 0.0;
#pragma unroll
    for (int d = 0; d < 40; ++d)
#pragma unroll
      for (int q = 0; q < RPL; ++q) {
        const double df = s[d] - tm[q][d];
        const double sq = df * df;
        o[q] = o[q] + sq;
      }
#pragma unroll
    for (int q = 0; q < RPL; ++q) {
      acc[q] += o[q];
#pragma unroll
      for (int d = 0; d < 40; ++d) asm volatile("" : "+v"(tm[q][d]));   // keep the loop body from being hoisted
    }
  }
  double r = 0;
  for (int q = 0; q < RPL; ++q) r += acc[q];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int RPL, bool SGPR>
void run(const char* name, const double* sv, const double* tv, double* out) {
  for (int wps = 1; wps <= 5; ++wps) {            // waves per SIMD = workgroups (4 waves) per CU
    size_t lds = 160 * 1024 / wps - 512;
    if (lds > 64 * 1024) hipFuncSetAttribute((const void*)k<RPL, SGPR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    int blocks = 256 * wps * 4, iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL((k<RPL, SGPR>), dim3(blocks), dim3(256), lds, 0, sv, tv, out, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
    }
    double ops = (double)blocks * 256 * iters * 120.0 * RPL;
    printf("%-28s RPL=%d waves/SIMD(max)=%d  %.3f ms  %.2f T lane-ops/s (%s)\n", name, RPL, wps, ms, ops / ms * 1e-9, hipGetErrorString(hipGetLastError()));
  }
}
int main() {
  double *sv, *tv, *out;
  hipMalloc(&sv, 8 * 64); hipMalloc(&tv, 8 * 40 * 1024); hipMalloc(&out, 8 * 256 * 5 * 4 * 256);
  hipMemset(sv, 0, 8 * 64); hipMemset(tv, 0, 8 * 40 * 1024);
  run<1, true>("sgpr operand", sv, tv, out);
  run<2, true>("sgpr operand", sv, tv, out);
  run<1, false>("vgpr operand", sv, tv, out);
  run<2, false>("vgpr operand", sv, tv, out);
  return 0;
}
