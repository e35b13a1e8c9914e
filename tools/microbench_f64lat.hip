// microbench_f64lat.hip -- issue interval and dependent latency (cycles, one wave on one SIMD) of the FP64 instructions the
// trajectory solver's scalar chain is made of: v_fma_f64, v_fmac_f64_dpp row_newbcast, v_mov_b64_dpp, v_rcp_f64, v_rsq_f64.
// build: hipcc -O3 --offload-arch=gfx950 tools/microbench_f64lat.hip -o tools/microbench_f64lat
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ long long now() {
  long long t;
  asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}
#define REP16(X) X X X X X X X X X X X X X X X X
__global__ void k(long long *out, double *sink) {
  const int lane = threadIdx.x;
  double a0 = 1.0 + lane * 1e-3, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  double m = 1e-9 * lane;
  long long t[12];
  t[0] = now();
  // 1: independent v_fma_f64 (8 registers round robin), 128 instructions
  for (int r = 0; r < 16; ++r)
    asm volatile("v_fma_f64 %0, %8, %8, %0\n v_fma_f64 %1, %8, %8, %1\n v_fma_f64 %2, %8, %8, %2\n v_fma_f64 %3, %8, %8, %3\n"
                 "v_fma_f64 %4, %8, %8, %4\n v_fma_f64 %5, %8, %8, %5\n v_fma_f64 %6, %8, %8, %6\n v_fma_f64 %7, %8, %8, %7\n"
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));
  t[1] = now();
  // 2: dependent v_fma_f64 chain, 128
  for (int r = 0; r < 16; ++r)
    asm volatile(REP16("v_fma_f64 %0, %0, %1, %1\n") REP16("") : "+v"(a0) : "v"(m));
  t[2] = now();
  for (int r = 0; r < 7; ++r) asm volatile(REP16("v_fma_f64 %0, %0, %1, %1\n") : "+v"(a0) : "v"(m));
  t[3] = now();   // (t3 - t2) = 112 dependent fma
  // 4: independent v_fmac_f64_dpp, 128
  for (int r = 0; r < 16; ++r)
    asm volatile("v_fmac_f64_dpp %0, %0, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %1, %1, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
                 "v_fmac_f64_dpp %2, %2, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %3, %3, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
                 "v_fmac_f64_dpp %4, %4, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %5, %5, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
                 "v_fmac_f64_dpp %6, %6, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %7, %7, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));
  t[4] = now();
  // 5: dependent v_fmac_f64_dpp chain (needs 2 wait states before a DPP read of a VALU result), 128
  for (int r = 0; r < 8; ++r)
    asm volatile(REP16("s_nop 1\n v_fmac_f64_dpp %0, %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf\n") : "+v"(a0) : "v"(m));
  t[5] = now();
  // 6: dependent v_rcp_f64 chain, 64
  for (int r = 0; r < 4; ++r) asm volatile(REP16("v_rcp_f64 %0, %0\n") : "+v"(a1));
  t[6] = now();
  // 7: dependent v_mov_b64_dpp + v_mul_f64 pairs, 64 pairs
  for (int r = 0; r < 4; ++r)
    asm volatile(REP16("s_nop 1\n v_mov_b64_dpp %1, %0 row_newbcast:5 row_mask:0xf bank_mask:0xf\n v_mul_f64 %0, %1, %2\n") : "+v"(a2), "+v"(a3) : "v"(m));
  t[7] = now();
  // 8: dependent v_rsq_f64, 64
  for (int r = 0; r < 4; ++r) asm volatile(REP16("v_rsq_f64 %0, %0\n") : "+v"(a4));
  t[8] = now();
  // 9: independent v_rcp_f64 (4 registers), 64
  for (int r = 0; r < 16; ++r) asm volatile("v_rcp_f64 %0, %0\n v_rcp_f64 %1, %1\n v_rcp_f64 %2, %2\n v_rcp_f64 %3, %3\n" : "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
  t[9] = now();
  // 10: dependent v_mul_f64 chain, 128
  for (int r = 0; r < 8; ++r) asm volatile(REP16("v_mul_f64 %0, %0, %1\n") : "+v"(a0) : "v"(m));
  t[10] = now();
  // 11: dependent v_add_f64 / v_cndmask pairs: skip
  if (lane == 0) {
    out[0] = (t[1] - t[0]);        // /128
    out[1] = (t[2] - t[1]);        // /256 (REP16 twice? no: REP16 + empty) -> /256? see host
    out[2] = (t[3] - t[2]);        // /112
    out[3] = (t[4] - t[3]);        // /128
    out[4] = (t[5] - t[4]);        // /128
    out[5] = (t[6] - t[5]);        // /64
    out[6] = (t[7] - t[6]);        // /64
    out[7] = (t[8] - t[7]);        // /64
    out[8] = (t[9] - t[8]);        // /64
    out[9] = (t[10] - t[9]);       // /128
  }
  sink[lane] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
int main() {
  long long *d, h[10];
  double *sink;
  (void)hipMalloc(&d, sizeof(h));
  (void)hipMalloc(&sink, 64 * 8);
  for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, sink);
  (void)hipDeviceSynchronize();
  (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  printf("v_fma_f64 independent:            %.1f cycles each\n", h[0] / 128.0);
  printf("v_fma_f64 dependent:              %.1f\n", h[2] / 112.0);
  printf("v_fmac_f64_dpp independent:       %.1f\n", h[3] / 128.0);
  printf("s_nop 1 + v_fmac_f64_dpp dependent: %.1f\n", h[4] / 128.0);
  printf("v_rcp_f64 dependent:              %.1f\n", h[5] / 64.0);
  printf("s_nop 1 + v_mov_b64_dpp + v_mul_f64 dependent pair: %.1f\n", h[6] / 64.0);
  printf("v_rsq_f64 dependent:              %.1f\n", h[7] / 64.0);
  printf("v_rcp_f64 independent:            %.1f\n", h[8] / 64.0);
  printf("v_mul_f64 dependent:              %.1f\n", h[9] / 128.0);
  return 0;
}
