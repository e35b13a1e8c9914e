// FP64 VALU op rates on MI355X: v_fma_f64 vs v_add_f64 vs v_mul_f64 (8 independent chains per lane, 2 blocks/CU).
#include <hip/hip_runtime.h>
#include <cstdio>
template <int OP>
__global__ void __launch_bounds__(256) k(double* out, int iters, double seed) {
  double a = seed + threadIdx.x * 1e-9, b = 1.0 - 1e-9 * threadIdx.x;
  double x[8];
  for (int i = 0; i < 8; ++i) x[i] = a + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (OP == 0) x[i] = __builtin_fma(x[i], b, a);
        if (OP == 1) x[i] = x[i] + b;
        if (OP == 2) x[i] = x[i] * b;
        if (OP == 3) { double d = x[i] - a; x[i] = x[i] + d * d; }   // sub, mul, add unfused? (compiled with contract off)
      }
  }
  double s = 0; for (int i = 0; i < 8; ++i) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
  int blocks = 256 * 4, threads = 256, iters = 20000;
  double* out; hipMalloc(&out, sizeof(double) * blocks * threads);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const char* names[4] = {"v_fma_f64", "v_add_f64", "v_mul_f64", "sub+mul+add"};
  for (int op = 0; op < 4; ++op) for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    if (op == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(threads), 0, 0, out, iters, 0.5);
    if (op == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(threads), 0, 0, out, iters, 0.5);
    if (op == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(threads), 0, 0, out, iters, 0.5);
    if (op == 3) hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(threads), 0, 0, out, iters, 0.5);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double instr = (double)blocks * threads / 64 * iters * 64 * (op == 3 ? 3 : 1);   // wave-instructions
    if (rep) printf("%-12s %.3f ms  %.2f T wave-lane-ops/s  (%.2f cycles per wave-instruction per SIMD at 2.4 GHz)\n", names[op], ms,
                    instr * 64 / ms * 1e-9, ms * 1e-3 * 2.4e9 * 1024 / instr);
  }
  return 0;
}
