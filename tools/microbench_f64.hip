// FP64 rate microbenchmark for MI355X (gfx950): VALU v_fma_f64, MFMA v_mfma_f64_16x16x4_f64,
// and both at once (separate waves on one SIMD) -- confirms the FP64 roof used by bench.py's roofline.
// Build: hipcc --offload-arch=gfx950 -O3 tools/microbench_f64.hip -o tools/microbench_f64
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)

// mode 0: all waves VALU; 1: all waves MFMA; 2: even waves MFMA, odd waves VALU
__global__ void __launch_bounds__(512) k_rate(double* out, int iters, int mode, double seed) {
  int wave = threadIdx.x >> 6;
  bool do_mfma = (mode == 1) || (mode == 2 && (wave & 4));   // waves 4..7 share SIMDs with waves 0..3
  double a = seed + threadIdx.x * 1e-9, b = 1.0 - 1e-9 * threadIdx.x;
  if (do_mfma) {
    d4 c0 = {0,0,0,0}, c1 = c0, c2 = c0, c3 = c0;
    for (int i = 0; i < iters; ++i) {
      c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
    }
    d4 s = c0 + c1 + c2 + c3;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
  } else {
    double x0 = a, x1 = a + 1, x2 = a + 2, x3 = a + 3, x4 = a + 4, x5 = a + 5, x6 = a + 6, x7 = a + 7;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {   // 64 FMAs per iteration, 8 independent chains
        x0 = __builtin_fma(x0, b, a); x1 = __builtin_fma(x1, b, a); x2 = __builtin_fma(x2, b, a); x3 = __builtin_fma(x3, b, a);
        x4 = __builtin_fma(x4, b, a); x5 = __builtin_fma(x5, b, a); x6 = __builtin_fma(x6, b, a); x7 = __builtin_fma(x7, b, a);
      }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
  }
}

int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  printf("device %s CUs %d clock %d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
  int blocks = p.multiProcessorCount * 2, threads = 512;
  double* out; CK(hipMalloc(&out, sizeof(double) * blocks * threads));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int thr : {256, 512}) for (int mode = 0; mode < 3; ++mode) {
    if (mode == 2 && thr == 256) continue;
    int iters = 20000;
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(k_rate, dim3(blocks), dim3(thr), 0, 0, out, iters, mode, 0.5);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      double waves = (double)blocks * thr / 64;
      double mf = 4.0 * 2048 * iters, vf = 64.0 * 64 * 2 * iters;  // flop per wave
      double flop = mode == 0 ? waves * vf : mode == 1 ? waves * mf : waves / 2 * (mf + vf);
      if (rep == 2) printf("threads %d mode %d (%s): %.3f ms  %.2f TFLOP/s f64\n", thr, mode,
                           mode == 0 ? "VALU fma" : mode == 1 ? "MFMA 16x16x4" : "MFMA+VALU co-resident", ms, flop / ms * 1e-9);
    }
  }
  return 0;
}
