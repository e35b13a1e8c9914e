// microbench_mfma_valu.hip -- what a VALU instruction costs NEXT TO v_mfma_f64_16x16x4_f64 (round 6).
// A wave issues 4 independent FP64 MFMAs per iteration, each followed by K independent VALU instructions of one kind;
// cycles of a SIMD per MFMA (s_memtime of the workgroup's slowest wave / MFMAs issued on the SIMD) for K = 0, 1, 2, 4, 8 at one
// and at two waves per SIMD (all CUs busy).  Fillers of one wave rotate over four registers (no dependent chain) except the
// 32-bit ones, which write one register from itself.
//   hipcc --offload-arch=gfx950 -O3 -o tools/microbench_mfma_valu tools/microbench_mfma_valu.hip && tools/microbench_mfma_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int KIND>
__device__ __forceinline__ void filler(double &a, double &b, float &f, int &i) {
  if constexpr (KIND == 1) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(a) : "v"(b));
  if constexpr (KIND == 2) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a) : "v"(b));
  if constexpr (KIND == 3) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a) : "v"(b));
  if constexpr (KIND == 4) asm volatile("v_fma_f32 %0, %1, %1, %1" : "=v"(f) : "v"(f));
  if constexpr (KIND == 5) asm volatile("v_add_u32 %0, %1, %1" : "=v"(i) : "v"(i));
  if constexpr (KIND == 6) asm volatile("v_mov_b32 %0, %1" : "=v"(i) : "v"(i));
  if constexpr (KIND == 7) asm volatile("v_max_f64 %0, %0, %1" : "+v"(a) : "v"(b));
  if constexpr (KIND == 8) asm volatile("v_cmp_lt_f64 vcc, %0, %1" ::"v"(a), "v"(b) : "vcc");
}

template <int KIND, int K>
__global__ void __launch_bounds__(512) probe(double *out, unsigned long long *cyc, int iters) {
  d4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  double x = 1.0 + threadIdx.x * 1e-9, y = 1.0 - threadIdx.x * 1e-9;
  double fa[4] = {x, y, x, y}, fb = 1.0000001;
  float ff = 1.0f;
  int fi = threadIdx.x;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      acc[m] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, acc[m], 0, 0, 0);
#pragma unroll
      for (int k = 0; k < K; ++k) filler<KIND>(fa[k & 3], fb, ff, fi);
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  double s = fa[0] + fa[1] + fa[2] + fa[3] + ff + fi;
  for (int m = 0; m < 4; ++m) s += acc[m][0] + acc[m][1] + acc[m][2] + acc[m][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (blockIdx.x == 7 && (threadIdx.x & 63) == 0) atomicMax(cyc, t1 - t0);      // the workgroup's slowest wave
}

template <int KIND, int K>
static double run(int threads, double *out, unsigned long long *cyc) {
  const int iters = 2000;
  hipLaunchKernelGGL((probe<KIND, K>), dim3(256), dim3(threads), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  hipMemset(cyc, 0, 8);
  hipLaunchKernelGGL((probe<KIND, K>), dim3(256), dim3(threads), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  unsigned long long h = 0;
  hipMemcpy(&h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  return (double)h / (4.0 * iters * (threads / 256));      // cycles of the SIMD per MFMA it was given
}

template <int KIND>
static void row(const char *name, double *out, unsigned long long *cyc) {
  for (int threads : {256, 512}) {
    printf("%-14s %d wave(s)/SIMD: cycles per MFMA at K = 0 1 2 4 8 fillers: %6.1f %6.1f %6.1f %6.1f %6.1f\n", name, threads / 256,
           run<KIND, 0>(threads, out, cyc), run<KIND, 1>(threads, out, cyc), run<KIND, 2>(threads, out, cyc), run<KIND, 4>(threads, out, cyc),
           run<KIND, 8>(threads, out, cyc));
  }
}

int main() {
  double *out;
  unsigned long long *cyc;
  hipMalloc(&out, sizeof(double) * 256 * 512);
  hipMalloc(&cyc, 8);
  row<1>("v_fma_f64", out, cyc);
  row<2>("v_mul_f64", out, cyc);
  row<3>("v_add_f64", out, cyc);
  row<7>("v_max_f64", out, cyc);
  row<8>("v_cmp_lt_f64", out, cyc);
  row<4>("v_fma_f32", out, cyc);
  row<5>("v_add_u32", out, cyc);
  row<6>("v_mov_b32", out, cyc);
  return 0;
}
