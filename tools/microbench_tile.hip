// microbench_tile.hip -- cycles of the 16x16 MFMA tile update  C -= X Y'  (k = 40) as the blocked trajectory solver
// runs it: one wave alone, and with the three other waves of the workgroup doing the same on other tiles.
// build: hipcc -O3 --offload-arch=gfx950 tools/microbench_tile.hip -o tools/microbench_tile
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ long long now() {
  long long t;
  asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}
template <int LS, int VAR>
__device__ __forceinline__ void tile(double *Cm, const double *X, const double *Y, int it, int jt, int lane) {
  constexpr int KS = 10;
  const int lrow = lane & 15, lq = lane >> 4;
  double *cp = Cm + (16 * it + lq) * LS + 16 * jt + lrow;
  d4 acc, acc2 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int r = 0; r < 4; ++r) acc[r] = cp[4 * r * LS];
  const double *xa = X + (16 * it + lrow) * LS + lq, *yb = Y + (16 * jt + lrow) * LS + lq;
  if (VAR == 0) {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      if (ks & 1) acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(-xa[4 * ks], yb[4 * ks], acc2, 0, 0, 0);
      else acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-xa[4 * ks], yb[4 * ks], acc, 0, 0, 0);
    }
  } else {
    double a[KS], b[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) { a[ks] = xa[4 * ks]; b[ks] = yb[4 * ks]; }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-a[ks], b[ks], acc, 0, 0, 0);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) cp[4 * r * LS] = acc[r] + acc2[r];
}
template <int LS, int VAR>
__global__ void k(long long *out, int others) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 6 * 48 * LS; i += blockDim.x) sm[i] = (i % 97) * 1e-3;
  __syncthreads();
  double *C = sm + (wave & 1) * 48 * LS, *X = sm + 2 * 48 * LS, *Y = sm + 3 * 48 * LS;
  if (wave != 0 && !others) return;
  long long t0 = now();
  for (int rep = 0; rep < 40; ++rep)
    for (int job = 0; job < 5; ++job) tile<LS, VAR>(C + (wave >> 1) * 16 * LS, X, Y, job % 2, job % 3, lane);
  long long t1 = now();
  if (lane == 0 && blockIdx.x == 0) out[wave] = (t1 - t0) / 200;
}
int main() {
  long long *d, h[4];
  (void)hipMalloc(&d, sizeof(h));
  auto run = [&](auto kern, const char *name, int LS, int nblk) {
    for (int others = 0; others < 2; ++others) {
      (void)hipMemset(d, 0, sizeof(h));
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 6 * 48 * LS * 8);
      for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(kern, dim3(nblk), dim3(256), 6 * 48 * LS * 8, 0, d, others);
      (void)hipDeviceSynchronize();
      (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
      printf("[%d blocks] %s, %s: cycles per tile: wave0 %lld wave1 %lld wave2 %lld wave3 %lld\n", nblk, name, others ? "4 waves" : "1 wave", h[0], h[1], h[2], h[3]);
    }
  };
  for (int nblk : {1, 256, 2048}) {
    run(k<50, 0>, "LS=50 interleaved loads", 50, nblk);
    run(k<50, 1>, "LS=50 loads first, one chain", 50, nblk);
  }
  return 0;
}
