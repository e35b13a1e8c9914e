// microbench_blkops.hip -- cycles per call of the blocked trajectory solver's MFMA building blocks (traj_solve_blk.hpp)
// run by 1 or 4 waves of one workgroup, against the FP64 MFMA minimum (64 cycles per v_mfma_f64_16x16x4 on gfx950).
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -I voiceconversion.jl_amd/csrc tools/microbench_blkops.hip -o tools/microbench_blkops
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <type_traits>
namespace vcmi {
struct TrajUtt { const double *X; double *Y; int64_t frame0; int32_t T; };
__device__ __forceinline__ double traj_rsqrt(double x) {
  double y = __builtin_amdgcn_rsq(x);
  y = y * fma(-0.5 * x * y, y, 1.5);
  y = y * fma(-0.5 * x * y, y, 1.5);
  return y;
}
#include "traj_solve_blk.hpp"
__device__ __forceinline__ long long now() {
  long long t;
  asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}
constexpr int REPS = 64;
template <int D>
__global__ void __launch_bounds__(512) k(long long *out, double *pan, int mode, int nwaves) {
  using C = BlkCfg<D>;
  extern __shared__ __attribute__((aligned(16))) double msm[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 6 * C::BUF; i += blockDim.x) msm[i] = 1e-3 * ((i * 7) % 13);
  __syncthreads();
  double *bufA = msm + (wave % 2) * C::BUF, *bufB = msm + 2 * C::BUF, *bufC = msm + (3 + (wave % 3)) * C::BUF;
  long long t0 = now();
  if (wave < nwaves)
    for (int rep = 0; rep < REPS; ++rep) {
      const int it = rep % C::NT;
      if (mode == 0) blk_update_rowgroup<D>(bufC, bufA, bufB, it, C::NT, lane);
      else if (mode == 1) blk_trsm_rowtile<D>(bufC, bufB, it, lane);
      else if (mode == 2) blk_lu_rowtile_to_panel<D>(bufA, bufB, it, D + 1, pan + (size_t)(wave % 4) * C::PAN, lane);
      else if (mode == 3) blk_update_tile<D>(bufC, bufA, bufB, it, (rep / 3) % C::NT, lane);
      else if (mode == 4) blk_update_rowgroup<D>(bufC, bufA, bufA, it, it + 1, lane);
      else if (mode >= 5) {     // 32 products on registers only: 4 independent chains (5), one chain (6)
        blk_d4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
        const double x = bufA[lane], y = bufB[lane];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, c0, 0, 0, 0);
          if (mode == 5) {
            c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, x, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, y, c3, 0, 0, 0);
          } else {
            c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, x, c0, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, c0, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, y, c0, 0, 0, 0);
          }
        }
        c0 += c1 + c2 + c3;
        bufC[lane] = c0[0] + c0[1] + c0[2] + c0[3];
      }
    }
  long long t1 = now();
  if (lane == 0) out[wave] = (t1 - t0) / REPS;
}
}  // namespace vcmi
int main() {
  long long *d, h[8];
  double *pan;
  (void)hipMalloc(&d, sizeof(h));
  constexpr int D = 40;
  using C = vcmi::BlkCfg<D>;
  (void)hipMalloc(&pan, 4 * C::PAN * 8);
  const size_t shm = 6 * C::BUF * 8;
  auto kern = vcmi::k<D>;
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
  const char *names[] = {"update_rowgroup njt=3 (30 mfma)", "trsm_rowtile (22 mfma)", "lu_rowtile_to_panel (18 mfma)", "update_tile (10 mfma)",
                         "update_rowgroup lower (10/20/30 mfma, avg 20)", "32 mfma on registers, 4 chains", "32 mfma on registers, 1 chain"};
  const int mf[] = {30, 22, 18, 10, 20, 32, 32};
  for (int mode = 0; mode < 7; ++mode)
    for (int nw = 1; nw <= 8; nw = nw == 1 ? 4 : nw * 2) {
      (void)hipMemset(d, 0, sizeof(h));
      for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(kern, dim3(1), dim3(nw > 4 ? 512 : 256), shm, 0, d, pan, mode, nw);
      (void)hipDeviceSynchronize();
      (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
      printf("%-48s waves %d: cycles/call %lld %lld %lld %lld | %lld %lld %lld %lld (mfma minimum %d)\n", names[mode], nw, h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7], 64 * mf[mode]);
    }
  return 0;
}
