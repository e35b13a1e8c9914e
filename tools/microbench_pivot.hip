// microbench_pivot.hip -- counts (s_memtime ticks, ~1.27 per shader cycle) per pivot column of the blocked trajectory
// solver's scalar chain (pv2_wave0 of voiceconversion.jl_amd/csrc/traj_solve_blk.hpp) run alone: the hand-over
// counter is preset so that no wait blocks.
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -I voiceconversion.jl_amd/csrc tools/microbench_pivot.hip -o tools/microbench_pivot
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
template <int D>
__global__ void __launch_bounds__(64) k(long long *out, double *chk) {
  using C = BlkCfg<D>;
  extern __shared__ __attribute__((aligned(16))) double msm[];
  double *b00 = msm;
  int *flags = reinterpret_cast<int *>(msm + C::BUF);
  const int lane = threadIdx.x;
  __shared__ int bad;
  long long total = 0;
  for (int rep = 0; rep < 9; ++rep) {
    for (int i = lane; i < C::BUF; i += 64) {
      const int r = i / C::LS, c = i % C::LS;
      b00[i] = (r == c) ? 50.0 + r : 1.0 / (1.0 + r + c);    // SPD (diagonally dominant); only the diagonal tiles are used
    }
    if (lane == 0) flags[0] = 1 << 30;
    __syncthreads();
    const long long t0 = now();
    pv2_wave0<D, false>(b00, flags, 0, lane, &bad);
    const long long t1 = now();
    if (rep > 0) total += t1 - t0;
    __syncthreads();
  }
  if (lane == 0) out[0] = total / (8 * D);
  if (lane < 16) chk[lane] = b00[lane * C::LS + (lane >> 1)];     // U_00 entries, for a look at the values
}
}  // namespace vcmi
int main() {
  long long *d, h[1];
  double *chk, hc[16];
  (void)hipMalloc(&d, sizeof(h));
  (void)hipMalloc(&chk, sizeof(hc));
  constexpr int D = 40;
  const size_t shm = (vcmi::BlkCfg<D>::BUF + 8) * 8;
  auto kern = vcmi::k<D>;
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(kern, dim3(1), dim3(64), shm, 0, d, chk);
  (void)hipDeviceSynchronize();
  (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  (void)hipMemcpy(hc, chk, sizeof(hc), hipMemcpyDeviceToHost);
  printf("scalar chain alone: %lld counts per pivot column (D = %d: %d blocks incl. tile load / store)\n", h[0], D, (D + 15) / 16);
  printf("U_00 sample:");
  for (int i = 0; i < 16; i += 3) printf(" %.12g", hc[i]);
  printf("\n");
  return 0;
}
