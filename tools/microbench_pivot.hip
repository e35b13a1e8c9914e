// microbench_pivot.hip -- cycles per column of the blocked trajectory solver's pivot waves (blk_pivot_s / blk_pivot_u of
// voiceconversion.jl_amd/csrc/traj_solve_blk.hpp) run alone on one CU.
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
__global__ void __launch_bounds__(256) k(long long *out, int mode) {
  using C = BlkCfg<D>;
  extern __shared__ __attribute__((aligned(16))) double msm[];
  double *b00 = msm, *ring = msm + C::BUF, *cbu = ring + C::RING;
  int *flags = reinterpret_cast<int *>(cbu + 64);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < C::BUF; i += 256) {
    const int r = i / C::LS, c = i % C::LS;
    b00[i] = (r == c) ? 50.0 + r : 1.0 / (1.0 + r + c);    // SPD (diagonally dominant)
  }
  if (tid == 0) flags[0] = flags[1] = 0;
  for (int i = tid; i < C::RING + 64; i += 256) ring[i] = 0.0;
  __syncthreads();
  long long t0 = now();
  for (int rep = 0; rep < 8; ++rep) {
    if (wave == 0) blk_pivot_s<D>(b00, ring, rep * D, lane, flags);
    else if (wave == 1 && mode >= 1) blk_pivot_u<D>(b00, ring, rep * D, cbu, lane);
    else __syncthreads();
    __syncthreads();
  }
  long long t1 = now();
  if (lane == 0) out[wave] = (t1 - t0) / (8 * D);
}
}  // namespace vcmi
int main() {
  long long *d, h[4];
  (void)hipMalloc(&d, sizeof(h));
  constexpr int D = 40;
  const size_t shm = (vcmi::BlkCfg<D>::BUF + vcmi::BlkCfg<D>::RING + 64 + 8) * 8;
  auto kern = vcmi::k<D>;
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
  for (int mode = 0; mode < 2; ++mode) {
    (void)hipMemset(d, 0, sizeof(h));
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(kern, dim3(1), dim3(256), shm, 0, d, mode);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("%s: cycles per pivot column: wave S %lld, wave U %lld (idle waves %lld %lld)\n", mode ? "S + U" : "S alone", h[0], h[1], h[2], h[3]);
  }
  return 0;
}
