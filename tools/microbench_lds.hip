// LDS read cost on MI355X by access pattern: distinct addresses vs full broadcast vs 4-lane groups sharing an address,
// for ds_read_b32 / b64 / b128.  One workgroup of 256 threads per CU; cycles per wave-instruction from s_memtime.
#include <hip/hip_runtime.h>
#include <cstdio>
template <typename T, int PAT>
__global__ void __launch_bounds__(256) k(double* out, long long* cyc, int iters) {
  __shared__ T buf[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) buf[i] = T{};
  __syncthreads();
  const int lane = threadIdx.x & 63;
  int idx = PAT == 0 ? lane : PAT == 1 ? 0 : PAT == 2 ? (lane & 15) : (lane >> 2);   // 2: 16 distinct x4 lanes strided, 3: groups of 4 adjacent lanes
  T acc{};
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      T v = buf[(idx + u * 64 + it) & 4095];
      if constexpr (sizeof(T) == 4) acc += v; else if constexpr (sizeof(T) == 8) acc += v; else { acc.x += v.x; acc.y += v.y; }
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  if constexpr (sizeof(T) <= 8) out[blockIdx.x * 256 + threadIdx.x] = (double)acc; else out[blockIdx.x * 256 + threadIdx.x] = acc.x + acc.y;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <typename T, int PAT> void run(const char* name, double* out, long long* cyc) {
  int iters = 2000;
  hipLaunchKernelGGL((k<T, PAT>), dim3(256), dim3(256), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  long long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double avg = 0; for (int i = 0; i < 256; ++i) avg += h[i]; avg /= 256;
  printf("%-28s %.1f cycles per wave-instruction (4 waves/CU issuing)\n", name, avg / (iters * 16.0));
}
int main() {
  double* out; long long* cyc; hipMalloc(&out, 8 * 256 * 256); hipMalloc(&cyc, 8 * 256);
  run<float, 0>("b32 distinct", out, cyc); run<float, 1>("b32 all-same", out, cyc); run<float, 2>("b32 16 addr (lane&15)", out, cyc); run<float, 3>("b32 16 addr (lane>>2)", out, cyc);
  run<double, 0>("b64 distinct", out, cyc); run<double, 1>("b64 all-same", out, cyc); run<double, 2>("b64 16 addr (lane&15)", out, cyc); run<double, 3>("b64 16 addr (lane>>2)", out, cyc);
  run<double2, 0>("b128 distinct", out, cyc); run<double2, 1>("b128 all-same", out, cyc); run<double2, 2>("b128 16 addr (lane&15)", out, cyc); run<double2, 3>("b128 16 addr (lane>>2)", out, cyc);
  return 0;
}
