// microbench_wave.hip -- single-wave latencies on one CU (what a latency-bound, one-wave-per-SIMD kernel sees):
// cycles per instruction for dependent / independent FP64 VALU and MFMA chains and LDS round trips, from s_memtime.
// build: hipcc -O3 --offload-arch=gfx950 tools/microbench_wave.hip -o tools/microbench_wave
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));
#define N 64
__device__ __forceinline__ long long now() {
  long long t;
  asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}
__global__ void k(long long *out, double *sink, int waves_busy) {
  __shared__ __attribute__((aligned(16))) double lds[4096];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = i * 0.001;
  __syncthreads();
  double x = lane * 1e-3 + 1.0, y = 1.0000001, z = 0.5;
  long long t0, t1;
  if (wave != 0) {           // optional background load on the other SIMDs: MFMA chains + LDS reads
    d4 acc = {0, 0, 0, 0};
    if (waves_busy) {
      for (int it = 0; it < 4000; ++it) {
        const double a = lds[(lane * 50 + it) & 4095], b = lds[(lane * 50 + it + 7) & 4095];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
      }
    }
    sink[threadIdx.x] = acc[0] + acc[1];
    return;
  }
  // 1. dependent fma chain
  __builtin_amdgcn_sched_barrier(0); t0 = now(); __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < N; ++i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x) : "v"(y), "v"(z));
  __builtin_amdgcn_sched_barrier(0); t1 = now(); __builtin_amdgcn_sched_barrier(0);
  if (lane == 0) out[0] = t1 - t0;
  // 2. independent fmas (8 chains)
  double v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = x + j;
  __builtin_amdgcn_sched_barrier(0); t0 = now(); __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < N / 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(v[j]) : "v"(y), "v"(z));
  __builtin_amdgcn_sched_barrier(0); t1 = now(); __builtin_amdgcn_sched_barrier(0);
  if (lane == 0) out[1] = t1 - t0;
#pragma unroll
  for (int j = 0; j < 8; ++j) x += v[j];
  // 3. dependent MFMA chain (16)
  d4 acc = {x, x, x, x};
  __builtin_amdgcn_sched_barrier(0); t0 = now(); __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < 16; ++i) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(y, z, acc, 0, 0, 0);
  x += acc[0];       // forces completion
  asm volatile("" : "+v"(x));
  __builtin_amdgcn_sched_barrier(0); t1 = now(); __builtin_amdgcn_sched_barrier(0);
  if (lane == 0) out[2] = t1 - t0;
  // 4. two independent MFMA chains (16 total)
  d4 a1 = {x, x, x, x}, a2 = {y, y, y, y};
  __builtin_amdgcn_sched_barrier(0); t0 = now(); __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, z, a1, 0, 0, 0);
    a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(z, y, a2, 0, 0, 0);
  }
  x += a1[0] + a2[0];
  asm volatile("" : "+v"(x));
  __builtin_amdgcn_sched_barrier(0); t1 = now(); __builtin_amdgcn_sched_barrier(0);
  if (lane == 0) out[3] = t1 - t0;
  // 5. four independent MFMA chains (16 total)
  d4 b1 = {x, x, x, x}, b2 = b1, b3 = b1, b4 = b1;
  __builtin_amdgcn_sched_barrier(0); t0 = now(); __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    b1 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, z, b1, 0, 0, 0);
    b2 = __builtin_amdgcn_mfma_f64_16x16x4f64(z, y, b2, 0, 0, 0);
    b3 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, y, b3, 0, 0, 0);
    b4 = __builtin_amdgcn_mfma_f64_16x16x4f64(z, z, b4, 0, 0, 0);
  }
  x += b1[0] + b2[0] + b3[0] + b4[0];
  asm volatile("" : "+v"(x));
  __builtin_amdgcn_sched_barrier(0); t1 = now(); __builtin_amdgcn_sched_barrier(0);
  if (lane == 0) out[4] = t1 - t0;
  // 6. LDS dependent read chain (pointer chase, ds_read_b64), 16 hops
  int idx = lane;
  __builtin_amdgcn_sched_barrier(0); t0 = now(); __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < 16; ++i) idx = (int)(lds[idx & 4095] * 1000.0) & 4095;
  __builtin_amdgcn_sched_barrier(0); t1 = now(); __builtin_amdgcn_sched_barrier(0);
  if (lane == 0) out[5] = t1 - t0;
  x += idx;
  // 7. LDS write -> read round trip, b128, 16 times (value carried through)
  d2 w = {x, y};
  __builtin_amdgcn_sched_barrier(0); t0 = now(); __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    *reinterpret_cast<d2 *>(&lds[2 * lane]) = w;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    w = *reinterpret_cast<d2 *>(&lds[2 * ((lane + 1) & 63)]);
  }
  __builtin_amdgcn_sched_barrier(0); t1 = now(); __builtin_amdgcn_sched_barrier(0);
  if (lane == 0) out[6] = t1 - t0;
  x += w.x;
  // 8. rsqrt + 2 Newton steps, dependent chain of 8
  double r = x;
  __builtin_amdgcn_sched_barrier(0); t0 = now(); __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    double q, h, e;
    asm volatile("v_rsq_f64 %0, %1" : "=v"(q) : "v"(r));
    asm volatile("v_mul_f64 %0, %1, -0.5" : "=v"(h) : "v"(r));
    asm volatile("v_mul_f64 %0, %1, %2" : "=v"(e) : "v"(h), "v"(q));
    asm volatile("v_fma_f64 %0, %1, %2, 1.0" : "=v"(e) : "v"(e), "v"(q));
    asm volatile("v_mul_f64 %0, %1, %2" : "=v"(q) : "v"(q), "v"(e));
    asm volatile("v_mul_f64 %0, %1, %2" : "=v"(e) : "v"(h), "v"(q));
    asm volatile("v_fma_f64 %0, %1, %2, 1.0" : "=v"(e) : "v"(e), "v"(q));
    asm volatile("v_mul_f64 %0, %1, %2" : "=v"(q) : "v"(q), "v"(e));
    asm volatile("v_add_f64 %0, %1, 1.0" : "=v"(r) : "v"(q));
  }
  __builtin_amdgcn_sched_barrier(0); t1 = now(); __builtin_amdgcn_sched_barrier(0);
  if (lane == 0) out[7] = t1 - t0;
  // 9. 7 independent ds_read_b128 then use (one batch latency)
  d2 g[7];
  __builtin_amdgcn_sched_barrier(0); t0 = now(); __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int j = 0; j < 7; ++j) g[j] = *reinterpret_cast<d2 *>(&lds[2 * ((lane + j * 64) & 2047)]);
  double gs = 0;
#pragma unroll
  for (int j = 0; j < 7; ++j) gs += g[j].x + g[j].y;
  asm volatile("" : "+v"(gs));
  __builtin_amdgcn_sched_barrier(0); t1 = now(); __builtin_amdgcn_sched_barrier(0);
  if (lane == 0) out[8] = t1 - t0;
  // 10. v_readlane pair + dependent use, 16 times
  double rl = x;
  __builtin_amdgcn_sched_barrier(0); t0 = now(); __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const double s = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(rl), i), __builtin_amdgcn_readlane(__double2loint(rl), i));
    rl = fma(rl, y, s);
  }
  __builtin_amdgcn_sched_barrier(0); t1 = now(); __builtin_amdgcn_sched_barrier(0);
  if (lane == 0) out[9] = t1 - t0;
  sink[threadIdx.x] = x + r + gs + rl;
}
int main() {
  long long *d, h[16];
  double *s;
  hipMalloc(&d, sizeof(h));
  hipMalloc(&s, 256 * 8);
  for (int busy = 0; busy < 2; ++busy) {
    for (int rep = 0; rep < 2; ++rep) {
      hipMemset(d, 0, sizeof(h));
      hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, d, s, busy);
      hipDeviceSynchronize();
    }
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("other waves %s (s_memtime ticks; clock64)\n", busy ? "busy (MFMA+LDS)" : "idle");
    printf("  dependent v_fma_f64 x64:          %lld  (%.1f / instr)\n", h[0], h[0] / 64.0);
    printf("  independent v_fma_f64 x64:        %lld  (%.1f / instr)\n", h[1], h[1] / 64.0);
    printf("  dependent MFMA f64 16x16x4 x16:   %lld  (%.1f / mfma)\n", h[2], h[2] / 16.0);
    printf("  2 chains MFMA x16:                %lld  (%.1f / mfma)\n", h[3], h[3] / 16.0);
    printf("  4 chains MFMA x16:                %lld  (%.1f / mfma)\n", h[4], h[4] / 16.0);
    printf("  LDS pointer chase x16 (b64):      %lld  (%.1f / hop)\n", h[5], h[5] / 16.0);
    printf("  LDS b128 write->read x16:         %lld  (%.1f / trip)\n", h[6], h[6] / 16.0);
    printf("  rsq + 2 Newton, chain x8:         %lld  (%.1f / rsqrt)\n", h[7], h[7] / 8.0);
    printf("  7 x ds_read_b128 batch + use:     %lld\n", h[8]);
    printf("  readlane pair + fma x16:          %lld  (%.1f / step)\n", h[9], h[9] / 16.0);
  }
  // calibrate ticks against wall time
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  return 0;
}
