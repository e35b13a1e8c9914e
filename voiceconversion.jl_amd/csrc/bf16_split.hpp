// bf16_split.hpp -- operands of v_mfma_f32_16x16x32_bf16 from FP64 values: the hi + lo split behind the certified screens
// (gmmmap_screen.hpp, estep_hard.hpp).  With x = hi + lo + r, |r| <= 2^-16 |x|, and the same for the other operand, the three
// products hi hi + hi lo + lo hi reproduce x y to 3 x 2^-16 |x y|; each product of two bf16 numbers is exact in FP32.
#pragma once
#include <hip/hip_runtime.h>

namespace vcmi {
// x -> bf16 hi (round to nearest even) and bf16 lo of the FP32 remainder (x_f32 - hi is exact in FP32)
__host__ __device__ inline void split_bf16(double x, unsigned short &h, unsigned short &l) {
  union {
    float f;
    unsigned u;
  } a, b, d;
  a.f = (float)x;
  h = (unsigned short)((a.u + 0x7FFFu + ((a.u >> 16) & 1u)) >> 16);
  b.u = (unsigned)h << 16;
  d.f = a.f - b.f;
  l = (unsigned short)((d.u + 0x7FFFu + ((d.u >> 16) & 1u)) >> 16);
}
#if defined(__HIPCC__)
// two values at once on the device: v_cvt_pk_bf16_f32 (gfx950, round to nearest even) for the hi parts, the exact FP32
// remainders through it again for the lo parts.  hi = {bf16(a) | bf16(b) << 16}, likewise lo.
__device__ __forceinline__ void split_bf16_pair(double a, double b, unsigned &hi, unsigned &lo) {
#if defined(__HIP_DEVICE_COMPILE__)
  const float af = (float)a, bf = (float)b;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(hi) : "v"(af), "v"(bf));
  const float ra = af - __uint_as_float(hi << 16), rb = bf - __uint_as_float(hi & 0xFFFF0000u);
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(lo) : "v"(ra), "v"(rb));
#else
  hi = lo = 0;                                                      // device only; the host pass just needs the declaration
  (void)a;
  (void)b;
#endif
}
#endif
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

}  // namespace vcmi
