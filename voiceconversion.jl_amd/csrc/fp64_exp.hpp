// fp64_exp.hpp -- e^x for x <= 0 on the FP64 vector pipe, as lean as the softmax loops need it (shared by gmmmap.hip and
// estep.hip).  The FP64 VALU shares its pipe with the FP64 MFMAs, so every instruction here is paid for in
// matrix time.
#pragma once
#include <hip/hip_runtime.h>

namespace vcmi {

// exp for the softmax weights; arguments are <= 0 (or -inf).
#ifndef VCMI_LEAN_EXP
#define VCMI_LEAN_EXP 1
#endif
// p * r + c with the constant c held in an SGPR pair (VOP3 takes it as an operand).  Left to itself the compiler emits
// v_fmac_f64 and first copies the 64-bit literal into the destination -- two v_mov_b32 per Horner step, 45 % of the
// VALU instructions of an exp; the scalar moves that replace them issue on the scalar port.
__device__ __forceinline__ double vc_fma_sconst(double p, double r, double c) {
  double d;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(p), "v"(r), "s"(c));
  return d;
}
__device__ __forceinline__ double vc_exp(double x) {
#if VCMI_LEAN_EXP
  // e^x for x <= 0: n = rint(x log2 e), r = x - n ln 2 (two-term Cody-Waite), degree-13 Taylor series on |r| <= 0.347
  // (truncation 4e-18), scaled by 2^n with v_ldexp_f64, which also delivers the underflow to 0.  The softmax epilogue
  // shares the FP64 pipe with the MFMAs, and two exps per mixture are 45 % of its instructions.
  x = fmax(x, -1000.0);                                  // also maps -inf; e^-1000 is 0 in double
  const double n = rint(x * 1.4426950408889634074);
  double r = fma(n, -6.93147180369123816490e-01, x);
  r = fma(n, -1.90821492927058770002e-10, r);
  double p = vc_fma_sconst(1.6059043836821613e-10, r, 2.08767569878681e-09);   // 1/13!, 1/12!
  p = vc_fma_sconst(p, r, 2.505210838544172e-08);        // 1/11!
  p = vc_fma_sconst(p, r, 2.755731922398589e-07);        // 1/10!
  p = vc_fma_sconst(p, r, 2.7557319223985893e-06);       // 1/9!
  p = vc_fma_sconst(p, r, 2.48015873015873e-05);         // 1/8!
  p = vc_fma_sconst(p, r, 1.984126984126984e-04);        // 1/7!
  p = vc_fma_sconst(p, r, 1.388888888888889e-03);        // 1/6!
  p = vc_fma_sconst(p, r, 8.333333333333333e-03);        // 1/5!
  p = vc_fma_sconst(p, r, 4.1666666666666664e-02);       // 1/4!
  p = vc_fma_sconst(p, r, 1.6666666666666666e-01);       // 1/3!
  p = fma(p, r, 0.5);
  p = fma(p, r, 1.0);
  p = fma(p, r, 1.0);
  return ldexp(p, (int)n);
#else
  return exp(x);
#endif
}

// e^x for x <= 0 from a 64-entry table of 2^(j/64) in LDS (tab) and a degree-5 polynomial: x = (64 n + j) ln2/64 + r,
// |r| <= ln2/128 (truncation r^6/720 < 4e-17), e^x = 2^n * tab[j] * p(r).  14 FP64 instructions instead of the 20 of
// vc_exp; the table read is an LDS access, which does not compete with the MFMAs for the FP64 pipe.
#ifndef VCMI_TABLE_EXP
#define VCMI_TABLE_EXP 1
#endif
static __device__ const double kExp2Tab[64] = {   // 2^(j/64), correctly rounded (mpmath)
    0x1.0000000000000p+0, 0x1.02c9a3e778061p+0, 0x1.059b0d3158574p+0, 0x1.0874518759bc8p+0,
    0x1.0b5586cf9890fp+0, 0x1.0e3ec32d3d1a2p+0, 0x1.11301d0125b51p+0, 0x1.1429aaea92de0p+0,
    0x1.172b83c7d517bp+0, 0x1.1a35beb6fcb75p+0, 0x1.1d4873168b9aap+0, 0x1.2063b88628cd6p+0,
    0x1.2387a6e756238p+0, 0x1.26b4565e27cddp+0, 0x1.29e9df51fdee1p+0, 0x1.2d285a6e4030bp+0,
    0x1.306fe0a31b715p+0, 0x1.33c08b26416ffp+0, 0x1.371a7373aa9cbp+0, 0x1.3a7db34e59ff7p+0,
    0x1.3dea64c123422p+0, 0x1.4160a21f72e2ap+0, 0x1.44e086061892dp+0, 0x1.486a2b5c13cd0p+0,
    0x1.4bfdad5362a27p+0, 0x1.4f9b2769d2ca7p+0, 0x1.5342b569d4f82p+0, 0x1.56f4736b527dap+0,
    0x1.5ab07dd485429p+0, 0x1.5e76f15ad2148p+0, 0x1.6247eb03a5585p+0, 0x1.6623882552225p+0,
    0x1.6a09e667f3bcdp+0, 0x1.6dfb23c651a2fp+0, 0x1.71f75e8ec5f74p+0, 0x1.75feb564267c9p+0,
    0x1.7a11473eb0187p+0, 0x1.7e2f336cf4e62p+0, 0x1.82589994cce13p+0, 0x1.868d99b4492edp+0,
    0x1.8ace5422aa0dbp+0, 0x1.8f1ae99157736p+0, 0x1.93737b0cdc5e5p+0, 0x1.97d829fde4e50p+0,
    0x1.9c49182a3f090p+0, 0x1.a0c667b5de565p+0, 0x1.a5503b23e255dp+0, 0x1.a9e6b5579fdbfp+0,
    0x1.ae89f995ad3adp+0, 0x1.b33a2b84f15fbp+0, 0x1.b7f76f2fb5e47p+0, 0x1.bcc1e904bc1d2p+0,
    0x1.c199bdd85529cp+0, 0x1.c67f12e57d14bp+0, 0x1.cb720dcef9069p+0, 0x1.d072d4a07897cp+0,
    0x1.d5818dcfba487p+0, 0x1.da9e603db3285p+0, 0x1.dfc97337b9b5fp+0, 0x1.e502ee78b3ff6p+0,
    0x1.ea4afa2a490dap+0, 0x1.efa1bee615a27p+0, 0x1.f50765b6e4540p+0, 0x1.fa7c1819e90d8p+0,
};
__device__ __forceinline__ double vc_exp_tab(double x, const double *tab) {
#if VCMI_TABLE_EXP
  x = fmax(x, -1000.0);
  const double kf = rint(x * 92.332482616893656877);        // 64 / ln 2
  double r = fma(kf, -1.083042469326756e-02, x);            // ln2/64, high part (21 trailing zero bits: kf * hi is exact)
  r = fma(kf, -2.9815858269852933e-12, r);                  // ln2/64, low part
  const int ki = (int)kf;
  const double t = tab[ki & 63];
  double p = vc_fma_sconst(8.333333333333333e-03, r, 4.1666666666666664e-02);
  p = vc_fma_sconst(p, r, 1.6666666666666666e-01);
  p = fma(p, r, 0.5);
  p = fma(p, r, 1.0);
  p = fma(p, r, 1.0);
  return ldexp(t * p, ki >> 6);
#else
  (void)tab;
  return vc_exp(x);
#endif
}

// All-reduce over the 16 lanes of a DPP row (lanes 16 g .. 16 g + 15) without LDS: two quad permutes (xor 1, xor 2), then
// row_half_mirror (i <-> 7 - i: joins the two quads of each half) and row_mirror (i <-> 15 - i: joins the halves).  Every
// lane of the row ends with the same value (the partners add / compare the same two operands).  __shfl_xor compiles to
// ds_bpermute_b32 pairs with an s_waitcnt each: four LDS round trips per reduction.
// (The permutations used here -- quad_perm, row_half_mirror, row_mirror -- give every lane a source, so the destination's old
// value never shows.  VCMI_DPP_NO_COPY = 1: mov_dpp with bound_ctrl instead of update_dpp(x, x, ...), whose tied operand costs
// a copy of x per 32-bit half -- the E-step kernels, where every VALU instruction next to the FP64 MFMAs is paid for in
// matrix-pipe time, define it.  The conversion kernels keep the tied form they were tuned and profiled
// with: their FP32 / integer permutations sit beside BF16 MFMAs, where nothing was to be gained (346-356 us per 10^6 frames for
// the screened convert kernel with either form, box by box).)
#ifndef VCMI_DPP_NO_COPY
#define VCMI_DPP_NO_COPY 0
#endif
template <int CTRL>
__device__ __forceinline__ double dpp_row_f64(double x) {
#if VCMI_DPP_NO_COPY
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(x), CTRL, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(x), CTRL, 0xF, 0xF, true);
#else
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xF, 0xF, false);
  hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xF, 0xF, false);
#endif
  return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ int dpp_row_i32(int x) {
#if VCMI_DPP_NO_COPY
  return __builtin_amdgcn_mov_dpp(x, CTRL, 0xF, 0xF, true);
#else
  return __builtin_amdgcn_update_dpp(x, x, CTRL, 0xF, 0xF, false);
#endif
}
__device__ __forceinline__ double row16_max(double x) {
  x = fmax(x, dpp_row_f64<0xB1>(x));
  x = fmax(x, dpp_row_f64<0x4E>(x));
  x = fmax(x, dpp_row_f64<0x141>(x));
  return fmax(x, dpp_row_f64<0x140>(x));
}
__device__ __forceinline__ double row16_sum(double x) {
  x += dpp_row_f64<0xB1>(x);
  x += dpp_row_f64<0x4E>(x);
  x += dpp_row_f64<0x141>(x);
  return x + dpp_row_f64<0x140>(x);
}
__device__ __forceinline__ int row16_min(int x) {
  x = min(x, dpp_row_i32<0xB1>(x));
  x = min(x, dpp_row_i32<0x4E>(x));
  x = min(x, dpp_row_i32<0x141>(x));
  return min(x, dpp_row_i32<0x140>(x));
}
__device__ __forceinline__ int row16_sum(int x) {
  x += dpp_row_i32<0xB1>(x);
  x += dpp_row_i32<0x4E>(x);
  x += dpp_row_i32<0x141>(x);
  return x + dpp_row_i32<0x140>(x);
}

}  // namespace vcmi
