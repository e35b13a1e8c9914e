// gmmmap_screen.hpp -- fvconvert for PEAKED models on grouped frames: screen four mixtures per MFMA tile, evaluate the few that
// survive (included by gmmmap.hip; shape 3 of vcmi_gmmmap_convert_plan).
//
// What the "peaked" loop of gmmmap_mfma_kernel (PRUNE = 2) spends its time on: of its ~10.5 MFMAs per (16-frame tile,
// mixture) all but ~0.7 only PROVE that the mixture's posterior is below e^-prune -- the last 16-row whitening tile, whose
// share of |z|^2 is a lower bound of |z|^2.  |z_m|^2 = (x - mu_m)' inv(Sxx_m) (x - mu_m) = sum_i kappa_i (v_i' (x - mu_m))^2
// over the eigenpairs of inv(Sxx_m): ANY partial sum is a lower bound, and the largest kappa (the directions of smallest
// variance) give the largest one per row.  So the proof is made with the rows sqrt(kappa_i) v_i' of the FOUR largest
// eigenpairs (prepare(), host: Jacobi), and the MFMA result layout does the rest: lane group g of a 16 x 16 result holds rows
// {g, 4 + g, 8 + g, 12 + g}, so a tile whose row 4 r + j is screening row r of mixture j gives lane group j
// the four rows of mixture j -- its bound is four multiply-adds in the lane, no cross-lane sum, and ONE tile of KS
// MFMAs screens FOUR mixtures (2.5 MFMAs per pair instead of 10).  Per workgroup (WAVES waves x FT tiles of 16 frames):
//   1. the mixtures of the workgroup's groups (gkey of its frames: one, or two or three where the workgroup straddles group
//      boundaries) are evaluated in full -- whitening, regression, softmax -- which makes the running maximum tight for
//      (almost) all of its frames;
//   2. every quad of mixtures is screened against (running maximum - prune); a mixture that no frame of the workgroup lets
//      through contributes less than e^-prune to every frame: exactly the terms the other shapes skip.  Survivors go into a
//      bitmap in LDS (rare: one or two per workgroup on the SURVEY 8d model);
//   3. the survivors are evaluated in full, in index order, with the "broad" loop's per-wave test in front of the regression.
// The running maximum only grows, so a mixture screened out against the maximum of step 1 is below e^-prune of the final
// maximum as well: the result is the dense loop's to rounding (the order of the sum differs: survivors in index order after
// the group's mixture).  Stages of QS quads are staged by LDS-DMA, double-buffered, one barrier per stage (16 mixtures).
#pragma once
#include <type_traits>
#include "bf16_split.hpp"
#include "lds_dma.hpp"

namespace vcmi {

// ROWS PER MIXTURE (rpm, chosen per model by prepare(): the fewest rows that still rule (almost) every wrong mixture out):
//   4: lane group j of a screening tile = mixture j's four strongest rows               ->  4 mixtures per tile
//   2: registers {0,1} of lane group j = mixture 2j's two strongest rows, {2,3} = mixture 2j+1's  ->  8 mixtures per tile
//   1: register r of lane group j = the strongest row of mixture 4j + r                  -> 16 mixtures per tile (KS MFMAs screen 16)
// screening tiles per stage (one barrier per stage: 16 / 32 / 64 mixtures at four): two beyond DP = 48, where a stage of four
// would no longer leave room for two workgroups per CU beside the 32 KB whitening block
__host__ __device__ constexpr int screen_quads(int DP) { return DP <= 48 ? 4 : 2; }

// stage layout in doubles: [QS x KS x 64 operand fragments | QS x 4 lane groups x 8 {cinit r = 0..3, lc of sub-mixture 0..3}], whole KB
__host__ __device__ constexpr int screen_frag_doubles(int DP) { return screen_quads(DP) * (DP / 4) * 64; }
__host__ __device__ constexpr int screen_stage_doubles(int DP) { return (screen_frag_doubles(DP) + screen_quads(DP) * 32 + 127) / 128 * 128; }
// which mixture (relative to the tile's first) and which of its screening rows (0 = the strongest) tile row i stands for
__host__ __device__ constexpr int screen_row_mixture(int i, int rpm) { return (4 / rpm) * (i & 3) + (i >> 2) / rpm; }
__host__ __device__ constexpr int screen_row_index(int i, int rpm) { return (i >> 2) % rpm; }

// ---- the screen on the BF16 matrix pipe (B16 = true; rpm = 4, DP <= 40) --------------------------------------------------
// The screen only has to produce a CERTIFIED lower bound of sum_i a_i^2, a_i = P_i x - c_i.  v_mfma_f32_16x16x32_bf16 runs
// at 16x the FP64 MFMA rate, so P and x are split into two bf16 pieces each (hi + lo: 16 of their 53 bits) and
//   a^ = Ph xh + Ph xl + Pl xh - c      (three K = 32 instructions over the first eight k-steps of the FP64 operand layout -- lane
//                                        group g, slot j <-> feature 4 j + g, exactly what the lane's xb[.][j] holds -- and one
//                                        more whose slots carry the three terms of k-steps 8, 9), accumulated in FP32.
// |a^ - a| <= (dropped Pl xl and the two split residuals: 3 x 2^-16; FP32 accumulation of <= 130 exact products: 2^-15)
//             x sum_k |P_ik||x_k|  +  2^-24 |c_i|   <=   eps_i := 2^-12 (|P_i| |x| + |c_i|)       (Cauchy-Schwarz; ~2 x the sum above),
// so  a_i^2 >= max(|a^_i| - eps_i, 0)^2  and  lc - sum_i max(|a^_i| - eps_i, 0)^2 / 2  is still an upper bound of the mixture's
// log-density: a mixture it rules out is ruled out.  eps is ~0.3 where the test needs |a| of 10 and more: what the screen
// decides hardly changes, its matrix work drops from 10 FP64 MFMAs (640 cycles) to 4 BF16 ones (64 cycles) per tile.
// The FP32 result layout gives lane group j rows 4 j .. 4 j + 3: tile row i <-> mixture i >> 2, screening row i & 3.
// The margins and the sum of squares are formed in FP32 (the FP64 vector pipe is the one the conversion itself needs): the
// constants are rounded UP on the host (and carry a factor 1 + 2^-20 for the FP32 roundings of eps), the sum is taken down by
// 1 - 2^-20 before it is used.
// Stage layout in doubles: per tile [Ph main | Pl main | tail] as 3 x 1 KB of bf16x8 per lane, then per tile and lane group
// 8 doubles {c_0..3 (4 floats), 2^-12 |P_0..3| (4 floats), 2^-12 |c_0..3| (4 floats), lc (double), pad}.
__host__ __device__ constexpr int screen16_tile_doubles() { return 3 * 128; }
__host__ __device__ constexpr int screen16_stage_doubles(int DP) { return screen_quads(DP) * (screen16_tile_doubles() + 32); }
__host__ __device__ constexpr bool screen16_has(int DP) { return DP >= 16 && DP <= 40 && DP % 4 == 0; }

template <int DP, int FT, int WAVES, bool B16 = false>
__global__ void __launch_bounds__(WAVES * 64) __attribute__((amdgpu_waves_per_eu(DP <= 40 ? (FT == 2 ? 3 : 4) : 2)))
gmmmap_screen_kernel(const double *__restrict__ packed, const double *__restrict__ packedQ, int rpm, int M, int D,
                     const double *__restrict__ X, int64_t ldx, int64_t T, double *__restrict__ Y, int64_t ldy, double prune,
                     unsigned long long *__restrict__ nreg, const int *__restrict__ perm, const int *__restrict__ gkey) {
  using TL = Tiling<DP, false>;
  constexpr int KS = TL::KS, NT = TL::NT, NU = TL::NU, BLK = TL::BLK;
  constexpr int QS = screen_quads(DP), QFR = screen_frag_doubles(DP), STG = B16 ? screen16_stage_doubles(DP) : screen_stage_doubles(DP);
  constexpr int BUF = (BLK > STG) ? BLK : STG;                  // doubles per buffer
  constexpr int NI_BLK = BLK / 128, NI_STG = STG / 128;         // 1 KB wave instructions per block / stage
  constexpr bool PAIRED = (FT == 2);
  static_assert(!B16 || (screen16_has(DP) && STG % 128 == 0), "the bf16 screen covers up to ten k-steps");
  extern __shared__ double smem[];                              // 2 * BUF doubles
  __shared__ double etab[64];
  __shared__ unsigned survivors[32];                            // bit m: mixture m passed the screen on some frame of the workgroup (M <= 1024)
  __shared__ unsigned keys[32];                                 // bit m: m is the group of some frame of the workgroup (evaluated in step 1)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const unsigned lane_off = 16u * (unsigned)lane;
  const int lcol = lane & 15, lgrp = lane >> 4;
  const int64_t frame0 = ((int64_t)blockIdx.x * WAVES + wave) * (16 * FT);
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)(reinterpret_cast<char *>(smem));

  if (tid < 64) etab[tid] = kExp2Tab[tid];
  if (tid < 32) survivors[tid] = 0u;
  else if (tid < 64) keys[tid - 32] = 0u;

  // group of the workgroup's first frame: the mixture evaluated first
  int mg = 0;
  {
    // (uniform addresses through the constant address space: two dependent SCALAR loads -- the chain perm -> gkey -> DMA of the
    // block does not queue behind, or hold up, the vector loads of the frames below; perm / gkey were written by earlier kernels)
    const int64_t f0 = (int64_t)blockIdx.x * WAVES * (16 * FT);
    const __attribute__((address_space(4))) int *perm_c = (const __attribute__((address_space(4))) int *)perm;
    const __attribute__((address_space(4))) int *gkey_c = (const __attribute__((address_space(4))) int *)gkey;
    mg = (f0 < T) ? gkey_c[perm_c[f0]] : 0;
    mg = (mg >= 0 && mg < M) ? mg : 0;
  }
  // block mg -> buffer 0, stage 0 -> buffer 1 (each wave issues every WAVES-th KB)
  auto dma_block = [&](int m, int buf) {
    const char *gb = reinterpret_cast<const char *>(packed + (size_t)m * BLK);
    const unsigned lb = lds0 + (unsigned)buf * (BUF * 8u);
#pragma unroll
    for (int i = 0; i < (NI_BLK + WAVES - 1) / WAVES; ++i) {
      const int k = wave_u + WAVES * i;
      if (k < NI_BLK) dma_1k(gb + 1024 * k, lb + 1024u * k, lane_off);
    }
  };
  auto dma_stage = [&](int s, int buf) {
    const char *gb = reinterpret_cast<const char *>(packedQ + (size_t)s * STG);
    const unsigned lb = lds0 + (unsigned)buf * (BUF * 8u);
#pragma unroll
    for (int i = 0; i < (NI_STG + WAVES - 1) / WAVES; ++i) {
      const int k = wave_u + WAVES * i;
      if (k < NI_STG) dma_1k(gb + 1024 * k, lb + 1024u * k, lane_off);
    }
  };
  dma_block(mg, 0);
  dma_stage(0, 1);
  __builtin_amdgcn_sched_barrier(0);

  // B operands: xb[f][ks] = X[k = 4 ks + lgrp][frame], zero outside (D, T)
  double xb[FT][KS];
  int64_t frow[FT];
  unsigned tiles_in_range = 0;
#pragma unroll
  for (int f = 0; f < FT; ++f) {
    const int64_t fr = frame0 + 16 * f + lcol;
    frow[f] = (fr < T) ? (int64_t)perm[fr] : fr;
    if (frame0 + 16 * f < T) tiles_in_range |= 1u << f;
  }
  // (round 6: rows as whole 128-byte lines where they are 16-byte aligned and unpadded -- load_frame_row, gmmmap.hip)
  const bool xlines = rows_as_lines(X, ldx, D, DP);
  auto load_x = [&]() {
#pragma unroll
    for (int f = 0; f < FT; ++f) {
      const int64_t fr = frame0 + 16 * f + lcol;
      load_frame_row<KS>(X + (fr < T ? frow[f] : (int64_t)0) * ldx, fr < T, xlines, D, lgrp, xb[f]);
    }
  };
  load_x();
  __syncthreads();                                               // the bitmaps are zeroed
  if (lgrp == 0) {
#pragma unroll
    for (int f = 0; f < FT; ++f) {
      const int64_t fr = frame0 + 16 * f + lcol;
      if (fr < T) {
        const int k = gkey[frow[f]];
        if (k >= 0 && k < M) atomicOr(&keys[k >> 5], 1u << (k & 31));
      }
    }
  }
  double yacc[FT][KS];
  double runmax[FT], den[FT];          // PAIRED: [0] holds tile 0's value in the even lane groups, tile 1's in the odd ones
#pragma unroll
  for (int f = 0; f < FT; ++f) {
    runmax[f] = -INFINITY;
    den[f] = 0.0;
#pragma unroll
    for (int j = 0; j < KS; ++j) yacc[f][j] = 0.0;
  }
  int nreg_wave = 0, nmfma_wave = 0, nmfma16_wave = 0;           // (v_mfma_f64_16x16x4 / v_mfma_f32_16x16x32_bf16 instructions issued)
  const __attribute__((address_space(4))) double *packed_c = (const __attribute__((address_space(4))) double *)packed;

  // ---- one mixture in full from the block in `cur`: whitening, (test,) regression, online softmax update -- the "broad" loop's body
  auto full_mixture = [&](const double *cur, double lc, bool tested) {
    if ((unsigned)((unsigned long long)__double_as_longlong(lc) >> 32) == 0xFFF00000u) return;      // zero weight: posterior exactly 0
    d4 acc[FT][NT];
#pragma unroll
    for (int t = 0; t < NU; ++t) {
      d4 c;
#pragma unroll
      for (int r = 0; r < 4; ++r) c[r] = cur[TL::CINIT_OFF + 16 * t + 4 * r + lgrp];
#pragma unroll
      for (int f = 0; f < FT; ++f) acc[f][t] = c;
    }
    constexpr int NUS = TL::tile_off(NU);
    int s = 0;
    double a_cur = cur[lane];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
      for (int t = 0; t < NU; ++t) {
        if (ks < TL::steps(t)) {
          const double a = a_cur;
          ++s;
          if (s < NUS) a_cur = cur[s * 64 + lane];
#pragma unroll
          for (int f = 0; f < FT; ++f) acc[f][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, xb[f][ks], acc[f][t], 0, 0, 0);
        }
      }
    }
    nmfma_wave += FT * NUS;
    double qv[FT];
#pragma unroll
    for (int f = 0; f < FT; ++f) {
      double qq = 0.0;
#pragma unroll
      for (int t = 0; t < NU; ++t) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (16 * t + 4 * r < DP) qq = fma(acc[f][t][r], acc[f][t][r], qq);
      }
      qv[f] = qq;
    }
    double lsel;
    if constexpr (PAIRED) lsel = lc - 0.5 * sum_lane_groups_pair(qv[0], qv[1]);
    else lsel = lc - 0.5 * sum_lane_groups(qv[0]);
    if (tested && __builtin_amdgcn_ballot_w64(lsel > runmax[0] - prune) == 0) return;
    nreg_wave += __builtin_popcount(tiles_in_range);
    nmfma_wave += FT * (TL::NSTEPS - NUS);
    if (__builtin_amdgcn_ballot_w64(lsel > runmax[0]) != 0) {      // lazy rescale (wave-uniform; the factor is exactly 1 elsewhere)
      const double nm = fmax(runmax[0], lsel);
      const double scs = vc_exp(runmax[0] - nm);
      den[0] *= scs;
      runmax[0] = nm;
      double sc[FT];
      if constexpr (PAIRED) unpair_lane_groups(scs, sc[0], sc[1]);
      else sc[0] = scs;
#pragma unroll
      for (int f = 0; f < FT; ++f) {
#pragma unroll
        for (int j = 0; j < KS; ++j) yacc[f][j] *= sc[f];
      }
    }
    double wg[FT];
    {
      const double e = vc_exp_tab(lsel - runmax[0], etab);
      den[0] += e;
      if constexpr (PAIRED) unpair_lane_groups(e, wg[0], wg[1]);
      else wg[0] = e;
    }
#pragma unroll
    for (int t = NU; t < NT; ++t) {
      d4 c;
#pragma unroll
      for (int r = 0; r < 4; ++r) c[r] = cur[TL::CINIT_OFF + 16 * t + 4 * r + lgrp];
#pragma unroll
      for (int f = 0; f < FT; ++f) acc[f][t] = c;
    }
    int sa = NUS;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
      for (int t = NU; t < NT; ++t) {
        const double a = cur[sa * 64 + lane];
        ++sa;
#pragma unroll
        for (int f = 0; f < FT; ++f) acc[f][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, xb[f][ks], acc[f][t], 0, 0, 0);
      }
    }
#pragma unroll
    for (int f = 0; f < FT; ++f) {
#pragma unroll
      for (int t = NU - 1; t < NT; ++t) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int p0 = 16 * t + 4 * r;
          if (p0 >= DP && p0 < 2 * DP) {
            const int j = (p0 - DP) / 4;
            yacc[f][j] = fma(wg[f], acc[f][t][r], yacc[f][j]);
          }
        }
      }
    }
  };

  // ---- every mixture of a bitmap (except `skip`) in full, in index order; blocks alternate between the buffers starting with
  // `first_buf`, which every wave must have left (the caller's barrier); ends with everyone out of both buffers' readers... the
  // bitmap must be complete and visible (a barrier since its last update)
  auto eval_bitmap = [&](const unsigned *bm, int first_buf, int skip, bool reload_x) -> int {
    const int nwords = (M + 31) / 32;
    int w = 0;
    unsigned bits = __builtin_amdgcn_readfirstlane(bm[0]);
    auto next = [&]() -> int {                                   // next mixture of the bitmap, -1 when there is none
      for (;;) {
        while (bits == 0u) {
          if (++w >= nwords) return -1;
          bits = __builtin_amdgcn_readfirstlane(bm[w]);
        }
        const int b = __builtin_ctz(bits);
        bits &= bits - 1u;
        if (32 * w + b != skip) return 32 * w + b;
      }
    };
    int cur_m = next(), p = first_buf, n = 0;
    if (cur_m < 0) return 0;
    __syncthreads();                                             // everyone has left buffer p
    dma_block(cur_m, p);
    // (B16: the FP64 operands of x were given up for the screen -- the survivors are rare -- and come back from L2 here)
    if (reload_x) load_x();
    while (cur_m >= 0) {
      ++n;
      const int nxt_m = next();
      const double lc = packed_c[(size_t)cur_m * BLK + TL::LC_OFF];
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();                                           // block cur_m is in buffer p; everyone has left buffer p ^ 1
      if (nxt_m >= 0) dma_block(nxt_m, p ^ 1);
      full_mixture(smem + p * BUF, lc, true);
      cur_m = nxt_m;
      p ^= 1;
    }
    return n;
  };

  // ---- 1. the group's own mixture
  const double lc_g = packed_c[(size_t)mg * BLK + TL::LC_OFF];
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  full_mixture(smem, lc_g, false);
  // the other groups present in the workgroup (it straddles a group boundary: rare on large calls, the rule on small ones)
  // (blocks go to buffer 0, 1, 0, ...: from the second one on they overwrite stage 0, which is then fetched again)
  const int nkeys = eval_bitmap(keys, 0, mg, false);
  // the thresholds of the screen, per tile in every lane (a lane group of a screening tile is a MIXTURE, not a tile)
  double thr[FT];
  if constexpr (PAIRED) {
    double r0, r1;
    unpair_lane_groups(runmax[0], r0, r1);
    thr[0] = r0 - prune;
    thr[1] = r1 - prune;
  } else {
    thr[0] = runmax[0] - prune;
  }
#pragma unroll
  for (int f = 0; f < FT; ++f)
    if (!(tiles_in_range >> f & 1u)) thr[f] = INFINITY;            // a tile beyond T: no mixture passes on its account
  // B16: the B operands of the screen from the lane's own FP64 operands (slot j of the K = 32 instructions <-> k-step j), and
  // |x| per frame for the error bound; xb itself is not needed again unless something survives
  constexpr int NMAIN = KS < 8 ? KS : 8, NTAIL = KS - NMAIN;     // k-steps in the three main instructions / in the tail one
  u32x4_t bh[B16 ? FT : 1], bl[B16 ? FT : 1], bt[B16 ? FT : 1];
  float nxf[B16 ? FT : 1];                                       // |x| of the lane's frame, rounded up
  if constexpr (B16) {
#pragma unroll
    for (int f = 0; f < FT; ++f) {
      unsigned short h[KS], l[KS];
      double q = 0.0;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        split_bf16(xb[f][ks], h[ks], l[ks]);
        q = fma(xb[f][ks], xb[f][ks], q);
      }
      nxf[f] = (float)(sqrt(sum_lane_groups(q)) * (1.0 + 0x1p-20));
#pragma unroll
      for (int w2 = 0; w2 < 4; ++w2) {
        const int j0 = 2 * w2, j1 = 2 * w2 + 1;
        bh[f][w2] = (unsigned)(j0 < NMAIN ? h[j0] : 0) | ((unsigned)(j1 < NMAIN ? h[j1] : 0) << 16);
        bl[f][w2] = (unsigned)(j0 < NMAIN ? l[j0] : 0) | ((unsigned)(j1 < NMAIN ? l[j1] : 0) << 16);
      }
      // tail slots: {xh8, xh9, xl8, xl9, xh8, xh9, 0, 0}  against  {Ph8, Ph9, Ph8, Ph9, Pl8, Pl9, 0, 0}
      unsigned short t0h = 0, t1h = 0, t0l = 0, t1l = 0;
      if constexpr (NTAIL > 0) {
        t0h = h[NMAIN];
        t0l = l[NMAIN];
      }
      if constexpr (NTAIL > 1) {
        t1h = h[NMAIN + 1];
        t1l = l[NMAIN + 1];
      }
      const unsigned th = (unsigned)t0h | ((unsigned)t1h << 16), tl = (unsigned)t0l | ((unsigned)t1l << 16);
      bt[f][0] = th;
      bt[f][1] = tl;
      bt[f][2] = th;
      bt[f][3] = 0u;
    }
  }
  __syncthreads();                                               // everyone is done with the block buffers
  if (nkeys >= 2) {
    dma_stage(0, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  // ---- 2. screen every quad: stage s is in buffer (s + 1) & 1, stage s + 1 is fetched into the other one meanwhile
  const int mpt = 16 / rpm;                                      // mixtures per screening tile
  const int nstages = (M + mpt * QS - 1) / (mpt * QS);
  for (int s = 0; s < nstages; ++s) {
    if (s + 1 < nstages) dma_stage(s + 1, s & 1);
    __builtin_amdgcn_sched_barrier(0);
    const double *stg = smem + ((s + 1) & 1) * BUF;
    // bit 4 q + u of a lane: sub-mixture u of the lane's group in tile q of the stage is NOT ruled out for the lane's frame on
    // some tile of the wave.  Collected branch-free (a compare and an or per test) and looked at once per stage, so that the
    // four tiles of a stage are one straight block: their operand reads, MFMAs and tests overlap
    unsigned lanebits = 0;
    auto screen_tile16 = [&](int q) {                            // B16: lane group j <-> mixture j of the tile, its four rows in the lane
      const char *tb = reinterpret_cast<const char *>(stg + q * screen16_tile_doubles());
      const double *cl = stg + QS * screen16_tile_doubles() + q * 32 + lgrp * 8;
      const u32x4_t aph = *reinterpret_cast<const u32x4_t *>(tb + 16 * lane);
      const u32x4_t apl = *reinterpret_cast<const u32x4_t *>(tb + 1024 + 16 * lane);
      const u32x4_t apt = *reinterpret_cast<const u32x4_t *>(tb + 2048 + 16 * lane);
      const f32x4_t cc = *reinterpret_cast<const f32x4_t *>(cl), np = *reinterpret_cast<const f32x4_t *>(cl + 2),
                    nc = *reinterpret_cast<const f32x4_t *>(cl + 4);
      const double lcq = cl[6];
      f32x4_t a[FT];
#pragma unroll
      for (int f = 0; f < FT; ++f) {
        a[f] = -cc;
        a[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, aph), __builtin_bit_cast(bf16x8_t, bh[f]), a[f], 0, 0, 0);
        a[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, aph), __builtin_bit_cast(bf16x8_t, bl[f]), a[f], 0, 0, 0);
        a[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, apl), __builtin_bit_cast(bf16x8_t, bh[f]), a[f], 0, 0, 0);
        if constexpr (NTAIL > 0)
          a[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, apt), __builtin_bit_cast(bf16x8_t, bt[f]), a[f], 0, 0, 0);
      }
#pragma unroll
      for (int f = 0; f < FT; ++f) {
        float lb = 0.0f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float t = fmaxf(fabsf(a[f][i]) - fmaf(np[i], nxf[f], nc[i]), 0.0f);          // certified |a_i| from below
          lb = fmaf(t, t, lb);
        }
        lanebits |= (fma(-0.5 * (1.0 - 0x1p-20), (double)lb, lcq) > thr[f]) ? (1u << (4 * q)) : 0u;
      }
    };
    auto screen_tile = [&](int q, auto rpm_c) {                  // (rows per mixture as a compile-time constant: no branch inside a tile)
      constexpr int RPM = decltype(rpm_c)::value;
      const double *fq = stg + q * (KS * 64) + lane;
      const double *cl = stg + QFR + q * 32 + lgrp * 8;
      double afr[KS];
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) afr[ks] = fq[ks * 64];
      d4 c;
#pragma unroll
      for (int r = 0; r < 4; ++r) c[r] = cl[r];
      double lcq[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) lcq[u] = cl[4 + u];
      d4 a[FT];
#pragma unroll
      for (int f = 0; f < FT; ++f) a[f] = c;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
        for (int f = 0; f < FT; ++f) a[f] = __builtin_amdgcn_mfma_f64_16x16x4f64(afr[ks], xb[f][ks], a[f], 0, 0, 0);
      }
      nmfma_wave += FT * KS;
#pragma unroll
      for (int f = 0; f < FT; ++f) {
        const double s0 = a[f][0] * a[f][0], s1 = a[f][1] * a[f][1], s2 = a[f][2] * a[f][2], s3 = a[f][3] * a[f][3];
        const double th = thr[f];                                // (+inf for a tile beyond T: nothing passes)
        if constexpr (RPM == 4) {
          lanebits |= (fma(-0.5, (s0 + s1) + (s2 + s3), lcq[0]) > th) ? (1u << (4 * q)) : 0u;
        } else if constexpr (RPM == 2) {
          lanebits |= (fma(-0.5, s0 + s1, lcq[0]) > th) ? (1u << (4 * q)) : 0u;
          lanebits |= (fma(-0.5, s2 + s3, lcq[1]) > th) ? (2u << (4 * q)) : 0u;
        } else {
          lanebits |= (fma(-0.5, s0, lcq[0]) > th) ? (1u << (4 * q)) : 0u;
          lanebits |= (fma(-0.5, s1, lcq[1]) > th) ? (2u << (4 * q)) : 0u;
          lanebits |= (fma(-0.5, s2, lcq[2]) > th) ? (4u << (4 * q)) : 0u;
          lanebits |= (fma(-0.5, s3, lcq[3]) > th) ? (8u << (4 * q)) : 0u;
        }
      }
    };
    const int nq = (M - QS * s * mpt + mpt - 1) / mpt;           // tiles of this stage that hold a mixture (the last stage may have fewer)
    auto screen_stage = [&](auto rpm_c) {
      if (nq >= QS) {
#pragma unroll
        for (int q = 0; q < QS; ++q) screen_tile(q, rpm_c);
      } else {
        for (int q = 0; q < nq; ++q) screen_tile(q, rpm_c);
      }
    };
    if constexpr (B16) {
      if (nq >= QS) {
#pragma unroll
        for (int q = 0; q < QS; ++q) screen_tile16(q);
      } else {
        for (int q = 0; q < nq; ++q) screen_tile16(q);
      }
      nmfma16_wave += (nq < QS ? nq : QS) * FT * (3 + (NTAIL > 0 ? 1 : 0));
    } else {
      if (rpm == 4) screen_stage(std::integral_constant<int, 4>{});
      else if (rpm == 2) screen_stage(std::integral_constant<int, 2>{});
      else screen_stage(std::integral_constant<int, 1>{});
    }
    if (__builtin_amdgcn_ballot_w64(lanebits != 0u) != 0) {      // rare: some mixture of the stage is not ruled out for some frame
      const int per = 4 / rpm;                                   // sub-mixtures per lane group
      while (lanebits) {                                         // (divergent: a few lanes, a few bits)
        const int bit = __builtin_ctz(lanebits);
        lanebits &= lanebits - 1u;
        const int m = (QS * s + (bit >> 2)) * mpt + per * lgrp + (bit & 3);
        if (m < M && !(keys[m >> 5] >> (m & 31) & 1u)) atomicOr(&survivors[m >> 5], 1u << (m & 31));
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  // ---- 3. the survivors in full, in index order (the bitmap is complete and visible: the loop above ended with a barrier)
  eval_bitmap(survivors, 0, -1, B16);

  if (nreg && lane == 0) {
    atomicAdd(nreg, (unsigned long long)nreg_wave);
    atomicAdd(nreg + 1, (unsigned long long)nmfma_wave);
    if (B16) atomicAdd(nreg + 2, (unsigned long long)nmfma16_wave);
  }
  if constexpr (PAIRED) {
    const double ds = den[0];
    unpair_lane_groups(ds, den[0], den[1]);
  }
#pragma unroll
  for (int f = 0; f < FT; ++f) {
    const int64_t fr = frame0 + 16 * f + lcol;
    const double inv = 1.0 / den[f];
    // rows of y as whole 128-byte lines where they can be (round 6: store_frame_row, gmmmap.hip)
    double yo[KS];
#pragma unroll
    for (int j = 0; j < KS; ++j) yo[j] = yacc[f][j] * inv;
    store_frame_row<KS>(Y + (fr < T ? frow[f] : (int64_t)0) * ldy, fr < T, rows_as_lines(Y, ldy, D, DP), D, lgrp, yo);
  }
}


// ------------------------------------------------------------------------------------------------
// predict (src/gmm.jl:44-47: the index of the first maximum of the log-weighted densities) on GROUPED frames with the same
// screen.  An arg-max needs no margin: mixture m is out as soon as its bound lc_m - |P_m (x - mu_m)|^2 / 2 lies below the best
// log-density found so far -- so the screen decides far more often than fvconvert's (which needs e^-prune of room), on broad
// models too.  Per workgroup: the whitening of the group keys' mixtures in full (running maximum + index), the four-row screen
// of every other mixture against the running maximum (>=: a tie must be looked at), survivors in full.  The result is EXACT:
// a mixture is skipped only when an upper bound of its log-density is strictly below a log-density that was evaluated, and
// ties go to the smaller index whatever the evaluation order.  U-only blocks (Tiling<DP, true>: packedU), DP up to 80 -- what
// the trajectory conversion's m-hat needs (static + delta source vectors).
// ------------------------------------------------------------------------------------------------
template <int DP, int WAVES>
__global__ void __launch_bounds__(WAVES * 64) __attribute__((amdgpu_waves_per_eu(DP <= 40 ? 3 : 2)))
gmmmap_screen_argmax_kernel(const double *__restrict__ packedU, const double *__restrict__ packedQ, int M, int D,
                            const double *__restrict__ X, int64_t ldx, int64_t T, int64_t *__restrict__ idx,
                            const int *__restrict__ perm, const int *__restrict__ gkey) {
  using TL = Tiling<DP, true>;
  constexpr int FT = 2;
  constexpr int KS = TL::KS, NU = TL::NT, BLK = TL::BLK;
  constexpr int QS = screen_quads(DP), QFR = screen_frag_doubles(DP), STG = screen_stage_doubles(DP);
  constexpr int BUF = (BLK > STG) ? BLK : STG;
  constexpr int NI_BLK = BLK / 128, NI_STG = STG / 128;
  extern __shared__ double smem[];
  __shared__ unsigned survivors[32];
  __shared__ unsigned keys[32];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const unsigned lane_off = 16u * (unsigned)lane;
  const int lcol = lane & 15, lgrp = lane >> 4;
  const int64_t frame0 = ((int64_t)blockIdx.x * WAVES + wave) * (16 * FT);
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)(reinterpret_cast<char *>(smem));
  if (tid < 32) survivors[tid] = 0u;
  else if (tid < 64) keys[tid - 32] = 0u;

  int mg = 0;
  {
    const int64_t f0 = (int64_t)blockIdx.x * WAVES * (16 * FT);
    const __attribute__((address_space(4))) int *perm_c = (const __attribute__((address_space(4))) int *)perm;
    const __attribute__((address_space(4))) int *gkey_c = (const __attribute__((address_space(4))) int *)gkey;
    mg = (f0 < T) ? gkey_c[perm_c[f0]] : 0;
    mg = (mg >= 0 && mg < M) ? mg : 0;
  }
  auto dma_block = [&](int m, int buf) {
    const char *gb = reinterpret_cast<const char *>(packedU + (size_t)m * BLK);
    const unsigned lb = lds0 + (unsigned)buf * (BUF * 8u);
#pragma unroll
    for (int i = 0; i < (NI_BLK + WAVES - 1) / WAVES; ++i) {
      const int k = wave_u + WAVES * i;
      if (k < NI_BLK) dma_1k(gb + 1024 * k, lb + 1024u * k, lane_off);
    }
  };
  auto dma_stage = [&](int s, int buf) {
    const char *gb = reinterpret_cast<const char *>(packedQ + (size_t)s * STG);
    const unsigned lb = lds0 + (unsigned)buf * (BUF * 8u);
#pragma unroll
    for (int i = 0; i < (NI_STG + WAVES - 1) / WAVES; ++i) {
      const int k = wave_u + WAVES * i;
      if (k < NI_STG) dma_1k(gb + 1024 * k, lb + 1024u * k, lane_off);
    }
  };
  dma_block(mg, 0);
  dma_stage(0, 1);
  __builtin_amdgcn_sched_barrier(0);

  double xb[FT][KS];
  int64_t frow[FT];
  unsigned tiles_in_range = 0;
#pragma unroll
  for (int f = 0; f < FT; ++f) {
    const int64_t fr = frame0 + 16 * f + lcol;
    frow[f] = (fr < T) ? (int64_t)perm[fr] : fr;
    if (frame0 + 16 * f < T) tiles_in_range |= 1u << f;
    load_frame_row<KS>(X + (fr < T ? frow[f] : (int64_t)0) * ldx, fr < T, rows_as_lines(X, ldx, D, DP), D, lgrp, xb[f]);
  }
  __syncthreads();                                               // the bitmaps are zeroed
  if (lgrp == 0) {
#pragma unroll
    for (int f = 0; f < FT; ++f) {
      const int64_t fr = frame0 + 16 * f + lcol;
      if (fr < T) {
        const int k = gkey[frow[f]];
        if (k >= 0 && k < M) atomicOr(&keys[k >> 5], 1u << (k & 31));
      }
    }
  }
  // selected layout: the even lane groups carry tile 0's running maximum and index, the odd ones tile 1's
  double runmax = -INFINITY;
  int bestm = 0;
  const __attribute__((address_space(4))) double *packed_c = (const __attribute__((address_space(4))) double *)packedU;

  // ---- the log-weighted density of one mixture for the wave's frames (all whitening tiles), then the arg-max update ----
  auto full_l = [&](const double *cur, double lc, int m) {
    if ((unsigned)((unsigned long long)__double_as_longlong(lc) >> 32) == 0xFFF00000u) return;      // zero weight: l = -inf, never a maximum
    d4 acc[FT][NU];
#pragma unroll
    for (int t = 0; t < NU; ++t) {
      d4 c;
#pragma unroll
      for (int r = 0; r < 4; ++r) c[r] = cur[TL::CINIT_OFF + 16 * t + 4 * r + lgrp];
#pragma unroll
      for (int f = 0; f < FT; ++f) acc[f][t] = c;
    }
    constexpr int NUS = TL::NSTEPS;
    int s = 0;
    double a_cur = cur[lane];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
      for (int t = 0; t < NU; ++t) {
        if (ks < TL::steps(t)) {
          const double a = a_cur;
          ++s;
          if (s < NUS) a_cur = cur[s * 64 + lane];
#pragma unroll
          for (int f = 0; f < FT; ++f) acc[f][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, xb[f][ks], acc[f][t], 0, 0, 0);
        }
      }
    }
    double qv[FT];
#pragma unroll
    for (int f = 0; f < FT; ++f) {
      double qq = 0.0;
#pragma unroll
      for (int t = 0; t < NU; ++t) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (16 * t + 4 * r < DP) qq = fma(acc[f][t][r], acc[f][t][r], qq);
      }
      qv[f] = qq;
    }
    const double lsel = lc - 0.5 * sum_lane_groups_pair(qv[0], qv[1]);
    // the first maximum in INDEX order, whatever the order of evaluation: a tie goes to the smaller index
    const bool take = (lsel > runmax) || (lsel == runmax && m < bestm);
    runmax = take ? lsel : runmax;
    bestm = take ? m : bestm;
  };
  auto eval_bitmap = [&](const unsigned *bm, int first_buf, int skip) -> int {
    const int nwords = (M + 31) / 32;
    int w = 0;
    unsigned bits = __builtin_amdgcn_readfirstlane(bm[0]);
    auto next = [&]() -> int {
      for (;;) {
        while (bits == 0u) {
          if (++w >= nwords) return -1;
          bits = __builtin_amdgcn_readfirstlane(bm[w]);
        }
        const int b = __builtin_ctz(bits);
        bits &= bits - 1u;
        if (32 * w + b != skip) return 32 * w + b;
      }
    };
    int cur_m = next(), p = first_buf, n = 0;
    if (cur_m < 0) return 0;
    __syncthreads();
    dma_block(cur_m, p);
    while (cur_m >= 0) {
      ++n;
      const int nxt_m = next();
      const double lc = packed_c[(size_t)cur_m * BLK + TL::LC_OFF];
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (nxt_m >= 0) dma_block(nxt_m, p ^ 1);
      full_l(smem + p * BUF, lc, cur_m);
      cur_m = nxt_m;
      p ^= 1;
    }
    return n;
  };

  // ---- 1. the mixtures of the workgroup's group keys
  const double lc_g = packed_c[(size_t)mg * BLK + TL::LC_OFF];
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  full_l(smem, lc_g, mg);
  const int nkeys = eval_bitmap(keys, 0, mg);
  double thr[FT];
  {
    double r0, r1;
    unpair_lane_groups(runmax, r0, r1);
    // (a hair of slack: the bound and the log-density it is compared with are rounded differently -- a mixture whose bound
    // misses the running maximum by less than that is simply evaluated)
    thr[0] = (tiles_in_range & 1u) ? r0 - (1e-7 + 1e-12 * fabs(r0)) : INFINITY;
    thr[1] = (tiles_in_range & 2u) ? r1 - (1e-7 + 1e-12 * fabs(r1)) : INFINITY;
  }
  __syncthreads();
  if (nkeys >= 2) {
    dma_stage(0, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  // ---- 2. the screen: four mixtures per tile, against the best log-density so far (>=: a tie has to be evaluated)
  const int nstages = (M + 4 * QS - 1) / (4 * QS);
  for (int s = 0; s < nstages; ++s) {
    if (s + 1 < nstages) dma_stage(s + 1, s & 1);
    __builtin_amdgcn_sched_barrier(0);
    const double *stg = smem + ((s + 1) & 1) * BUF;
    unsigned lanebits = 0;
    auto screen_tile = [&](int q) {
      const double *fq = stg + q * (KS * 64) + lane;
      const double *cl = stg + QFR + q * 32 + lgrp * 8;
      d4 c;
#pragma unroll
      for (int r = 0; r < 4; ++r) c[r] = cl[r];
      const double lcq = cl[4];
      d4 a[FT];
#pragma unroll
      for (int f = 0; f < FT; ++f) a[f] = c;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const double afr = fq[ks * 64];
#pragma unroll
        for (int f = 0; f < FT; ++f) a[f] = __builtin_amdgcn_mfma_f64_16x16x4f64(afr, xb[f][ks], a[f], 0, 0, 0);
      }
#pragma unroll
      for (int f = 0; f < FT; ++f) {
        const double s0 = a[f][0] * a[f][0], s1 = a[f][1] * a[f][1], s2 = a[f][2] * a[f][2], s3 = a[f][3] * a[f][3];
        lanebits |= (fma(-0.5, (s0 + s1) + (s2 + s3), lcq) >= thr[f]) ? (1u << q) : 0u;
      }
    };
    const int nq = (M - QS * s * 4 + 3) / 4;
    if (nq >= QS) {
#pragma unroll
      for (int q = 0; q < QS; ++q) screen_tile(q);
    } else {
      for (int q = 0; q < nq; ++q) screen_tile(q);
    }
    if (__builtin_amdgcn_ballot_w64(lanebits != 0u) != 0) {
      while (lanebits) {
        const int bit = __builtin_ctz(lanebits);
        lanebits &= lanebits - 1u;
        const int m = (QS * s + bit) * 4 + lgrp;
        if (m < M && !(keys[m >> 5] >> (m & 31) & 1u)) atomicOr(&survivors[m >> 5], 1u << (m & 31));
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  // ---- 3. the survivors in full
  eval_bitmap(survivors, 0, -1);

  if (lgrp < FT) {                                               // lane group 0 holds tile 0's result, lane group 1 tile 1's
    const int64_t fr = frame0 + 16 * lgrp + lcol;
    if (fr < T) idx[lgrp == 0 ? frow[0] : frow[1]] = (int64_t)bestm + 1;
  }
}

}  // namespace vcmi
