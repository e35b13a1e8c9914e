// gmmmap_screen_pers.hpp -- shape 3 of fvconvert (gmmmap_screen.hpp) with PERSISTENT, mutually independent waves (round 6;
// included by gmmmap.hip).
//
// Round 5's gmmmap_screen_kernel ran one workgroup per 128 frames: perm[f0] -> gkey[.] -> LDS-DMA of the group's block, the
// gather of its frames, a barrier per screening stage around a DMA of the stage's operands (the same 53 KB for every
// workgroup), a workgroup-wide bitmap of survivors evaluated by all four waves behind more barriers.  Its counters
// (profiles/r06_ab/convert_sq_counters.txt): a wave issues instructions 18 % of its time, waits for memory / barriers 40 %.
// A first persistent version that kept the workgroup's shared block buffer (one barrier per job) was no faster: one wave in
// six has a survivor of its own screen to evaluate, and every job waited at the barrier for the slowest of its waves.
// Here NOTHING couples the waves of a workgroup after the prologue:
//   * ALL screening stages are staged into LDS once per workgroup (bf16 four-row screen: 13 KB per 16 mixtures); the screen
//     of a wave-job (32 frames: two tiles) is one straight run of LDS reads and v_mfma_f32_16x16x32_bf16;
//   * every mixture a wave evaluates in full -- the group of its frames (all of them on a call grouped over 10^6 frames), the
//     other groups of a wave that straddles a boundary, the survivors of its own screen -- streams its operand fragments from
//     L2 / L1 through a ring of eight loads in flight (the waves of a CU work on neighbouring frames of the grouped order, i.e.
//     on the same 22 KB block, which the vector L1 then serves); no block buffer in LDS, no barrier, no bitmap;
//   * wave-jobs are taken with a fixed stride (wave-job k -> wave k mod (waves of the grid)): the tiles, the evaluation sets
//     and every sum are a function of the data alone;
//   * the gather is software-pipelined two jobs deep: rows (perm, gkey) of job k + 2 and frames of job k + 1 are requested
//     while job k is screened, into a second set of operand registers.
// Arithmetic per (frame, mixture) exactly as gmmmap_screen_kernel<DP, 2, ., true>; the set of mixtures evaluated for a frame
// is its wave's (not its workgroup's) keys and survivors, so the two kernels agree to the rounding of terms below e^-prune.
// Covers the bf16 screen with four rows per mixture (DP <= 40) and M <= 64 (the wave's set of evaluated mixtures is one 64-bit
// mask); everything else stays with gmmmap_screen_kernel.
#pragma once
#include "gmmmap_screen.hpp"

#ifndef VCMI_PERS_SCHED
#define VCMI_PERS_SCHED 0    // 0: the four tiles of a stage are one scheduling region; 1 / 2: a fence after every / every second tile
#endif
#ifndef VCMI_PERS_EXP
#define VCMI_PERS_EXP 0      // experiment switches of tools/pers_exp.sh (never set in the product build): 1 no screen, 2 no full
#endif                      // evaluation, 4 no stores, 8 frames read in grouped-order positions (no gather), 16 rows stored likewise
namespace vcmi {

#if VCMI_PERS_EXP & 32
__device__ unsigned long long pers_prof[8];      // wave cycles: round 0 (keys), operand prep, screen, survivors, stores + rotate, jobs
#define PERS_T(k) { const unsigned long long t_ = __builtin_readcyclecounter(); pp[k] += t_ - tlast; tlast = t_; }
#else
#define PERS_T(k)
#endif

__host__ __device__ constexpr int screen_pers_waves() { return 4; }
template <int DP>
__host__ __device__ constexpr size_t screen_pers_lds_bytes(int M) {
  return (size_t)((M + 4 * screen_quads(DP) - 1) / (4 * screen_quads(DP))) * screen16_stage_doubles(DP) * sizeof(double);
}
__host__ __device__ constexpr size_t screen_pers_lds_budget() { return 78 * 1024; }      // two workgroups per CU beside the static arrays

template <int DP, int WAVES>
__global__ void __launch_bounds__(WAVES * 64) __attribute__((amdgpu_waves_per_eu(2, 2)))
gmmmap_screen_pers_kernel(const double *__restrict__ packed, const double *__restrict__ packedQ16, int M, int D,
                          const double *__restrict__ X, int64_t ldx, int64_t T, double *__restrict__ Y, int64_t ldy, double prune,
                          unsigned long long *__restrict__ nreg, const int *__restrict__ perm, const int *__restrict__ gkey) {
  using TL = Tiling<DP, false>;
  constexpr int FT = 2;
  constexpr int KS = TL::KS, NT = TL::NT, NU = TL::NU, BLK = TL::BLK;
  constexpr int QS = screen_quads(DP), STG = screen16_stage_doubles(DP);
  constexpr int RING = 8;                                        // operand fragments of a mixture in flight from L2
  static_assert(screen16_has(DP), "the persistent kernel runs the bf16 screen");
  extern __shared__ double smem[];                               // [nstages * STG]
  __shared__ double etab[64];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const unsigned lane_off = 16u * (unsigned)lane;
  const int lcol = lane & 15, lgrp = lane >> 4;
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)(reinterpret_cast<char *>(smem));
  const int nstages = (M + 4 * QS - 1) / (4 * QS);
  const int64_t njobs = (T + 16 * FT - 1) / (16 * FT);           // wave-jobs of 32 frames
  const int64_t stride = (int64_t)gridDim.x * WAVES;

  if (tid < 64) etab[tid] = kExp2Tab[tid];
  {   // every stage of the screen -> LDS, once (each wave issues every WAVES-th KB)
    const int ni = nstages * (STG / 128);
    const char *gb = reinterpret_cast<const char *>(packedQ16);
    for (int k = wave_u; k < ni; k += WAVES) dma_1k(gb + 1024 * (size_t)k, lds0 + 1024u * (unsigned)k, lane_off);
  }
  const __attribute__((address_space(4))) double *packed_c = (const __attribute__((address_space(4))) double *)packed;

  // the wave's frames of a job: positions fr = 32 j + 16 f + lcol of the grouped order, rows perm[fr] of the caller's (-1 beyond T),
  // and their group keys
  auto load_rows = [&](int64_t j, int *fro, int *kfo) {
#pragma unroll
    for (int f = 0; f < FT; ++f) {
      const int64_t fr = j * (16 * FT) + 16 * f + lcol;
      fro[f] = (j < njobs && fr < T) ? ((VCMI_PERS_EXP & 8) ? (int)fr : perm[fr]) : -1;
    }
#pragma unroll
    for (int f = 0; f < FT; ++f) kfo[f] = (fro[f] >= 0) ? gkey[fro[f]] : -1;
  };
  // plain global loads straight into operand registers: nothing waits for the data until the first MFMA that uses it.  A
  // position beyond T reads row 0 -- its column of every product is independent of the others and is never stored --, a
  // feature beyond D (only the last k-step can hold one: DP - D < 4) reads feature 0 and is zeroed by the one select below
  auto load_x = [&](const int *fro, double (*xo)[KS]) {
#pragma unroll
    for (int f = 0; f < FT; ++f) {
      const double *xr = X + (int64_t)(fro[f] >= 0 ? fro[f] : 0) * ldx;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int k = 4 * ks + lgrp;
        if (ks < KS - 1) {
          xo[f][ks] = xr[k];
        } else {
          const double v = xr[k < D ? k : 0];
          xo[f][ks] = (k < D) ? v : 0.0;
        }
      }
    }
  };

  int64_t j = (int64_t)blockIdx.x * WAVES + wave;                // this wave's first job (a wave of the last workgroup may have none)
  double xb[FT][KS], xn[FT][KS];
  int frow[FT], kf[FT], frow_n[FT], kf_n[FT];
  load_rows(j, frow, kf);
  load_rows(j + stride, frow_n, kf_n);
  load_x(frow, xb);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();                                               // the stages (and the exp table) are in LDS: the only barrier

  int nreg_wave = 0, nmfma_wave = 0, nmfma16_wave = 0;
  constexpr int NMAIN = KS < 8 ? KS : 8, NTAIL = KS - NMAIN;     // k-steps in the three main bf16 instructions / in the tail one

#if VCMI_PERS_EXP & 32
  unsigned long long pp[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_readcyclecounter();
#endif
  for (; j < njobs; j += stride) {
    double yacc[FT][KS];
    double runmax[FT], den[FT];          // [0] holds tile 0's value in the even lane groups, tile 1's in the odd ones (paired layout)
#pragma unroll
    for (int f = 0; f < FT; ++f) {
      runmax[f] = -INFINITY;
      den[f] = 0.0;
#pragma unroll
      for (int jj = 0; jj < KS; ++jj) yacc[f][jj] = 0.0;
    }
    unsigned tiles_in_range = 0;
#pragma unroll
    for (int f = 0; f < FT; ++f)
      if (j * (16 * FT) + 16 * f < T) tiles_in_range |= 1u << f;

    // ---- one mixture in full, operand fragments from L2 through a ring of RING loads in flight: whitening, (test,) regression,
    // online softmax update -- gmmmap_screen_kernel's body
    auto full_mixture = [&](int m, bool tested) {
      const double *cur = packed + (size_t)m * BLK;
      const double lc = packed_c[(size_t)m * BLK + TL::LC_OFF];
      if ((unsigned)((unsigned long long)__double_as_longlong(lc) >> 32) == 0xFFF00000u) return;      // zero weight: posterior exactly 0
      constexpr int NUS = TL::tile_off(NU), NRS = TL::NSTEPS - NUS;
      double ring[RING];
#pragma unroll
      for (int i = 0; i < RING; ++i) ring[i] = (i < TL::NSTEPS) ? cur[i * 64 + lane] : 0.0;
      d4 acc[FT][NT];
#pragma unroll
      for (int t = 0; t < NU; ++t) {
        d4 c;
#pragma unroll
        for (int r = 0; r < 4; ++r) c[r] = cur[TL::CINIT_OFF + 16 * t + 4 * r + lgrp];
#pragma unroll
        for (int f = 0; f < FT; ++f) acc[f][t] = c;
      }
      int s = 0;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
        for (int t = 0; t < NU; ++t) {
          if (ks < TL::steps(t)) {
            const double a = ring[s % RING];
            // (the ring runs on into the regression fragments: they follow the whitening ones in the block)
            if (s + RING < TL::NSTEPS) ring[s % RING] = cur[(s + RING) * 64 + lane];
            ++s;
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
      const double lsel = lc - 0.5 * sum_lane_groups_pair(qv[0], qv[1]);
      if (tested && __builtin_amdgcn_ballot_w64(lsel > runmax[0] - prune) == 0) return;
      nreg_wave += __builtin_popcount(tiles_in_range);
      nmfma_wave += FT * NRS;
      if (__builtin_amdgcn_ballot_w64(lsel > runmax[0]) != 0) {      // lazy rescale (wave-uniform; the factor is exactly 1 elsewhere)
        const double nm = fmax(runmax[0], lsel);
        const double scs = vc_exp(runmax[0] - nm);
        den[0] *= scs;
        runmax[0] = nm;
        double sc[FT];
        unpair_lane_groups(scs, sc[0], sc[1]);
#pragma unroll
        for (int f = 0; f < FT; ++f) {
#pragma unroll
          for (int jj = 0; jj < KS; ++jj) yacc[f][jj] *= sc[f];
        }
      }
      double wg[FT];
      {
        const double e = vc_exp_tab(lsel - runmax[0], etab);
        den[0] += e;
        unpair_lane_groups(e, wg[0], wg[1]);
      }
#pragma unroll
      for (int t = NU; t < NT; ++t) {
        d4 c;
#pragma unroll
        for (int r = 0; r < 4; ++r) c[r] = cur[TL::CINIT_OFF + 16 * t + 4 * r + lgrp];
#pragma unroll
        for (int f = 0; f < FT; ++f) acc[f][t] = c;
      }
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
        for (int t = NU; t < NT; ++t) {
          const double a = ring[s % RING];
          if (s + RING < TL::NSTEPS) ring[s % RING] = cur[(s + RING) * 64 + lane];
          ++s;
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
              const int jj = (p0 - DP) / 4;
              yacc[f][jj] = fma(wg[f], acc[f][t][r], yacc[f][jj]);
            }
          }
        }
      }
    };

    // Two rounds of "a set of mixtures, in index order" share ONE copy of the evaluation body:
    //   round 0: the group keys of the wave's 32 frames -- one mixture on a call grouped over many frames, the first untested;
    //   round 1: the screen against (running maximum - prune), then its survivors for the wave's frames (one wave-job in six
    //            has one on the SURVEY 8(d) model).
    unsigned long long evaluated = 0;
#pragma nounroll
    for (int round = 0; round < 2; ++round) {
      unsigned long long todo = 0;
      if (round == 0) {
        bool pend0 = kf[0] >= 0 && kf[0] < M, pend1 = kf[1] >= 0 && kf[1] < M;
        for (;;) {                                               // wave-wide set of the lanes' keys
          const unsigned long long b0 = __builtin_amdgcn_ballot_w64(pend0), b1 = __builtin_amdgcn_ballot_w64(pend1);
          if ((b0 | b1) == 0) break;
          const int m = b0 ? __builtin_amdgcn_readlane(kf[0], __builtin_ctzll(b0)) : __builtin_amdgcn_readlane(kf[1], __builtin_ctzll(b1));
          todo |= 1ull << m;
          pend0 = pend0 && kf[0] != m;
          pend1 = pend1 && kf[1] != m;
        }
        if (todo == 0) todo = 1ull;                              // (no valid key: cannot happen for a job inside T)
      } else {
        PERS_T(0)
        // the next job's frames: in flight under the screen, into the second operand set
        load_x(frow_n, xn);
        // the thresholds of the screen, per tile in every lane (a lane group of a screening tile is a MIXTURE, not a tile)
        double thr[FT];
        {
          double r0, r1;
          unpair_lane_groups(runmax[0], r0, r1);
          thr[0] = (tiles_in_range & 1u) ? r0 - prune : INFINITY;      // a tile beyond T: no mixture passes on its account
          thr[1] = (tiles_in_range & 2u) ? r1 - prune : INFINITY;
        }
        // B operands of the screen from the lane's own FP64 operands (slot jj of the K = 32 instructions <-> k-step jj), |x| per frame
        u32x4_t bh[FT], bl[FT], bt[FT];
        float nxf[FT];
#pragma unroll
        for (int f = 0; f < FT; ++f) {
          unsigned h[(KS + 1) / 2], l[(KS + 1) / 2];             // pairs of k-steps: {2 p, 2 p + 1}
          double q = 0.0;
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) q = fma(xb[f][ks], xb[f][ks], q);
#pragma unroll
          for (int p = 0; p < (KS + 1) / 2; ++p) split_bf16_pair(xb[f][2 * p], (2 * p + 1 < KS) ? xb[f][2 * p + 1] : 0.0, h[p], l[p]);
          nxf[f] = (float)(sqrt(sum_lane_groups(q)) * (1.0 + 0x1p-20));
#pragma unroll
          for (int w2 = 0; w2 < 4; ++w2) {
            bh[f][w2] = (2 * w2 < NMAIN) ? h[w2] : 0u;
            bl[f][w2] = (2 * w2 < NMAIN) ? l[w2] : 0u;
          }
          // tail slots: {xh8, xh9, xl8, xl9, xh8, xh9, 0, 0}  against  {Ph8, Ph9, Ph8, Ph9, Pl8, Pl9, 0, 0}
          unsigned th = 0u, tl = 0u;
          if constexpr (NTAIL > 0) {
            th = h[NMAIN / 2];
            tl = l[NMAIN / 2];
          }
          bt[f][0] = th;
          bt[f][1] = tl;
          bt[f][2] = th;
          bt[f][3] = 0u;
        }
        PERS_T(1)
        // the screen: every stage from LDS.  Bit m of `mine`: mixture m is not ruled out for this lane's frame
        unsigned long long mine = 0;
        for (int s = 0; s < ((VCMI_PERS_EXP & 1) ? 0 : nstages); ++s) {
          const double *stg = smem + (size_t)s * STG;
          const int nq = (M - 4 * QS * s + 3) / 4;               // tiles of this stage that hold a mixture
          auto screen_tile = [&](int q) {
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
            bool pass = false;
#pragma unroll
            for (int f = 0; f < FT; ++f) {
              float lb = 0.0f;
#pragma unroll
              for (int i = 0; i < 4; ++i) {
                const float t = fmaxf(fabsf(a[f][i]) - fmaf(np[i], nxf[f], nc[i]), 0.0f);          // certified |a_i| from below
                lb = fmaf(t, t, lb);
              }
              pass = pass || (fma(-0.5 * (1.0 - 0x1p-20), (double)lb, lcq) > thr[f]);
            }
            const int m = (4 * s + q) * 4 + lgrp;              // (QS = 4: tile q of stage s holds mixtures 16 s + 4 q .. + 3)
            mine |= (pass && m < M) ? (1ull << m) : 0ull;
          };
          if (nq >= QS) {                                        // (a full stage: four tiles in one straight block, their reads, MFMAs and tests overlap)
#pragma unroll
            for (int q = 0; q < QS; ++q) {
              screen_tile(q);
#if VCMI_PERS_SCHED == 1
              __builtin_amdgcn_sched_barrier(0);
#elif VCMI_PERS_SCHED == 2
              if (q & 1) __builtin_amdgcn_sched_barrier(0);
#endif
            }
          } else {
            for (int q = 0; q < nq; ++q) screen_tile(q);
          }
          nmfma16_wave += (nq < QS ? nq : QS) * FT * (3 + (NTAIL > 0 ? 1 : 0));
        }
        PERS_T(2)
        mine &= ~evaluated;
        if (__builtin_amdgcn_ballot_w64(mine != 0ull) != 0) {
          for (;;) {                                             // wave-wide OR of the lanes' masks
            const unsigned long long bal = __builtin_amdgcn_ballot_w64((mine & ~todo) != 0ull);
            if (bal == 0) break;
            const int src = __builtin_ctzll(bal);
            const unsigned lo = __builtin_amdgcn_readlane((unsigned)mine, src), hi = __builtin_amdgcn_readlane((unsigned)(mine >> 32), src);
            todo |= ((unsigned long long)hi << 32) | lo;
          }
        }
      }
      bool first = (round == 0);
      while (todo) {
        const int m = __builtin_ctzll(todo);
        todo &= todo - 1ull;
        if (!(VCMI_PERS_EXP & 2)) full_mixture(m, !first);
        first = false;
        evaluated |= 1ull << m;
      }
    }

    PERS_T(3)
    // ---- the job's rows of Y, in the caller's order
    {
      const double ds = den[0];
      double d0, d1;
      unpair_lane_groups(ds, d0, d1);
      const double inv[FT] = {1.0 / d0, 1.0 / d1};
#pragma unroll
      for (int f = 0; f < FT; ++f) {
        if (frow[f] >= 0 && (!(VCMI_PERS_EXP & 4) || prune < -1.0)) {      // (EXP 4: a condition the compiler cannot fold: the work stays, the stores never run)
          double *yr = Y + (int64_t)((VCMI_PERS_EXP & 16) ? (int)(j * 32 + 16 * f + lcol) : frow[f]) * ldy + lgrp;
#pragma unroll
          for (int jj = 0; jj < KS; ++jj) {
            if (4 * jj + lgrp < D) yr[4 * jj] = yacc[f][jj] * inv[f];
          }
        }
      }
    }
    PERS_T(4)
    // rotate the pipeline: the next job's frames become the operands, its rows the rows; the rows of the job after are requested
#pragma unroll
    for (int f = 0; f < FT; ++f) {
      frow[f] = frow_n[f];
      kf[f] = kf_n[f];
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) xb[f][ks] = xn[f][ks];
    }
    PERS_T(6)
    load_rows(j + 2 * stride, frow_n, kf_n);
    PERS_T(7)
#if VCMI_PERS_EXP & 32
    pp[5] += 1;
#endif
  }
#if VCMI_PERS_EXP & 32
  if (lane == 0)
    for (int k = 0; k < 8; ++k) atomicAdd(&pers_prof[k], pp[k]);
#endif
  if (nreg && lane == 0) {
    atomicAdd(nreg, (unsigned long long)nreg_wave);
    atomicAdd(nreg + 1, (unsigned long long)nmfma_wave);
    atomicAdd(nreg + 2, (unsigned long long)nmfma16_wave);
  }
}

}  // namespace vcmi
