// estep_small.hpp -- the diagonal E-step for SMALL models: M <= 32 mixtures, i.e. one or two mixture tiles of 16 (included by
// estep.hip; Dj <= 80).  The reference's own sizes are here: bin/train_gmm.jl:84-89 trains 32 mixtures on 80 joint dimensions
// (test/models/clb_and_slt_gmm32_order40.jld).
//
// estep_mfma_kernel<DJ, 0, SHARE = true> runs such a model with four (M <= 32) or eight (M <= 16) of its eight waves on one
// tile, which balances the MFMAs -- but everything around them is laid out for 128 slots: the softmax walks eight slot groups
// per lane with sixteen lanes on a frame (two slots of the eight exist), takes two passes per wave and block, and the one
// workgroup of a CU moves through step A | barrier | softmax | barrier | step B | barrier in lock step, so the FP64 matrix pipe
// idles through every softmax (phase timers at M = 32: 39 % of the workgroup's time in the softmax, 26 % / 34 % in the steps
// that hold all the MFMAs; 0.37 of the FP64 MFMA peak -- DESIGN 3.3 round 6).
//
// Here the workgroup is as small as the model: 2 MT waves (MT = mixture tiles), 32 frames per block, two waves to a tile
// (wave = (tile, half): step A on frame tile `half`, every other k-step of step B).  Then
//   * the softmax is ONE pass per wave and block with four values per lane and 4 MT lanes on a frame (16 / MT frames per wave),
//     all loops as long as the model is wide;
//   * a CU holds two (MT = 2: 256 threads, 55 KB of LDS each) or three (MT = 1) workgroups that run at their own pace: one
//     is in its softmax (vector pipe, LDS) while the other issues MFMAs -- the overlap the barriers deny a single workgroup.
// The arithmetic of a frame is that of estep_mfma_kernel (same expanded form, same refinement rule, same table exp); the
// partial statistics have the same row layout (row = workgroup x half) and go through estep_reduce_kernel.
#pragma once

namespace vcmi {

template <int DJ, int MT>
struct EstepSmallCfg {
  static_assert(MT == 1 || MT == 2, "one or two mixture tiles");
  static constexpr int KS = 2 * DJ / 4, NDT = 2 * DJ / 16;
  static constexpr int NW = 2 * MT;                              // waves per workgroup
  static constexpr int FB = 32;                                  // frames per block: two frame tiles of 16
  static constexpr int RSX = (DJ + 2 + 13) / 32 * 32 + 18;       // as EstepCfg: 16-byte rows, conflict-free column reads
  static constexpr int XBUF = (FB * RSX * 8 + 1023) / 1024 * 128;
  static constexpr int LPF = 4 * MT;                             // softmax lanes per frame (four slots each)
  // LDS row stride of l / gamma: 20 MT doubles -- the softmax's half-wave (32 / LPF frames x LPF consecutive doubles) lands on
  // 64 distinct banks (MT = 2: rows 80 dwords apart -> 0, 16, 32, 48 mod 64; MT = 1: 40 dwords -> 0, 40, 16, 56, 32, 8, 48, 24)
  static constexpr int RSG = 20 * MT;
  static constexpr int WG_PER_CU = MT == 2 ? 2 : 3;
  // [x 2][l / gamma FB x RSG][etab 64][thresholds 16 MT] doubles
  static constexpr size_t LDS_BYTES = ((size_t)2 * XBUF + (size_t)FB * RSG + 64 + 16 * MT) * sizeof(double);
  static_assert(LDS_BYTES * WG_PER_CU <= 160 * 1024, "LDS");
};

// all-reduce over aligned groups of N = 4 or 8 lanes: the first two / three steps of row16_* (fp64_exp.hpp)
template <int N>
__device__ __forceinline__ double rowN_max(double x) {
  x = fmax(x, dpp_row_f64<0xB1>(x));
  x = fmax(x, dpp_row_f64<0x4E>(x));
  if constexpr (N == 8) x = fmax(x, dpp_row_f64<0x141>(x));
  return x;
}
template <int N>
__device__ __forceinline__ double rowN_sum(double x) {
  x += dpp_row_f64<0xB1>(x);
  x += dpp_row_f64<0x4E>(x);
  if constexpr (N == 8) x += dpp_row_f64<0x141>(x);
  return x;
}
template <int N>
__device__ __forceinline__ int rowN_min(int x) {
  x = min(x, dpp_row_i32<0xB1>(x));
  x = min(x, dpp_row_i32<0x4E>(x));
  if constexpr (N == 8) x = min(x, dpp_row_i32<0x141>(x));
  return x;
}
template <int N>
__device__ __forceinline__ int rowN_sum(int x) {
  x += dpp_row_i32<0xB1>(x);
  x += dpp_row_i32<0x4E>(x);
  if constexpr (N == 8) x += dpp_row_i32<0x141>(x);
  return x;
}

// Wpack / cinit / refmu / refiv / refc / part / Ndev: as estep_mfma_kernel (estep_prep_kernel's operands; rows of partial
// statistics (blockIdx.x * 2 + half) * plen)
template <int DJ, int MT>
__global__ void __launch_bounds__(128 * MT) __attribute__((amdgpu_waves_per_eu(2, 2)))
estep_small_kernel(const double *__restrict__ X, int64_t N, int M, const double *__restrict__ Wpack,
                   const double *__restrict__ cinit, double *__restrict__ part, int64_t plen,
                   const double *__restrict__ refmu, const double *__restrict__ refiv, const double *__restrict__ refc,
                   int dj, unsigned long long *__restrict__ mfma_count, const int64_t *__restrict__ Ndev) {
  if (Ndev) {
    N = *Ndev;
    if (N == 0) return;
  }
  using C = EstepSmallCfg<DJ, MT>;
#ifdef VCMI_ESTEP_PROF
  const unsigned long long tk0_ = __builtin_readcyclecounter(), rk0_ = __builtin_amdgcn_s_memrealtime();
#endif
  constexpr int KS = C::KS, NDT = C::NDT, FB = C::FB, RSX = C::RSX, RSG = C::RSG, XBUF = C::XBUF, NW = C::NW, LPF = C::LPF;
  extern __shared__ double smem[];
  double *xbuf = smem;                     // [2][XBUF]: [FB][RSX] images
  double *lg = smem + 2 * XBUF;            // [FB][RSG]   l, then gamma
  double *red = lg;                        // [NW] scratch of the log-likelihood reduction (epilogue only)
  double *etab = lg + FB * RSG;            // [64] 2^(j/64) for vc_exp_tab
  double *tthr = etab + 64;                // [16 MT] refinement thresholds
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lcol = lane & 15, lgrp = lane >> 4;
  if (tid < 64) etab[tid] = kExp2Tab[tid];
  if (tid < 16 * MT) tthr[tid] = (tid < M) ? refc[2 * tid + 1] : -INFINITY;
  const int tile = wave % MT, half = wave / MT;

  double wfrag[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) wfrag[ks] = Wpack[((size_t)tile * KS + ks) * 64 + lane];
  const double cm = cinit[16 * tile + lcol];
  const d4 cin = {cm, cm, cm, cm};
  d4 sacc[NDT];
#pragma unroll
  for (int j = 0; j < NDT; ++j) sacc[j] = d4{0, 0, 0, 0};
  double s0l = 0.0, llacc = 0.0, sprod = 1.0;
  int nprod = 0, nmfma = 0;

  const int64_t nblocks = (N + FB - 1) / FB;
  constexpr int ROWB = RSX * 8, NCHUNK = XBUF / 128;
  auto stage = [&](int64_t f0, double *dst) {          // LDS-DMA of one block, 1 KB per wave-instruction (see estep_mfma_kernel)
    const char *base = reinterpret_cast<const char *>(X + f0 * dj);
    const int last = (int)((N - 1 - f0 < FB - 1) ? N - 1 - f0 : FB - 1);
    int lane_v = lane;
    asm volatile("" : "+v"(lane_v));
#pragma unroll
    for (int i = 0; i < (NCHUNK + NW - 1) / NW; ++i) {
      const int q = wave + NW * i;
      if (q < NCHUNK) {                                          // wave-uniform
        const int o = 1024 * q + 16 * lane_v, row = o / ROWB, col = o - row * ROWB;
        const int rowc = row < last ? row : last;
        const unsigned off = (col < dj * 8) ? (unsigned)(rowc * (dj * 8) + col) : 0u;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(base + off),
                                         (__attribute__((address_space(3))) void *)(reinterpret_cast<char *>(dst) + 1024 * q), 16, 0, 0);
      }
    }
  };
  if (blockIdx.x < nblocks) stage((int64_t)blockIdx.x * FB, xbuf);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int cur = 0;
#ifdef VCMI_ESTEP_PROF
  unsigned long long pt_[6] = {0, 0, 0, 0, 0, 0};     // probe build: cycle counts in A | barrier | softmax | barrier | B | barrier
  const unsigned long long tk1_ = __builtin_readcyclecounter();
#define VCMI_PT(i) { const unsigned long long t_ = __builtin_readcyclecounter(); pt_[i] += t_ - tl_; tl_ = t_; }
#else
#define VCMI_PT(i)
#endif
  for (int64_t blk = blockIdx.x; blk < nblocks; blk += gridDim.x, cur ^= 1) {
#ifdef VCMI_ESTEP_PROF
    unsigned long long tl_ = __builtin_readcyclecounter();
#endif
    const int64_t f0 = blk * FB;
    const double *xs = xbuf + cur * XBUF;
    // ---- step A: l[f][m] = c_m + sum_k Xe[f][k] W[m][k],  Xe = [x^2 | x]: this wave's frame tile and mixture tile.  The two
    //      halves of the contraction run as two accumulator chains (a chain's MFMAs wait for one another) ----
    {
      d4 acc = cin, acc2 = {0, 0, 0, 0};
      nmfma += KS;
      const double *xr = xs + (16 * half + lcol) * RSX + lgrp;
#pragma unroll
      for (int ks = 0; ks < KS / 2; ++ks) {
        const double x = xr[4 * ks];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x * x, wfrag[ks], acc, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, wfrag[KS / 2 + ks], acc2, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) lg[(16 * half + 4 * r + lgrp) * RSG + 16 * tile + lcol] = acc[r] + acc2[r];
    }
    VCMI_PT(0)
    __syncthreads();
    VCMI_PT(1)
    if (blk + gridDim.x < nblocks) stage((blk + gridDim.x) * FB, xbuf + (cur ^ 1) * XBUF);
    // ---- softmax over the 16 MT slots of each frame: LPF lanes per frame, lane l owns the slots l + LPF i.  Straight-line
    //      code but for the (rare) exact re-evaluation: the wave is alone on its SIMD with its workgroup in this phase, every
    //      dependent LDS round trip is exposed ----
    {
      const int l = lane & (LPF - 1), f = (64 / LPF) * wave + lane / LPF;
      double *row = lg + f * RSG + l;
      double v[4];
      double u = -INFINITY;
      bool below[4];                           // value under its mixture's refinement threshold (tthr; -inf: never)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        v[i] = row[LPF * i];
        below[i] = v[i] < tthr[l + LPF * i];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) u = fmax(u, v[i]);
      u = rowN_max<LPF>(u);
      {
        // refinement: when several mixtures are within kRefine of the frame's maximum, exactly those whose expanded form is
        // not provably good enough (value below the mixture's threshold: estep_prep_kernel) are re-evaluated term by term
        constexpr double kRefine = 36.0;
        const double thr = u - kRefine;
        int nc = 0;
        bool need = false;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          nc += (v[i] > thr) ? 1 : 0;
          need = need || (v[i] > thr && below[i]);
        }
        nc = rowN_sum<LPF>(nc);
        need = need && nc > 1;
        if (__builtin_amdgcn_ballot_w64(need) != 0) {      // models with ordinary variances never get here
          const double *xf = xs + f * RSX;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            if (need && v[i] > thr && v[i] < tthr[l + LPF * i]) {
              const int m = l + LPF * i;
              const double *mp = refmu + (size_t)dj * m, *ip = refiv + (size_t)dj * m;
              double q = 0.0;
#pragma unroll 2
              for (int d = 0; d < dj; ++d) {
                const double df = xf[d] - mp[d];
                q = fma(df * df, ip[d], q);
              }
              v[i] = refc[2 * m] - 0.5 * q;
            }
          }
          u = -INFINITY;
#pragma unroll
          for (int i = 0; i < 4; ++i) u = fmax(u, v[i]);
          u = rowN_max<LPF>(u);
        }
      }
      double s = 0.0;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        v[i] = vc_exp_tab(v[i] - u, etab);    // -inf (slots beyond M, zero weights) and < -745 give exactly 0
        s += v[i];
      }
      s = rowN_sum<LPF>(s);
      const bool livef = (f0 + f < N);
      const double inv = (livef && s > 0.0) ? 1.0 / s : 0.0;      // frames beyond N, models without any weight: gamma = 0
#pragma unroll
      for (int i = 0; i < 4; ++i) row[LPF * i] = v[i] * inv;
      // log-likelihood: sum of u + log s; the s of a lane's frames (in [1, 32]) are multiplied up, one log per sixteen blocks
      if (l == 0 && livef) {
        llacc += u;
        sprod *= s;
      }
      if (++nprod == 16) {
        llacc += log(sprod);
        sprod = 1.0;
        nprod = 0;
      }
    }
    VCMI_PT(2)
    __syncthreads();
    VCMI_PT(3)
    // ---- step B: S[m][c] += sum_f gamma[f][m] Xe[f][c],  Xe = [x | x^2]; this wave takes every other k-step of 4 frames.
    //      No test for responsibilities that are all zero (estep_mfma_kernel's skip, and the grouping of the frames that makes
    //      it bite): with at most 32 mixtures a frame that one mixture owns went down the hard-assignment path, and without
    //      the branch the next k-step's operands are on their way while this one's products run ----
    {
      double gmv[FB / 8];
#pragma unroll
      for (int kk = 0; kk < FB / 8; ++kk) gmv[kk] = lg[(4 * (2 * kk + half) + lgrp) * RSG + 16 * tile + lcol];
      nmfma += NDT * (FB / 8);
      double xn[NDT / 2];
#pragma unroll
      for (int j = 0; j < NDT / 2; ++j) xn[j] = xs[(4 * half + lgrp) * RSX + lcol + 16 * j];
#pragma unroll
      for (int kk = 0; kk < FB / 8; ++kk) {
        const double gm = gmv[kk];
        double xc[NDT / 2];
#pragma unroll
        for (int j = 0; j < NDT / 2; ++j) xc[j] = xn[j];
        if (kk + 1 < FB / 8) {                 // the next k-step's operands: in flight under this one's products
#pragma unroll
          for (int j = 0; j < NDT / 2; ++j) xn[j] = xs[(4 * (2 * kk + 2 + half) + lgrp) * RSX + lcol + 16 * j];
        }
#pragma unroll
        for (int j = 0; j < NDT / 2; ++j) {
          sacc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(gm, xc[j], sacc[j], 0, 0, 0);
          sacc[NDT / 2 + j] = __builtin_amdgcn_mfma_f64_16x16x4f64(gm, xc[j] * xc[j], sacc[NDT / 2 + j], 0, 0, 0);
        }
        s0l += gm;
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    VCMI_PT(4)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the next block's LDS-DMA has landed
    __syncthreads();
    VCMI_PT(5)
  }
#ifdef VCMI_ESTEP_PROF
  const unsigned long long tk2_ = __builtin_readcyclecounter();
  if (blockIdx.x == 3 && lane == 0)
    printf("estep small prof wave %d: A %llu | bar %llu | softmax %llu | bar %llu | B %llu | bar %llu\n", wave, pt_[0], pt_[1], pt_[2], pt_[3], pt_[4], pt_[5]);
#endif
#undef VCMI_PT

  if (mfma_count && lane == 0) atomicAdd(mfma_count, (unsigned long long)nmfma);
  // ---- this wave's partial statistics: row (workgroup, half); rows m = 16 tile + lgrp + 4 r, cols = 16 j + lcol ----
  double *P = part + ((size_t)blockIdx.x * 2 + half) * plen;
  s0l += __shfl_xor(s0l, 16);
  s0l += __shfl_xor(s0l, 32);
  if (lgrp == 0 && 16 * tile + lcol < M) P[16 * tile + lcol] = s0l;
  if (lane == 0 && tile == 0 && half > 0) P[plen - 1] = 0.0;          // the log-likelihood travels in row half = 0
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int m = 16 * tile + 4 * r + lgrp;
    if (m < M) {
#pragma unroll
      for (int j = 0; j < NDT; ++j) {
        const int c = 16 * j + lcol;   // column of [x | x^2]
        if (c < DJ) {
          if (c < dj) P[M + (size_t)m * dj + c] = sacc[j][r];
        } else if (c - DJ < dj) {
          P[M + (size_t)M * dj + (size_t)m * dj + (c - DJ)] = sacc[j][r];
        }
      }
    }
  }
  // log-likelihood: fixed-order reduction inside the workgroup (butterflies within a wave, the waves in order; thread 0: tile 0, half 0)
  llacc += log(sprod);
#pragma unroll
  for (int sh = 1; sh < 64; sh <<= 1) llacc += __shfl_xor(llacc, sh);
  if (lane == 0) red[wave] = llacc;
  __syncthreads();
  if (tid == 0) {
    double ll = 0.0;
    for (int i = 0; i < NW; ++i) ll += red[i];
    P[plen - 1] = ll;
  }
#ifdef VCMI_ESTEP_PROF
  if ((blockIdx.x == 3 || blockIdx.x == 259 || blockIdx.x == 511 || blockIdx.x == 128) && lane == 0 && (wave == 0 || wave == 3)) {
    const unsigned long long tk3_ = __builtin_readcyclecounter(), rk3_ = __builtin_amdgcn_s_memrealtime();
    printf("estep small life wg %d wave %d: prologue %llu loop %llu epilogue %llu cycles; realtime start %llu end %llu (100 MHz)\n", (int)blockIdx.x, wave,
           tk1_ - tk0_, tk2_ - tk1_, tk3_ - tk2_, rk0_ % 100000000ull, rk3_ % 100000000ull);
  }
#endif
}

}  // namespace vcmi
