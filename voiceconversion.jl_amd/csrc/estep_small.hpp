// estep_small.hpp -- the diagonal E-step for SMALL models: M <= 32 mixtures, i.e. one or two mixture tiles of 16 (included by
// estep.hip; Dj <= 80).  The reference's own sizes are here: bin/train_gmm.jl:84-89 trains 32 mixtures on 80 joint dimensions
// (test/models/clb_and_slt_gmm32_order40.jld).
//
// estep_mfma_kernel<DJ, 0, SHARE = true> runs such a model with four (M <= 32) or eight (M <= 16) of its eight waves on one
// tile, which balances the MFMAs -- but everything around them is laid out for 128 slots: the softmax walks eight slot groups
// per lane with sixteen lanes on a frame (two slots of the eight exist), takes two passes per wave and block, and the one
// workgroup of a CU moves through step A | barrier | softmax | barrier | step B | barrier in lock step (phase timers at M = 32:
// 39 % of the workgroup's time in the softmax, 26 % / 34 % in the steps that hold all the MFMAs; 0.37 of the FP64 MFMA peak).
//
// Here the workgroup is as small as the model: 2 MT waves (MT = mixture tiles), 32 frames per block, two waves to a tile
// (wave = (tile, half): step A on frame tile `half`, every other k-step of step B); a CU holds two of them, which run at their
// own pace.  What the measurements of round 6 say about this kernel (profiles/r06_ab/estep_small_kernel.txt):
//   * The FP64 vector instructions and the FP64 MFMAs share ONE pipe, and so does everything else a wave issues to the VALU:
//     over all variants the time of a SIMD is  64 cycles x MFMAs + ~8 cycles x other VALU instructions, whatever the order
//     and whichever of the two resident waves issues them.  Overlapping "the softmax of one workgroup with the MFMAs of the
//     other" buys the latencies (LDS round trips, dependent chains), not the instructions.  So the kernel counts VALU
//     instructions: one softmax pass per wave and block with four values per lane and 4 MT lanes on a frame, loops as long
//     as the model is wide; the refinement test folded into the exps' own operands, its re-evaluation out of line; 1/s by
//     v_rcp_f64 and two Newton steps; lane permutations without the copy of the old value; the LDS-DMA's source offsets
//     from a table in LDS instead of a dozen integer instructions per chunk (470 -> 220 per wave and block).
//   * A wave is alone on its SIMD with its workgroup.  Step A of the NEXT block (MFMAs, operands from LDS) is interleaved
//     piece by piece with the softmax of the current one, so the wave has an MFMA to issue while an exp's chain or an LDS
//     read is under way; three x buffers make the next block's frames available one iteration early.
//   * The older of a SIMD's two waves wins every arbitration: left alone, the first workgroup of a CU finished its blocks in
//     0.82 M cycles and the second in 1.16 M.  s_setprio by phase (high in the softmax + step A phase, low in step B's MFMA
//     stream) evens them out (1.02 M / 1.06 M).
// estep_fixture (M = 32, Dj = 80, 1.25e6 frames): 0.815 ms in estep_mfma_kernel -> 0.51 ms here (step: 0.89 -> 0.61 ms).
// M <= 16 (the size bin/train_gmm.jl defaults to): one mixture tile, blocks of ONE frame tile, the two waves of a workgroup split
// step A's contraction by k-steps (two planes of partial sums in LDS) -- 39.7 KB of LDS per workgroup with the x images packed
// and 16-bit DMA offsets, four workgroups = eight waves per CU: 0.63 ms (estep_mfma_kernel) -> 0.39-0.41 ms per 1.25e6 frames.
// The arithmetic of a frame is that of estep_mfma_kernel (same expanded form, same refinement rule, same table exp); the
// partial statistics have the same row layout (row = workgroup x half) and go through estep_reduce_kernel.
#pragma once

#ifndef VCMI_SMALL_PRIO
#define VCMI_SMALL_PRIO 1
#endif
#ifndef VCMI_SMALL_B_BATCH
#define VCMI_SMALL_B_BATCH 1
#endif

namespace vcmi {

template <int DJ, int MT>
struct EstepSmallCfg {
  static_assert(MT == 1 || MT == 2, "one or two mixture tiles");
  static constexpr int KS = 2 * DJ / 4, NDT = 2 * DJ / 16;
  static constexpr int NW = 2 * MT;                              // waves per workgroup
  // frames per block.  MT = 2: two frame tiles of 16, a wave = (mixture tile, frame tile) in step A.  MT = 1: ONE frame tile,
  // and the two waves split the contraction of step A by k-steps (partial sums in two planes of LDS, added by the softmax) --
  // half the LDS per workgroup, so that four of them fit a CU and every SIMD has its two waves (with 32-frame blocks a CU
  // held two workgroups of two waves: one wave per SIMD, 0.55 ms per 1.25e6 frames at M = 16)
  static constexpr int FB = MT == 1 ? 16 : 32;
  static constexpr bool SPLITK = MT == 1;
  static constexpr int PLANES = SPLITK ? 2 : 1;
  static constexpr int RSX = (DJ + 2 + 13) / 32 * 32 + 18;       // as EstepCfg: 16-byte rows, conflict-free column reads
  static constexpr int NCHUNK = (FB * RSX * 8 + 1023) / 1024;    // 1 KB wave-instructions of the LDS-DMA per x image
  // doubles between two x images.  MT = 1: exactly the image -- the last wave-instruction of an image is cut off behind it
  // (kLastLanes) -- because there every byte counts: four workgroups per CU have 40 KB each
  static constexpr int XBUF = SPLITK ? FB * RSX : NCHUNK * 128;
  static constexpr int kLastLanes = SPLITK ? (FB * RSX * 8 - 1024 * (NCHUNK - 1)) / 16 : 64;
  static constexpr int NV = 16 * MT * FB / (64 * NW);            // softmax values per lane: 4 (MT = 2) / 2 (MT = 1)
  static constexpr int LPF = 16 * MT / NV;                       // softmax lanes per frame
  static_assert(LPF == 8 && 64 / LPF * NW == FB, "one softmax pass per wave and block");
  // LDS row stride of l / gamma (doubles): the softmax's half-wave (four frames x eight consecutive doubles) lands on 64
  // distinct banks -- MT = 2: rows 80 dwords apart -> 0, 16, 32, 48 mod 64; MT = 1: 48 dwords -> 0, 48, 32, 16
  static constexpr int RSG = MT == 2 ? 40 : 24;
  static constexpr int WG_PER_CU = MT == 2 ? 2 : 4;
  // [x 3][l / gamma PLANES x FB x RSG][etab 64][thresholds 16 MT] doubles + [DMA source offsets: NCHUNK x 64 lanes] shorts
  static constexpr size_t LDS_BYTES = ((size_t)3 * XBUF + (size_t)PLANES * FB * RSG + 64 + 16 * MT) * sizeof(double) + (size_t)NCHUNK * 64 * sizeof(unsigned short);
  static_assert(FB * 80 * 8 < 65536 && (XBUF * 8) % 16 == 0, "offsets fit 16 bits; images are 16-byte aligned");
  static_assert(LDS_BYTES * WG_PER_CU <= 160 * 1024, "LDS");
};

// a lane permutation within DPP rows (quad_perm / row_half_mirror: every lane has a source) without the copy of the destination's
// old value that update_dpp's tied operand costs -- two VALU instructions per double and step
template <int CTRL>
__device__ __forceinline__ double dpp_perm_f64(double x) {
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(x), CTRL, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(x), CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}

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
  constexpr int KS = C::KS, NDT = C::NDT, FB = C::FB, RSX = C::RSX, RSG = C::RSG, XBUF = C::XBUF, NW = C::NW, LPF = C::LPF, NV = C::NV;
  constexpr bool kSplitK = C::SPLITK;
  constexpr int NS = KS / 2;                            // k-steps of x^2 (and of x) in a step A
  constexpr int NSL = kSplitK ? NS / 2 : NS;            // ... of which a wave takes these (MFMA slots: one k-step of either chain each)
  static_assert(!kSplitK || NS % 2 == 0, "k-steps");
  extern __shared__ double smem[];
  double *xbuf = smem;                     // [3][XBUF]: [FB][RSX] images
  double *lg = smem + 3 * XBUF;            // [FB][RSG]   l, then gamma
  double *red = lg;                        // [NW] scratch of the log-likelihood reduction (epilogue only)
  double *etab = lg + C::PLANES * FB * RSG;  // [64] 2^(j/64) for vc_exp_tab
  double *tthr = etab + 64;                // [16 MT] refinement thresholds
  unsigned short *otab = reinterpret_cast<unsigned short *>(tthr + 16 * MT);   // [NCHUNK][64] source offsets of the LDS-DMA (full blocks)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // in a scalar register: per-wave addresses and branches are uniform
  const int lcol = lane & 15, lgrp = lane >> 4;
  if (tid < 64) etab[tid] = kExp2Tab[tid];
  if (tid < 16 * MT) tthr[tid] = (tid < M) ? refc[2 * tid + 1] : -INFINITY;
  const int tile = wave % MT, half = wave / MT;

  // this wave's weight fragments: slots [0, NSL) multiply x^2, [NSL, 2 NSL) multiply x (kSplitK: the k-steps half * NSL ... of either)
  const int kofs = kSplitK ? half * NSL : 0;
  double wfrag[2 * NSL];
#pragma unroll
  for (int k = 0; k < NSL; ++k) {
    wfrag[k] = Wpack[((size_t)tile * KS + kofs + k) * 64 + lane];
    wfrag[NSL + k] = Wpack[((size_t)tile * KS + NS + kofs + k) * 64 + lane];
  }
  const double cm = (kSplitK && half) ? 0.0 : cinit[16 * tile + lcol];        // (kSplitK: the constant travels with the first partial sum)
  const d4 cin = {cm, cm, cm, cm};
  d4 sacc[NDT];
#pragma unroll
  for (int j = 0; j < NDT; ++j) sacc[j] = d4{0, 0, 0, 0};
  double s0l = 0.0, llacc = 0.0, sprod = 1.0;
  int nprod = 0, nmfma = 0;

  const int64_t nblocks = (N + FB - 1) / FB;
  constexpr int ROWB = RSX * 8, NCHUNK = C::NCHUNK;
  // LDS-DMA of one block, 1 KB per wave-instruction (see estep_mfma_kernel).  The per-lane source offset of every chunk is the
  // same for all full blocks: it comes from a table in LDS (one ds_read per chunk instead of a dozen integer instructions --
  // every VALU instruction of this kernel is paid for in matrix-pipe time); only the call's last, partial block computes it.
  auto chunk_off = [&](int q, int lane_, int last) -> unsigned {
    const int o = 1024 * q + 16 * lane_, row = o / ROWB, col = o - row * ROWB;
    const int rowc = row < last ? row : last;                    // (also the rows >= FB of the padding)
    return (col < dj * 8) ? (unsigned)(rowc * (dj * 8) + col) : 0u;
  };
  for (int e = tid; e < NCHUNK * 64; e += 64 * NW) otab[e] = (unsigned short)chunk_off(e >> 6, e & 63, FB - 1);
  auto stage = [&](int64_t f0, double *dst) {
    const char *base = reinterpret_cast<const char *>(X + f0 * dj);
    auto dma = [&](int q, unsigned off) {
      // (kLastLanes < 64: the images are packed, the lanes of the last chunk behind the image must not write -- what lies there
      // is the next image, or l / gamma)
      if (C::kLastLanes == 64 || q != NCHUNK - 1 || lane < C::kLastLanes)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(base + off),
                                         (__attribute__((address_space(3))) void *)(reinterpret_cast<char *>(dst) + 1024 * q), 16, 0, 0);
    };
    if (N - f0 >= FB) {                                          // workgroup-uniform
      // (the table address is formed here, per block, from the opaque lane id: hoisted out of the loop the six addresses are
      // spilled, and a scratch reload's s_waitcnt vmcnt(0) also waits for the DMA issued just before it -- 950 cycles a chunk)
      int lane_t = lane;
      asm volatile("" : "+v"(lane_t));
      const unsigned short *ot = otab + lane_t;
      unsigned off[(NCHUNK + NW - 1) / NW];
#pragma unroll
      for (int i = 0; i < (NCHUNK + NW - 1) / NW; ++i) {
        const int q = wave + NW * i;
        off[i] = ot[64 * (q < NCHUNK ? q : 0)];
      }
#pragma unroll
      for (int i = 0; i < (NCHUNK + NW - 1) / NW; ++i) {
        const int q = wave + NW * i;
        if (q < NCHUNK) dma(q, off[i]);                          // wave-uniform
      }
    } else {
      int lane_v = lane;
      asm volatile("" : "+v"(lane_v));                           // (nothing of this branch is to be computed ahead of the loop)
#pragma unroll 1
      for (int q = wave; q < NCHUNK; q += NW) dma(q, chunk_off(q, lane_v, (int)(N - 1 - f0)));
    }
  };
  // ---- the pipeline.  Block i of this workgroup lives in x buffer i % 3.  Iteration i:
  //        [DMA of block i+2 issued]
  //        S1: softmax(i) on l_i in LDS  INTERLEAVED, instruction by instruction, with step A(i+1) (-> registers)
  //        barrier;  S2: step B(i);  barrier;  S3: l_{i+1} -> LDS, DMA landed;  barrier
  //      A wave is alone on its SIMD with its workgroup: whatever its softmax waits for (LDS round trips, the dependent FP64
  //      chains of the exps) is time the matrix pipe idles unless the same wave has MFMAs to issue in between.
  const int64_t blk0 = blockIdx.x, bstride = gridDim.x;
  __syncthreads();                                               // (the offset table)
  if (blk0 < nblocks) stage(blk0 * FB, xbuf);
  if (blk0 + bstride < nblocks) stage((blk0 + bstride) * FB, xbuf + XBUF);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  // step A: this lane's operand row / first k offset in an x image, its rows of l in LDS (kSplitK: the wave's plane)
  const int arow = ((kSplitK ? 0 : 16 * half) + lcol) * RSX + lgrp + 4 * kofs;
  double *const lgA = lg + (kSplitK ? half * FB * RSG : 16 * half * RSG) + lgrp * RSG + 16 * tile + lcol;
  if (blk0 < nblocks) {                                        // step A(0), on its own
    d4 acc = cin, acc2 = {0, 0, 0, 0};
    nmfma += 2 * NSL;
    const double *xr = xbuf + arow;
#pragma unroll
    for (int k = 0; k < NSL; ++k) {
      const double x = xr[4 * k];
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x * x, wfrag[k], acc, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, wfrag[NSL + k], acc2, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) lgA[4 * r * RSG] = acc[r] + acc2[r];
  }
  __syncthreads();
  int cur = 0;
#ifdef VCMI_ESTEP_PROF
  int nslow_ = 0;
  unsigned long long pt_[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};     // probe build: cycle counts in S1 | barrier | B | barrier | S3 | barrier
  const unsigned long long tk1_ = __builtin_readcyclecounter();
#define VCMI_PT(i) { const unsigned long long t_ = __builtin_readcyclecounter(); pt_[i] += t_ - tl_; tl_ = t_; }
#else
#define VCMI_PT(i)
#endif
  for (int64_t blk = blk0; blk < nblocks; blk += bstride) {
#ifdef VCMI_ESTEP_PROF
    unsigned long long tl_ = __builtin_readcyclecounter();
#endif
    const int64_t f0 = blk * FB;
    const int nxt = cur == 2 ? 0 : cur + 1, nn = nxt == 2 ? 0 : nxt + 1;
    const double *xs = xbuf + cur * XBUF;
    const bool has_next = blk + bstride < nblocks;
    if (blk + 2 * bstride < nblocks) stage((blk + 2 * bstride) * FB, xbuf + nn * XBUF);
    VCMI_PT(6)                                 // (probe: the DMA issue)
    // ---- S1 ----
#if VCMI_SMALL_PRIO
    __builtin_amdgcn_s_setprio(2);             // the phase with the dependent chains goes first where both workgroups of the CU want to issue
#endif
    d4 acc = cin, acc2 = {0, 0, 0, 0};        // step A(i+1): l[f][m] = c_m + sum_k Xe[f][k] W[m][k], Xe = [x^2 | x], two chains
    {
      // (without a next block the products run on whatever the buffer holds and are dropped)
      const double *xr = xbuf + nxt * XBUF + arow;
      double xq[3];
      xq[0] = xr[0];
      xq[1] = xr[4];
      auto mfma_slot = [&](int k) {           // k-step k of both chains; the operand of slot k + 2 is requested
        if (k + 2 < NSL) xq[(k + 2) % 3] = xr[4 * (k + 2)];
        const double x = xq[k % 3];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x * x, wfrag[k], acc, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, wfrag[NSL + k], acc2, 0, 0, 0);
      };
      // softmax over the 16 MT slots of each frame: LPF lanes per frame, lane l owns the slots l + LPF i.  e^(v - u) by
      // vc_exp_tab's steps (fp64_exp.hpp), the four values of a lane side by side, one piece per MFMA slot.
      const int l = lane & (LPF - 1), f = (64 / LPF) * wave + lane / LPF;
      double *row = lg + f * RSG + l;
      constexpr double kRefine = 36.0;        // e^-36 = 2e-16: a mixture further below the maximum cannot change a sum
      double v[NV], tt[NV], t[NV], kf[NV], r[NV], pl[NV], u = -INFINITY, s = 0.0, inv = 0.0;
      int ki[NV];
      bool below[NV], needl = false;
      const bool livef = (f0 + f < N);
      auto piece = [&](int p) {
        if (p == 0) {
#pragma unroll
          for (int i = 0; i < NV; ++i) {
            v[i] = row[LPF * i];
            if constexpr (kSplitK) t[i] = row[FB * RSG + LPF * i];      // the other wave's partial sum
            tt[i] = tthr[l + LPF * i];
          }
        } else if (p == 1) {
          if constexpr (kSplitK) {
#pragma unroll
            for (int i = 0; i < NV; ++i) v[i] += t[i];
          }
          u = v[0];
#pragma unroll
          for (int i = 1; i < NV; ++i) u = fmax(u, v[i]);
#pragma unroll
          for (int i = 0; i < NV; ++i) below[i] = v[i] < tt[i];       // under the mixture's refinement threshold (-inf: never)
        } else if (p == 2) {
          u = fmax(u, dpp_perm_f64<0xB1>(u));
        } else if (p == 3) {
          u = fmax(u, dpp_perm_f64<0x4E>(u));
        } else if (p == 4) {
          u = fmax(u, dpp_perm_f64<0x141>(u));
        } else if (p == 5) {
#pragma unroll
          for (int i = 0; i < NV; ++i) {
            t[i] = fmax(v[i] - u, -1000.0);       // (also maps -inf: slots beyond M, zero weights)
            needl = needl || (below[i] && t[i] > -kRefine);
          }
        } else if (p == 6) {
#pragma unroll
          for (int i = 0; i < NV; ++i) kf[i] = rint(t[i] * 92.332482616893656877);
        } else if (p == 7) {
#pragma unroll
          for (int i = 0; i < NV; ++i) r[i] = fma(kf[i], -1.083042469326756e-02, t[i]);
        } else if (p == 8) {
#pragma unroll
          for (int i = 0; i < NV; ++i) {
            r[i] = fma(kf[i], -2.9815858269852933e-12, r[i]);
            ki[i] = (int)kf[i];
            t[i] = etab[ki[i] & 63];
          }
        } else if (p == 9) {
#pragma unroll
          for (int i = 0; i < NV; ++i) pl[i] = vc_fma_sconst(8.333333333333333e-03, r[i], 4.1666666666666664e-02);
        } else if (p == 10) {
#pragma unroll
          for (int i = 0; i < NV; ++i) pl[i] = vc_fma_sconst(pl[i], r[i], 1.6666666666666666e-01);
        } else if (p == 11) {
#pragma unroll
          for (int i = 0; i < NV; ++i) pl[i] = fma(pl[i], r[i], 0.5);
        } else if (p == 12) {
#pragma unroll
          for (int i = 0; i < NV; ++i) pl[i] = fma(pl[i], r[i], 1.0);
        } else if (p == 13) {
#pragma unroll
          for (int i = 0; i < NV; ++i) pl[i] = fma(pl[i], r[i], 1.0);
        } else if (p == 14) {
#pragma unroll
          for (int i = 0; i < NV; ++i) v[i] = ldexp(t[i] * pl[i], ki[i] >> 6);      // exactly 0 below -745
        } else if (p == 15) {
          s = v[0];
#pragma unroll
          for (int i = 1; i < NV; ++i) s += v[i];
          s += dpp_perm_f64<0xB1>(s);
        } else if (p == 16) {
          s += dpp_perm_f64<0x4E>(s);
          s += dpp_perm_f64<0x141>(s);
        } else {
          // 1 / s, s in [1, 32]: the hardware's estimate and two Newton steps (the division's scaling and fix-up cases cannot occur)
          const double sc = s > 0.0 ? s : 1.0;
          double q = __builtin_amdgcn_rcp(sc);
          q = fma(fma(-sc, q, 1.0), q, q);
          q = fma(fma(-sc, q, 1.0), q, q);
          inv = (livef && s > 0.0) ? q : 0.0;              // frames beyond N, models without any weight: gamma = 0
        }
      };
      constexpr int NP = 18;
#pragma unroll
      for (int sl = 0; sl < NSL; ++sl) {
#pragma unroll
        for (int p = sl * NP / NSL; p < (sl + 1) * NP / NSL; ++p) piece(p);
        mfma_slot(sl);
        __builtin_amdgcn_sched_barrier(0);
      }
      VCMI_PT(7)                               // (probe: the slots)
      // Refinement (rare: models with ordinary variances never get here).  Several mixtures share the frame (s > 1: another one
      // within ~36 nats of the maximum) and one of those within kRefine has an expanded-form value that is not provably good
      // enough (below its mixture's threshold, estep_prep_kernel): exactly those are re-evaluated term by term, (x - mu)^2 / var
      // summed over d as the reference formula reads, and the frame's softmax is taken again.
      if (__builtin_amdgcn_ballot_w64(needl && s > 1.0) != 0) {
#ifdef VCMI_ESTEP_PROF
        ++nslow_;
#endif
        const double *xf = xs + f * RSX;
        double vv[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) vv[i] = row[LPF * i] + (kSplitK ? row[FB * RSG + LPF * i] : 0.0);
        auto vmax = [&]() {
          double m = vv[0];
#pragma unroll
          for (int i = 1; i < NV; ++i) m = fmax(m, vv[i]);
          return rowN_max<LPF>(m);
        };
        double uu = vmax();
        const double thr = uu - kRefine;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
          const int m = l + LPF * i;
          if (needl && s > 1.0 && vv[i] > thr && vv[i] < tthr[m]) {
            const double *mp = refmu + (size_t)dj * m, *ip = refiv + (size_t)dj * m;
            double q = 0.0;
#pragma unroll 2
            for (int d = 0; d < dj; ++d) {
              const double df = xf[d] - mp[d];
              q = fma(df * df, ip[d], q);
            }
            vv[i] = refc[2 * m] - 0.5 * q;
          }
        }
        uu = vmax();
        double ss = 0.0;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
          vv[i] = vc_exp_tab(vv[i] - uu, etab);
          ss += vv[i];
        }
        ss = rowN_sum<LPF>(ss);
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] = vv[i];
        u = uu;
        s = ss;
        inv = (livef && ss > 0.0) ? 1.0 / ss : 0.0;
      }
#pragma unroll
      for (int i = 0; i < NV; ++i) row[LPF * i] = v[i] * inv;
      if (has_next) nmfma += 2 * NSL;
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
#if VCMI_SMALL_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
    VCMI_PT(0)
    __syncthreads();
    VCMI_PT(1)
    // ---- S2, step B: S[m][c] += sum_f gamma[f][m] Xe[f][c],  Xe = [x | x^2]; this wave takes every other k-step of 4 frames.
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
        double x2[NDT / 2];
#pragma unroll
        for (int j = 0; j < NDT / 2; ++j) x2[j] = xc[j] * xc[j];
#pragma unroll
        for (int j = 0; j < NDT / 2; ++j) {
          sacc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(gm, xc[j], sacc[j], 0, 0, 0);
          sacc[NDT / 2 + j] = __builtin_amdgcn_mfma_f64_16x16x4f64(gm, x2[j], sacc[NDT / 2 + j], 0, 0, 0);
        }
        s0l += gm;
#if VCMI_SMALL_B_BATCH
        __builtin_amdgcn_sched_group_barrier(0x100, NDT / 2, 0);      // the next k-step's reads
        __builtin_amdgcn_sched_group_barrier(0x002, NDT / 2 + 1, 0);  // the squares (and s0l), back to back
        __builtin_amdgcn_sched_group_barrier(0x008, NDT, 0);          // the products, back to back
#endif
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    VCMI_PT(2)
    __syncthreads();
    VCMI_PT(3)
    // ---- S3: l of the next block into LDS (its responsibilities-to-be), the DMA of the block after it has landed ----
    if (has_next) {
#pragma unroll
      for (int r = 0; r < 4; ++r) lgA[4 * r * RSG] = acc[r] + acc2[r];
    }
    VCMI_PT(4)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    VCMI_PT(5)
    cur = nxt;
  }
#ifdef VCMI_ESTEP_PROF
  const unsigned long long tk2_ = __builtin_readcyclecounter();
  if (blockIdx.x == 3 && lane == 0)
    printf("estep small prof wave %d: S1 %llu | bar %llu | B %llu | bar %llu | S3 %llu | bar %llu; refinement branch taken %d times; stage %llu slots %llu\n", wave, pt_[0], pt_[1], pt_[2], pt_[3], pt_[4], pt_[5], nslow_, pt_[6], pt_[7]);
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
