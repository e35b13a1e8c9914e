// estep_hard.hpp -- the diagonal E-step for frames that ONE mixture owns (included by estep.hip; M <= 128, Dj <= 80, even).
//
// On peaked data -- the BASELINE frames are: mixtures tens of thousands of nats apart -- every frame has exactly one
// responsibility that is not EXACTLY zero (exp underflows below -745), and it is exactly 1.  estep_mfma_kernel still forms all
// 128 log-densities of every frame in FP64 (step A: 53-60 % of its time) to find that out.  Here the finding-out is a
// certified screen on the BF16 matrix pipe, and the frames it settles never see an FP64 MFMA:
//   1. estep_hard_key_kernel: l^_m = c_m + W_m . [x^2 ; x] for all mixtures from bf16-split operands (hi + lo of W and of
//      [x^2 ; x]: three v_mfma_f32_16x16x32_bf16 per 32 of the 2 Dj contraction steps, FP32 accumulation), with the error margin
//      eps_m = 2^-12 (|W_m| |[x^2 ; x]| + |c_m|) (the bound of gmmmap_screen.hpp, ~2 x what the arithmetic needs).  The frame is
//      HARD when every mixture but the best one satisfies  l^_m + eps_m < (l^_best - eps_best) - 746: then exp(l_m - max) is
//      exactly 0 in FP64 for all of them -- also as estep_mfma_kernel evaluates it -- and the best one's responsibility is
//      exactly 1.  key = that mixture; every other frame gets key = M ("soft");
//   2. the stable counting sort of gmmmap.hip (grouping.hpp) by key;
//   3. estep_hard_stats_kernel: per mixture and 256-frame piece of its hard frames: count, sum x, sum x^2 (weights are exactly 1)
//      and sum_d (x_d - mu_d)^2 / var_d for the log-likelihood (the winner's log-density term by term, as the reference formula
//      reads) -- one coalesced pass over the rows through perm; estep_hard_reduce_kernel adds a mixture's pieces in order;
//   4. the soft frames (none on the BASELINE data; all of them where mixtures overlap) are gathered and go through
//      estep_mfma_kernel as before (its frame count read from device memory: nothing in the call waits for the GPU), and its
//      partial statistics are added on top.
// The statistics equal the one-kernel path's to rounding (other summation order; deterministic: the sort is stable, every
// sum has a fixed order).  Where the screen settles little -- overlapping mixtures: a frame is soft as soon as a second
// mixture is within 746 nats -- the detour costs its two passes over X; the caller (estep_mfma_launch) watches the soft
// fraction of the previous call and stays on the one-kernel path while it is high.
#pragma once
#include "bf16_split.hpp"
#include "grouping.hpp"

namespace vcmi {

constexpr int kHardPiece = 256;        // hard frames per workgroup of the statistics kernel
constexpr int kHardMaxM = 128;         // mixtures (the one-kernel path's limit; more go through the groups of estep_mfma_groups_launch)
// W in bf16 for the key kernel: per mixture tile mt (16 mixtures) and instruction i (x dimensions 16 i .. 16 i + 15):
// [hi 64 x 16 B | lo 64 x 16 B]; slot j of lane group g <-> j < 4: -1/(2 var) of dimension 16 i + 4 g + j (multiplies x^2),
// j >= 4: mu / var of dimension 16 i + 4 g + (j - 4) (multiplies x).  Then per tile 48 floats: c (16), 2^-12 |W_m| (16, rounded
// up), 2^-12 |c_m| (16, rounded up); a mixture without weight or beyond M: c = -1e30, margins 0 (its l^ is certified hopeless).
template <int DJ>
struct EstepHardCfg {
  static constexpr int NI = (DJ + 15) / 16;                  // K = 32 instructions per term
  static constexpr int TILE_BYTES = NI * 2048 + 192;         // operands of one mixture tile
  static constexpr size_t lds_bytes(int mt) { return (size_t)mt * TILE_BYTES; }
};

template <int DJ>
__global__ void __launch_bounds__(256)
estep_hard_prep_kernel(const double *__restrict__ raw, const double *__restrict__ cinit, int M, int dj, unsigned char *__restrict__ W16) {
  using C = EstepHardCfg<DJ>;
  const double *mu = raw + M, *var = mu + (size_t)dj * M;
  const int MT = (M + 15) / 16;
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e < MT * C::NI * 64) {
    const int l = e & 63, i = (e >> 6) % C::NI, mt = (e >> 6) / C::NI;
    const int m = 16 * mt + (l & 15), g = l >> 4;
    unsigned short *hi = reinterpret_cast<unsigned short *>(W16 + (size_t)mt * C::TILE_BYTES + (size_t)i * 2048) + l * 8, *lo = hi + 512;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int d = 16 * i + 4 * g + (j & 3);
      double v = 0.0;
      if (m < M && d < dj) {
        const double ivv = 1.0 / var[d + (size_t)dj * m];
        v = j < 4 ? -0.5 * ivv : mu[d + (size_t)dj * m] * ivv;
      }
      split_bf16(v, hi[j], lo[j]);
    }
  }
  // per mixture: c and the two margins -- one wave each (lanes over the dimensions, summed in a fixed order), in the blocks
  // behind the operands' (the launch adds 4 MT of them: 4 waves per block, 16 MT mixture slots)
  const int nb0 = (MT * C::NI * 64 + 255) / 256;
  if ((int)blockIdx.x >= nb0) {
    const int m = ((int)blockIdx.x - nb0) * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (m >= 16 * MT) return;
    const int mt = m >> 4, r = m & 15;
    float *cf = reinterpret_cast<float *>(W16 + (size_t)mt * C::TILE_BYTES + (size_t)C::NI * 2048);
    const double c = m < M ? cinit[m] : -INFINITY;
    const bool dead = !(c > -INFINITY);
    double q = 0.0;
    if (!dead)
      for (int d = lane; d < dj; d += 64) {
        const double ivv = 1.0 / var[d + (size_t)dj * m], a = 0.5 * ivv, b = mu[d + (size_t)dj * m] * ivv;
        q = fma(a, a, fma(b, b, q));
      }
#pragma unroll
    for (int sh = 1; sh < 64; sh <<= 1) q += __shfl_xor(q, sh);
    if (lane == 0) {
      auto up = [](double v) { return __uint_as_float(__float_as_uint((float)(v * (1.0 + 0x1p-20))) + 1u); };      // next float up (v >= 0)
      cf[r] = dead ? -1e30f : (float)c;
      cf[16 + r] = dead ? 0.0f : up(sqrt(q) * 0x1p-12);
      cf[32 + r] = dead ? 0.0f : up(fabs(c) * 0x1p-12);
    }
  }
}

// keys + one histogram per chunk of kGroupChunk frames (keys 0 .. M: M = soft), as gmmmap_group_key_kernel
constexpr int kHardKeyThreads = 512;      // one workgroup per CU (the operands of all mixtures fill half its LDS): eight waves of two frame tiles each (256 registers: sixteen waves of two tiles spill at Dj = 80)
template <int DJ>
__global__ void __launch_bounds__(kHardKeyThreads)
estep_hard_key_kernel(const unsigned char *__restrict__ W16, int M, int dj, const double *__restrict__ X, int64_t N,
                      int *__restrict__ key, int *__restrict__ chunkhist, int64_t nrun, int64_t cstride, const int64_t *__restrict__ gate,
                      int split) {
  if (gate && *gate == 0) return;                             // (estep_path.hpp: the call takes the other path)
  // nrun chunks are looked at, chunk c of the run = chunk c * cstride of the frames: all of them (cstride = 1), or a SAMPLE
  // spread over the frames (key = nullptr: only the histograms are wanted -- estep_path_decide_kernel).  split = 1: a run unit is
  // ONE pass (a quarter of a chunk: unit u = pass u % 4 of chunk u / 4) -- the sample then spreads over four times as many CUs
  using C = EstepHardCfg<DJ>;
  constexpr int NI = C::NI;
  extern __shared__ double hsm[];
  const int MT = (M + 15) / 16, MK = M + 1;
  const int nd = (int)(C::lds_bytes(MT) / 8);
  int *hist = reinterpret_cast<int *>(hsm + nd);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lcol = lane & 15, lgrp = lane >> 4;
  for (int e = tid; e < nd; e += kHardKeyThreads) hsm[e] = reinterpret_cast<const double *>(W16)[e];
  __syncthreads();
  // the largest margins of the model: NWmax = max 2^-12 |W_m|, NCmax = max 2^-12 |c_m| (every wave for itself)
  float nwmax = 0.0f, ncmax = 0.0f;
  for (int m = lane; m < 16 * MT; m += 64) {
    const float *cf = reinterpret_cast<const float *>(reinterpret_cast<const char *>(hsm) + (size_t)(m >> 4) * C::TILE_BYTES + (size_t)NI * 2048);
    nwmax = fmaxf(nwmax, cf[16 + (m & 15)]);
    ncmax = fmaxf(ncmax, cf[32 + (m & 15)]);
  }
#pragma unroll
  for (int sh = 1; sh < 64; sh <<= 1) {
    nwmax = fmaxf(nwmax, __shfl_xor(nwmax, sh));
    ncmax = fmaxf(ncmax, __shfl_xor(ncmax, sh));
  }
  constexpr int kPasses = kGroupChunk / (16 * 2 * (kHardKeyThreads / 64));
  for (int64_t cu = blockIdx.x; cu < nrun; cu += gridDim.x) {
    const int64_t c = split ? cu / kPasses : cu;
    for (int m = tid; m < MK; m += kHardKeyThreads) hist[m] = 0;
    __syncthreads();
    // TWO frame tiles (32 frames) per wave and pass: every operand fragment read from LDS feeds six MFMAs instead of three -- the
    // kernel is bound by those reads (80 ds_read_b128 of 1 KB per 16 frames at Dj = 80, M = 128: round 6, profiles/r06_ab)
    constexpr int FT = 2;
    static_assert(kPasses == kGroupChunk / (16 * FT * (kHardKeyThreads / 64)), "passes per chunk");
    // the rows of a pass are requested one pass ahead (into the registers the previous pass's rows have just left: they are
    // converted to bf16 operands first), so that their latency runs under the MFMAs of the pass before
    typedef double kd2 __attribute__((ext_vector_type(2)));
    kd2 xraw[FT][NI][2];
    const int it_first = split ? (int)(cu % kPasses) : 0, it_end = split ? it_first + 1 : kPasses;
    auto request = [&](int it) {
      const int64_t fr0 = c * cstride * kGroupChunk + 16 * FT * ((kHardKeyThreads / 64) * it + wave) + lcol;
#pragma unroll
      for (int f = 0; f < FT; ++f) {
        const int64_t fr = fr0 + 16 * f;
        // 16-byte loads on clamped addresses (dj is even and the rows are 16-byte aligned on this path: a pair is inside the row
        // or outside it as a whole), masked when they are used -- no branch around a load
        const double *xr = X + (fr < N ? fr : N - 1) * dj;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const int d = 16 * i + 4 * lgrp + 2 * j;
            xraw[f][i][j] = *reinterpret_cast<const kd2 *>(xr + (d < dj ? d : 0));
          }
        }
      }
    };
    request(it_first);
    for (int it = it_first; it < it_end; ++it) {
      const int64_t fr0 = c * cstride * kGroupChunk + 16 * FT * ((kHardKeyThreads / 64) * it + wave) + lcol;
      if (fr0 - lcol >= N) break;                                   // (wave-uniform)
      // B operands: slot j < 4: x^2, j >= 4: x, of dimensions 16 i + 4 g + (j & 3); and |[x^2 ; x]|^2
      u32x4_t bh[FT][NI], bl[FT][NI];
      float nxe[FT];
#pragma unroll
      for (int f = 0; f < FT; ++f) {
        double q = 0.0;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
          double x[4];
#pragma unroll
          for (int j = 0; j < 4; j += 2) {
            const int d = 16 * i + 4 * lgrp + j;
            const bool in = d < dj;
            const kd2 v = xraw[f][i][j >> 1];
            x[j] = in ? v.x : 0.0;
            x[j + 1] = in ? v.y : 0.0;
          }
          double x2[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            x2[j] = x[j] * x[j];
            q = fma(x2[j], x2[j], fma(x[j], x[j], q));
          }
          unsigned ph[4], pl[4];
          split_bf16_pair(x2[0], x2[1], ph[0], pl[0]);               // slots 0 .. 3: x^2, 4 .. 7: x
          split_bf16_pair(x2[2], x2[3], ph[1], pl[1]);
          split_bf16_pair(x[0], x[1], ph[2], pl[2]);
          split_bf16_pair(x[2], x[3], ph[3], pl[3]);
#pragma unroll
          for (int w2 = 0; w2 < 4; ++w2) {
            bh[f][i][w2] = ph[w2];
            bl[f][i][w2] = pl[w2];
          }
        }
        q += __shfl_xor(q, 16);
        q += __shfl_xor(q, 32);
        nxe[f] = (float)(sqrt(q) * (1.0 + 0x1p-20));
      }
      if (it + 1 < it_end) request(it + 1);                        // (beyond N: clamped to the last row, never used)
      // The margin of every mixture is at most E = NWmax |[x^2 ; x]| + NCmax (the largest 2^-12 |W_m| and 2^-12 |c_m| of the model,
      // floats 0 and 1 behind the operands), so it suffices to know the two largest l^ of the frame: hard iff
      // second + E < (best - E) - 746.  This lane sees rows 4 lgrp .. 4 lgrp + 3 of every mixture tile.
      float b1[FT], b2[FT];                                         // the largest and second largest l^ among the lane's mixtures
      int bm[FT];
#pragma unroll
      for (int f = 0; f < FT; ++f) {
        b1[f] = -INFINITY;
        b2[f] = -INFINITY;
        bm[f] = 0;
      }
      for (int mt = 0; mt < MT; ++mt) {
        const char *tb = reinterpret_cast<const char *>(hsm) + (size_t)mt * C::TILE_BYTES;
        f32x4_t acc[FT];
#pragma unroll
        for (int f = 0; f < FT; ++f) acc[f] = *reinterpret_cast<const f32x4_t *>(tb + NI * 2048 + 16 * lgrp);
#pragma unroll
        for (int i = 0; i < NI; ++i) {
          const u32x4_t ah = *reinterpret_cast<const u32x4_t *>(tb + i * 2048 + 16 * lane), al = *reinterpret_cast<const u32x4_t *>(tb + i * 2048 + 1024 + 16 * lane);
#pragma unroll
          for (int f = 0; f < FT; ++f) {
            acc[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, ah), __builtin_bit_cast(bf16x8_t, bh[f][i]), acc[f], 0, 0, 0);
            acc[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, ah), __builtin_bit_cast(bf16x8_t, bl[f][i]), acc[f], 0, 0, 0);
#ifndef ESTEP_KEY_TWO_TERMS      // (timing experiment: without the W-lo term -- and its LDS read -- the margins would not hold)
            acc[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, al), __builtin_bit_cast(bf16x8_t, bh[f][i]), acc[f], 0, 0, 0);
#endif
          }
        }
#pragma unroll
        for (int f = 0; f < FT; ++f) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float v = acc[f][r];
            const bool nb = v > b1[f];
            b2[f] = nb ? b1[f] : fmaxf(b2[f], v);
            bm[f] = nb ? 16 * mt + 4 * lgrp + r : bm[f];
            b1[f] = nb ? v : b1[f];
          }
        }
      }
#pragma unroll
      for (int f = 0; f < FT; ++f) {
        const int64_t fr = fr0 + 16 * f;
        // across the four lane groups of the frame (ties: the smaller index -- any choice is certified or none is)
#pragma unroll
        for (int sh = 16; sh < 64; sh <<= 1) {
          const float o1 = __shfl_xor(b1[f], sh), o2 = __shfl_xor(b2[f], sh);
          const int om = __shfl_xor(bm[f], sh);
          const bool take = o1 > b1[f] || (o1 == b1[f] && om < bm[f]);
          b2[f] = fmaxf(fmaxf(b2[f], o2), take ? b1[f] : o1);
          bm[f] = take ? om : bm[f];
          b1[f] = take ? o1 : b1[f];
        }
        const float E = fmaf(nwmax, nxe[f], ncmax) * 1.000001f;
        const float blo = b1[f] - E, hi2 = b2[f] + E;
        if (lgrp == 0 && fr < N) {
          // hard: every other mixture is certified more than 746 nats below the best one (exp underflows to exactly 0 below
          // -745.2; the margin also covers the 1e-7 the one-kernel path's own log-densities may be off), and the best is finite
          const bool hard = bm[f] < M && blo > -1e29f && hi2 < blo - 746.0f;
          const int k = hard ? bm[f] : M;
          if (key) key[fr] = k;
          atomicAdd(&hist[k], 1);
        }
      }
    }
    __syncthreads();
    for (int m = tid; m < MK; m += kHardKeyThreads) chunkhist[cu * MK + m] = hist[m];
    __syncthreads();
  }
}

// the sampled look: out[0] = frames of the sample without an owner, out[1] = frames of the sample
__global__ void estep_hard_probe_sum_kernel(const int *__restrict__ chunkhist, int nsample, int MK, int *__restrict__ out) {
  int soft = 0, all = 0;
  for (int e = threadIdx.x; e < nsample * MK; e += 64) {
    const int v = chunkhist[e];
    all += v;
    soft += (e % MK == MK - 1) ? v : 0;
  }
#pragma unroll
  for (int sh = 1; sh < 64; sh <<= 1) {
    soft += __shfl_xor(soft, sh);
    all += __shfl_xor(all, sh);
  }
  if (threadIdx.x == 0) {
    out[0] = soft;
    out[1] = all;
  }
}

// piece table: workgroup b -> (mixture, first position, count) of its piece of the sorted hard frames; pieces of a mixture are
// consecutive.  total[k] = frames of key k (k = M: soft), in key order the sorted positions are prefix sums of total.
// part16 row b: [count | S1 (dj) | S2 (dj) | sum_d T_d] -- rows of mixtures' pieces in order; rows beyond the last piece are
// not written (estep_hard_reduce_kernel walks the same table).
template <int DJ>
__global__ void __launch_bounds__(256)
estep_hard_stats_kernel(const double *__restrict__ X, int dj, int M, const int *__restrict__ perm, const int *__restrict__ total,
                        const double *__restrict__ mu, const double *__restrict__ iv, double *__restrict__ part, int64_t prow,
                        const int64_t *__restrict__ gate) {
  if (gate && *gate == 0) return;
  constexpr int RPP = 256 / DJ;                                 // rows in flight per pass
  __shared__ int piece[3], tot[kHardMaxM];
  __shared__ double red[3][RPP > 1 ? RPP : 1][DJ];
  if ((int)threadIdx.x < M) tot[threadIdx.x] = total[threadIdx.x];       // (one load per thread instead of a serial walk through global memory)
  __syncthreads();
  if (threadIdx.x == 0) {
    int b = blockIdx.x, pos = 0, m = 0;
    for (; m < M; ++m) {
      const int np = (tot[m] + kHardPiece - 1) / kHardPiece;
      if (b < np) break;
      b -= np;
      pos += tot[m];
    }
    piece[0] = m;
    piece[1] = m < M ? pos + b * kHardPiece : 0;
    const int left = m < M ? tot[m] - b * kHardPiece : 0;
    piece[2] = left < kHardPiece ? left : kHardPiece;
  }
  __syncthreads();
  const int m = piece[0], p0 = piece[1], n = piece[2];
  if (m >= M) return;                                           // (workgroup-uniform: beyond the last piece)
  const int slot = threadIdx.x / DJ, d = threadIdx.x - slot * DJ;
  double s1 = 0.0, s2 = 0.0, tt = 0.0;
  if (slot < RPP && d < dj) {
    const double mud = mu[d + (size_t)dj * m], ivd = iv[d + (size_t)dj * m];
    // eight rows in flight (their perm entries first, then the rows), accumulated in row order
    for (int f0 = slot; f0 < n; f0 += 8 * RPP) {
      int pr[8];
      double xv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int f = f0 + u * RPP;
        pr[u] = perm[p0 + (f < n ? f : n - 1)];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) xv[u] = X[(int64_t)pr[u] * dj + d];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (f0 + u * RPP < n) {
          const double x = xv[u], df = x - mud;
          s1 += x;
          s2 = fma(x, x, s2);
          tt = fma(df * df, ivd, tt);
        }
      }
    }
  }
  if (slot < RPP) {
    red[0][slot][d] = s1;
    red[1][slot][d] = s2;
    red[2][slot][d] = tt;
  }
  __syncthreads();
  double *P = part + (size_t)blockIdx.x * prow;
  if (threadIdx.x < dj) {                                       // the slots in order
    double a = 0.0, b2 = 0.0, c = 0.0;
#pragma unroll
    for (int s = 0; s < RPP; ++s) {
      a += red[0][s][threadIdx.x];
      b2 += red[1][s][threadIdx.x];
      c += red[2][s][threadIdx.x];
    }
    P[1 + threadIdx.x] = a;
    P[1 + dj + threadIdx.x] = b2;
    red[2][0][threadIdx.x] = c;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int k = 0; k < dj; ++k) t += red[2][0][k];
    P[0] = (double)n;
    P[1 + 2 * dj] = t;
  }
}

// one workgroup per mixture: its pieces added in order -> stats = [S0 (M) | S1 (dj,M) | S2 (dj,M) | .]; llm[m] = the mixture's
// share of the log-likelihood, count * c'_m - sum T / 2 with c'_m = log w - (dj log 2 pi + sum log var) / 2 (refc)
__global__ void __launch_bounds__(256)
estep_hard_reduce_kernel(const double *__restrict__ part, int64_t prow, const int *__restrict__ total, int M, int dj,
                         const double *__restrict__ refc, double *__restrict__ stats, double *__restrict__ llm, const int64_t *__restrict__ gate) {
  if (gate && *gate == 0) return;
  const int m = blockIdx.x;
  __shared__ int tot[kHardMaxM];
  if ((int)threadIdx.x < M) tot[threadIdx.x] = total[threadIdx.x];
  __syncthreads();
  int first = 0;
  for (int k = 0; k < m; ++k) first += (tot[k] + kHardPiece - 1) / kHardPiece;
  const int np = (tot[m] + kHardPiece - 1) / kHardPiece;
  const int e = threadIdx.x;                                    // element of the row: 0 count, 1 .. dj S1, dj+1 .. 2dj S2, 2dj+1 T
  if (e > 2 * dj + 1) return;
  // four partial sums over every fourth piece, combined in a fixed order; sixteen loads in flight
  double q0 = 0.0, q1 = 0.0, q2 = 0.0, q3 = 0.0;
  const double *pp = part + (size_t)first * prow + e;
  int r = 0;
  for (; r + 15 < np; r += 16) {
    double v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = pp[(size_t)(r + u) * prow];
#pragma unroll
    for (int u = 0; u < 16; u += 4) {
      q0 += v[u];
      q1 += v[u + 1];
      q2 += v[u + 2];
      q3 += v[u + 3];
    }
  }
  for (; r + 3 < np; r += 4) {
    const double v0 = pp[(size_t)r * prow], v1 = pp[(size_t)(r + 1) * prow], v2 = pp[(size_t)(r + 2) * prow], v3 = pp[(size_t)(r + 3) * prow];
    q0 += v0;
    q1 += v1;
    q2 += v2;
    q3 += v3;
  }
  for (; r < np; ++r) q0 += pp[(size_t)r * prow];
  const double s = (q0 + q1) + (q2 + q3);
  if (e == 0) stats[m] = s;
  else if (e <= dj) stats[M + (size_t)m * dj + (e - 1)] = s;
  else if (e <= 2 * dj) stats[M + (size_t)M * dj + (size_t)m * dj + (e - 1 - dj)] = s;
  else llm[m] = np > 0 ? (double)tot[m] * refc[2 * m] - 0.5 * s : 0.0;
}
// the hard frames' log-likelihood (mixtures in order) into the statistics' last element; the soft frames' is added by the
// one-kernel path's reduction afterwards
__global__ void estep_hard_ll_kernel(const double *__restrict__ llm, int M, double *__restrict__ stats, int64_t plen, const int64_t *__restrict__ gate) {
  if (gate && *gate == 0) return;
  __shared__ double v[kHardMaxM];
  for (int m = threadIdx.x; m < M; m += blockDim.x) v[m] = llm[m];
  __syncthreads();
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    double ll = 0.0;
    for (int m = 0; m < M; ++m) ll += v[m];
    stats[plen - 1] = ll;
  }
}

// the soft frames (sorted positions behind the hard ones), rows copied into a dense matrix for estep_mfma_kernel; nsoft[0] =
// their number (read by that kernel from device memory)
__global__ void __launch_bounds__(256)
estep_hard_gather_kernel(const double *__restrict__ X, int dj, int M, const int *__restrict__ perm, const int *__restrict__ total,
                         double *__restrict__ Xs, int64_t *__restrict__ nsoft, int64_t N, const int64_t *__restrict__ gate) {
  if (gate && *gate == 0) return;                             // (nsoft stays 0: estep_path_decide_kernel)
  const int64_t n = total[M], base = N - n;                     // (every frame has a key: the hard ones come first)
  if (blockIdx.x == 0 && threadIdx.x == 0) nsoft[0] = n;
  const int64_t ne = n * dj;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < ne; e += (int64_t)gridDim.x * 256) {
    const int64_t f = e / dj;
    const int d = (int)(e - f * dj);
    Xs[e] = X[(int64_t)perm[base + f] * dj + d];
  }
}

}  // namespace vcmi
