// estep_onepass.hpp -- the hard-assignment path of the diagonal E-step (estep_hard.hpp) in ONE pass over X (round 6; included by
// estep.hip behind estep_hard.hpp; M <= 128, DJ in {32, 48, 64, 80}).
//
// Round 5's path read X twice and ran thirteen kernels: keys (a certified bf16 screen: which frames does ONE mixture own?) ->
// scan -> scatter (a global stable sort by owner) -> per-mixture sums over the sorted rows through the permutation -> reduce ->
// log-likelihood -> gather of the frames without an owner -> the FP64 kernel on those -> reduce.  The sort existed to turn
// "sum the rows of mixture m" into contiguous pieces.  It is not needed: the accumulators can stay where they are and the
// FRAMES come to them --
//   * a persistent workgroup of 1024 threads per CU owns the 2 x 80 x 128 running sums of ITS frames in registers: thread
//     (m, p) = (tid / 8, tid % 8) holds sum (x - mu_m) and sum (x - mu_m)^2 over the DJ / 8 dimensions of part p of mixture m
//     (20 doubles at DJ = 80; centred sums: the owner's log-density then needs no cancelling terms; mu_m comes with the rows);
//   * per round of 256 frames: each of the sixteen waves screens 16 frames (the arithmetic of estep_hard_key_kernel, operands of
//     all mixtures resident in LDS), writes their owners (or "none") to LDS; one barrier; every thread scans the round's 256
//     owners IN FRAME ORDER and adds the frames its mixture owns -- their rows come back from L2 (this CU read them a moment
//     ago), 80 contiguous bytes per thread; the owners are double-buffered, so a round costs one barrier;
//   * frames without an owner go, in frame order, into the chunk's list of indices (1024 frames per chunk, chunks handed to the
//     workgroups with a fixed stride: every sum is a function of the data alone);
//   * estep_onepass_finish_kernel adds the workgroups' sums in order, uncentres them (S1 = S1' + n mu, S2 = S2' + 2 mu S1' + n
//     mu^2), forms the owned frames' log-likelihood n c'_m - sum_d S2'_d / (2 var_d) and the prefix of the chunks' soft counts;
//     the soft rows are copied into a dense matrix for estep_mfma_kernel as before.
// X is read from HBM once (the second touch is an L2 hit), and the launches of the path drop from thirteen to eight.
// WHICH path a call takes is decided from THIS call's data on the device (estep_path_decide_kernel: the screen on a sample of
// 16 chunks), never from what earlier calls found: every kernel of the path starts with a look at ctl[0].
#pragma once
#include "estep_hard.hpp"

namespace vcmi {

#ifdef VCMI_ONEPASS_PROF
__device__ unsigned long long onepass_prof[8];      // wave cycles: A (screen), barrier, B (sums), soft list; rounds
#define OP_T(k) { const unsigned long long t_ = __builtin_readcyclecounter(); pp[k] += t_ - tlast; tlast = t_; }
#else
#define OP_T(k)
#endif

constexpr int kOnePassThreads = 1024;
constexpr int kOnePassRound = 256;            // frames per round: 16 waves x 16
// ctl (int64, device): [0] 1: the hard-assignment path runs / 0: every frame through estep_mfma_kernel; [1] frames of the
// "everything soft" launch (N or 0); [2] soft frames found by the one-pass kernel (set by the finish kernel)
enum { kCtlHard = 0, kCtlAllSoft = 1, kCtlNSoft = 2, kCtlLen = 4 };

// decision from the sample's histograms (hist[c][k], k = M: no owner): hard iff at most a quarter of the sample has no owner.
// force: -1 decide, 0 / 1 set.
__global__ void estep_path_decide_kernel(const int *__restrict__ hist, int nsample, int MK, int force, int64_t N, int64_t *__restrict__ ctl) {
  int soft = 0, all = 0;
  if (force < 0) {
    for (int e = threadIdx.x; e < nsample * MK; e += 64) {
      const int v = hist[e];
      all += v;
      soft += (e % MK == MK - 1) ? v : 0;
    }
#pragma unroll
    for (int sh = 1; sh < 64; sh <<= 1) {
      soft += __shfl_xor(soft, sh);
      all += __shfl_xor(all, sh);
    }
  }
  if (threadIdx.x == 0) {
    const bool hard = force < 0 ? (all > 0 && 4 * (int64_t)soft <= (int64_t)all) : force != 0;
    ctl[kCtlHard] = hard ? 1 : 0;
    ctl[kCtlAllSoft] = hard ? 0 : N;
    ctl[kCtlNSoft] = 0;
  }
}

// part (per workgroup): [M][1 + 2 dj] doubles: count | S1' (dj) | S2' (dj)
template <int DJ>
__global__ void __launch_bounds__(kOnePassThreads)
estep_onepass_kernel(const unsigned char *__restrict__ W16, int M, int dj, const double *__restrict__ X, int64_t N,
                     const double *__restrict__ mu, const int64_t *__restrict__ ctl, double *__restrict__ part, int *__restrict__ softidx,
                     int *__restrict__ softcount, int64_t nchunks) {
  if (ctl[kCtlHard] == 0) return;
  using C = EstepHardCfg<DJ>;
  constexpr int NI = C::NI, DP8 = DJ / 8;                    // dimensions per accumulator thread (eight threads per mixture)
  static_assert(DJ % 16 == 0, "an accumulator thread's slice must be a whole number of 16-byte pairs");
  extern __shared__ double hsm[];
  const int MT = (M + 15) / 16;
  const int nd = (int)(C::lds_bytes(MT) / 8);
  int *keys = reinterpret_cast<int *>(hsm + nd);             // [2][kOnePassRound]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lcol = lane & 15, lgrp = lane >> 4;
  for (int e = tid; e < nd; e += kOnePassThreads) hsm[e] = reinterpret_cast<const double *>(W16)[e];
  // this thread's accumulators: mixture am, dimensions d0 .. d0 + DP8 - 1 (centred on mu_m, which is fetched with the rows: 1024
  // threads leave 128 registers each, and the running sums are 4 DP8 of them)
  const int am = tid >> 3, d0 = (tid & 7) * DP8;
  const bool acc_on = am < M;
  double s1[DP8], s2[DP8];
  double cnt = 0.0;
#pragma unroll
  for (int i = 0; i < DP8; ++i) {
    s1[i] = 0.0;
    s2[i] = 0.0;
  }
  __syncthreads();
  // the largest margins of the model (estep_hard_key_kernel): NWmax = max 2^-12 |W_m|, NCmax = max 2^-12 |c_m|
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
  int round = 0;
#ifdef VCMI_ONEPASS_PROF
  unsigned long long pp[5] = {0, 0, 0, 0, 0}, tlast = __builtin_readcyclecounter();
#endif
  for (int64_t c = blockIdx.x; c < nchunks; c += gridDim.x) {
    int nsoft_chunk = 0;                                      // (kept by wave 0)
    const int64_t cf0 = c * kGroupChunk;
    for (int it = 0; it < kGroupChunk / kOnePassRound; ++it, ++round) {
      const int64_t r0 = cf0 + (int64_t)kOnePassRound * it;
      if (r0 >= N) break;                                     // (workgroup-uniform)
      int *kr = keys + (round & 1) * kOnePassRound;
      // ---- A: the owners of this wave's 16 frames
      {
        const int64_t fr = r0 + 16 * wave + lcol;
        u32x4_t bh[NI], bl[NI];
        double q = 0.0;
        const double *xr = X + (fr < N ? fr : N - 1) * dj;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
          double x[4];
          typedef double kd2 __attribute__((ext_vector_type(2)));
#pragma unroll
          for (int j = 0; j < 4; j += 2) {
            const int d = 16 * i + 4 * lgrp + j;
            const bool in = d < dj;
            const kd2 v = *reinterpret_cast<const kd2 *>(xr + (in ? d : 0));
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
          split_bf16_pair(x2[0], x2[1], ph[0], pl[0]);             // slots 0 .. 3: x^2, 4 .. 7: x
          split_bf16_pair(x2[2], x2[3], ph[1], pl[1]);
          split_bf16_pair(x[0], x[1], ph[2], pl[2]);
          split_bf16_pair(x[2], x[3], ph[3], pl[3]);
#pragma unroll
          for (int w2 = 0; w2 < 4; ++w2) {
            bh[i][w2] = ph[w2];
            bl[i][w2] = pl[w2];
          }
        }
        q += __shfl_xor(q, 16);
        q += __shfl_xor(q, 32);
        const float nxe = (float)(sqrt(q) * (1.0 + 0x1p-20));
        float b1 = -INFINITY, b2 = -INFINITY;                       // the largest and second largest l^ among the lane's mixtures
        int bm = 0;
        for (int mt = 0; mt < MT; ++mt) {
          const char *tb = reinterpret_cast<const char *>(hsm) + (size_t)mt * C::TILE_BYTES;
          f32x4_t acc = *reinterpret_cast<const f32x4_t *>(tb + NI * 2048 + 16 * lgrp);
#pragma unroll
          for (int i = 0; i < NI; ++i) {
            const u32x4_t ah = *reinterpret_cast<const u32x4_t *>(tb + i * 2048 + 16 * lane), al = *reinterpret_cast<const u32x4_t *>(tb + i * 2048 + 1024 + 16 * lane);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, ah), __builtin_bit_cast(bf16x8_t, bh[i]), acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, ah), __builtin_bit_cast(bf16x8_t, bl[i]), acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, al), __builtin_bit_cast(bf16x8_t, bh[i]), acc, 0, 0, 0);
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float v = acc[r];
            const bool nb = v > b1;
            b2 = nb ? b1 : fmaxf(b2, v);
            bm = nb ? 16 * mt + 4 * lgrp + r : bm;
            b1 = nb ? v : b1;
          }
        }
#pragma unroll
        for (int sh = 16; sh < 64; sh <<= 1) {
          const float o1 = __shfl_xor(b1, sh), o2 = __shfl_xor(b2, sh);
          const int om = __shfl_xor(bm, sh);
          const bool take = o1 > b1 || (o1 == b1 && om < bm);
          b2 = fmaxf(fmaxf(b2, o2), take ? b1 : o1);
          bm = take ? om : bm;
          b1 = take ? o1 : b1;
        }
        const float E = fmaf(nwmax, nxe, ncmax) * 1.000001f;
        const float blo = b1 - E, hi2 = b2 + E;
        if (lgrp == 0) {
          // hard: every other mixture is certified more than 746 nats below the best one (estep_hard.hpp); -1: beyond N
          const bool hard = bm < M && blo > -1e29f && hi2 < blo - 746.0f;
          kr[16 * wave + lcol] = fr < N ? (hard ? bm : M) : -1;
        }
      }
      OP_T(0)
      __syncthreads();                                          // the round's owners are in LDS (the other buffer is the previous round's)
      OP_T(1)
      // ---- B: every accumulator thread takes the frames of the round that its mixture owns, in frame order.  The eight threads
      // of a mixture each compare an eighth of the round's owners and exchange the 32-bit masks inside their group; the rows then
      // come back from L2 two frames at a time together with the mixture's means: one exposed load latency per pair of frames
      // (the first version took the frames one by one: 16 serialised L2 round trips per wave and round, 1.16 ms per 1.25e6 frames)
      {
        unsigned mask[8];
        {
          const int part = tid & 7;
          unsigned mm = 0u;
          if (acc_on) {
#pragma unroll
            for (int f4 = 0; f4 < 32; f4 += 4) {
              const int4 k4 = *reinterpret_cast<const int4 *>(kr + 32 * part + f4);
              mm |= (k4.x == am ? 1u : 0u) << f4 | (k4.y == am ? 2u : 0u) << f4 | (k4.z == am ? 4u : 0u) << f4 | (k4.w == am ? 8u : 0u) << f4;
            }
          }
#pragma unroll
          for (int q = 0; q < 8; ++q) mask[q] = (unsigned)__shfl((int)mm, (lane & ~7) | q);
        }
        typedef double kd2 __attribute__((ext_vector_type(2)));
        constexpr int HP = DP8 / 2;                               // 16-byte pairs per slice
        constexpr int KB = 2;                                     // frames per batch
        int w = 0;                                                // first mask word that may still hold a frame
        for (;;) {
          while (w < 7 && mask[w] == 0u) ++w;                     // (per-thread; eight words at most over the whole round)
          const bool more = mask[w] != 0u;
          if (__builtin_amdgcn_ballot_w64(more) == 0) break;
          int pos[KB];
          bool on[KB];
#pragma unroll
          for (int k = 0; k < KB; ++k) {                          // the next KB owned frames of this thread, in order
            unsigned mw = 0u;
            int ww = 0;
#pragma unroll
            for (int q = 7; q >= 0; --q) {
              const bool take = mask[q] != 0u;
              mw = take ? mask[q] : mw;
              ww = take ? q : ww;
            }
            on[k] = mw != 0u;
            pos[k] = 32 * ww + (on[k] ? __builtin_ctz(mw) : 0);
            const unsigned clr = mw & (mw - 1u);
#pragma unroll
            for (int q = 0; q < 8; ++q) mask[q] = (on[k] && q == ww) ? clr : mask[q];
          }
          kd2 v[KB][HP], mv[HP];
#pragma unroll
          for (int i = 0; i < HP; ++i) mv[i] = (acc_on && d0 + 2 * i < dj) ? *reinterpret_cast<const kd2 *>(mu + (size_t)dj * am + d0 + 2 * i) : kd2{0.0, 0.0};
#pragma unroll
          for (int k = 0; k < KB; ++k) {
            const double *xr = X + (r0 + (on[k] ? pos[k] : 0)) * dj + d0;
#pragma unroll
            for (int i = 0; i < HP; ++i) v[k][i] = (d0 + 2 * i < dj) ? *reinterpret_cast<const kd2 *>(xr + 2 * i) : kd2{0.0, 0.0};
          }
#pragma unroll
          for (int k = 0; k < KB; ++k) {
            if (on[k]) {
#pragma unroll
              for (int i = 0; i < HP; ++i) {
                const double a = v[k][i].x - mv[i].x, b = v[k][i].y - mv[i].y;
                s1[2 * i] += a;
                s1[2 * i + 1] += b;
                s2[2 * i] = fma(a, a, s2[2 * i]);
                s2[2 * i + 1] = fma(b, b, s2[2 * i + 1]);
              }
              cnt += 1.0;
            }
          }
        }
      }
      OP_T(2)
      // the frames without an owner, in frame order, into the chunk's list (wave 0: two ballots per round)
      if (wave == 0) {
        const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
        for (int h = 0; h < kOnePassRound / 64; ++h) {
          const int kk = kr[64 * h + lane];
          const unsigned long long bb = __builtin_amdgcn_ballot_w64(kk == M);
          if (kk == M) softidx[cf0 + nsoft_chunk + __builtin_popcountll(bb & below)] = (int)(r0 + 64 * h + lane);
          nsoft_chunk += __builtin_popcountll(bb);
        }
      }
      OP_T(3)
#ifdef VCMI_ONEPASS_PROF
      pp[4] += 1;
#endif
    }
    if (wave == 0 && lane == 0) softcount[c] = nsoft_chunk;
  }
#ifdef VCMI_ONEPASS_PROF
  if (lane == 0)
    for (int k = 0; k < 5; ++k) atomicAdd(&onepass_prof[k], pp[k]);
#endif
  if (acc_on) {
    double *P = part + ((size_t)blockIdx.x * M + am) * (1 + 2 * dj);
    if ((tid & 7) == 0) P[0] = cnt;
#pragma unroll
    for (int i = 0; i < DP8; ++i) {
      if (d0 + i < dj) {
        P[1 + d0 + i] = s1[i];
        P[1 + dj + d0 + i] = s2[i];
      }
    }
  }
}

// Workgroup m < M: the workgroups' sums of mixture m in order (four interleaved partial sums, combined in a fixed order), uncentred
// into stats = [S0 (M) | S1 (dj,M) | S2 (dj,M) | .]; llm[m] = the mixture's share of the log-likelihood, n c'_m - sum_d S2'_d /
// (2 var_d) with c'_m = refc[2 m].  Workgroup M: the exclusive prefix of the chunks' soft counts and their total (ctl[kCtlNSoft]).
__global__ void __launch_bounds__(256)
estep_onepass_finish_kernel(const double *__restrict__ part, int nwg, int M, int dj, const double *__restrict__ mu, const double *__restrict__ iv,
                            const double *__restrict__ refc, const int *__restrict__ softcount, int64_t nchunks, int64_t *__restrict__ softoffs,
                            int64_t *__restrict__ ctl, double *__restrict__ stats, double *__restrict__ llm) {
  if (ctl[kCtlHard] == 0) return;
  const int m = blockIdx.x, tid = threadIdx.x;
  if (m == M) {
    __shared__ int64_t psum[256];
    const int64_t per = (nchunks + 255) / 256, lo = std::min<int64_t>(nchunks, tid * per), hi = std::min<int64_t>(nchunks, lo + per);
    int64_t s = 0;
    for (int64_t c = lo; c < hi; ++c) s += softcount[c];
    psum[tid] = s;
    __syncthreads();
    if (tid == 0) {
      int64_t run = 0;
      for (int i = 0; i < 256; ++i) {
        const int64_t v = psum[i];
        psum[i] = run;
        run += v;
      }
      ctl[kCtlNSoft] = run;
    }
    __syncthreads();
    int64_t run = psum[tid];
    for (int64_t c = lo; c < hi; ++c) {
      softoffs[c] = run;
      run += softcount[c];
    }
    return;
  }
  __shared__ double sh1[kHardMaxM > 160 ? kHardMaxM : 160], sh2[160], red[160];
  const int prow = 1 + 2 * dj;
  const int e = tid;                                          // element of the mixture's row: 0 count, 1 .. dj S1', dj + 1 .. 2 dj S2'
  double s = 0.0;
  if (e < prow) {
    double q0 = 0.0, q1 = 0.0, q2 = 0.0, q3 = 0.0;
    const double *pp = part + (size_t)m * prow + e;
    const size_t stride = (size_t)M * prow;
    int r = 0;
    for (; r + 3 < nwg; r += 4) {
      const double v0 = pp[(size_t)r * stride], v1 = pp[(size_t)(r + 1) * stride], v2 = pp[(size_t)(r + 2) * stride], v3 = pp[(size_t)(r + 3) * stride];
      q0 += v0;
      q1 += v1;
      q2 += v2;
      q3 += v3;
    }
    for (; r < nwg; ++r) q0 += pp[(size_t)r * stride];
    s = (q0 + q1) + (q2 + q3);
    if (e == 0) sh1[0] = s;                                   // (count in sh1[0]; the sums start at index 1)
    else if (e <= dj) sh1[e] = s;
    else sh2[e - dj] = s;
  }
  __syncthreads();
  const double n = sh1[0];
  if (e >= 1 && e <= dj) {
    const int d = e - 1;
    const double mud = mu[d + (size_t)dj * m], a = sh1[e], b = sh2[e];
    stats[M + (size_t)m * dj + d] = fma(n, mud, a);                                              // S1 = S1' + n mu
    stats[M + (size_t)M * dj + (size_t)m * dj + d] = fma(mud, fma(n, mud, 2.0 * a), b);          // S2 = S2' + 2 mu S1' + n mu^2
    red[d] = b * iv[d + (size_t)dj * m];
  }
  __syncthreads();
  if (e == 0) {
    stats[m] = n;
    double t = 0.0;
    for (int d = 0; d < dj; ++d) t += red[d];
    llm[m] = n > 0.0 ? n * refc[2 * m] - 0.5 * t : 0.0;
  }
}

// the owned frames' log-likelihood (mixtures in order) into the statistics' last element; the soft frames' is added by the
// one-kernel path's reduction afterwards
__global__ void estep_onepass_ll_kernel(const double *__restrict__ llm, int M, const int64_t *__restrict__ ctl, double *__restrict__ stats, int64_t plen) {
  if (ctl[kCtlHard] == 0) return;
  __shared__ double v[kHardMaxM];
  for (int m = threadIdx.x; m < M; m += blockDim.x) v[m] = llm[m];
  __syncthreads();
  if (threadIdx.x == 0) {
    double ll = 0.0;
    for (int m = 0; m < M; ++m) ll += v[m];
    stats[plen - 1] = ll;
  }
}

// the soft frames' rows -> a dense matrix for estep_mfma_kernel (one workgroup walks chunks with a stride)
__global__ void __launch_bounds__(256)
estep_onepass_gather_kernel(const double *__restrict__ X, int dj, const int *__restrict__ softidx, const int *__restrict__ softcount,
                            const int64_t *__restrict__ softoffs, int64_t nchunks, const int64_t *__restrict__ ctl, double *__restrict__ Xs) {
  if (ctl[kCtlHard] == 0 || ctl[kCtlNSoft] == 0) return;
  for (int64_t c = blockIdx.x; c < nchunks; c += gridDim.x) {
    const int n = softcount[c];
    const int64_t off = softoffs[c];
    const int ne = n * dj;
    for (int e = threadIdx.x; e < ne; e += 256) {
      const int f = e / dj, d = e - f * dj;
      Xs[(off + f) * dj + d] = X[(int64_t)softidx[c * kGroupChunk + f] * dj + d];
    }
  }
}

}  // namespace vcmi
