// estep.hip -- E-step of EM for a diagonal-covariance GMM on joint features, MI355X (gfx950).
//
// Call site in the reference: bin/train_gmm.jl:103 `gmm[:fit](dataset.X')` (scikit-learn through PyCall); the
// joint feature matrix is built by src/datasets.jl:52-77.  The arithmetic itself is third-party (sklearn.mixture);
// what is implemented here is its published definition:
//   l_nm = log w_m - (Dj log 2pi + sum_d log var_md + sum_d (x_nd - mu_md)^2 / var_md) / 2
//   lse_n = logsumexp_m l_nm,   gamma_nm = exp(l_nm - lse_n)
//   S0_m = sum_n gamma_nm,  S1_md = sum_n gamma_nm x_nd,  S2_md = sum_n gamma_nm x_nd^2,  loglik = sum_n lse_n
// Output buffer layout (device): [S0 (M) | S1 (Dj,M) | S2 (Dj,M) | loglik] -- contiguous so that the multi-GPU
// path is ONE all-reduce(sum) of M(1+2Dj)+1 doubles over RCCL.
//
// Two implementations:
//  * MFMA kernel (Dj = 32, 48, 64, 80, 160 and M <= 128; v_mfma_f64_16x16x4_f64): both the log-density
//    l = [x^2, x] . [-iv/2 ; mu iv] + c  and the statistics  gamma' [x, x^2]  are dense FP64 contractions.
//    A workgroup of 8 waves owns a strided set of 64-frame blocks; wave w owns mixtures 16w..16w+15: their
//    weight fragments (80 VGPRs) and their 16 x 160 statistics accumulators (80 VGPRs) stay in registers for
//    the whole kernel.  Per-workgroup partial statistics are reduced in a fixed order by a second kernel,
//    so results are bit-identical run to run (no FP64 atomics).
//  * generic kernels (any Dj, M): gamma to an HBM workspace in chunks, then per-(m,d) sequential accumulation
//    over fixed frame segments, then a fixed-order reduction.
#include "vcmi_common.hpp"
#include <atomic>
#include "gmmmap_handle.hpp"
#include "devgroup.hpp"
#include "hostpipe.hpp"
#define VCMI_DPP_NO_COPY 1           // (fp64_exp.hpp: lane permutations without the copy of the old value)
#include "fp64_exp.hpp"
#ifndef VCMI_ESTEP_EXP_SKIP
#define VCMI_ESTEP_EXP_SKIP 1      // wave-uniform skip of the softmax exps of slot groups that are hopeless for the four frames of a pass (A/B: -0.9 %)
#endif

#include <algorithm>
#include <cmath>

namespace vcmi {

static constexpr double kLog2Pi = 1.8378770664093454835606594728112;
typedef double d4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------------
// generic path
// ------------------------------------------------------------------------------------------------
static constexpr int kChunk = 1 << 16;   // frames per workspace chunk
static constexpr int kSeg = 512;         // frames per deterministic accumulation segment

// one lane per frame: responsibilities gamma (chunk-local (n,M) row-major) and lse
__global__ void __launch_bounds__(64)
estep_gamma_kernel(const double *__restrict__ X, int64_t n0, int64_t nfr, int Dj, int M, const double *__restrict__ mu,
                   const double *__restrict__ iv, const double *__restrict__ cst, double *__restrict__ G,
                   double *__restrict__ LSE) {
  extern __shared__ double xs[];   // [Dj][64]
  const int lane = threadIdx.x;
  const int64_t nl = (int64_t)blockIdx.x * 64 + lane;
  const bool live = nl < nfr;
  for (int d = 0; d < Dj; ++d) xs[d * 64 + lane] = live ? X[(n0 + nl) * Dj + d] : 0.0;
  double *g = G + nl * M;
  double u = -INFINITY;
  for (int m = 0; m < M; ++m) {
    double q = 0.0;
    for (int d = 0; d < Dj; ++d) {
      const double df = xs[d * 64 + lane] - mu[(size_t)m * Dj + d];
      q = fma(df * df, iv[(size_t)m * Dj + d], q);
    }
    const double l = cst[m] - 0.5 * q;
    if (live) g[m] = l;
    u = fmax(u, l);
  }
  if (!live) return;
  double s = 0.0;
  for (int m = 0; m < M; ++m) s += exp(g[m] - u);
  const double lse = u + log(s);
  for (int m = 0; m < M; ++m) g[m] = exp(g[m] - lse);
  LSE[nl] = lse;
}

// thread per statistic element e = m*Dj + d, sequential over the frames of one segment
__global__ void __launch_bounds__(256)
estep_stats_kernel(const double *__restrict__ X, int64_t n0, int64_t nfr, int Dj, int M, const double *__restrict__ G,
                   const double *__restrict__ LSE, double *__restrict__ part, int64_t plen) {
  const int seg = blockIdx.x;
  const int e = blockIdx.y * 256 + threadIdx.x;
  const int64_t f0 = (int64_t)seg * kSeg, f1 = (f0 + kSeg < nfr) ? f0 + kSeg : nfr;
  double *P = part + (size_t)seg * plen;
  if (e < M * Dj) {
    const int m = e / Dj, d = e % Dj;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    for (int64_t f = f0; f < f1; ++f) {
      const double g = G[f * M + m], x = X[(n0 + f) * Dj + d];
      s0 += g;
      s1 = fma(g, x, s1);
      s2 = fma(g * x, x, s2);
    }
    if (d == 0) P[m] = s0;
    P[M + e] = s1;
    P[M + (size_t)M * Dj + e] = s2;
  }
  if (e == 0) {
    double ll = 0.0;
    for (int64_t f = f0; f < f1; ++f) ll += LSE[f];
    P[plen - 1] = ll;
  }
}

// stats[e] (+)= sum over the partial rows (accumulate = 0: stats need no memset before), in a FIXED order (deterministic): four threads per element each add up every
// fourth row in increasing order, then ((p0 + p1) + (p2 + p3)).  (One thread per element walking all rows one after the
// other left two thirds of the CUs idle and took 64 us for the 256 rows of the benchmark E-step -- 4 % of the step.)
__global__ void __launch_bounds__(256)
estep_reduce_kernel(const double *__restrict__ part, int nrows, int64_t plen, double *__restrict__ stats, int accumulate,
                    const int64_t *__restrict__ only_if) {
  if (only_if && *only_if == 0) return;      // (the soft frames of estep_hard.hpp: without any, the partials were not written)
  __shared__ double psum[4][64];
  const int el = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int64_t e = (int64_t)blockIdx.x * 64 + el;
  double s = 0.0;
  if (e < plen) {
    const double *p = part + e;
    int r = q;
    for (; r + 28 < nrows; r += 32) {              // eight independent loads in flight, added in row order
      const double v0 = p[(size_t)r * plen], v1 = p[(size_t)(r + 4) * plen], v2 = p[(size_t)(r + 8) * plen], v3 = p[(size_t)(r + 12) * plen],
                   v4 = p[(size_t)(r + 16) * plen], v5 = p[(size_t)(r + 20) * plen], v6 = p[(size_t)(r + 24) * plen], v7 = p[(size_t)(r + 28) * plen];
      s += v0; s += v1; s += v2; s += v3; s += v4; s += v5; s += v6; s += v7;
    }
    for (; r < nrows; r += 4) s += p[(size_t)r * plen];
  }
  psum[q][el] = s;
  __syncthreads();
  if (q == 0 && e < plen) stats[e] = (accumulate ? stats[e] : 0.0) + ((psum[0][el] + psum[1][el]) + (psum[2][el] + psum[3][el]));
}
static inline void estep_reduce_launch(const double *part, int nrows, int64_t plen, double *stats, hipStream_t st, int accumulate = 1,
                                       const int64_t *only_if = nullptr) {
  hipLaunchKernelGGL(estep_reduce_kernel, dim3((unsigned)((plen + 63) / 64)), dim3(256), 0, st, part, nrows, plen, stats, accumulate, only_if);
}

// More than 128 mixtures (groups of 128, one PHASE 3 launch each): the responsibilities G_g[f][.] are normalised within
// group g and lse_g[f] is the group's log-sum-exp.  Here, per frame: L = log sum_g e^(lse_g) (max-shifted, groups in
// order), G_g[f][.] *= e^(lse_g - L), and the frame's L goes into a per-workgroup partial of the log-likelihood (summed
// in fixed order by estep_sum_kernel).  One wave per frame; G is [group][n][128], lse [group][n].
__global__ void __launch_bounds__(256)
estep_group_combine_kernel(double *__restrict__ G, const double *__restrict__ lse, int ngroups, int64_t n, int64_t gstride,
                           double *__restrict__ llpart) {
  __shared__ double red[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double acc = 0.0;
  for (int64_t f = (int64_t)blockIdx.x * 4 + wave; f < n; f += (int64_t)gridDim.x * 4) {
    double u = -INFINITY;
    for (int g = 0; g < ngroups; ++g) u = fmax(u, lse[(size_t)g * n + f]);
    double s = 0.0;
    for (int g = 0; g < ngroups; ++g) s += exp(lse[(size_t)g * n + f] - u);
    const double L = u + log(s);
    for (int g = 0; g < ngroups; ++g) {
      const double lg = lse[(size_t)g * n + f];
      double *row = G + (size_t)g * gstride + f * 128;
      if (lg == -INFINITY) {                    // a group without any weight: its responsibilities are exactly zero
        row[lane] = 0.0;
        row[lane + 64] = 0.0;
        continue;
      }
      const double sc = exp(lg - L);
      row[lane] *= sc;
      row[lane + 64] *= sc;
    }
    acc += L;                                   // (every lane holds the same value)
  }
  if (lane == 0) red[wave] = acc;
  __syncthreads();
  if (threadIdx.x == 0) llpart[blockIdx.x] = ((red[0] + red[1]) + red[2]) + red[3];
}

// partial statistics of one mixture group (rows of plen_g = Mg (1 + 2 dj) + 1 doubles: [S0 | S1 (dj,Mg) | S2 (dj,Mg) | unused])
// summed in row order and added into the full layout at mixture m0
__global__ void __launch_bounds__(256)
estep_group_reduce_kernel(const double *__restrict__ part, int nrows, int64_t plen_g, int Mg, int dj, int m0, int M,
                          double *__restrict__ stats) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x, per = (int64_t)Mg * dj;
  if (e >= plen_g - 1) return;
  double s = 0.0;
  for (int r = 0; r < nrows; ++r) s += part[(size_t)r * plen_g + e];
  const int64_t dst = e < Mg ? m0 + e : e < Mg + per ? M + (int64_t)m0 * dj + (e - Mg) : M + (int64_t)M * dj + (int64_t)m0 * dj + (e - Mg - per);
  stats[dst] += s;
}

__global__ void estep_sum_kernel(const double *__restrict__ v, int64_t n, double *__restrict__ out);

// Odd joint dimension: the MFMA kernels stream X by 16-byte LDS-DMA rows, i.e. need an even row length.  The frames are
// copied with one extra dimension that is identically 0 under a unit-variance, zero-mean parameter: it adds
// (0 - 0)^2 / 1 = 0 to every distance and log 1 = 0 to every log-determinant -- the statistics of the real dimensions are
// those of the unpadded problem; its own statistics are dropped again and the extra -log(2 pi)/2 per frame is added back.
__global__ void __launch_bounds__(256)
estep_pad_x_kernel(const double *__restrict__ X, int64_t n, int dj, double *__restrict__ Xp) {
  const int djp = dj + 1;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n * djp; e += (int64_t)gridDim.x * 256) {
    const int64_t f = e / djp;
    const int d = (int)(e - f * djp);
    Xp[e] = d < dj ? X[f * dj + d] : 0.0;
  }
}
// stats_p ([S0 (M) | S1 (dj+1,M) | S2 (dj+1,M) | ll]) added into stats ([S0 | S1 (dj,M) | S2 (dj,M) | ll])
__global__ void __launch_bounds__(256)
estep_unpad_stats_kernel(const double *__restrict__ sp, int M, int dj, int64_t nframes, double *__restrict__ stats) {
  const int djp = dj + 1;
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x, per = (int64_t)M * dj;
  if (e < M) stats[e] += sp[e];
  if (e < per) {
    const int64_t m = e / dj, d = e - m * dj;
    stats[M + e] += sp[M + m * djp + d];
    stats[M + per + e] += sp[M + (int64_t)M * djp + m * djp + d];
  }
  // the padding dimension put one more -log(2 pi)/2 into every frame's log-density
  if (e == 0) stats[M + 2 * per] += sp[M + 2 * (int64_t)M * djp] + (double)nframes * (0.5 * kLog2Pi);
}

// ------------------------------------------------------------------------------------------------
// MFMA path, Dj a multiple of 16 (at Dj = 80: K = 160 = [x^2 | x], 40 k-steps; 10 statistic column tiles [x | x^2]).
// Up to Dj = 80 one kernel; Dj = 160 as two (EstepCfg::SPLIT below).
// ------------------------------------------------------------------------------------------------
template <int DJ>
struct EstepCfg {
  static constexpr int KS = 2 * DJ / 4;          // k-steps of the log-density contraction
  static constexpr int NDT = 2 * DJ / 16;        // 16-wide column tiles of the statistics [x | x^2]
  // A workgroup keeps its slice of W (M x 2 DJ) and of the statistics (M x 2 DJ) in registers: 2 x 164 KB at DJ = 80,
  // M = 128.  Beyond that the two do not fit the 512 KB register file of a CU together, and the E-step runs as TWO
  // kernels -- responsibilities (W in registers), then statistics (accumulators in registers) -- with gamma (N x 128)
  // through HBM in between: 2 x 1 KB per frame of extra traffic against 5 KB of x^2 / x operands per frame.
  static constexpr bool SPLIT = DJ > 80;
  static constexpr int FB = SPLIT ? 32 : 64;     // frames per block
  // LDS row stride of x (doubles): a multiple of 16 bytes (LDS-DMA rows) and == 18 mod 32, i.e. 36 mod 64 in dwords ->
  // step A's 16-row column reads hit distinct banks
  static constexpr int RSX = (DJ + 2 + 13) / 32 * 32 + 18;
  static constexpr int XBUF = (FB * RSX * 8 + 1023) / 1024 * 128;   // doubles per x buffer: whole 1 KB wave-instructions of the DMA
  static constexpr int MMAX = 128;               // 8 waves x 16 mixtures
  static constexpr int RSG = MMAX + 16;          // LDS row stride of gamma; == 16 mod 32 -> f-groups land 32 banks apart
  // two x buffers (block k+1 streams in by LDS-DMA while block k is processed) + l/gamma; the log-likelihood scratch of
  // the epilogue aliases the l/gamma area
  // + the 2^(j/64) table of vc_exp_tab (64 doubles) + step B's frame order: winning tile per frame (FB ints) and one
  // permutation per wave (8 x FB ints)
  // + the refinement thresholds of the 128 mixture slots (see estep_prep_kernel)
  static constexpr size_t LDS_BYTES = ((size_t)2 * XBUF + (size_t)FB * RSG + 64 + MMAX) * sizeof(double) + (size_t)9 * FB * sizeof(int);
  static_assert(RSX >= DJ + 2 && RSX % 2 == 0, "row stride");
  static_assert(LDS_BYTES <= 160 * 1024, "LDS");
};

// Wpack: [mt (8)][ks (KS)][lane (64)] A-operand fragments of W = [-iv/2 | mu*iv] (rows = mixtures), zero rows for m >= M
// cinit: [128] log-density constants c_m (-inf rows for m >= M so that their gamma is exactly 0)
// PHASE 0: the whole E-step in one kernel.  PHASE 1 / 2 (EstepCfg::SPLIT): responsibilities -> G (frames x 128, row-major)
// and the log-likelihood / statistics from G.  PHASE 3 (more than 128 mixtures: one launch per GROUP of 128): as PHASE 1,
// normalised within the group, plus the group's log-sum-exp of every frame written to `part[frame]`; the groups are
// combined by estep_group_combine_kernel and the statistics taken per group by PHASE 2.
template <int DJ, int PHASE, bool SHARE>
__global__ void __launch_bounds__(512)
estep_mfma_kernel(const double *__restrict__ X, int64_t N, int M, const double *__restrict__ Wpack,
                  const double *__restrict__ cinit, double *__restrict__ part, int64_t plen,
                  const double *__restrict__ refmu, const double *__restrict__ refiv, const double *__restrict__ refc,
                  double *__restrict__ G, int dj, int mtp_arg, unsigned long long *__restrict__ mfma_count,
                  const int64_t *__restrict__ Ndev) {
  if (Ndev) {                                // (the soft frames of estep_hard.hpp: their number exists on the device only)
    N = *Ndev;
    if (N == 0) return;                      // none: no partials either -- estep_reduce_kernel is told the same way and leaves the statistics alone
  }
  // mfma_count (optional, measurement): v_mfma_f64_16x16x4 instructions issued, counted by wave-uniform scalar adds
  // mtp = 1, 2, 4 or 8 mixture tiles of 16 (>= M / 16): with fewer than eight (SHARE), 8 / mtp waves share a tile -- wave
  // w takes tile w % mtp and every (8 / mtp)-th frame tile of step A / k-step of step B, and writes its own row of
  // partial statistics -- so that a 16- or 64-mixture model does not pay for 128 slots.  (A template flag: the branches
  // cost the full-width kernel 3.5 % when they are run-time decisions.)
  const int mtp = SHARE ? mtp_arg : 8;
  // dj <= DJ is the data's joint dimension (even: rows of X are 16-byte aligned), DJ the instantiated one: the columns
  // dj .. DJ-1 of the LDS image hold other (finite) values of X, meet zero weights in step A and statistics in step B
  // that are never written out
  using C = EstepCfg<DJ>;
  constexpr int KS = C::KS, NDT = C::NDT, FB = C::FB, RSX = C::RSX, RSG = C::RSG, XBUF = C::XBUF;
  constexpr bool kGamma = PHASE != 2, kStats = (PHASE == 0 || PHASE == 2);
  constexpr int kStepBUnroll = PHASE == 0 ? 4 : FB / 4;
  extern __shared__ double smem[];
  double *xbuf = smem;                     // [2][XBUF]: [FB][RSX] images
  double *lg = smem + 2 * XBUF;            // [FB][RSG]   l, then gamma
  double *red = lg;                        // [8][64] scratch for the log-likelihood reduction (epilogue only)
  double *etab = lg + FB * RSG;            // [64] 2^(j/64) for vc_exp_tab (fp64_exp.hpp)
  // PHASE 0: step B takes the block's frames GROUPED BY THE MIXTURE TILE THAT WINS THEM.  Its k-steps are 4 frames x the wave's
  // 16 mixtures and are skipped when all 64 responsibilities are exactly zero; in the order the frames arrive, a tile's
  // few frames are spread over most k-steps (BASELINE data: 41 % of the k-steps survive), grouped they sit in two or three.
  // A sum over frames does not depend on their order (rounding aside: the statistics stay deterministic, run to run).
  // (fkey / fperm / kSortB: below, after the lane indices)
  if (threadIdx.x < 64) etab[threadIdx.x] = kExp2Tab[threadIdx.x];       // (visible after the barrier behind the first stage)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lcol = lane & 15, lgrp = lane >> 4;
  double *tthr = etab + 64;                         // [MMAX] refinement thresholds (refc[2 m + 1]; -inf: the expanded form is always good enough)
  if (kGamma && threadIdx.x < C::MMAX) tthr[threadIdx.x] = ((int)threadIdx.x < M) ? refc[2 * threadIdx.x + 1] : -INFINITY;
  int *fkey = reinterpret_cast<int *>(tthr + C::MMAX);   // [FB] winning tile of each frame (softmax phase)
  int *fperm = fkey + FB + wave * FB;               // [FB] this wave's copy of the grouped order
  constexpr bool kSortB = (PHASE == 0);
  const int tile = SHARE ? (wave & (mtp - 1)) : wave, sub = SHARE ? wave / mtp : 0, wpt = SHARE ? 8 / mtp : 1;      // mixture tile, position among the tile's waves

  // this wave's weight fragments and log-density constants stay in registers for the whole kernel
  double wfrag[kGamma ? KS : 1];
  if constexpr (kGamma) {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) wfrag[ks] = Wpack[((size_t)tile * KS + ks) * 64 + lane];
  }
  // step A computes the TRANSPOSED tile (rows = frames, cols = this wave's mixtures) by swapping the MFMA operands:
  // the result then lands in LDS with lanes along consecutive mixtures -> conflict-free stores
  const double cm = kGamma ? cinit[16 * tile + lcol] : 0.0;
  const d4 cin = {cm, cm, cm, cm};

  d4 sacc[kStats ? NDT : 1];   // statistics tiles: rows = this wave's 16 mixtures, cols = 16 of the 2*DJ columns [x | x^2]
  double s0l = 0.0;   // sum over this lane's frames of gamma[f][m = 16 wave + lcol]
#pragma unroll
  for (int j = 0; j < (kStats ? NDT : 1); ++j) sacc[j] = d4{0, 0, 0, 0};
  double llacc = 0.0, sprod = 1.0;
  int nprod = 0;
  int nmfma = 0;

  const int64_t nblocks = (N + FB - 1) / FB;
  // ---- x staging by LDS-DMA (global_load_lds_dwordx4: 16 bytes per lane straight into LDS, no VGPRs -- the kernel has
  //      none to spare): a wave-instruction fills 1 KB of the padded [FB][RSX] image; lanes that fall into a row's
  //      16-byte pad, and frames beyond N, fetch a valid address (frame N-1 / the block start) and are never used ----
  constexpr int ROWB = RSX * 8, NCHUNK = XBUF / 128;      // (the last wave-instruction may run into the buffer's padding)
  auto stage = [&](int64_t f0, double *dst) {
    const char *base = reinterpret_cast<const char *>(X + f0 * dj);          // workgroup-uniform 64-bit base,
    const int last = (int)((N - 1 - f0 < FB - 1) ? N - 1 - f0 : FB - 1);     // 32-bit per-lane offsets
    // The per-lane (row, column) of every chunk is loop-invariant; hoisted out of the block loop it costs 9 VGPRs this
    // kernel does not have -- they were spilled and came back from scratch behind four exposed s_waitcnt vmcnt per
    // block.  The opaque copy of the lane id makes them a handful of integer instructions per block instead.
    int lane_v = lane;
    asm volatile("" : "+v"(lane_v));
#pragma unroll
    for (int i = 0; i < (NCHUNK + 7) / 8; ++i) {
      const int q = wave + 8 * i;
      if (q < NCHUNK) {                                          // wave-uniform
        const int o = 1024 * q + 16 * lane_v, row = o / ROWB, col = o - row * ROWB;
        const int rowc = row < last ? row : last;      // (also rows >= FB of the padding)
        const unsigned off = (col < dj * 8) ? (unsigned)(rowc * (dj * 8) + col) : 0u;
        const char *src = base + off;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                         (__attribute__((address_space(3))) void *)(reinterpret_cast<char *>(dst) + 1024 * q),
                                         16, 0, 0);
      }
    }
  };
  if (blockIdx.x < nblocks) stage((int64_t)blockIdx.x * FB, xbuf);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int cur = 0;
#ifdef VCMI_ESTEP_PROF
  unsigned long long pt_[6] = {0, 0, 0, 0, 0, 0};     // probe build: s_memtime counts in A | barrier | softmax | barrier | B | barrier
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
    double gpre[kGamma ? 1 : FB / 4];         // PHASE 2: this lane's responsibilities of the block, fetched before the products
    if constexpr (kGamma) {
      // ---- step A: l[m][f] = c_m + sum_k W[m][k] Xe[k][f],  Xe = [x^2 ; x] ----
#pragma unroll
      for (int ft = 0; ft < FB / 16; ++ft) {
        if constexpr (SHARE) {
          if ((ft & (wpt - 1)) != (sub & (wpt - 1)) || (wpt > FB / 16 && sub >= FB / 16)) continue;     // wave-uniform
        }
        d4 acc = cin;
        nmfma += KS;
        const double *xr = xs + (16 * ft + lcol) * RSX + lgrp;
#pragma unroll
        for (int ks = 0; ks < KS / 2; ++ks) {
          const double x = xr[4 * ks];
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x * x, wfrag[ks], acc, 0, 0, 0);
          if constexpr (C::SPLIT) {         // 160 registers hold W: keep the operand reads from running far ahead
            if ((ks & 3) == 3) __builtin_amdgcn_sched_barrier(0);
          }
        }
#pragma unroll
        for (int ks = 0; ks < KS / 2; ++ks) {
          const double x = xr[4 * ks];
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x, wfrag[KS / 2 + ks], acc, 0, 0, 0);
          if constexpr (C::SPLIT) {
            if ((ks & 3) == 3) __builtin_amdgcn_sched_barrier(0);
          }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) lg[(16 * ft + 4 * r + lgrp) * RSG + 16 * tile + lcol] = acc[r];
      }
      VCMI_PT(0)
      __syncthreads();
      VCMI_PT(1)
      // ---- softmax over the 128 mixture slots of each frame.  16 lanes per frame, lane lcol owns slots lcol + 16 i:
      //      per instruction a 32-lane group touches 2 frame rows x 16 consecutive doubles, which with RSG == 16 mod 32
      //      is conflict-free (same pattern as step B's gamma reads) ----
      // the next block streams in from here on (issued in the phase with the lowest register pressure; it has the
      // softmax and step B to land)
      if (blk + gridDim.x < nblocks) stage((blk + gridDim.x) * FB, xbuf + (cur ^ 1) * XBUF);
#pragma unroll
      for (int ps = 0; ps < FB / 32; ++ps) {
        const int f = 32 * ps + 4 * wave + lgrp;
        double *row = lg + f * RSG + lcol;
        double v[C::MMAX / 16];
        double u = -INFINITY;
#pragma unroll
        for (int i = 0; i < C::MMAX / 16; ++i) {
          v[i] = (!SHARE || i < mtp) ? row[16 * i] : -INFINITY;          // slots of tiles that do not exist: gamma = 0
          u = fmax(u, v[i]);
        }
        int wtile = 0;                        // (kSortB) the first slot of this lane that attains its maximum
        if constexpr (kSortB) {
          const double ul = u;
#pragma unroll
          for (int i = C::MMAX / 16 - 1; i >= 0; --i) wtile = (v[i] == ul) ? i : wtile;
        }
        const double ulane = u;
        u = row16_max(u);                    // (DPP: no LDS round trips; fp64_exp.hpp)
        if constexpr (kSortB) {
          wtile = row16_min((ulane == u) ? wtile : 99);
          if (lcol == 0) fkey[f] = wtile < 8 ? wtile : 0;
        }
        // ---- refinement.  The GEMM form  x^2 (-1/2var) + x (mu/var) + c  cancels: with var down to min_covar = 1e-7 and
        //      |mu| ~ 10 its terms reach 1e9 and l carries an absolute error of ~1e-7.  That is harmless while one mixture
        //      owns the frame (gamma = 1 whatever l is) and wrong when several compete: the responsibilities inherit the
        //      error.  So when more than one mixture is within kRefine of the frame's maximum, exactly those are
        //      re-evaluated term by term, (x - mu)^2 / var summed over d, as the reference formula reads --
        //      unless the expanded form is provably good enough for the (frame, mixture): l at or above the mixture's threshold
        //      (estep_prep_kernel: an error bound of 1e-10 from the value itself; models with ordinary variances, the
        //      reference's trained ones among them, never re-evaluate anything).
        //      Round 6: the TEST costs nothing extra on the way -- a compare on the exps' own operand and one against the
        //      threshold per slot; "several compete" is read off the sum (s > 1: another mixture within ~36 nats) -- and the
        //      re-evaluation, with a second softmax of the frame, sits behind one wave-uniform branch.  (Before, every frame
        //      with competing mixtures -- every frame of real joint mel-cepstra -- walked its slots through LDS one by one to
        //      find nothing: ~150 of a pass's ~400 VALU instructions, each paid for in matrix-pipe time.)
        constexpr double kRefine = 36.0;     // e^-36 = 2e-16: a mixture further below the maximum cannot change a sum
        bool needl = false;                  // one of this lane's slots is within kRefine of the maximum and below its threshold
        double s = 0.0;
#pragma unroll
        for (int i = 0; i < C::MMAX / 16; ++i) {
          const double dlt = v[i] - u;
          if (!SHARE || i < mtp) needl = needl || (dlt > -kRefine && v[i] < tthr[lcol + 16 * i]);
#if VCMI_ESTEP_EXP_SKIP
          // the 16 mixtures of slot group i are hopeless for all four frames of the pass (e^x = 0 below -745.2): no exp at all
          if (__builtin_amdgcn_ballot_w64(dlt > -745.2) == 0) {
            v[i] = 0.0;
            continue;
          }
#endif
          v[i] = vc_exp_tab(dlt, etab);   // 20 instructions against the 42 of exp(); -inf and < -745 give exactly 0
          s += v[i];
        }
        s = row16_sum(s);
        // PHASE 3 (one group of 128 of a larger model): a mixture that is alone near the top of ITS group may still compete
        // with one of another group, which this launch cannot see -- so every value within kRefine of the group's maximum is
        // made exact, the maximum itself included (a mixture within 36 nats of the frame's maximum over all groups is within
        // 36 nats of its own group's maximum): cross-group competition then sees exact log-densities too
        const bool need = needl && (PHASE == 3 || s > 1.0);
        if (__builtin_amdgcn_ballot_w64(need) != 0) {
          const double thr = u - kRefine;
          const double *xf = xs + f * RSX;
#pragma unroll 1
          for (int i = 0; i < mtp; ++i) {
            const double li = row[16 * i];           // (still the log-densities: the responsibilities are stored below)
            if (need && li > thr && li < tthr[lcol + 16 * i]) {
              const int m = lcol + 16 * i;
              const double *mp = refmu + (size_t)dj * m, *ip = refiv + (size_t)dj * m;
              double q = 0.0;
#pragma unroll 2
              for (int d = 0; d < dj; ++d) {
                const double df = xf[d] - mp[d];
                q = fma(df * df, ip[d], q);
              }
              row[16 * i] = refc[2 * m] - 0.5 * q;
            }
          }
          u = -INFINITY;
#pragma unroll
          for (int i = 0; i < C::MMAX / 16; ++i) {
            v[i] = (!SHARE || i < mtp) ? row[16 * i] : -INFINITY;
            u = fmax(u, v[i]);
          }
          u = row16_max(u);
          s = 0.0;
#pragma unroll
          for (int i = 0; i < C::MMAX / 16; ++i) {
            v[i] = vc_exp_tab(v[i] - u, etab);
            s += v[i];
          }
          s = row16_sum(s);
        }
        const bool livef = (f0 + f < N);
        // frames beyond N contribute gamma = 0; so does a slot group whose mixtures ALL have zero weight (every l = -inf:
        // u = -inf, s = 0 -- a 128-group of a padded model, M = 129 with w[129] = 0): gamma = 0 and log-sum-exp = -inf, not NaN
        // 1 / s, s in [1, 128]: the hardware's estimate and two Newton steps (none of the division's scaling / fix-up cases can occur)
        const double sc_ = s > 0.0 ? s : 1.0;
        double rq_ = __builtin_amdgcn_rcp(sc_);
        rq_ = fma(fma(-sc_, rq_, 1.0), rq_, rq_);
        rq_ = fma(fma(-sc_, rq_, 1.0), rq_, rq_);
        const double inv = (livef && s > 0.0) ? rq_ : 0.0;
#pragma unroll
        for (int i = 0; i < C::MMAX / 16; ++i)
          if (!SHARE || i < mtp) row[16 * i] = v[i] * inv;
        if constexpr (PHASE == 3) {
          if (lcol == 0 && livef) part[f0 + f] = u + log(s);      // the group's log-sum-exp of this frame
        } else {
          // log-likelihood: sum of u + log s over the frames.  log() is ~100 instructions and s lies in [1, 128], so the
          // s of a lane's frames are multiplied up (16 of them stay below 1e34) and ONE log is taken per eight blocks
          if (lcol == 0 && livef) {
            llacc += u;
            sprod *= s;
          }
        }
      }
      if constexpr (PHASE != 3) {
        if (++nprod == 8) {                   // (wave-uniform)
          llacc += log(sprod);
          sprod = 1.0;
          nprod = 0;
        }
      }
      VCMI_PT(2)
      __syncthreads();
      VCMI_PT(3)
      if constexpr (kSortB) {
        // counting sort of the block's FB frames by winning tile, every wave for itself (lane = frame; ballots + mbcnt):
        // fperm[position] = frame.  Stable, so the order -- and with it every sum -- is a function of the data alone.
        static_assert(FB == 64, "one lane per frame");
        const int k = fkey[lane];
        int base = 0, pos = 0;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          const unsigned long long bt = __builtin_amdgcn_ballot_w64(k == t);
          const int below = __builtin_amdgcn_mbcnt_hi((unsigned)(bt >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bt, 0));
          pos = (k == t) ? base + below : pos;
          base += __builtin_popcountll(bt);
        }
        fperm[pos] = lane;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the wave's own LDS writes are visible to its reads below
      }
      if constexpr (PHASE == 1 || PHASE == 3) {  // responsibilities -> G, rows of 128, coalesced
#pragma unroll
        for (int i = 0; i < FB * C::MMAX / 512; ++i) {
          const int e = tid + 512 * i, f = e / C::MMAX, m = e - f * C::MMAX;
          if (f0 + f < N) G[(f0 + f) * C::MMAX + m] = (m < 16 * mtp) ? lg[f * RSG + m] : 0.0;
        }
      }
    } else {
      if (blk + gridDim.x < nblocks) stage((blk + gridDim.x) * FB, xbuf + (cur ^ 1) * XBUF);
#pragma unroll
      for (int ks = 0; ks < FB / 4; ++ks) {
        const int64_t f = f0 + 4 * ks + lgrp;
        gpre[ks] = (f < N) ? G[f * C::MMAX + 16 * tile + lcol] : 0.0;     // frames beyond N contribute gamma = 0
      }
    }
    // ---- step B: S[m][c] += sum_f gamma[f][m] Xe[f][c],  Xe = [x | x^2];  S0[m] += sum_f gamma[f][m] ----
    if constexpr (kStats) {
#pragma unroll kStepBUnroll
      for (int ks = 0; ks < FB / 4; ++ks) {
        if constexpr (SHARE) {
          if ((ks & (wpt - 1)) != sub) continue;             // wave-uniform: this tile's waves share the k-steps
        }
        const int f = kSortB ? fperm[4 * ks + lgrp] : 4 * ks + lgrp;
        double gm;
        if constexpr (PHASE == 0) gm = lg[f * RSG + 16 * tile + lcol];
        else gm = gpre[ks];
        // responsibilities that underflowed to exactly 0 (l_m more than 745 nats under the frame's maximum: the usual case
        // for all but a few of the 128 mixtures) add exactly nothing: when that holds for the tile's 16 mixtures on all
        // four frames of the k-step, its 2 NDT/2 products are skipped (wave-uniform; the statistics are bit-identical)
        if (__builtin_amdgcn_ballot_w64(gm != 0.0) == 0) continue;
        nmfma += NDT;
        const double *xr = xs + f * RSX + lcol;
#pragma unroll
        for (int j = 0; j < NDT / 2; ++j) {
          const double x = xr[16 * j];
          sacc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(gm, x, sacc[j], 0, 0, 0);
          sacc[NDT / 2 + j] = __builtin_amdgcn_mfma_f64_16x16x4f64(gm, x * x, sacc[NDT / 2 + j], 0, 0, 0);
        }
        s0l += gm;
      }
    }
    VCMI_PT(4)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the next block's LDS-DMA has landed
    __syncthreads();
    VCMI_PT(5)
  }
#ifdef VCMI_ESTEP_PROF
  if (blockIdx.x == 3 && (tid & 63) == 0)
    printf("estep prof wave %d: A %llu | bar %llu | softmax %llu | bar %llu | B %llu | bar %llu\n", wave, pt_[0], pt_[1], pt_[2], pt_[3], pt_[4], pt_[5]);
#endif
#undef VCMI_PT

  if (mfma_count && lane == 0) atomicAdd(mfma_count, (unsigned long long)nmfma);
  // ---- write this workgroup's partial statistics: rows m = 16 wave + lgrp + 4 r, cols = 16 j + lcol ----
  // row blockIdx.x * wpt + sub of the partial statistics: every mixture tile is written by the wave (tile, sub)
  double *P = part + ((size_t)blockIdx.x * wpt + sub) * plen;
  if constexpr (kStats) {
    s0l += __shfl_xor(s0l, 16);
    s0l += __shfl_xor(s0l, 32);
    if (lgrp == 0 && 16 * tile + lcol < M) P[16 * tile + lcol] = s0l;
    if (lane == 0 && tile == 0 && sub > 0) P[plen - 1] = 0.0;        // the log-likelihood travels in row sub = 0
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
  }
  if constexpr (kGamma && PHASE != 3) {   // log-likelihood: fixed-order reduction inside the workgroup (thread 0: tile 0, sub 0)
    // (butterflies within a wave, then the eight waves in order: a fixed order; thread 0 walking 512 LDS entries took 14 us at
    // the end of every workgroup)
    llacc += log(sprod);
#pragma unroll
    for (int sh = 1; sh < 64; sh <<= 1) llacc += __shfl_xor(llacc, sh);
    if (lane == 0) red[wave] = llacc;
    __syncthreads();
    if (tid == 0) {
      double ll = 0.0;
      for (int i = 0; i < 8; ++i) ll += red[i];
      P[plen - 1] = ll;
    }
  }
}

}  // namespace vcmi
#include "estep_wave.hpp"
#include "estep_small.hpp"
#include "estep_hard.hpp"
#include "estep_path.hpp"
namespace vcmi {

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
// Pinned staging (a ring of slots) for the (tiny) model parameters of the MFMA path: the copy to the device is a real
// asynchronous copy, the host buffers outlive the call, and nothing in the call waits for the GPU -- consecutive
// E-steps queue back to back (the first version packed on the host and drained the stream before every launch:
// 0.4 ms of a 2.3 ms step).
struct EstepStaging {
  // a ring of eight: a caller issuing E-steps back to back blocks on the event of the slot it is about to reuse, and a
  // blocked host thread wakes up late on a busy machine -- with eight calls queued the GPU does not run dry meanwhile
  static constexpr int kSlots = 8;
  double *host[kSlots] = {};
  hipEvent_t copied[kSlots] = {};
  size_t cap = 0;
  int next = 0;
  int reserve(size_t n) {
    if (n <= cap) return VCMI_OK;
    release();
    for (int i = 0; i < kSlots; ++i) {
      VCMI_HIP(hipHostMalloc(reinterpret_cast<void **>(&host[i]), n * sizeof(double), hipHostMallocDefault));
      VCMI_HIP(hipEventCreateWithFlags(&copied[i], hipEventDisableTiming));
    }
    cap = n;
    return VCMI_OK;
  }
  void release() {
    for (int i = 0; i < kSlots; ++i) {
      if (host[i]) (void)hipHostFree(host[i]);
      if (copied[i]) (void)hipEventDestroy(copied[i]);
      host[i] = nullptr;
      copied[i] = nullptr;
    }
    cap = 0;
  }
  ~EstepStaging() { release(); }
};

struct EstepScratch {
  DevBuf<double> mu, iv, cst, G, LSE, part, Wpack, cinit, X, stats, raw, refiv, refc, Xpad, statsp;
  DevBuf<unsigned long long> mfma_count;      // optional measurement counter (vcmi_debug_estep_mfma); null: the kernels count nothing
  // the hard-assignment path (estep_hard.hpp, estep_path.hpp): operands, the sample's histograms, the path control words, sort
  // scratch, pieces and the soft frames' matrix
  DevBuf<unsigned char> W16;
  DevBuf<int> probe;                           // sample histograms (16 x (M + 1))
  DevBuf<int64_t> ctl;                         // kCtlLen control words (estep_path.hpp)
  DevBuf<int> hkeys;                           // key (N) | perm (N) | chunkhist (nchunks x (M + 1)) | total (M + 1)
  DevBuf<double> hpart, hllm, Xsoft;
  bool last_hard = false;                      // the last diagonal E-step of this thread launched the hard-assignment path's kernels
  EstepStaging stage;
  StreamOrder order;   // calls of one thread on different streams share the buffers above
};
static EstepScratch &scratch() {
  static thread_local EstepScratch s;
  return s;
}
// vcmi_estep_set_path: which of the two paths the diagonal E-step of this host thread takes (default: decided per call, from
// the call's own data)
static int &estep_path_choice_ref() {
  static thread_local int p = VCMI_ESTEP_AUTO;
  return p;
}
static int estep_path_choice() { return estep_path_choice_ref(); }


// model parameters: pinned host slot -> device
__global__ void __launch_bounds__(256) estep_param_copy_kernel(const double *__restrict__ src, double *__restrict__ dst, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) dst[i] = src[i];
}

// raw = [w (M) | mu (DJ,M) | var (DJ,M)] on the device -> the MFMA kernel's operands:
//   Wpack[mt][ks][lane]: A-operand fragments of W[m][k], k < DJ -> -1/(2 var) (multiplies x^2), k >= DJ -> mu/var
//   cinit[m] = log w - (DJ log 2pi + sum log var)/2 - sum mu^2/(2 var)   (-inf for m >= M and for zero weights)
template <int DJ>
__global__ void __launch_bounds__(256)
estep_prep_kernel(const double *__restrict__ raw, int M, int dj, double *__restrict__ Wpack, double *__restrict__ cinit,
                  double *__restrict__ refiv, double *__restrict__ refc) {
  using C = EstepCfg<DJ>;
  const double *w = raw, *mu = raw + M, *var = mu + (size_t)dj * M;      // dj <= DJ: the data's dimension
  const int e = blockIdx.x * 256 + threadIdx.x;
  // operands of the exact re-evaluation (estep_mfma_kernel, "refinement"): 1/var in the parameters' own (dj,M) layout and
  // the constant WITHOUT the -mu^2/(2 var) term
  if (e < M * dj) refiv[e] = 1.0 / var[e];
  if (e < 8 * C::KS * 64) {
    const int l = e & 63, ks = (e >> 6) % C::KS, mt = (e >> 6) / C::KS;
    const int m = 16 * mt + (l & 15), k = 4 * ks + (l >> 4);
    double v = 0.0;
    const int d = k < DJ ? k : k - DJ;
    if (m < M && d < dj) {                      // zero weights for the padding dimensions dj .. DJ-1
      const double ivv = 1.0 / var[d + (size_t)dj * m];
      v = k < DJ ? -0.5 * ivv : mu[d + (size_t)dj * m] * ivv;
    }
    Wpack[e] = v;
  }
  // the constants: sixteen lanes per mixture share the sums over d (log var is the expensive part: one thread per mixture
  // walking all dj dimensions, twice, took 50 us), combined by xor butterflies -- a fixed order, the same in every lane
  if (e < 16 * C::MMAX) {
    const int m = e >> 4, l = e & 15;
    double sl = 0.0, t = 0.0;
    if (m < M)
      for (int d = l; d < dj; d += 16) {
        const double vv = var[d + (size_t)dj * m], mm = mu[d + (size_t)dj * m];
        sl += log(vv);
        t += mm * mm * (1.0 / vv);
      }
#pragma unroll
    for (int sh = 8; sh >= 1; sh >>= 1) {
      sl += __shfl_xor(sl, sh);
      t += __shfl_xor(t, sh);
    }
    if (l == 0) {
      const double base = (m < M) ? (w[m] > 0.0 ? log(w[m]) : -INFINITY) - 0.5 * (dj * kLog2Pi + sl) : -INFINITY;
      if (m < M) {
        refc[2 * m] = base;                 // the constant WITHOUT the -mu^2/(2 var) term (exact re-evaluation)
        // When is the expanded form  l^ = sum_d (a_d x_d^2 + b_d x_d) + cinit  good enough?  Its rounding error is at most
        // (2 dj + 6) u T,  T = sum_d |a_d x_d^2| + |b_d x_d| + |cinit|  (u = 2^-53; the accumulation, the rounded operands, x^2).
        // With A = sum x^2/var, C = sum mu^2/var (`t` above) and q = sum (x - mu)^2/var = 2 (base - l):  sum |a x^2| = A / 2,
        // sum |b x| <= sqrt(A C), sqrt(A) <= sqrt(q) + sqrt(C)  =>  T <= (sqrt(q) + 2 sqrt(C))^2 / 2 + |base| <= q + 4 C + |base|.
        // The error stays below 1e-10 (tests: statistics to 1e-9) when q + 4 C + |base| <= Tmax = 1e-10 / ((2 dj + 6) u), i.e. when
        //   l >= base - (Tmax - 4 C - |base|) / 2   =: the threshold -- above `base` (never reached) for the tight-variance models
        // the re-evaluation exists for, far below any competing value for ordinary ones.
        const double tmax = 1e-10 / ((2.0 * dj + 6.0) * 0x1p-53);
        refc[2 * m + 1] = base > -INFINITY ? base - 0.5 * (tmax - 4.0 * t - fabs(base)) : -INFINITY;
      }
      cinit[m] = (m < M) ? base - 0.5 * t : -INFINITY;
    }
  }
}

// The MFMA path for one joint dimension (parameters staged through pinned buffers, prep kernel, the E-step kernel with one
// workgroup per CU, fixed-order reduction of the workgroups' partial statistics).
template <int DJ>
static int estep_mfma_launch(EstepScratch &sc, const double *dX, int64_t N, int dj, int M, const double *w, const double *mu,
                             const double *var, double *dstats, int64_t plen, hipStream_t st) {
  using C = EstepCfg<DJ>;
  int dev = 0, cus = 256;
  (void)hipGetDevice(&dev);
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  const int64_t nblocks = (N + C::FB - 1) / C::FB;
  const int grid = (int)std::min<int64_t>(nblocks, cus);
  const size_t nraw = (size_t)M * (1 + 2 * dj);
  VCMI_TRY(sc.raw.reserve(EstepStaging::kSlots * nraw));                  // one device copy per staging buffer
  VCMI_TRY(sc.Wpack.reserve((size_t)8 * C::KS * 64));
  VCMI_TRY(sc.cinit.reserve((size_t)C::MMAX));
  VCMI_TRY(sc.refiv.reserve((size_t)M * dj));
  VCMI_TRY(sc.refc.reserve((size_t)2 * M));
  // mixture tiles of 16, rounded up to a power of two: 8 / mtp waves share a tile and each writes its own partial row
  int mtp = 1;
  while (16 * mtp < M) mtp *= 2;
  const int wpt = 8 / mtp;
  VCMI_TRY(sc.part.reserve((size_t)grid * wpt * plen));
  VCMI_TRY(sc.stage.reserve(nraw));
  const int b = sc.stage.next;
  sc.stage.next = (sc.stage.next + 1) % EstepStaging::kSlots;
  VCMI_HIP(hipEventSynchronize(sc.stage.copied[b]));   // the copy that last used this slot (eight calls ago) is done
  double *h = sc.stage.host[b], *draw = sc.raw.p + (size_t)b * nraw;
  memcpy(h, w, sizeof(double) * M);
  memcpy(h + M, mu, sizeof(double) * M * dj);
  memcpy(h + M + (size_t)M * dj, var, sizeof(double) * M * dj);
  // (a copy kernel reading the pinned slot, not hipMemcpyAsync: the copy engine's hand-over to the compute queue costs
  // ~20 us of idle GPU per call, a kernel on the same queue a few.  Round 4 let the prep kernel read the pinned slot itself
  // and write the device copy -- one launch less -- and lost 7 us per call: its threads read every parameter three or four
  // times, across the link.)
  hipLaunchKernelGGL(estep_param_copy_kernel, dim3((unsigned)((nraw + 255) / 256)), dim3(256), 0, st, h, draw, nraw);
  VCMI_HIP(hipEventRecord(sc.stage.copied[b], st));
  hipLaunchKernelGGL(estep_prep_kernel<DJ>, dim3((8 * C::KS * 64 + 255) / 256), dim3(256), 0, st, draw, M, dj, sc.Wpack.p,
                     sc.cinit.p, sc.refiv.p, sc.refc.p);
  VCMI_HIP(hipGetLastError());
  const double *dmu = draw + M;      // (the means of the re-evaluation are the uploaded parameters themselves: raw = [w | mu (Dj,M) | var (Dj,M)])
  if constexpr (!C::SPLIT) {
    // The one-kernel E-step of `nfr` frames at Xp (their number read from *ndev on the device where that is given: the grid
    // then covers the CUs) + the fixed-order reduction of its partial statistics into dstats.  M <= 32: estep_small.hpp.
    auto soft = [&](const double *Xp, int64_t nfr, const int64_t *ndev, int accumulate, hipStream_t st) -> int {
      if (M <= 32 && !debug_flag(kDbgEstepNoSmall)) {
        auto go = [&](auto cfg, auto kern) -> int {
          using CS = decltype(cfg);
          const int g = (int)std::min<int64_t>((nfr + CS::FB - 1) / CS::FB, (int64_t)cus * CS::WG_PER_CU);
          VCMI_TRY(sc.part.reserve((size_t)g * 2 * plen));
          VCMI_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)CS::LDS_BYTES));
          hipLaunchKernelGGL(kern, dim3(g), dim3(64 * CS::NW), CS::LDS_BYTES, st, Xp, nfr, M, sc.Wpack.p, sc.cinit.p, sc.part.p, plen, dmu,
                             sc.refiv.p, sc.refc.p, dj, sc.mfma_count.p, ndev);
          VCMI_HIP(hipGetLastError());
          estep_reduce_launch(sc.part.p, g * 2, plen, dstats, st, accumulate, ndev);
          VCMI_HIP(hipGetLastError());
          return VCMI_OK;
        };
        return M <= 16 ? go(EstepSmallCfg<DJ, 1>{}, estep_small_kernel<DJ, 1>) : go(EstepSmallCfg<DJ, 2>{}, estep_small_kernel<DJ, 2>);
      }
      const int g = (int)std::min<int64_t>((nfr + C::FB - 1) / C::FB, cus);
      VCMI_TRY(sc.part.reserve((size_t)g * wpt * plen));
      auto kern = (mtp < 8) ? estep_mfma_kernel<DJ, 0, true> : estep_mfma_kernel<DJ, 0, false>;
      VCMI_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES));
      hipLaunchKernelGGL(kern, dim3(g), dim3(512), C::LDS_BYTES, st, Xp, nfr, M, sc.Wpack.p, sc.cinit.p, sc.part.p, plen, dmu,
                         sc.refiv.p, sc.refc.p, (double *)nullptr, dj, mtp, sc.mfma_count.p, ndev);
      VCMI_HIP(hipGetLastError());
      estep_reduce_launch(sc.part.p, g * wpt, plen, dstats, st, accumulate, ndev);
      VCMI_HIP(hipGetLastError());
      return VCMI_OK;
    };
    // ---- frames that one mixture owns never see an FP64 MFMA (estep_hard.hpp, estep_path.hpp); the rest goes on below, gathered ----
    static constexpr int64_t kHardMinFrames = 65536;
    // Which path?  The hard-assignment path pays where most frames have an owner and costs its pass over X on top of the whole
    // one-kernel E-step where they do not (real joint mel-cepstra: DESIGN 3.3).  Decided from THIS call's data, on the device:
    // the screen on a SAMPLE of 16 chunks spread over the frames (~20 us), estep_path_decide_kernel writes the control words,
    // every kernel of either path starts with a look at them -- the launch sequence is fixed, nothing waits for the GPU, and
    // identical inputs give identical bits whatever this thread (or any other) ran before.  vcmi_estep_set_path pins one.
    const int path = debug_flag(kDbgEstepNoHard) ? VCMI_ESTEP_SOFT : estep_path_choice();
    bool hard_on = N >= kHardMinFrames && N < ((int64_t)1 << 31) && M <= kHardMaxM && path != VCMI_ESTEP_SOFT;
    // One mixture tile (M <= 16, the size bin/train_gmm.jl defaults to): estep_small_kernel takes 0.30 ms per 1.25e6 frames
    // whatever the data, the hard-assignment path 0.42 where every frame has an owner -- nothing to decide, and the sample screen,
    // the decision and the gated launches (0.09 ms) are not paid (vcmi_estep_set_path(VCMI_ESTEP_HARD) still takes that path)
    if (path == VCMI_ESTEP_AUTO && M <= 16 && !debug_flag(kDbgEstepNoSmall)) hard_on = false;
    sc.last_hard = false;
    using CH = EstepHardCfg<DJ>;
    const int MT = (M + 15) / 16, MK = M + 1;
    const int64_t nchunks = (N + kGroupChunk - 1) / kGroupChunk;
    if (hard_on) {
      // the soft frames' dense matrix is as large as X in the worst case: a device without room for it takes the one-kernel path
      // (the reservation used to fail the whole E-step)
      if (sc.Xsoft.reserve((size_t)N * dj) != VCMI_OK) {
        (void)hipGetLastError();
        hard_on = false;
      }
    }
    if (hard_on) {
      const int64_t npmax = (N + kHardPiece - 1) / kHardPiece + M, prow = 2 * (int64_t)dj + 2;
      const int64_t nsample = std::min<int64_t>(16, nchunks), cstride = nchunks / nsample;
      VCMI_TRY(sc.W16.reserve((size_t)MT * CH::TILE_BYTES));
      constexpr int kSamplePasses = kGroupChunk / (32 * (kHardKeyThreads / 64));      // run units per sampled chunk (one pass each)
      VCMI_TRY(sc.probe.reserve((size_t)nsample * kSamplePasses * MK));
      VCMI_TRY(sc.ctl.reserve((size_t)kCtlLen));
      VCMI_TRY(sc.hkeys.reserve((size_t)2 * N + (size_t)(nchunks + 1) * MK));
      VCMI_TRY(sc.hpart.reserve((size_t)npmax * prow));
      VCMI_TRY(sc.hllm.reserve((size_t)M));
      int64_t *ctl = sc.ctl.p;
      const int64_t *gate = ctl + kCtlHard;
      int *key = sc.hkeys.p, *perm = key + N, *chunkhist = perm + N, *total = chunkhist + nchunks * MK;
      hipLaunchKernelGGL(estep_hard_prep_kernel<DJ>, dim3((unsigned)((MT * CH::NI * 64 + 255) / 256 + 4 * MT)), dim3(256), 0, st, draw, sc.cinit.p, M,
                         dj, sc.W16.p);
      const size_t kshmem = CH::lds_bytes(MT) + (size_t)MK * sizeof(int);
      auto kk = estep_hard_key_kernel<DJ>;
      VCMI_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kk), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kshmem));
      if (path == VCMI_ESTEP_AUTO)       // the screen on a sample of the call's frames -> the control words
        hipLaunchKernelGGL(kk, dim3((unsigned)(nsample * kSamplePasses)), dim3(kHardKeyThreads), kshmem, st, sc.W16.p, M, dj, dX, N, (int *)nullptr,
                           sc.probe.p, nsample * kSamplePasses, cstride, (const int64_t *)nullptr, 1);
      hipLaunchKernelGGL(estep_path_decide_kernel, dim3(1), dim3(kDecideThreads), 0, st, sc.probe.p, (int)(nsample * kSamplePasses), MK,
                         path == VCMI_ESTEP_AUTO ? -1 : 1, N, ctl);
      // The hard-assignment path proper: every kernel looks at ctl[kCtlHard] first.  (With the decision left to the device one of
      // the two chains -- these launches, or the one-kernel E-step of every frame -- returns at once, launch by launch, ~4.4 us
      // each.  Running the two chains on two streams, forked behind the decision and joined at the end, was measured in round 6:
      // the fork / join events cost more than the idle launches -- estep_fixture 0.619 against 0.608 ms, estep 0.556 against 0.526.)
      hipStream_t hs = st;
      hipLaunchKernelGGL(kk, dim3((unsigned)std::min<int64_t>(nchunks, (int64_t)cus)), dim3(kHardKeyThreads), kshmem, hs, sc.W16.p, M, dj, dX, N, key,
                         chunkhist, nchunks, (int64_t)1, gate, 0);
      hipLaunchKernelGGL(gmmmap_group_scan_kernel, dim3((unsigned)MK), dim3(256), 0, hs, chunkhist, nchunks, MK, total, gate);
      hipLaunchKernelGGL(gmmmap_group_scatter_kernel, dim3((unsigned)nchunks), dim3(256), (size_t)17 * MK * sizeof(int), hs, key, N, MK,
                         chunkhist, total, perm, gate);
      hipLaunchKernelGGL(estep_hard_stats_kernel<DJ>, dim3((unsigned)npmax), dim3(256), 0, hs, dX, dj, M, perm, total, dmu, sc.refiv.p,
                         sc.hpart.p, prow, gate);
      hipLaunchKernelGGL(estep_hard_reduce_kernel, dim3((unsigned)M), dim3(256), 0, hs, sc.hpart.p, prow, total, M, dj, sc.refc.p, dstats,
                         sc.hllm.p, gate);
      hipLaunchKernelGGL(estep_hard_ll_kernel, dim3(1), dim3(64), 0, hs, sc.hllm.p, M, dstats, plen, gate);
      hipLaunchKernelGGL(estep_hard_gather_kernel, dim3((unsigned)(cus * 4)), dim3(256), 0, hs, dX, dj, M, perm, total, sc.Xsoft.p, ctl + kCtlNSoft, N, gate);
      VCMI_HIP(hipGetLastError());
      sc.last_hard = true;
      // the soft frames through the one-kernel path, their number read on the device; its partials are added on top ...
      // (both one-kernel launches share the partial rows: only one of them has frames; reserved for the larger grid first)
      VCMI_TRY(soft(sc.Xsoft.p, /*every CU: the count is the device's*/ (int64_t)1 << 40, (const int64_t *)(ctl + kCtlNSoft), /*accumulate=*/1, hs));
      // ... or, where the sample found few owners, every frame (ctl[kCtlAllSoft] = N; 0 otherwise: both launches return at once)
      if (path == VCMI_ESTEP_AUTO) VCMI_TRY(soft(dX, N, (const int64_t *)(ctl + kCtlAllSoft), /*accumulate=*/0, st));
      return VCMI_OK;
    }
    if (mtp == 8 && debug_flag(kDbgEstepWaveKernel)) {      // (measured slower than the three-barrier kernel: 1.59 against 1.32 ms -- DESIGN 3.3 round 5)
      // more than 64 mixtures: every wave owns a tile of 16 -- the one-barrier-per-block kernel (estep_wave.hpp)
      using CW = EstepWaveCfg<DJ>;
      const int gridw = (int)std::min<int64_t>((N + CW::FB - 1) / CW::FB, cus);
      VCMI_TRY(sc.part.reserve((size_t)gridw * plen));
      auto kw = estep_wave_kernel<DJ>;
      VCMI_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kw), hipFuncAttributeMaxDynamicSharedMemorySize, (int)CW::LDS_BYTES));
      hipLaunchKernelGGL(kw, dim3(gridw), dim3(512), CW::LDS_BYTES, st, dX, N, M, sc.Wpack.p, sc.cinit.p, sc.part.p, plen, dmu, sc.refiv.p,
                         sc.refc.p, dj, sc.mfma_count.p);
      VCMI_HIP(hipGetLastError());
      estep_reduce_launch(sc.part.p, gridw, plen, dstats, st, /*accumulate=*/0);
      VCMI_HIP(hipGetLastError());
      return VCMI_OK;
    }
    VCMI_TRY(soft(dX, N, (const int64_t *)nullptr, /*accumulate=*/0, st));     // (dstats was not zeroed)
  } else {
    // two kernels per chunk of frames, the responsibilities (frames x 128 doubles) through HBM in between
    constexpr int64_t kSplitChunk = 1 << 20;
    const int64_t ch = std::min<int64_t>(N, kSplitChunk);
    VCMI_TRY(sc.G.reserve((size_t)ch * C::MMAX));
    auto kg = (mtp < 8) ? estep_mfma_kernel<DJ, 1, true> : estep_mfma_kernel<DJ, 1, false>;
    auto ks = (mtp < 8) ? estep_mfma_kernel<DJ, 2, true> : estep_mfma_kernel<DJ, 2, false>;
    VCMI_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kg), hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES));
    VCMI_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(ks), hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES));
    for (int64_t n0 = 0; n0 < N; n0 += kSplitChunk) {
      const int64_t nfr = std::min<int64_t>(kSplitChunk, N - n0);
      const int g2 = (int)std::min<int64_t>((nfr + C::FB - 1) / C::FB, cus);
      hipLaunchKernelGGL(kg, dim3(g2), dim3(512), C::LDS_BYTES, st, dX + n0 * dj, nfr, M, sc.Wpack.p, sc.cinit.p, sc.part.p, plen,
                         dmu, sc.refiv.p, sc.refc.p, sc.G.p, dj, mtp, sc.mfma_count.p, (const int64_t *)nullptr);
      hipLaunchKernelGGL(ks, dim3(g2), dim3(512), C::LDS_BYTES, st, dX + n0 * dj, nfr, M, sc.Wpack.p, sc.cinit.p, sc.part.p, plen,
                         dmu, sc.refiv.p, sc.refc.p, sc.G.p, dj, mtp, sc.mfma_count.p, (const int64_t *)nullptr);
      VCMI_HIP(hipGetLastError());
      estep_reduce_launch(sc.part.p, g2 * wpt, plen, dstats, st, /*accumulate=*/n0 > 0);
      VCMI_HIP(hipGetLastError());
    }
  }
  return VCMI_OK;
}

// More than 128 mixtures: groups of 128, each with its own operand blocks.  Per chunk of frames: PHASE 3 per group
// (responsibilities within the group -> G_g, the group's log-sum-exp per frame), estep_group_combine_kernel (frame-wise
// log-sum-exp over the groups, rescaling of G, log-likelihood), PHASE 2 per group (statistics from G_g, partials reduced
// in fixed order into the full layout).  Every log-density within 36 nats of its group's maximum is re-evaluated term by
// term (the kernel's refinement, unconditional in PHASE 3), so mixtures that compete across two groups are exact as well.
template <int DJ>
static int estep_mfma_groups_launch(EstepScratch &sc, const double *dX, int64_t N, int dj, int M, const double *w, const double *mu,
                                    const double *var, double *dstats, int64_t plen, hipStream_t st) {
  using C = EstepCfg<DJ>;
  int dev = 0, cus = 256;
  (void)hipGetDevice(&dev);
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  const int ng = (M + C::MMAX - 1) / C::MMAX;
  const int64_t chunk = std::min<int64_t>(N, (int64_t)1 << 18);
  const size_t nraw = (size_t)M * (1 + 2 * dj), wlen = (size_t)8 * C::KS * 64;
  constexpr int kCombineGrid = 1024;
  VCMI_TRY(sc.raw.reserve(EstepStaging::kSlots * nraw));
  VCMI_TRY(sc.Wpack.reserve(wlen * ng));
  VCMI_TRY(sc.cinit.reserve((size_t)C::MMAX * ng));
  VCMI_TRY(sc.refiv.reserve((size_t)M * dj));
  VCMI_TRY(sc.refc.reserve((size_t)2 * M));
  VCMI_TRY(sc.G.reserve((size_t)ng * chunk * C::MMAX));
  VCMI_TRY(sc.LSE.reserve((size_t)ng * chunk + kCombineGrid));
  VCMI_TRY(sc.part.reserve((size_t)cus * ((size_t)C::MMAX * (1 + 2 * dj) + 1)));
  VCMI_TRY(sc.stage.reserve(nraw));
  const int b = sc.stage.next;
  sc.stage.next = (sc.stage.next + 1) % EstepStaging::kSlots;
  VCMI_HIP(hipEventSynchronize(sc.stage.copied[b]));
  double *h = sc.stage.host[b], *draw = sc.raw.p + (size_t)b * nraw;
  std::vector<size_t> goff((size_t)ng + 1, 0);
  for (int g = 0; g < ng; ++g) {      // every group's [w | mu (dj,Mg) | var (dj,Mg)] contiguous
    const int m0 = g * C::MMAX, Mg = std::min(C::MMAX, M - m0);
    double *hg = h + goff[(size_t)g];
    memcpy(hg, w + m0, sizeof(double) * Mg);
    memcpy(hg + Mg, mu + (size_t)dj * m0, sizeof(double) * Mg * dj);
    memcpy(hg + Mg + (size_t)Mg * dj, var + (size_t)dj * m0, sizeof(double) * Mg * dj);
    goff[(size_t)g + 1] = goff[(size_t)g] + (size_t)Mg * (1 + 2 * dj);
  }
  // (a copy kernel reading the pinned slot, not hipMemcpyAsync: the copy engine's hand-over to the compute queue costs
  // ~20 us of idle GPU per call, a kernel on the same queue a few)
  hipLaunchKernelGGL(estep_param_copy_kernel, dim3((unsigned)((nraw + 255) / 256)), dim3(256), 0, st, h, draw, nraw);
  VCMI_HIP(hipEventRecord(sc.stage.copied[b], st));
  for (int g = 0; g < ng; ++g) {
    const int m0 = g * C::MMAX, Mg = std::min(C::MMAX, M - m0);
    hipLaunchKernelGGL(estep_prep_kernel<DJ>, dim3((unsigned)((wlen + 255) / 256)), dim3(256), 0, st, draw + goff[(size_t)g], Mg, dj,
                       sc.Wpack.p + wlen * g, sc.cinit.p + (size_t)C::MMAX * g, sc.refiv.p + (size_t)m0 * dj, sc.refc.p + 2 * m0);
  }
  VCMI_HIP(hipGetLastError());
  auto k3 = estep_mfma_kernel<DJ, 3, false>;
  auto k2 = estep_mfma_kernel<DJ, 2, false>;
  VCMI_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k3), hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES));
  VCMI_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k2), hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES));
  double *llpart = sc.LSE.p + (size_t)ng * chunk;
  for (int64_t n0 = 0; n0 < N; n0 += chunk) {
    const int64_t nfr = std::min<int64_t>(chunk, N - n0);
    const int g2 = (int)std::min<int64_t>((nfr + C::FB - 1) / C::FB, cus);
    for (int g = 0; g < ng; ++g) {
      const int m0 = g * C::MMAX, Mg = std::min(C::MMAX, M - m0);
      hipLaunchKernelGGL(k3, dim3(g2), dim3(512), C::LDS_BYTES, st, dX + n0 * dj, nfr, Mg, sc.Wpack.p + wlen * g,
                         sc.cinit.p + (size_t)C::MMAX * g, sc.LSE.p + (size_t)g * nfr, (int64_t)0, draw + goff[(size_t)g] + Mg,
                         sc.refiv.p + (size_t)m0 * dj, sc.refc.p + 2 * m0, sc.G.p + (size_t)g * chunk * C::MMAX, dj, 8, sc.mfma_count.p, (const int64_t *)nullptr);
    }
    hipLaunchKernelGGL(estep_group_combine_kernel, dim3(kCombineGrid), dim3(256), 0, st, sc.G.p, sc.LSE.p, ng, nfr,
                       (int64_t)chunk * C::MMAX, llpart);
    hipLaunchKernelGGL(estep_sum_kernel, dim3(1), dim3(256), 0, st, llpart, (int64_t)kCombineGrid, dstats + (plen - 1));
    for (int g = 0; g < ng; ++g) {
      const int m0 = g * C::MMAX, Mg = std::min(C::MMAX, M - m0);
      const int64_t plen_g = (int64_t)Mg * (1 + 2 * dj) + 1;
      hipLaunchKernelGGL(k2, dim3(g2), dim3(512), C::LDS_BYTES, st, dX + n0 * dj, nfr, Mg, sc.Wpack.p, sc.cinit.p, sc.part.p, plen_g,
                         draw, sc.refiv.p, sc.refc.p, sc.G.p + (size_t)g * chunk * C::MMAX, dj, 8, sc.mfma_count.p, (const int64_t *)nullptr);
      hipLaunchKernelGGL(estep_group_reduce_kernel, dim3((unsigned)((plen_g + 255) / 256)), dim3(256), 0, st, sc.part.p, g2, plen_g, Mg,
                         dj, m0, M, dstats);
    }
    VCMI_HIP(hipGetLastError());
  }
  return VCMI_OK;
}

static int estep_device_run(const double *dX, int64_t N, int Dj, int M, const double *w, const double *mu, const double *var,
                            double *dstats, hipStream_t st);

static int estep_device(const double *dX, int64_t N, int Dj, int M, const double *w, const double *mu, const double *var,
                        double *dstats, hipStream_t st) {
  EstepScratch &sc = scratch();
  VCMI_TRY(sc.order.enter(st));
  const int rc = estep_device_run(dX, N, Dj, M, w, mu, var, dstats, st);
  (void)sc.order.leave(st);
  return rc;
}

static int estep_device_run(const double *dX, int64_t N, int Dj, int M, const double *w, const double *mu, const double *var,
                            double *dstats, hipStream_t st) {
  if (N < 0 || Dj < 1 || M < 1) return fail(VCMI_ERR_DIM, "E-step: N=%lld Dj=%d M=%d invalid", (long long)N, Dj, M);
  if (!w || !mu || !var || !dstats || (N > 0 && !dX)) return fail(VCMI_ERR_ARG, "E-step: NULL argument");
  EstepScratch &sc = scratch();
  const int64_t plen = (int64_t)M * (1 + 2 * Dj) + 1;
  for (int m = 0; m < M; ++m)
    for (int d = 0; d < Dj; ++d)
      if (!(var[d + (size_t)Dj * m] > 0.0))
        return fail(VCMI_ERR_NOT_PD, "E-step: variance (%d,%d) is not positive", d + 1, m + 1);
  // (estep_mfma_launch's first reduction overwrites dstats: no memset -- two fill kernels -- in front of it)
  const bool overwrites = N > 0 && M <= EstepCfg<80>::MMAX && Dj % 2 == 0 && Dj <= 160 && !debug_flag(kDbgEstepGeneric);
  if (!overwrites) VCMI_HIP(hipMemsetAsync(dstats, 0, plen * sizeof(double), st));
  if (N == 0) return VCMI_OK;

  // MFMA instantiations for Dj = 32, 48, 64, 80 (one kernel) and 160 (two kernels); any even Dj up to 160 runs in the next
  // larger one with zero weights in the padding dimensions; odd Dj, Dj > 160 and M > 128 take the generic kernels below.
  if (Dj % 2 == 1 && Dj + 1 <= 160 && !debug_flag(kDbgEstepGeneric)) {
    // odd joint dimension: one zero dimension more (see estep_pad_x_kernel), in chunks that bound the padded copy
    const int djp = Dj + 1;
    const int64_t plen_p = (int64_t)M * (1 + 2 * djp) + 1, chunk = std::min<int64_t>(N, (int64_t)1 << 20);
    std::vector<double> mup((size_t)M * djp, 0.0), varp((size_t)M * djp, 1.0);
    for (int m = 0; m < M; ++m)
      for (int d = 0; d < Dj; ++d) {
        mup[(size_t)m * djp + d] = mu[d + (size_t)Dj * m];
        varp[(size_t)m * djp + d] = var[d + (size_t)Dj * m];
      }
    VCMI_TRY(sc.Xpad.reserve((size_t)chunk * djp));
    VCMI_TRY(sc.statsp.reserve((size_t)plen_p));
    for (int64_t n0 = 0; n0 < N; n0 += chunk) {
      const int64_t nfr = std::min(chunk, N - n0);
      hipLaunchKernelGGL(estep_pad_x_kernel, dim3((unsigned)std::min<int64_t>((nfr * djp + 255) / 256, 65536)), dim3(256), 0, st,
                         dX + n0 * Dj, nfr, Dj, sc.Xpad.p);
      VCMI_TRY(estep_device_run(sc.Xpad.p, nfr, djp, M, w, mup.data(), varp.data(), sc.statsp.p, st));     // (zeroes statsp first)
      hipLaunchKernelGGL(estep_unpad_stats_kernel, dim3((unsigned)(((int64_t)M * Dj + 255) / 256)), dim3(256), 0, st, sc.statsp.p, M, Dj,
                         nfr, dstats);
      VCMI_HIP(hipGetLastError());
    }
    // (mup / varp are staged into pinned memory by the inner call before it returns)
    return VCMI_OK;
  }
  if (M > EstepCfg<80>::MMAX && Dj % 2 == 0 && Dj <= 160 && !debug_flag(kDbgEstepGeneric)) {      // groups of 128 mixtures
    if (Dj <= 32) return estep_mfma_groups_launch<32>(sc, dX, N, Dj, M, w, mu, var, dstats, plen, st);
    if (Dj <= 48) return estep_mfma_groups_launch<48>(sc, dX, N, Dj, M, w, mu, var, dstats, plen, st);
    if (Dj <= 64) return estep_mfma_groups_launch<64>(sc, dX, N, Dj, M, w, mu, var, dstats, plen, st);
    if (Dj <= 80) return estep_mfma_groups_launch<80>(sc, dX, N, Dj, M, w, mu, var, dstats, plen, st);
    if (Dj <= 112) return estep_mfma_groups_launch<112>(sc, dX, N, Dj, M, w, mu, var, dstats, plen, st);
    return estep_mfma_groups_launch<160>(sc, dX, N, Dj, M, w, mu, var, dstats, plen, st);
  }
  if (M <= EstepCfg<80>::MMAX && Dj % 2 == 0 && Dj <= 160 && !debug_flag(kDbgEstepGeneric)) {
    // the smallest instantiation that holds Dj (an even Dj keeps the rows of X 16-byte aligned for the LDS-DMA)
    if (Dj <= 32) return estep_mfma_launch<32>(sc, dX, N, Dj, M, w, mu, var, dstats, plen, st);
    if (Dj <= 48) return estep_mfma_launch<48>(sc, dX, N, Dj, M, w, mu, var, dstats, plen, st);
    if (Dj <= 64) return estep_mfma_launch<64>(sc, dX, N, Dj, M, w, mu, var, dstats, plen, st);
    if (Dj <= 80) return estep_mfma_launch<80>(sc, dX, N, Dj, M, w, mu, var, dstats, plen, st);
    // (the two-kernel form beyond 80; 112 for the dimensions between -- Dj = 82 ... 112 ran in the 160-wide one: 1.5 x the work)
    if (Dj <= 112) return estep_mfma_launch<112>(sc, dX, N, Dj, M, w, mu, var, dstats, plen, st);
    return estep_mfma_launch<160>(sc, dX, N, Dj, M, w, mu, var, dstats, plen, st);
  }

  std::vector<double> hiv((size_t)M * Dj), hc(M);
  for (int m = 0; m < M; ++m) {
    double sl = 0.0;
    for (int d = 0; d < Dj; ++d) {
      const double v = var[d + (size_t)Dj * m];
      sl += std::log(v);
      hiv[(size_t)m * Dj + d] = 1.0 / v;
    }
    hc[m] = (w[m] > 0.0 ? std::log(w[m]) : -INFINITY) - 0.5 * (Dj * kLog2Pi + sl);
  }
  // generic path
  std::vector<double> hmu((size_t)M * Dj);
  for (int m = 0; m < M; ++m)
    for (int d = 0; d < Dj; ++d) hmu[(size_t)m * Dj + d] = mu[d + (size_t)Dj * m];
  const size_t shmem = (size_t)Dj * 64 * sizeof(double);
  if (shmem > 150 * 1024) return fail(VCMI_ERR_ARG, "E-step: joint dimension %d too large", Dj);
  VCMI_TRY(sc.mu.reserve(hmu.size()));
  VCMI_TRY(sc.iv.reserve(hiv.size()));
  VCMI_TRY(sc.cst.reserve(hc.size()));
  const int64_t ch = std::min<int64_t>(N, kChunk);
  const int maxseg = (int)((ch + kSeg - 1) / kSeg);
  VCMI_TRY(sc.G.reserve((size_t)ch * M));
  VCMI_TRY(sc.LSE.reserve((size_t)ch));
  VCMI_TRY(sc.part.reserve((size_t)maxseg * plen));
  VCMI_TRY(staged_upload(sc.mu.p, hmu.data(), hmu.size() * 8, st));      // (through the pinned ring: hostpipe.hpp, upload_now)
  VCMI_TRY(staged_upload(sc.iv.p, hiv.data(), hiv.size() * 8, st));      // (through the pinned ring: hostpipe.hpp, upload_now)
  VCMI_TRY(staged_upload(sc.cst.p, hc.data(), hc.size() * 8, st));      // (through the pinned ring: hostpipe.hpp, upload_now)
  VCMI_HIP(hipStreamSynchronize(st));
  if (shmem > 64 * 1024)
    VCMI_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(estep_gamma_kernel),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
  for (int64_t n0 = 0; n0 < N; n0 += kChunk) {
    const int64_t nfr = std::min<int64_t>(kChunk, N - n0);
    const int nseg = (int)((nfr + kSeg - 1) / kSeg);
    hipLaunchKernelGGL(estep_gamma_kernel, dim3((unsigned)((nfr + 63) / 64)), dim3(64), shmem, st, dX, n0, nfr, Dj, M,
                       sc.mu.p, sc.iv.p, sc.cst.p, sc.G.p, sc.LSE.p);
    hipLaunchKernelGGL(estep_stats_kernel, dim3(nseg, (M * Dj + 255) / 256), dim3(256), 0, st, dX, n0, nfr, Dj, M, sc.G.p,
                       sc.LSE.p, sc.part.p, plen);
    estep_reduce_launch(sc.part.p, nseg, plen, dstats, st);
    VCMI_HIP(hipGetLastError());
  }
  return VCMI_OK;
}


// ================================================================================================
// Full-covariance E-step -- what `gmm[:fit](dataset.X')` does per EM iteration in the reference as shipped
// (bin/train_gmm.jl:84-89 builds sklearn.mixture.GMM(covariance_type="full"); :103 runs EM).  SURVEY 8(f) rank 1.
//   l_nm = log w_m + log N(x_n; mu_m, Sigma_m)   (Cholesky whitening, the fvconvert log-density kernel, MODE 1)
//   gamma = softmax_m(l),  S0_m = sum gamma,  S1_m = sum gamma x,  S2_m = sum gamma x x',  loglik = sum_n lse_n
// Output buffer: [S0 (M) | S1 (Dj,M) | S2 (Dj,Dj,M) | loglik] (one all-reduce).  S2_m is a weighted Gram matrix:
// wave w of an 8-wave workgroup owns mixture 8*mg + w and accumulates the 15 lower 16x16 tiles of its 80x80 S2 with
// v_mfma_f64_16x16x4_f64 (A operand = gamma_f * x_f[i], B operand = x_f[j], k = 4 frames per step); the x tile
// loaded for the A operand is the same register as the B operand of the matching column tile, so a k-step costs
// Dj/16 LDS reads + Dj/16 multiplies for Dj/16*(Dj/16+1)/2 MFMAs.  S0/S1 ride along as per-lane sums of the A operands.
// Per-(mixture group, frame segment) partials are reduced in fixed order -> bit-identical run to run.
// ================================================================================================

// log-weighted densities (n,M) -> gamma in place; wave per frame (lanes across mixtures: coalesced rows), fixed grid.
// The per-frame log-sum-exp values are summed per wave in frame order, then per workgroup: lsepart[blockIdx.x].
static constexpr int kSoftmaxGrid = 2048;
__global__ void __launch_bounds__(256)
estep_full_softmax_kernel(double *__restrict__ LP, int M, int64_t n, double *__restrict__ lsepart, unsigned *__restrict__ fmask,
                          int nm) {
  // fmask (optional; mixture groups of nm, at most 32 of them): bit g of fmask[frame] = some mixture of group g has a
  // responsibility that is not exactly zero -- the statistics kernel then visits, per group, only those frames
  __shared__ double wsum[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double acc = 0.0;
  if (M <= 32) {
    // Small models (the reference's own: 16 mixtures by default, 32 in its trained ones): 8, 16 or 32 lanes per frame, so a
    // wave takes 8, 4 or 2 frames per turn instead of leaving most of its lanes idle.  The butterflies over lpf lanes give
    // the bits of the 64-lane ones (those only add the zeros of the idle lanes first).
    const int lpf = M <= 8 ? 8 : (M <= 16 ? 16 : 32), fpw = 64 / lpf, sub = lane / lpf, sl = lane % lpf;
    for (int64_t f0 = ((int64_t)blockIdx.x * 4 + wave) * fpw; f0 < n; f0 += (int64_t)gridDim.x * 4 * fpw) {
      const int64_t fr = f0 + sub;
      const bool on = fr < n && sl < M;
      double *l = LP + (fr < n ? fr : n - 1) * M;
      const double lv = on ? l[sl] : -INFINITY;
      double u = lv;
      for (int o = lpf / 2; o >= 1; o >>= 1) u = fmax(u, __shfl_xor(u, o));
      double sm = on ? exp(lv - u) : 0.0;
      for (int o = lpf / 2; o >= 1; o >>= 1) sm += __shfl_xor(sm, o);
      const double ls = u + log(sm);
      unsigned bits = 0;
      if (on) {
        const double gm = exp(lv - ls);
        l[sl] = gm;
        if (gm != 0.0) bits = 1u << ((sl / nm) & 31);
      }
      if (fmask) {
        for (int o = lpf / 2; o >= 1; o >>= 1) bits |= (unsigned)__shfl_xor((int)bits, o);
        if (sl == 0 && fr < n) fmask[fr] = bits;
      }
      if (fr < n) acc += ls;
    }
    // the sub-groups' sums in sub-group order (lanes 0, lpf, 2 lpf, ...)
    double t = 0.0;
    for (int sg = 0; sg < fpw; ++sg) t += __shfl(acc, sg * lpf);
    acc = t;
  } else
  for (int64_t fr = (int64_t)blockIdx.x * 4 + wave; fr < n; fr += (int64_t)gridDim.x * 4) {
    double *l = LP + fr * M;
    double u = -INFINITY;
    for (int m = lane; m < M; m += 64) u = fmax(u, l[m]);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) u = fmax(u, __shfl_xor(u, o));
    double sm = 0.0;
    for (int m = lane; m < M; m += 64) sm += exp(l[m] - u);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) sm += __shfl_xor(sm, o);
    const double ls = u + log(sm);
    unsigned bits = 0;
    for (int m = lane; m < M; m += 64) {
      const double gm = exp(l[m] - ls);
      l[m] = gm;
      if (gm != 0.0) bits |= 1u << ((m / nm) & 31);
    }
    if (fmask) {
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) bits |= (unsigned)__shfl_xor((int)bits, o);
      if (lane == 0) fmask[fr] = bits;
    }
    acc += ls;
  }
  if (lane == 0) wsum[wave] = acc;
  __syncthreads();
  if (threadIdx.x == 0) lsepart[blockIdx.x] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
}

// deterministic sum of n doubles into out[0] (+=): one workgroup, strided partials then a sequential tail
__global__ void __launch_bounds__(256)
estep_sum_kernel(const double *__restrict__ v, int64_t n, double *__restrict__ out) {
  __shared__ double part[256];
  double s = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += 256) s += v[i];
  part[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = out[0];
    for (int i = 0; i < 256; ++i) t += part[i];
    out[0] = t;
  }
}

// ---- frame lists of the statistics kernel (round 4).  Its workgroup (a mixture group, a frame segment) used to stage EVERY
// frame of its segment -- X travels once per mixture group, 8 x 320 MB through the L2 at M = 64 -- to find that 94 % of the
// 4-frame k-steps carry only exact zeros for its mixtures: the kernel was bound by the fetch / stash / barrier chain of the
// blocks it then skipped.  With fmask (softmax kernel) the frames of a group are listed once -- in frame order: chunk
// histograms, a prefix per group, a stable compaction by ballot / mbcnt, nothing depends on the scheduler -- and the
// statistics kernel walks its group's list.  list[g * n + pos] = frame; total[g] = length.
constexpr int kListChunk = 1024;
__global__ void __launch_bounds__(256)
estep_full_list_count_kernel(const unsigned *__restrict__ fmask, int64_t n, int G, int *__restrict__ chunkcnt) {
  __shared__ int hist[32];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid < 32) hist[tid] = 0;
  __syncthreads();
  const int64_t f0 = (int64_t)blockIdx.x * kListChunk;
  for (int i = 0; i < kListChunk / 256; ++i) {
    const int64_t fr = f0 + 64 * (wave + 4 * i) + lane;
    const unsigned b = fr < n ? fmask[fr] : 0u;
    for (int g = 0; g < G; ++g) {
      const int c = __builtin_popcountll(__builtin_amdgcn_ballot_w64((b >> g) & 1u));
      if (lane == 0 && c) atomicAdd(&hist[g], c);
    }
  }
  __syncthreads();
  if (tid < G) chunkcnt[(size_t)blockIdx.x * G + tid] = hist[tid];
}
// one workgroup per group: exclusive prefix of chunkcnt[.][g] over the chunks (in place), total[g]
__global__ void __launch_bounds__(256)
estep_full_list_scan_kernel(int *__restrict__ chunkcnt, int64_t nchunks, int G, int *__restrict__ total) {
  __shared__ int part[256];
  const int g = blockIdx.x, tid = threadIdx.x;
  const int64_t per = (nchunks + 255) / 256, lo = std::min<int64_t>(nchunks, tid * per), hi = std::min<int64_t>(nchunks, lo + per);
  int sum = 0;
  for (int64_t c = lo; c < hi; ++c) sum += chunkcnt[c * G + g];
  part[tid] = sum;
  __syncthreads();
  if (tid == 0) {
    int run = 0;
    for (int i = 0; i < 256; ++i) {
      const int v = part[i];
      part[i] = run;
      run += v;
    }
    total[g] = run;
  }
  __syncthreads();
  int run = part[tid];
  for (int64_t c = lo; c < hi; ++c) {
    const int v = chunkcnt[c * G + g];
    chunkcnt[c * G + g] = run;
    run += v;
  }
}
__global__ void __launch_bounds__(256)
estep_full_list_fill_kernel(const unsigned *__restrict__ fmask, int64_t n, int G, const int *__restrict__ chunkoff,
                            int *__restrict__ list) {
  __shared__ int rowcnt[16][32];           // frames of group g in row r of the chunk -> exclusive prefix over the rows
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t f0 = (int64_t)blockIdx.x * kListChunk;
  unsigned b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = wave + 4 * i;
    const int64_t fr = f0 + 64 * r + lane;
    b[i] = fr < n ? fmask[fr] : 0u;
    for (int g = 0; g < G; ++g) {
      const int c = __builtin_popcountll(__builtin_amdgcn_ballot_w64((b[i] >> g) & 1u));
      if (lane == 0) rowcnt[r][g] = c;
    }
  }
  __syncthreads();
  if (tid < G) {
    int run = 0;
    for (int r = 0; r < 16; ++r) {
      const int v = rowcnt[r][tid];
      rowcnt[r][tid] = run;
      run += v;
    }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = wave + 4 * i;
    const int64_t fr = f0 + 64 * r + lane;
    for (int g = 0; g < G; ++g) {
      const unsigned long long mask = __builtin_amdgcn_ballot_w64((b[i] >> g) & 1u);
      if ((b[i] >> g) & 1u) {
        const int below = __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
        list[(size_t)g * n + chunkoff[(size_t)blockIdx.x * G + g] + rowcnt[r][g] + below] = (int)fr;
      }
    }
  }
}

static constexpr int kFullFB = 32;   // frames per staged block (double-buffered in LDS)

// lower tiles of a (16 NTL)^2 symmetric matrix in row-major order: tile t -> (row tile a, column tile j <= a)
__host__ __device__ constexpr int full_tile_a(int t) {
  int a = 0;
  while ((a + 1) * (a + 2) / 2 <= t) ++a;
  return a;
}
__host__ __device__ constexpr int full_tile_j(int t) { return t - full_tile_a(t) * (full_tile_a(t) + 1) / 2; }

// the tiles [T0, T0 + NTP) as compile-time tables: row / column tile of each, and which x tiles they read as rows / columns
template <int T0, int NTP>
struct FullTileList {
  int a[NTP > 0 ? NTP : 1], j[NTP > 0 ? NTP : 1];
  unsigned rows, cols;
  constexpr FullTileList() : a{}, j{}, rows(0), cols(0) {
    for (int t = 0; t < NTP; ++t) {
      a[t] = full_tile_a(T0 + t);
      j[t] = full_tile_j(T0 + t);
      rows |= 1u << a[t];
      cols |= 1u << j[t];
    }
  }
};

template <int DJ, int PARTS>
struct FullStatsCfg {
  static constexpr int NTL = DJ / 16, NTILES = NTL * (NTL + 1) / 2;
  static constexpr int TPP = (NTILES + PARTS - 1) / PARTS;     // tiles per part (consecutive tiles: few distinct operands)
  static constexpr int NM = 8 / PARTS;                          // mixtures per workgroup
  // row stride == 16 (mod 32) doubles: the four 16-lane groups of an operand read (4 consecutive frames) then fall in
  // disjoint halves of the 64 LDS banks per half-wave
  static constexpr int RSX = (DJ % 32 == 16) ? DJ : DJ + 16;
  static constexpr size_t LDS_BYTES = ((size_t)2 * kFullFB * RSX + 2 * kFullFB * 8) * sizeof(double);
};

// The body of one wave: mixture `m`, tiles [PART TPP, (PART+1) TPP) of its S2 (and, for the last part, S0 and S1).
// Wave w of an 8-wave workgroup owns mixture NM mg + w / PARTS and part w % PARTS: at DJ = 80 one wave holds all 15 lower
// tiles of its mixture (PARTS = 1); at DJ = 160 the 55 tiles (220 accumulator registers) are shared by four waves.
template <int DJ, int PARTS, int PART>
__device__ __forceinline__ void estep_full_stats_body(const double *__restrict__ X, int64_t n0, int64_t f_begin, int64_t f_end,
                                                      int M, int mg, const double *__restrict__ G, double *__restrict__ P,
                                                      double *xs, double *gs, int dj, const int *__restrict__ lst,
                                                      unsigned long long *__restrict__ mfma_count) {
  // lst (optional): positions [f_begin, f_end) index this mixture group's frame list instead of the frames themselves
  using C = FullStatsCfg<DJ, PARTS>;
  constexpr int NTL = C::NTL, RSX = C::RSX, FB = kFullFB, NM = C::NM;
  constexpr int T0 = PART * C::TPP, T1 = (T0 + C::TPP < C::NTILES) ? T0 + C::TPP : C::NTILES, NTP = T1 - T0;
  constexpr bool kFirstMoments = PART == PARTS - 1;        // the last part has the fewest tiles: it also sums S0 and S1
  constexpr int NPF = (FB * DJ + 511) / 512;               // staged doubles per thread per block
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lcol = lane & 15, lgrp = lane >> 4;
  const int ml = wave / PARTS, m = mg * NM + ml;           // this wave's mixture (may be >= M: then gamma is staged as 0)

  d4 acc[NTP > 0 ? NTP : 1];
#pragma unroll
  for (int t = 0; t < NTP; ++t) acc[t] = d4{0, 0, 0, 0};
  double s1[NTL], s0 = 0.0;
#pragma unroll
  for (int a = 0; a < NTL; ++a) s1[a] = 0.0;
  int nmfma = 0;               // MFMAs this wave issued (measurement: mfma_count, optional)

  double pf[NPF], pg = 0.0;
  const int gf = tid / NM, gq = tid % NM;                 // gamma staging: FB frames x NM mixtures
  const int gm_idx = mg * NM + gq;
  // element i of this thread: row (frame of the block) and column of the staged image (dj <= DJ is the data's dimension;
  // the columns dj .. DJ-1 of the LDS image are never written: they only reach accumulators that are not stored)
  int rowi[NPF], coli[NPF];
#pragma unroll
  for (int i = 0; i < NPF; ++i) {
    const int e = tid + 512 * i;
    rowi[i] = e / dj;
    coli[i] = e - rowi[i] * dj;
  }
  // the frames of the block that is fetched NEXT (list mode: read one block ahead of the rows they address)
  int fidx[NPF], gfidx = 0;                               // (a call's chunk has at most 2^20 frames)
  auto load_idx = [&](int64_t fb) {
#pragma unroll
    for (int i = 0; i < NPF; ++i) {
      const int64_t pos = fb + rowi[i];
      fidx[i] = (lst && rowi[i] < FB && pos < f_end) ? lst[pos] : (int)pos;
    }
    gfidx = (lst && tid < FB * NM && fb + gf < f_end) ? lst[fb + gf] : (int)(fb + gf);
  };
  auto fetch = [&](int64_t fb) {                           // global -> registers
#pragma unroll
    for (int i = 0; i < NPF; ++i)
      pf[i] = (rowi[i] < FB && fb + rowi[i] < f_end) ? X[(n0 + (int64_t)fidx[i]) * dj + coli[i]] : 0.0;
    pg = (tid < FB * NM && fb + gf < f_end && gm_idx < M) ? G[(int64_t)gfidx * M + gm_idx] : 0.0;
  };
  auto stash = [&](int buf) {                              // registers -> LDS
#pragma unroll
    for (int i = 0; i < NPF; ++i)
      if (rowi[i] < FB) xs[buf * FB * RSX + rowi[i] * RSX + coli[i]] = pf[i];
    if (tid < FB * NM) gs[buf * FB * 8 + gf * NM + gq] = pg;
  };
  constexpr FullTileList<T0, NTP> TL{};                   // which x tiles this part reads, which of them it scales by gamma

  if (dj < DJ) {                                          // padding columns: finite values (they are multiplied, never stored)
    for (int e = tid; e < 2 * FB * RSX; e += 512) xs[e] = 0.0;
    __syncthreads();
  }
  if (f_begin < f_end) {
    load_idx(f_begin);
    fetch(f_begin);
    stash(0);
    load_idx(f_begin + FB);
  }
  __syncthreads();
  int buf = 0;
  for (int64_t fb = f_begin; fb < f_end; fb += FB, buf ^= 1) {
    const bool more = fb + FB < f_end;
    if (more) {
      fetch(fb + FB);
      load_idx(fb + 2 * FB);
    }
    const double *xb = xs + buf * FB * RSX, *gb = gs + buf * FB * 8;
#pragma unroll 2
    for (int ks = 0; ks < FB / 4; ++ks) {
      const int f = 4 * ks + lgrp;
      const double gm = gb[f * NM + ml];
      // the four frames of the k-step all have gamma == 0 exactly for this wave's mixture (l_m more than 745 nats under
      // the frame's maximum): the products add exactly nothing -- skipped (wave-uniform; bit-identical statistics)
      if (__builtin_amdgcn_ballot_w64(gm != 0.0) == 0) continue;
      nmfma += NTP;
      const double *xr = xb + f * RSX + lcol;
      double xv[NTL], ax[NTL];
#pragma unroll
      for (int a = 0; a < NTL; ++a) {
        if (kFirstMoments || (((TL.rows | TL.cols) >> a) & 1u)) xv[a] = xr[16 * a];
        if (kFirstMoments || ((TL.rows >> a) & 1u)) ax[a] = gm * xv[a];
        if (kFirstMoments) s1[a] += ax[a];
      }
      if (kFirstMoments) s0 += gm;
#pragma unroll
      for (int t = 0; t < NTP; ++t)
        acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(ax[TL.a[t]], xv[TL.j[t]], acc[t], 0, 0, 0);
    }
    if (more) stash(buf ^ 1);
    __syncthreads();
  }
  if (mfma_count && lane == 0) atomicAdd(mfma_count, (unsigned long long)nmfma);
  if (m >= M) return;
  // partial statistics of this (mixture, segment) in the final layout [S0 | S1 | S2 | loglik]
  if (kFirstMoments) {
    s0 += __shfl_xor(s0, 16);
    s0 += __shfl_xor(s0, 32);
    if (lane == 0) P[m] = s0;
#pragma unroll
    for (int a = 0; a < NTL; ++a) {
      double v = s1[a];
      v += __shfl_xor(v, 16);
      v += __shfl_xor(v, 32);
      if (lgrp == 0 && 16 * a + lcol < dj) P[M + (size_t)m * dj + 16 * a + lcol] = v;
    }
  }
  double *S2 = P + M + (size_t)M * dj + (size_t)m * dj * dj;      // (dj,dj) column-major
#pragma unroll
  for (int t = 0; t < NTP; ++t) {
    const int a = TL.a[t], j = TL.j[t];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = 16 * a + lgrp + 4 * r, jc = 16 * j + lcol;    // D[i][jc]
      if ((a != j || i >= jc) && i < dj && jc < dj) {              // diagonal tiles: lower part only, then mirrored
        S2[i + (size_t)dj * jc] = acc[t][r];
        S2[jc + (size_t)dj * i] = acc[t][r];
      }
    }
  }
}

template <int DJ, int PARTS>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
estep_full_stats_kernel(const double *__restrict__ X, int64_t n0, int64_t n, int M, const double *__restrict__ G,
                        double *__restrict__ part, int64_t plen, int dj, const int *__restrict__ lists,
                        const int *__restrict__ totals, unsigned long long *__restrict__ mfma_count) {
  static_assert(DJ % 16 == 0, "full-covariance MFMA statistics need Dj to be a multiple of 16");
  static_assert(PARTS == 1 || PARTS == 2 || PARTS == 4, "waves per mixture");
  using C = FullStatsCfg<DJ, PARTS>;
  extern __shared__ double fsm[];
  double *xs = fsm;                                     // [2][FB][RSX]
  double *gs = fsm + 2 * kFullFB * C::RSX;              // [2][FB][8]
  // grid = (frame segments, mixture groups): consecutive workgroups go to consecutive XCDs, so with the segment as the
  // FAST index the mixture groups that read one segment of X share an XCD (when the segment count is a multiple of 8)
  // and X reaches that L2 once instead of once per mixture group
  const int mg = blockIdx.y, seg = blockIdx.x, nsegs = gridDim.x;
  // with frame lists the segment is a range of POSITIONS in this mixture group's list (its length is on the device)
  const int64_t len = lists ? (int64_t)totals[mg] : n;
  const int *lst = lists ? lists + (size_t)mg * n : nullptr;
  const int64_t seglen = (len + nsegs - 1) / nsegs;
  const int64_t f_begin = std::min<int64_t>(len, seg * seglen), f_end = (f_begin + seglen < len) ? f_begin + seglen : len;
  double *P = part + (size_t)seg * plen;
  const int prt = (threadIdx.x >> 6) % PARTS;          // wave-uniform; every branch runs the same number of barriers
  if (PARTS == 1 || prt == 0) estep_full_stats_body<DJ, PARTS, 0>(X, n0, f_begin, f_end, M, mg, G, P, xs, gs, dj, lst, mfma_count);
  else if (PARTS == 2 || prt == 1) estep_full_stats_body<DJ, PARTS, 1>(X, n0, f_begin, f_end, M, mg, G, P, xs, gs, dj, lst, mfma_count);
  else if (prt == 2) estep_full_stats_body<DJ, PARTS, (PARTS > 2 ? 2 : 0)>(X, n0, f_begin, f_end, M, mg, G, P, xs, gs, dj, lst, mfma_count);
  else estep_full_stats_body<DJ, PARTS, (PARTS > 3 ? 3 : 0)>(X, n0, f_begin, f_end, M, mg, G, P, xs, gs, dj, lst, mfma_count);
}

// generic statistics (any Dj): thread per lower-triangle element of one mixture's S2 (+ S1, S0), sequential over the
// frames of one segment
__global__ void __launch_bounds__(256)
estep_full_stats_generic_kernel(const double *__restrict__ X, int64_t n0, int64_t n, int Dj, int M,
                                const double *__restrict__ G, double *__restrict__ part, int64_t plen) {
  const int m = blockIdx.x;
  const int64_t seglen = (n + gridDim.y - 1) / gridDim.y;
  const int64_t f_begin = blockIdx.y * seglen, f_end = (f_begin + seglen < n) ? f_begin + seglen : n;
  double *P = part + (size_t)blockIdx.y * plen;
  double *S2 = P + M + (size_t)M * Dj + (size_t)m * Dj * Dj;
  const int ntri = Dj * (Dj + 1) / 2;
  for (int e = threadIdx.x; e < ntri + Dj + 1; e += 256) {
    double s = 0.0;
    if (e < ntri) {
      int i = (int)((sqrt(8.0 * e + 1.0) - 1.0) / 2.0);
      while (i * (i + 1) / 2 > e) --i;
      while ((i + 1) * (i + 2) / 2 <= e) ++i;
      const int j = e - i * (i + 1) / 2;
      for (int64_t f = f_begin; f < f_end; ++f) s = fma(G[f * M + m] * X[(n0 + f) * Dj + i], X[(n0 + f) * Dj + j], s);
      S2[i + (size_t)Dj * j] = s;
      S2[j + (size_t)Dj * i] = s;
    } else if (e < ntri + Dj) {
      const int d = e - ntri;
      for (int64_t f = f_begin; f < f_end; ++f) s = fma(G[f * M + m], X[(n0 + f) * Dj + d], s);
      P[M + (size_t)m * Dj + d] = s;
    } else {
      for (int64_t f = f_begin; f < f_end; ++f) s += G[f * M + m];
      P[m] = s;
    }
  }
}

struct EstepFullScratch {
  DevBuf<double> LP, lse, part, X, stats, params;
  DevBuf<int> flag;
  DevBuf<unsigned long long> mfma_count;   // optional measurement counter of the statistics kernel (vcmi_debug_estep_full_mfma)
  DevBuf<int> lists;        // frame lists of the statistics kernel: [fmask (n) | chunk counts (nchunks, G) | totals (G) | lists (G, n)]
  vcmi_gmmmap *px = nullptr;
  StreamOrder order;   // calls of one thread on different streams share the buffers above
  ~EstepFullScratch() { delete px; }
};
static EstepFullScratch &full_scratch() {
  static thread_local EstepFullScratch s;
  return s;
}

// statistics of N device-resident frames under the prepared p(x) handle -> dstats (zeroed here); asynchronous on st
static int estep_full_core_run(vcmi_gmmmap *px, const double *dX, int64_t N, int Dj, int M, double *dstats, hipStream_t st);

static int estep_full_core(vcmi_gmmmap *px, const double *dX, int64_t N, int Dj, int M, double *dstats, hipStream_t st) {
  EstepFullScratch &sc = full_scratch();
  VCMI_TRY(sc.order.enter(st));
  const int rc = estep_full_core_run(px, dX, N, Dj, M, dstats, st);
  (void)sc.order.leave(st);
  return rc;
}

static int estep_full_core_run(vcmi_gmmmap *px, const double *dX, int64_t N, int Dj, int M, double *dstats, hipStream_t st) {
  const int64_t plen = (int64_t)M * (1 + Dj + (int64_t)Dj * Dj) + 1;
  VCMI_HIP(hipMemsetAsync(dstats, 0, plen * sizeof(double), st));
  if (N == 0) return VCMI_OK;
  EstepFullScratch &sc = full_scratch();
  const int64_t chunk = std::min<int64_t>(N, (int64_t)1 << 20);
  VCMI_TRY(sc.LP.reserve((size_t)chunk * M));
  VCMI_TRY(sc.lse.reserve((size_t)kSoftmaxGrid));
  // frame segments (grid.x): one 8-wave workgroup per CU in a single round, whatever the mixture count; a workgroup
  // holds 8 mixtures up to Dj = 80 and 2 (four waves per mixture) beyond
  // MFMA statistics for every Dj <= 160, in the smallest of the instantiations 32, 48, 64, 80 (one wave per mixture) and
  // 96, 128, 160 (four) that holds it
  const bool mfma = Dj <= 160 && !debug_flag(kDbgEstepGeneric);
  const int nm = (!mfma || Dj <= 80) ? 8 : 2;
  const int mgroups = (M + nm - 1) / nm;
  const int nseg = std::max(1, (256 + mgroups - 1) / mgroups);
  VCMI_TRY(sc.part.reserve((size_t)nseg * plen));
  for (int64_t n0 = 0; n0 < N; n0 += chunk) {
    const int64_t n = std::min<int64_t>(chunk, N - n0);
    VCMI_TRY(gmmmap_logdens_device(px, dX + n0 * Dj, Dj, n, sc.LP.p, st));
    // frame lists per mixture group (MFMA statistics, at most 32 groups, enough frames to matter)
    const bool use_lists = mfma && mgroups <= 32 && n >= 4096 && !debug_flag(kDbgEstepFullNoLists);
    unsigned *fmask = nullptr;
    int *chunkcnt = nullptr, *totals = nullptr, *lists = nullptr;
    const int64_t nlc = (n + kListChunk - 1) / kListChunk;
    if (use_lists) {
      VCMI_TRY(sc.lists.reserve((size_t)n + (size_t)nlc * mgroups + mgroups + (size_t)mgroups * n));
      fmask = reinterpret_cast<unsigned *>(sc.lists.p);
      chunkcnt = sc.lists.p + n;
      totals = chunkcnt + (size_t)nlc * mgroups;
      lists = totals + mgroups;
    }
    hipLaunchKernelGGL(estep_full_softmax_kernel, dim3(kSoftmaxGrid), dim3(256), 0, st, sc.LP.p, M, n, sc.lse.p, fmask, nm);
    hipLaunchKernelGGL(estep_sum_kernel, dim3(1), dim3(256), 0, st, sc.lse.p, (int64_t)kSoftmaxGrid, dstats + (plen - 1));
    if (use_lists) {
      hipLaunchKernelGGL(estep_full_list_count_kernel, dim3((unsigned)nlc), dim3(256), 0, st, fmask, n, mgroups, chunkcnt);
      hipLaunchKernelGGL(estep_full_list_scan_kernel, dim3((unsigned)mgroups), dim3(256), 0, st, chunkcnt, nlc, mgroups, totals);
      hipLaunchKernelGGL(estep_full_list_fill_kernel, dim3((unsigned)nlc), dim3(256), 0, st, fmask, n, mgroups, chunkcnt, lists);
    }
    VCMI_HIP(hipMemsetAsync(sc.part.p, 0, (size_t)nseg * plen * sizeof(double), st));
    const dim3 grid(nseg, mgroups);
    if (mfma) {
      auto launch = [&](auto kern, size_t lds) -> int {
        static std::atomic<bool> attr_done[64];           // per instantiation (the lambda is generic) and per device
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (!attr_done[dev & 63].load(std::memory_order_acquire)) {
          VCMI_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
          attr_done[dev & 63].store(true, std::memory_order_release);
        }
        hipLaunchKernelGGL(kern, grid, dim3(512), lds, st, dX, n0, n, M, sc.LP.p, sc.part.p, plen, Dj, (const int *)lists,
                           (const int *)totals, sc.mfma_count.p);
        return VCMI_OK;
      };
      if (Dj <= 32) VCMI_TRY(launch(estep_full_stats_kernel<32, 1>, FullStatsCfg<32, 1>::LDS_BYTES));
      else if (Dj <= 48) VCMI_TRY(launch(estep_full_stats_kernel<48, 1>, FullStatsCfg<48, 1>::LDS_BYTES));
      else if (Dj <= 64) VCMI_TRY(launch(estep_full_stats_kernel<64, 1>, FullStatsCfg<64, 1>::LDS_BYTES));
      else if (Dj <= 80) VCMI_TRY(launch(estep_full_stats_kernel<80, 1>, FullStatsCfg<80, 1>::LDS_BYTES));
      else if (Dj <= 96) VCMI_TRY(launch(estep_full_stats_kernel<96, 4>, FullStatsCfg<96, 4>::LDS_BYTES));
      else if (Dj <= 128) VCMI_TRY(launch(estep_full_stats_kernel<128, 4>, FullStatsCfg<128, 4>::LDS_BYTES));
      else VCMI_TRY(launch(estep_full_stats_kernel<160, 4>, FullStatsCfg<160, 4>::LDS_BYTES));
    } else {
      hipLaunchKernelGGL(estep_full_stats_generic_kernel, dim3(M, nseg), dim3(256), 0, st, dX, n0, n, Dj, M, sc.LP.p,
                         sc.part.p, plen);
    }
    // the loglik slot of the partial rows is zero, so the generic reduction leaves dstats[plen-1] (set above) intact
    estep_reduce_launch(sc.part.p, nseg, plen, dstats, st);
    VCMI_HIP(hipGetLastError());
  }
  return VCMI_OK;
}

static int read_pd_flag(const int *d_flag, hipStream_t st) {
  int h = 0;
  VCMI_HIP(hipMemcpyAsync(&h, d_flag, sizeof(int), hipMemcpyDeviceToHost, st));
  VCMI_HIP(hipStreamSynchronize(st));
  if (h) return fail(VCMI_ERR_NOT_PD, "covariance of mixture %d is not positive definite", h);
  return VCMI_OK;
}

// one E-step from HOST parameters: upload (Dj*Dj*M doubles), Cholesky whitening of every mixture on the device
// (px_prep_kernel; host fallback for very large Dj), statistics, then the stream is drained (the p(x) handle and the
// parameter staging buffers persist per host thread and are rewritten by the next call).
static int estep_full_device(const double *dX, int64_t N, int Dj, int M, const double *w, const double *mu,
                             const double *sigma, double *dstats, hipStream_t st) {
  if (N < 0 || Dj < 1 || M < 1) return fail(VCMI_ERR_DIM, "E-step: N=%lld Dj=%d M=%d invalid", (long long)N, Dj, M);
  if (!w || !mu || !sigma || !dstats || (N > 0 && !dX)) return fail(VCMI_ERR_ARG, "E-step: NULL argument");
  EstepFullScratch &sc = full_scratch();
  if (N == 0) return estep_full_core(nullptr, dX, 0, Dj, M, dstats, st);
  VCMI_TRY(sc.order.enter(st));   // the parameter staging and the p(x) handle are rewritten before the core runs
  if (gmm_px_device_prepare_supported(Dj)) {
    const size_t dd = (size_t)Dj * Dj;
    VCMI_TRY(sc.params.reserve((size_t)M * (1 + Dj + dd)));
    VCMI_TRY(sc.flag.reserve(1));
    double *dw = sc.params.p, *dmu = dw + M, *dsig = dmu + (size_t)M * Dj;
    VCMI_TRY(staged_upload(dw, w, sizeof(double) * M, st));      // (through the pinned ring: hostpipe.hpp, upload_now)
    VCMI_TRY(staged_upload(dmu, mu, sizeof(double) * M * Dj, st));      // (through the pinned ring: hostpipe.hpp, upload_now)
    VCMI_TRY(staged_upload(dsig, sigma, sizeof(double) * M * dd, st));      // (through the pinned ring: hostpipe.hpp, upload_now)
    VCMI_HIP(hipMemsetAsync(sc.flag.p, 0, sizeof(int), st));
    VCMI_TRY(gmm_px_prepare_device(&sc.px, dw, dmu, dsig, Dj, M, sc.flag.p, st));
    VCMI_TRY(estep_full_core(sc.px, dX, N, Dj, M, dstats, st));
    return read_pd_flag(sc.flag.p, st);
  }
  VCMI_TRY(gmm_px_create(w, mu, sigma, Dj, M, &sc.px));
  VCMI_TRY(estep_full_core(sc.px, dX, N, Dj, M, dstats, st));
  VCMI_HIP(hipStreamSynchronize(st));
  return VCMI_OK;
}

// ------------------------------------------------------------------------------------------------
// EM state resident on the device (bin/train_gmm.jl:84-103: sklearn.mixture.GMM(covariance_type="full",
// min_covar).fit): parameters, statistics and whitening blocks stay in HBM; one iteration is
//   estep (local statistics) -> [caller all-reduces the statistics buffer over RCCL] -> mstep (+ whitening prep).
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
em_mstep_full_kernel(const double *__restrict__ stats, int Dj, int M, double min_covar, double *__restrict__ w,
                     double *__restrict__ mu, double *__restrict__ sigma) {
  __shared__ double red[256];
  __shared__ double mus[256];
  const int tid = threadIdx.x, m = blockIdx.x;
  const double eps = 2.220446049250313e-16;
  double t = 0.0;
  for (int k = tid; k < M; k += 256) t += stats[k];
  red[tid] = t;
  __syncthreads();
  for (int o = 128; o >= 1; o >>= 1) {
    if (tid < o) red[tid] += red[tid + o];
    __syncthreads();
  }
  const double tot = red[0], s0 = stats[m];
  const double inv = 1.0 / (s0 + 10 * eps);
  const double *S1 = stats + M + (size_t)m * Dj;
  const double *S2 = stats + M + (size_t)M * Dj + (size_t)m * Dj * Dj;
  for (int d = tid; d < Dj; d += 256) {
    const double v = S1[d] * inv;
    mus[d] = v;
    mu[(size_t)m * Dj + d] = v;
  }
  if (tid == 0) w[m] = s0 / (tot + 10 * eps) + eps;
  __syncthreads();
  for (int e = tid; e < Dj * Dj; e += 256) {
    const int c = e / Dj, r = e - c * Dj;
    sigma[(size_t)m * Dj * Dj + e] = S2[e] * inv - mus[r] * mus[c] + (r == c ? min_covar : 0.0);
  }
}

}  // namespace vcmi

struct vcmi_gmm_em {
  int Dj = 0, M = 0, device = 0;
  double min_covar = 0.0;
  bool prepared = false;
  vcmi::DevBuf<double> params;   // [w (M) | mu (Dj,M) | sigma (Dj,Dj,M)]
  vcmi::DevBuf<int> flag;
  vcmi_gmmmap *px = nullptr;
  ~vcmi_gmm_em() { delete px; }
  double *w() { return params.p; }
  double *mu() { return params.p + M; }
  double *sigma() { return params.p + M + (size_t)M * Dj; }
  int64_t plen() const { return (int64_t)M * (1 + Dj + (int64_t)Dj * Dj) + 1; }
};

namespace vcmi {
static int em_prepare(vcmi_gmm_em *h, hipStream_t st) {
  if (gmm_px_device_prepare_supported(h->Dj)) {
    VCMI_TRY(gmm_px_prepare_device(&h->px, h->w(), h->mu(), h->sigma(), h->Dj, h->M, h->flag.p, st));
  } else {
    // dimensions without a device preparation (Dj > 160, and 98 < Dj whose padded size has an MFMA instantiation --
    // there is none today): Cholesky on the host
    const size_t dd = (size_t)h->Dj * h->Dj;
    std::vector<double> hw(h->M), hmu((size_t)h->M * h->Dj), hs((size_t)h->M * dd);
    VCMI_HIP(hipStreamSynchronize(st));
    VCMI_HIP(hipMemcpy(hw.data(), h->w(), hw.size() * 8, hipMemcpyDeviceToHost));
    VCMI_HIP(hipMemcpy(hmu.data(), h->mu(), hmu.size() * 8, hipMemcpyDeviceToHost));
    VCMI_HIP(hipMemcpy(hs.data(), h->sigma(), hs.size() * 8, hipMemcpyDeviceToHost));
    VCMI_TRY(gmm_px_create(hw.data(), hmu.data(), hs.data(), h->Dj, h->M, &h->px));
  }
  h->prepared = true;
  return VCMI_OK;
}
}  // namespace vcmi

using namespace vcmi;

extern "C" int64_t vcmi_estep_stats_len(int Dj, int M) { return (int64_t)M * (1 + 2 * Dj) + 1; }

extern "C" int vcmi_estep_diag_dev(const double *dX, int64_t N, int Dj, int M, const double *w, const double *mu,
                                   const double *var, double *dstats, void *stream) {
  return estep_device(dX, N, Dj, M, w, mu, var, dstats, as_stream(stream));
}

// Host-pointer E-step shared by the diagonal and the full-covariance entry points: X goes up through the pinned staging
// ring; with a device group (vcmi_set_devices) member i takes the contiguous frame block [lo, hi), computes its local
// statistics, and ONE ncclAllReduce(sum) of the packed buffer over RCCL leaves the global statistics on every member
//  -- member 0 returns them.  Two group phases, so that a member that failed never leaves the others
// waiting inside the collective.
template <class Scratch, class DevFn>
static int estep_host(Scratch &(*get_scratch)(), const double *X, int64_t N, int Dj, int64_t plen, const DevFn &dev_fn,
                      std::vector<double> &h) {
  h.assign((size_t)plen, 0.0);
  auto local = [&](int64_t lo, int64_t hi) -> int {
    Scratch &sc = get_scratch();
    const int64_t n = hi - lo;
    VCMI_TRY(sc.X.reserve((size_t)std::max<int64_t>(n, 1) * Dj));
    VCMI_TRY(sc.stats.reserve((size_t)plen));
    if (n > 0) VCMI_TRY(staged_upload(sc.X.p, X + (size_t)lo * Dj, (size_t)n * Dj * 8, nullptr));
    VCMI_TRY(dev_fn(sc.X.p, n, sc.stats.p));
    VCMI_HIP(hipStreamSynchronize(nullptr));
    return VCMI_OK;
  };
  const int m = group_size();
  if (m == 0) {
    VCMI_TRY(check_device());
    VCMI_TRY(local(0, N));
    VCMI_HIP(hipMemcpy(h.data(), get_scratch().stats.p, (size_t)plen * 8, hipMemcpyDeviceToHost));
    return VCMI_OK;
  }
  VCMI_TRY(group_run(m, [&](int i) -> int {
    int64_t lo, hi;
    shard_range(N, i, m, &lo, &hi);
    return local(lo, hi);
  }));
  return group_run(m, [&](int i) -> int {
    Scratch &sc = get_scratch();
    VCMI_TRY(group_allreduce_sum(i, sc.stats.p, (size_t)plen, nullptr));
    if (i == 0) VCMI_HIP(hipMemcpy(h.data(), sc.stats.p, (size_t)plen * 8, hipMemcpyDeviceToHost));
    return VCMI_OK;
  });
}

extern "C" int vcmi_estep_diag(const double *X, int64_t N, int Dj, int M, const double *w, const double *mu,
                               const double *var, double *S0, double *S1, double *S2, double *loglik) {
  if (!S0 || !S1 || !S2 || !loglik) return fail(VCMI_ERR_ARG, "vcmi_estep_diag: NULL output");
  if (N < 0 || Dj < 1 || M < 1) return fail(VCMI_ERR_DIM, "E-step: N=%lld Dj=%d M=%d invalid", (long long)N, Dj, M);
  if (N > 0 && !X) return fail(VCMI_ERR_ARG, "vcmi_estep_diag: NULL frames");
  const int64_t plen = vcmi_estep_stats_len(Dj, M);
  std::vector<double> h;
  VCMI_TRY(estep_host<EstepScratch>(scratch, X, N, Dj, plen, [&](const double *dX, int64_t n, double *dstats) -> int {
    return estep_device(dX, n, Dj, M, w, mu, var, dstats, nullptr);
  }, h));
  memcpy(S0, h.data(), sizeof(double) * M);
  memcpy(S1, h.data() + M, sizeof(double) * M * Dj);
  memcpy(S2, h.data() + M + (size_t)M * Dj, sizeof(double) * M * Dj);
  *loglik = h[(size_t)plen - 1];
  return VCMI_OK;
}

extern "C" int64_t vcmi_estep_full_stats_len(int Dj, int M) { return (int64_t)M * (1 + Dj + (int64_t)Dj * Dj) + 1; }

extern "C" int vcmi_estep_full_dev(const double *dX, int64_t N, int Dj, int M, const double *w, const double *mu,
                                   const double *sigma, double *dstats, void *stream) {
  return estep_full_device(dX, N, Dj, M, w, mu, sigma, dstats, as_stream(stream));
}

extern "C" int vcmi_estep_full(const double *X, int64_t N, int Dj, int M, const double *w, const double *mu,
                               const double *sigma, double *S0, double *S1, double *S2, double *loglik) {
  if (!S0 || !S1 || !S2 || !loglik) return fail(VCMI_ERR_ARG, "vcmi_estep_full: NULL output");
  if (N < 0 || Dj < 1 || M < 1) return fail(VCMI_ERR_DIM, "E-step: N=%lld Dj=%d M=%d invalid", (long long)N, Dj, M);
  if (N > 0 && !X) return fail(VCMI_ERR_ARG, "vcmi_estep_full: NULL frames");
  const int64_t plen = vcmi_estep_full_stats_len(Dj, M);
  std::vector<double> h;
  VCMI_TRY(estep_host<EstepFullScratch>(full_scratch, X, N, Dj, plen, [&](const double *dX, int64_t n, double *dstats) -> int {
    return estep_full_device(dX, n, Dj, M, w, mu, sigma, dstats, nullptr);
  }, h));
  memcpy(S0, h.data(), sizeof(double) * M);
  memcpy(S1, h.data() + M, sizeof(double) * M * Dj);
  memcpy(S2, h.data() + M + (size_t)M * Dj, sizeof(double) * M * Dj * Dj);
  *loglik = h[(size_t)plen - 1];
  return VCMI_OK;
}

// ---- device-resident EM state --------------------------------------------------------------------
extern "C" int vcmi_gmm_em_create(int Dj, int M, const double *w, const double *mu, const double *sigma, double min_covar,
                                  vcmi_gmm_em **out) {
  if (!w || !mu || !sigma || !out) return fail(VCMI_ERR_ARG, "vcmi_gmm_em_create: NULL argument");
  *out = nullptr;
  if (Dj < 1 || M < 1) return fail(VCMI_ERR_DIM, "vcmi_gmm_em_create: Dj=%d M=%d invalid", Dj, M);
  if (Dj > 256)   // em_mstep_full_kernel stages the mean vector in a 256-entry LDS array
    return fail(VCMI_ERR_DIM, "vcmi_gmm_em_create: joint dimension %d exceeds the device EM limit (256)", Dj);
  if (!(min_covar >= 0.0)) return fail(VCMI_ERR_ARG, "vcmi_gmm_em_create: min_covar must be >= 0");
  VCMI_TRY(check_device());
  vcmi_gmm_em *h = new (std::nothrow) vcmi_gmm_em();
  if (!h) return fail(VCMI_ERR_OOM, "out of host memory");
  h->Dj = Dj;
  h->M = M;
  h->min_covar = min_covar;
  (void)hipGetDevice(&h->device);
  const size_t dd = (size_t)Dj * Dj;
  int rc = h->params.alloc((size_t)M * (1 + Dj + dd));
  if (rc == VCMI_OK) rc = h->flag.alloc(1);
  if (rc != VCMI_OK) {
    delete h;
    return rc;
  }
  hipError_t e = upload_now_hip(h->w(), w, sizeof(double) * M);
  if (e == hipSuccess) e = upload_now_hip(h->mu(), mu, sizeof(double) * M * Dj);
  if (e == hipSuccess) e = upload_now_hip(h->sigma(), sigma, sizeof(double) * M * dd);
  if (e == hipSuccess) e = hipMemset(h->flag.p, 0, sizeof(int));
  if (e != hipSuccess) {
    delete h;
    return fail(VCMI_ERR_HIP, "vcmi_gmm_em_create: %s", hipGetErrorString(e));
  }
  *out = h;
  return VCMI_OK;
}

extern "C" int vcmi_gmm_em_destroy(vcmi_gmm_em *h) {
  delete h;
  return VCMI_OK;
}

extern "C" int vcmi_gmm_em_estep_dev(vcmi_gmm_em *h, const double *dX, int64_t N, double *dstats, void *stream) {
  if (!h || !dstats) return fail(VCMI_ERR_ARG, "vcmi_gmm_em_estep_dev: NULL argument");
  if (N < 0 || (N > 0 && !dX)) return fail(VCMI_ERR_ARG, "vcmi_gmm_em_estep_dev: bad frame block");
  hipStream_t st = as_stream(stream);
  if (!h->prepared) VCMI_TRY(em_prepare(h, st));
  return estep_full_core(h->px, dX, N, h->Dj, h->M, dstats, st);
}

extern "C" int vcmi_gmm_em_mstep(vcmi_gmm_em *h, const double *dstats, void *stream, double *loglik) {
  if (!h || !dstats) return fail(VCMI_ERR_ARG, "vcmi_gmm_em_mstep: NULL argument");
  hipStream_t st = as_stream(stream);
  hipLaunchKernelGGL(em_mstep_full_kernel, dim3(h->M), dim3(256), 0, st, dstats, h->Dj, h->M, h->min_covar, h->w(), h->mu(),
                     h->sigma());
  VCMI_HIP(hipGetLastError());
  VCMI_TRY(em_prepare(h, st));
  double ll = 0.0;
  VCMI_HIP(hipMemcpyAsync(&ll, dstats + (h->plen() - 1), sizeof(double), hipMemcpyDeviceToHost, st));
  VCMI_TRY(read_pd_flag(h->flag.p, st));   // synchronises: covers the initial and the new parameters
  if (loglik) *loglik = ll;
  return VCMI_OK;
}

extern "C" int vcmi_gmm_em_get(vcmi_gmm_em *h, double *w, double *mu, double *sigma) {
  if (!h || !w || !mu || !sigma) return fail(VCMI_ERR_ARG, "vcmi_gmm_em_get: NULL argument");
  VCMI_HIP(hipDeviceSynchronize());
  const size_t dd = (size_t)h->Dj * h->Dj;
  VCMI_HIP(hipMemcpy(w, h->w(), sizeof(double) * h->M, hipMemcpyDeviceToHost));
  VCMI_HIP(hipMemcpy(mu, h->mu(), sizeof(double) * h->M * h->Dj, hipMemcpyDeviceToHost));
  VCMI_HIP(hipMemcpy(sigma, h->sigma(), sizeof(double) * h->M * dd, hipMemcpyDeviceToHost));
  return VCMI_OK;
}

// Measurement hook (not part of include/vcmi.h; bench.py prices the diagonal E-step's matrix pipe with it): *issued (may be
// NULL) receives the v_mfma_f64_16x16x4 instructions the MFMA E-step kernels of THIS host thread have issued since the
// counter was last enabled; then enable != 0 (re)starts it at zero, enable == 0 switches it off.  Synchronises.
extern "C" int vcmi_debug_estep_mfma(int enable, int64_t *issued) {
  using namespace vcmi;
  EstepScratch &sc = scratch();
  if (issued) {
    *issued = 0;
    if (sc.mfma_count.p) {
      unsigned long long h = 0;
      VCMI_HIP(hipMemcpy(&h, sc.mfma_count.p, sizeof(h), hipMemcpyDeviceToHost));
      *issued = (int64_t)h;
    }
  }
  if (enable) {
    if (!sc.mfma_count.p) VCMI_TRY(sc.mfma_count.alloc(1));
    VCMI_HIP(hipMemset(sc.mfma_count.p, 0, sizeof(unsigned long long)));
  } else {
    sc.mfma_count.release();
  }
  return VCMI_OK;
}

// Measurement hook (not part of include/vcmi.h): did the last diagonal E-step of this host thread take the hard-assignment
// path (estep_hard.hpp), and how many of its frames were soft (went through estep_mfma_kernel)?  *soft = -1: the one-kernel
// path.  Synchronises with the device.
extern "C" int vcmi_debug_estep_last_soft(int64_t *soft) {
  using namespace vcmi;
  if (!soft) return fail(VCMI_ERR_ARG, "vcmi_debug_estep_last_soft: NULL argument");
  EstepScratch &sc = scratch();
  *soft = -1;
  if (sc.last_hard && sc.ctl.p) {
    int64_t h[kCtlLen];
    VCMI_HIP(hipDeviceSynchronize());
    VCMI_HIP(hipMemcpy(h, sc.ctl.p, sizeof(h), hipMemcpyDeviceToHost));
    if (h[kCtlHard]) *soft = h[kCtlNSoft];
  }
  return VCMI_OK;
}

extern "C" int vcmi_estep_set_path(int path) {
  using namespace vcmi;
  if (path != VCMI_ESTEP_AUTO && path != VCMI_ESTEP_HARD && path != VCMI_ESTEP_SOFT)
    return fail(VCMI_ERR_ARG, "vcmi_estep_set_path: %d is not VCMI_ESTEP_AUTO / _HARD / _SOFT", path);
  estep_path_choice_ref() = path;
  return VCMI_OK;
}
extern "C" int vcmi_estep_get_path(int *path) {
  using namespace vcmi;
  if (!path) return fail(VCMI_ERR_ARG, "vcmi_estep_get_path: NULL argument");
  *path = estep_path_choice_ref();
  return VCMI_OK;
}

// The same for the full-covariance STATISTICS kernel (its log-density kernel issues a fixed, known number of MFMAs).
extern "C" int vcmi_debug_estep_full_mfma(int enable, int64_t *issued) {
  using namespace vcmi;
  EstepFullScratch &sc = full_scratch();
  if (issued) {
    *issued = 0;
    if (sc.mfma_count.p) {
      unsigned long long h = 0;
      VCMI_HIP(hipMemcpy(&h, sc.mfma_count.p, sizeof(h), hipMemcpyDeviceToHost));
      *issued = (int64_t)h;
    }
  }
  if (enable) {
    if (!sc.mfma_count.p) VCMI_TRY(sc.mfma_count.alloc(1));
    VCMI_HIP(hipMemset(sc.mfma_count.p, 0, sizeof(unsigned long long)));
  } else {
    sc.mfma_count.release();
  }
  return VCMI_OK;
}
