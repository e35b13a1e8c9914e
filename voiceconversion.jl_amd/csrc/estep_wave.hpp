// estep_wave.hpp -- the diagonal E-step in ONE barrier per block (included by estep.hip; M > 64, Dj <= 80).
//
// estep_mfma_kernel<DJ, 0> separates its three phases by workgroup barriers because its softmax walks a frame across all 128
// mixtures -- i.e. across the eight waves that computed them: step A | barrier | softmax | barrier | step B | barrier, with
// nothing overlapping (up to 20 % of a wave's time in the barriers, a softmax phase without a single MFMA).
// Here every wave keeps the whole path of ITS 16 mixtures to itself and the three steps of a block are spread over two
// iterations, so that ONE barrier per 32-frame block carries both exchanges the softmax needs:
//   iteration b, before the barrier
//     step A(b)   l[f][m] for the block's frames and the wave's mixtures (MFMA, W in registers) -> a wave-PRIVATE plane in LDS;
//                 the wave's own maximum per frame m_w(f) (two lanes per frame) and #{l > m_w - 36} -> the shared table;
//     exps(b-1)   the frame's maximum over ALL mixtures u = max_w m_w is known since the last barrier: e = exp(l - u) in place
//                 (hopeless values -- almost all of them -- are skipped 64 at a time), the wave's sum s_w(f) -> the shared
//                 table.  Frames with competing mixtures (more than one within 36 nats of u, counted conservatively from
//                 the table) first re-evaluate exactly those term by term, as estep_mfma_kernel does; the softmax is
//                 shift-invariant, so u stays the reference and nothing has to be exchanged again;
//   THE barrier; the LDS-DMA of block b+2 is issued behind it (four x buffers: block b-2's is free now);
//   iteration b, after the barrier
//     step B(b-1) S = sum_w s_w, gamma[f][m] = e[f][m] / S (never stored), frames grouped by the tile that wins them, k-steps
//                 whose 64 responsibilities are exactly zero skipped: S[m][.] += sum_f gamma[f][m] [x, x^2] (MFMA).
// Between two barriers a wave thus runs B(b-1), A(b+1) and the exps of block b at its own pace; the two waves of a SIMD are
// rarely in the same phase.  LDS at Dj = 80: 4 x 20.5 KB of x + 8 waves x 2 planes x 4.25 KB + 9.3 KB of tables = 159.3 KB.
#pragma once

namespace vcmi {

template <int DJ>
struct EstepWaveCfg {
  static constexpr int KS = 2 * DJ / 4, NDT = 2 * DJ / 16;
  static constexpr int FB = 32;                                  // frames per block
  static constexpr int RSX = (DJ + 2 + 13) / 32 * 32 + 18;       // as EstepCfg: 16-byte rows, conflict-free column reads
  static constexpr int XBUF = FB * RSX;                          // doubles per x buffer (the last 1 KB wave-instruction of the DMA is masked)
  static constexpr int NXB = 4;
  // a private plane holds value (frame f, mixture c of the wave's 16) at (c & 7) * PROW + 2 f + (c >> 3): the softmax lanes
  // (lane = 2 f + half, value i = c & 7) read and write stride-1 across the wave, step A's stores (rows of 8 dwords, 8 dwords
  // apart) are conflict-free too; the 4 spare doubles of each row hold 1 / S of the block's frames (step B)
  static constexpr int PROW = 2 * FB + 4;
  static constexpr int PLANE = 8 * PROW;                         // doubles per private plane (l, then e in place); two per wave
  static constexpr int XSET = 8 * FB;                            // doubles per table set: [wave][frame]
  // doubles: [x NXB][planes 8 x 2][maxima 2 sets][sums 2 sets][etab 64]; bytes: [counts 2 x 8 x FB][frame order 8 x FB]
  static constexpr size_t LDS_DOUBLES = (size_t)NXB * XBUF + 16 * PLANE + 4 * XSET + 64;
  static constexpr size_t LDS_BYTES = LDS_DOUBLES * 8 + (size_t)2 * 8 * FB + 8 * FB;
  static_assert(LDS_BYTES <= 160 * 1024, "LDS");
};

template <int DJ>
__global__ void __launch_bounds__(512)
estep_wave_kernel(const double *__restrict__ X, int64_t N, int M, const double *__restrict__ Wpack,
                  const double *__restrict__ cinit, double *__restrict__ part, int64_t plen,
                  const double *__restrict__ refmu, const double *__restrict__ refiv, const double *__restrict__ refc, int dj,
                  unsigned long long *__restrict__ mfma_count) {
  using C = EstepWaveCfg<DJ>;
  constexpr int KS = C::KS, NDT = C::NDT, FB = C::FB, RSX = C::RSX, XBUF = C::XBUF, PROW = C::PROW;
  constexpr double kRefine = 36.0;
  extern __shared__ double smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // in a scalar register: every per-wave LDS address below is uniform
  const int lcol = lane & 15, lgrp = lane >> 4;
  double *xbuf = smem;
  double *planes = smem + C::NXB * XBUF + wave * 2 * C::PLANE;                    // this wave's two planes [FB][16]
  double *xm = smem + C::NXB * XBUF + 16 * C::PLANE;                              // [2][8][FB] the waves' maxima per frame
  double *xsum = xm + 2 * C::XSET;                                                // [2][8][FB] the waves' sums per frame
  double *etab = xsum + 2 * C::XSET;
  unsigned char *xcnt = reinterpret_cast<unsigned char *>(etab + 64);             // [2][8][FB] #{l > m_w - 36} (<= 16)
  unsigned char *fperm = xcnt + 2 * 8 * FB + wave * FB;                           // this wave's grouped frame order
  double *red = smem + C::NXB * XBUF;                                             // epilogue: log-likelihood scratch (aliases the planes)
  if (tid < 64) etab[tid] = kExp2Tab[tid];

  double wfrag[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) wfrag[ks] = Wpack[((size_t)wave * KS + ks) * 64 + lane];
  const double cm = cinit[16 * wave + lcol];
  const d4 cin = {cm, cm, cm, cm};
  d4 sacc[NDT];
#pragma unroll
  for (int j = 0; j < NDT; ++j) sacc[j] = d4{0, 0, 0, 0};
  double s0l = 0.0, llacc = 0.0, sprod = 1.0;
  int nprod = 0, nmfma = 0;

  const int64_t nblocks = (N + FB - 1) / FB;
  constexpr int ROWB = RSX * 8, NCHUNK = (XBUF * 8 + 1023) / 1024;
  auto stage = [&](int64_t f0, double *dst) {
    const char *base = reinterpret_cast<const char *>(X + f0 * dj);
    const int last = (int)((N - 1 - f0 < FB - 1) ? N - 1 - f0 : FB - 1);
    int lane_v = lane;
    asm volatile("" : "+v"(lane_v));                   // (keeps the per-lane (row, column) of the chunks out of the loop-invariant registers)
#pragma unroll
    for (int i = 0; i < (NCHUNK + 7) / 8; ++i) {
      const int q = wave + 8 * i;
      if (q < NCHUNK) {                                // wave-uniform
        const int o = 1024 * q + 16 * lane_v, row = o / ROWB, col = o - row * ROWB;
        const int rowc = row < last ? row : last;
        const unsigned off = (col < dj * 8) ? (unsigned)(rowc * (dj * 8) + col) : 0u;
        if (o < XBUF * 8)                              // (the buffers are not padded to whole KB: the last instruction is partial)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(base + off),
                                         (__attribute__((address_space(3))) void *)(reinterpret_cast<char *>(dst) + 1024 * q), 16, 0, 0);
      }
    }
  };
  const int64_t stride = gridDim.x;
  // this workgroup's blocks: blockIdx.x, + stride, ...  (nb of them; block k of the workgroup lives in x buffer k % 4)
  const int64_t nb = ((int64_t)blockIdx.x < nblocks) ? (nblocks - 1 - blockIdx.x) / stride + 1 : 0;
  auto block_f0 = [&](int64_t k) { return ((int64_t)blockIdx.x + k * stride) * FB; };
  if (nb > 0) stage(block_f0(0), xbuf);
  if (nb > 1) stage(block_f0(1), xbuf + XBUF);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  const int fl = lane >> 1, hh = lane & 1;             // softmax: two lanes per frame, eight mixtures each
  auto wave_lds_sync = [&]() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); };   // a wave's LDS writes -> its own later reads

  // carried from the first half of an iteration (block k-1's maximum and winning tile, read from the table BEFORE the barrier:
  // behind it a faster wave may already be writing the same set for block k+1) to the second half
  double u_prev = 0.0;
  int wtile_prev = 0;
#ifdef VCMI_ESTEP_PROF
  unsigned long long pt_[6] = {0, 0, 0, 0, 0, 0};     // probe build: s_memtime counts in A | max | exps | barrier | combine + order | B
  unsigned long long tl_ = __builtin_readcyclecounter();
#define VCMI_WPT(i) { const unsigned long long t_ = __builtin_readcyclecounter(); pt_[i] += t_ - tl_; tl_ = t_; }
#else
#define VCMI_WPT(i)
#endif
  for (int64_t k = 0; k <= nb; ++k) {
    const int set = (int)(k & 1);
    if (k < nb) {
      // ---- step A(k): the wave's 16 mixtures for the block's frames -> plane k & 1; the wave's own maximum per frame ----
      const double *xs = xbuf + (int)(k & 3) * XBUF;
      double *Lp = planes + set * C::PLANE;
#pragma unroll
      for (int ft = 0; ft < FB / 16; ++ft) {
        d4 acc = cin;
        nmfma += KS;
        const double *xr = xs + (16 * ft + lcol) * RSX + lgrp;
#pragma unroll
        for (int ks = 0; ks < KS / 2; ++ks) {
          const double x = xr[4 * ks];
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x * x, wfrag[ks], acc, 0, 0, 0);
        }
#pragma unroll
        for (int ks = 0; ks < KS / 2; ++ks) {
          const double x = xr[4 * ks];
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x, wfrag[KS / 2 + ks], acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) Lp[(lcol & 7) * PROW + 2 * (16 * ft + 4 * r + lgrp) + (lcol >> 3)] = acc[r];
      }
      wave_lds_sync();
      VCMI_WPT(0)
      const double *lr = Lp + lane;                    // value i of this lane (frame fl, mixtures 8 hh + i) at lr[i * PROW]
      double m = lr[0];
#pragma unroll
      for (int i = 1; i < 8; ++i) m = fmax(m, lr[i * PROW]);
      m = fmax(m, dpp_row_f64<0xB1>(m));               // the other half of the frame (lane ^ 1)
      int cnt = 0;
#pragma unroll
      for (int i = 0; i < 8; ++i) cnt += (lr[i * PROW] > m - kRefine) ? 1 : 0;
      cnt += dpp_row_i32<0xB1>(cnt);
      if (hh == 0) {
        xm[set * C::XSET + wave * FB + fl] = m;
        xcnt[(set * 8 + wave) * FB + fl] = (unsigned char)cnt;
      }
      wave_lds_sync();
      VCMI_WPT(1)
    }
    if (k >= 1) {
      // ---- exps of block k-1 against the maximum over all mixtures (in the table since the last barrier) ----
      const int ps = set ^ 1;
      const double *xs = xbuf + (int)((k - 1) & 3) * XBUF;
      double *Lp = planes + ps * C::PLANE;
      const bool livef = block_f0(k - 1) + fl < N;
      double u = xm[ps * C::XSET + fl];
#pragma unroll
      for (int w = 1; w < 8; ++w) u = fmax(u, xm[ps * C::XSET + w * FB + fl]);
      int wt = 0, nc = 0;
#pragma unroll
      for (int w = 7; w >= 0; --w) {
        const double mw = xm[ps * C::XSET + w * FB + fl];
        wt = (mw == u) ? w : wt;
        nc += (mw > u - kRefine) ? (int)xcnt[(ps * 8 + w) * FB + fl] : 0;
      }
      u_prev = u;
      wtile_prev = wt;
      double *lr = Lp + lane;                          // value i at lr[i * PROW]
      // refinement (see estep_mfma_kernel): frames with competing mixtures re-evaluate exactly those term by term from the
      // LDS copy of x.  u stays the reference of the exps (the softmax is shift-invariant; the exact values differ from
      // the GEMM form's by ~1e-7 at most), so no maximum has to be exchanged again.
      if (__builtin_amdgcn_ballot_w64(nc > 1 && livef) != 0) {
        if (nc > 1 && livef) {
          const double thr = u - kRefine;
          const double *xf = xs + fl * RSX;
#pragma unroll 1
          for (int i = 0; i < 8; ++i) {
            if (lr[i * PROW] > thr) {
              const int m = 16 * wave + 8 * hh + i;
              const double *mp = refmu + (size_t)dj * m, *ip = refiv + (size_t)dj * m;
              double q = 0.0;
#pragma unroll 2
              for (int d = 0; d < dj; ++d) {
                const double df = xf[d] - mp[d];
                q = fma(df * df, ip[d], q);
              }
              lr[i * PROW] = refc[2 * m] - 0.5 * q;
            }
          }
        }
        wave_lds_sync();
      }
      double s = 0.0;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const double d = lr[i * PROW] - u;
        // all 64 values of the pass hopeless (e^x = 0 below -745.2; u = -inf gives NaN, which compares false): no exp at all
        if (__builtin_amdgcn_ballot_w64(d > -745.2) == 0) {
          lr[i * PROW] = 0.0;
          continue;
        }
        const double e = (u > -INFINITY) ? vc_exp_tab(d, etab) : 0.0;
        lr[i * PROW] = e;
        s += e;
      }
      s += dpp_row_f64<0xB1>(s);
      if (hh == 0) xsum[ps * C::XSET + wave * FB + fl] = s;
      wave_lds_sync();
      VCMI_WPT(2)
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's share of block k+1's LDS-DMA has landed
    __syncthreads();                                   // THE barrier of the iteration
    VCMI_WPT(3)
    if (k + 2 < nb) stage(block_f0(k + 2), xbuf + (int)((k + 2) & 3) * XBUF);      // into block k-2's buffer: everyone has left it
    if (k >= 1) {
      // ---- step B(k-1): 1 / S, frame order, log-likelihood, then the statistics ----
      const int ps = set ^ 1;
      const double *xs = xbuf + (int)((k - 1) & 3) * XBUF;
      double *Ep = planes + ps * C::PLANE;
      const bool livef = block_f0(k - 1) + fl < N;
      double S = xsum[ps * C::XSET + fl];
#pragma unroll
      for (int w = 1; w < 8; ++w) S += xsum[ps * C::XSET + w * FB + fl];
      if (hh == 0) Ep[(fl >> 2) * PROW + 2 * FB + (fl & 3)] = (livef && S > 0.0) ? 1.0 / S : 0.0;      // (the plane's spare doubles)
      {
        const int key = wtile_prev;
        int base = 0, pos = 0;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          const unsigned long long bt = __builtin_amdgcn_ballot_w64(key == t && hh == 0);
          const int below = __builtin_amdgcn_mbcnt_hi((unsigned)(bt >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bt, 0));
          pos = (key == t) ? base + below : pos;
          base += __builtin_popcountll(bt);
        }
        if (hh == 0) fperm[pos] = (unsigned char)fl;
      }
      if (wave == 0) {
        if (hh == 0 && livef && S > 0.0) {
          llacc += u_prev;
          sprod *= S;
        }
        if (++nprod == 16) {                           // S in [1, 128]: sixteen of them stay below 1e34 -- one log per 16 blocks
          llacc += log(sprod);
          sprod = 1.0;
          nprod = 0;
        }
      }
      wave_lds_sync();
      VCMI_WPT(4)
#pragma unroll 2
      for (int ks = 0; ks < FB / 4; ++ks) {
        const int f = fperm[4 * ks + lgrp];
        const double gm = Ep[(lcol & 7) * PROW + 2 * f + (lcol >> 3)] * Ep[(f >> 2) * PROW + 2 * FB + (f & 3)];
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
      VCMI_WPT(5)
    }
  }
#ifdef VCMI_ESTEP_PROF
  if (blockIdx.x == 3 && lane == 0)
    printf("estep wave prof wave %d: A %llu | max %llu | exps %llu | barrier %llu | combine %llu | B %llu\n", wave, pt_[0], pt_[1], pt_[2], pt_[3], pt_[4], pt_[5]);
#endif
#undef VCMI_WPT

  if (mfma_count && lane == 0) atomicAdd(mfma_count, (unsigned long long)nmfma);
  double *P = part + (size_t)blockIdx.x * plen;
  s0l += __shfl_xor(s0l, 16);
  s0l += __shfl_xor(s0l, 32);
  if (lgrp == 0 && 16 * wave + lcol < M) P[16 * wave + lcol] = s0l;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int m = 16 * wave + 4 * r + lgrp;
    if (m < M) {
#pragma unroll
      for (int j = 0; j < NDT; ++j) {
        const int c = 16 * j + lcol;
        if (c < DJ) {
          if (c < dj) P[M + (size_t)m * dj + c] = sacc[j][r];
        } else if (c - DJ < dj) {
          P[M + (size_t)M * dj + (size_t)m * dj + (c - DJ)] = sacc[j][r];
        }
      }
    }
  }
  // log-likelihood: wave 0's even lanes hold it; fixed-order sum by thread 0
  __syncthreads();                                     // everyone has left the planes (red aliases them)
  if (wave == 0) red[lane] = llacc + log(sprod);
  __syncthreads();
  if (tid == 0) {
    double ll = 0.0;
    for (int i = 0; i < 64; i += 2) ll += red[i];
    P[plen - 1] = ll;
  }
}

}  // namespace vcmi
