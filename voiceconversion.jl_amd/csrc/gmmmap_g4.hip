// gmmmap_g4.hip -- fvconvert batch kernel with 4-mixture row grouping (opt-in: vcmi_gmmmap_set_kernel(g, 3)).
// Measured on MI355X (D=40, M=64, 1e6 frames): 7.7 % fewer MFMAs than gmmmap_mfma_kernel but the same 5.4-5.6 ms --
// the saving is eaten by the extra piece barriers -- so it is NOT the default; kept as a parity-tested alternative.
//
// Same math and same MFMA mapping as gmmmap_mfma_kernel (gmmmap.hip: D[16 rows x 16 frames] += W[16x4] X[4x16],
// reference src/gmmmap.jl:101-118), different ROW TILING.  The whitening factor U_m is lower triangular: row r needs
// only k <= r.  A 16-row tile of ONE mixture pays for the full k extent of its last row (42 MFMA steps per mixture
// at D = 40 against an ideal 37.8).  Here a tile takes 4 consecutive rows (4g..4g+3) from each of FOUR mixtures: all
// 16 rows have the same k extent, so tile g costs exactly g+1 k-steps -- 55 steps per 4 mixtures instead of 68 --
// and in the C/D layout (row = (lane>>4) + 4*reg) register `reg` of the accumulator belongs to mixture `reg`, so the
// per-mixture |z|^2 reduction is still "square, add over registers, two cross-lane steps".  The regression rows of the
// 4 mixtures are concatenated (4*Dp rows = Dp/4 full tiles, no padding).  Total 38.75 steps per mixture at D = 40
// (97.5 % of ideal).  Four log-densities arrive together, so the online softmax does one max/rescale check per 4
// mixtures.
//
// Operand stream: per group of 4 mixtures a sequence of PIECES (<= 32 MFMA steps each, fixed stride PB doubles),
// each the unit of LDS double-buffering: [fragments in issue order | accumulator initial values | 4 log-constants].
#include "vcmi_common.hpp"
#include "gmmmap_handle.hpp"

#include <cmath>

namespace vcmi {

typedef double d4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------------
// piece plan, shared verbatim by the host packer (run time) and the kernel (compile time)
// ------------------------------------------------------------------------------------------------
struct G4Plan {
  static constexpr int MAXP = 16, PMAX = 32;   // pieces per group (upper bound), MFMA steps per piece (upper bound)
  int DP = 0, RG = 0, KS = 0;
  int np = 0;                 // pieces per group
  int nup = 0;                // of which whitening pieces (they come first)
  int kind[MAXP] = {};        // 0 = whitening (U) piece, 1 = regression (A) piece
  int t0[MAXP] = {};          // first tile of the piece
  int nt[MAXP] = {};          // tiles in the piece
  int steps[MAXP] = {};       // MFMA steps in the piece
  int PB = 0;                 // piece stride in doubles
  int CINIT_OFF = 0, LC_OFF = 0;
  __host__ __device__ constexpr explicit G4Plan(int dp) : DP(dp), RG(dp / 4), KS(dp / 4) {
    int g = 0;
    while (g < RG) {          // U tiles: tile g costs g+1 steps; greedy fill up to PMAX
      int s = 0, n = 0;
      while (g + n < RG && (n == 0 || s + (g + n + 1) <= PMAX)) { s += g + n + 1; ++n; }
      kind[np] = 0; t0[np] = g; nt[np] = n; steps[np] = s; ++np;
      g += n;
    }
    nup = np;
    const int per = (PMAX / KS) > 0 ? (PMAX / KS) : 1;   // A tiles per piece (each costs KS steps)
    int j = 0;
    while (j < RG) {          // 4*DP rows / 16 = DP/4 = RG regression tiles
      const int n = (RG - j < per) ? RG - j : per;
      kind[np] = 1; t0[np] = j; nt[np] = n; steps[np] = n * KS; ++np;
      j += n;
    }
    int maxsteps = 0, maxtiles = 0;
    for (int p = 0; p < np; ++p) {
      if (steps[p] > maxsteps) maxsteps = steps[p];
      if (nt[p] > maxtiles) maxtiles = nt[p];
    }
    CINIT_OFF = maxsteps * 64;
    LC_OFF = CINIT_OFF + maxtiles * 16;
    PB = ((LC_OFF + 4 + 1023) / 1024) * 1024;   // whole double2 per thread for 256- and 512-thread groups
  }
};

// ------------------------------------------------------------------------------------------------
// kernel
// ------------------------------------------------------------------------------------------------
template <int DP, int FT, int P>
struct G4Piece;

template <int DP, int FT, int WAVES>
__global__ void __launch_bounds__(WAVES * 64) __attribute__((amdgpu_waves_per_eu(FT == 1 ? 4 : 2)))
gmmmap_g4_kernel(const double *__restrict__ packed, int NG, int D, const double *__restrict__ X, int64_t ldx, int64_t T,
                 double *__restrict__ Y, int64_t ldy) {
  constexpr G4Plan PL(DP);
  constexpr int KS = PL.KS, PB = PL.PB, NP = PL.np;
  constexpr int NTHREADS = WAVES * 64;
  constexpr int NV = PB / 2 / NTHREADS;
  static_assert(PB % (2 * NTHREADS) == 0, "piece stride must be a whole number of double2 per thread");
  extern __shared__ double smem[];   // 2 * PB doubles

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lcol = lane & 15, lgrp = lane >> 4;
  const int64_t frame0 = ((int64_t)blockIdx.x * WAVES + wave) * (16 * FT);

  double xb[FT][KS];
#pragma unroll
  for (int f = 0; f < FT; ++f) {
    const int64_t fr = frame0 + 16 * f + lcol;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int k = 4 * ks + lgrp;
      xb[f][ks] = (fr < T && k < D) ? X[fr * ldx + k] : 0.0;
    }
  }
  double yacc[FT][KS], runmax[FT], den[FT];
#pragma unroll
  for (int f = 0; f < FT; ++f) {
    runmax[f] = -INFINITY;
    den[f] = 0.0;
#pragma unroll
    for (int j = 0; j < KS; ++j) yacc[f][j] = 0.0;
  }
  {   // stage piece 0
    const double2 *src = reinterpret_cast<const double2 *>(packed);
    double2 *dst = reinterpret_cast<double2 *>(smem);
#pragma unroll
    for (int i = 0; i < NV; ++i) dst[tid + i * NTHREADS] = src[tid + i * NTHREADS];
  }
  __syncthreads();

  const int npieces = NG * NP;
  int gp = 0;
  for (int G = 0; G < NG; ++G) {
    double q[FT][4], wg[FT][4];
#pragma unroll
    for (int f = 0; f < FT; ++f)
#pragma unroll
      for (int r = 0; r < 4; ++r) q[f][r] = wg[f][r] = 0.0;
    double lc[4] = {0, 0, 0, 0};
    G4Piece<DP, FT, 0>::template run<WAVES>(packed, smem, npieces, gp, xb, yacc, runmax, den, q, wg, lc);
  }

#pragma unroll
  for (int f = 0; f < FT; ++f) {
    const int64_t fr = frame0 + 16 * f + lcol;
    const double inv = 1.0 / den[f];
    if (fr < T) {
#pragma unroll
      for (int j = 0; j < KS; ++j) {
        const int row = 4 * j + lgrp;
        if (row < D) Y[fr * ldy + row] = yacc[f][j] * inv;
      }
    }
  }
}

// one piece (P compile-time): prefetch the next piece, compute from the current LDS buffer, publish, barrier; recurse
template <int DP, int FT, int P>
struct G4Piece {
  template <int WAVES>
  static __device__ __forceinline__ void run(const double *__restrict__ packed, double *smem, int npieces, int &gp,
                                             double (&xb)[FT][DP / 4], double (&yacc)[FT][DP / 4], double (&runmax)[FT],
                                             double (&den)[FT], double (&q)[FT][4], double (&wg)[FT][4], double (&lc)[4]) {
    constexpr G4Plan PL(DP);
    constexpr int KS = PL.KS, PB = PL.PB, NP = PL.np;
    if constexpr (P < NP) {
      constexpr int NTHREADS = WAVES * 64, NV = PB / 2 / NTHREADS;
      constexpr int KIND = PL.kind[P], T0 = PL.t0[P], NTL = PL.nt[P];
      const int tid = threadIdx.x, lane = tid & 63, lgrp = lane >> 4;
      const double *cur = smem + (gp & 1) * PB;
      double2 *nxt = reinterpret_cast<double2 *>(smem + ((gp + 1) & 1) * PB);
      double2 pre[NV];
      {
        const int gn = (gp + 1 < npieces) ? gp + 1 : gp;   // the last piece re-reads itself (branch-free body)
        const double2 *src = reinterpret_cast<const double2 *>(packed + (size_t)gn * PB);
#pragma unroll
        for (int i = 0; i < NV; ++i) pre[i] = src[tid + i * NTHREADS];
      }
      if (P == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) lc[r] = cur[PL.LC_OFF + r];
      }
      int s = 0;
      if (KIND == 0) {
        // ---- whitening tiles, two at a time (independent accumulator chains); tile g costs g+1 k-steps ----
#pragma unroll
        for (int c0 = 0; c0 < NTL; c0 += 2) {
          constexpr int dummy = 0;
          (void)dummy;
          d4 acc[2][FT];
#pragma unroll
          for (int c = 0; c < 2; ++c) {
            if (c0 + c < NTL) {
              d4 ci;
#pragma unroll
              for (int r = 0; r < 4; ++r) ci[r] = cur[PL.CINIT_OFF + 16 * (c0 + c) + 4 * r + lgrp];
#pragma unroll
              for (int f = 0; f < FT; ++f) acc[c][f] = ci;
            }
          }
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
            for (int c = 0; c < 2; ++c) {
              if (c0 + c < NTL && ks <= T0 + c0 + c) {
                const double a = cur[s * 64 + lane];
                ++s;
#pragma unroll
                for (int f = 0; f < FT; ++f) acc[c][f] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, xb[f][ks], acc[c][f], 0, 0, 0);
              }
            }
          }
#pragma unroll
          for (int c = 0; c < 2; ++c)
            if (c0 + c < NTL)
#pragma unroll
              for (int f = 0; f < FT; ++f)
#pragma unroll
                for (int r = 0; r < 4; ++r) q[f][r] = fma(acc[c][f][r], acc[c][f][r], q[f][r]);
        }
        if (P == PL.nup - 1) {
          // ---- all |z|^2 of the group's 4 mixtures are complete: posterior weights (online softmax, lazy rescale) ----
#pragma unroll
          for (int f = 0; f < FT; ++f) {
            double l[4];
            double gmax = -INFINITY;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              double qq = q[f][r];
              qq += __shfl_xor(qq, 16);
              qq += __shfl_xor(qq, 32);
              l[r] = lc[r] - 0.5 * qq;                     // lc = -inf for zero-weight / padding mixtures
              gmax = fmax(gmax, l[r]);
            }
            if (__builtin_amdgcn_ballot_w64(gmax > runmax[f]) != 0) {   // wave-uniform: some frame has a new maximum
              const double nm = fmax(runmax[f], gmax);
              const double sc = (nm == -INFINITY) ? 1.0 : exp(runmax[f] - nm);
              den[f] *= sc;
              runmax[f] = nm;
#pragma unroll
              for (int j = 0; j < KS; ++j) yacc[f][j] *= sc;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              wg[f][r] = (l[r] == -INFINITY) ? 0.0 : exp(l[r] - runmax[f]);
              den[f] += wg[f][r];
            }
          }
        }
      } else {
        // ---- regression tiles of the concatenated [A_m0; A_m1; A_m2; A_m3] rows; all NTL tiles interleaved ----
        d4 acc[NTL][FT];
#pragma unroll
        for (int c = 0; c < NTL; ++c) {
          d4 ci;
#pragma unroll
          for (int r = 0; r < 4; ++r) ci[r] = cur[PL.CINIT_OFF + 16 * c + 4 * r + lgrp];
#pragma unroll
          for (int f = 0; f < FT; ++f) acc[c][f] = ci;
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
          for (int c = 0; c < NTL; ++c) {
            const double a = cur[s * 64 + lane];
            ++s;
#pragma unroll
            for (int f = 0; f < FT; ++f) acc[c][f] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, xb[f][ks], acc[c][f], 0, 0, 0);
          }
        }
        // y += w_m * (A_m x + b_m): register r of tile j holds rows R = 16 j + 4 r + lgrp of the concatenation
#pragma unroll
        for (int c = 0; c < NTL; ++c)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            constexpr int dummy2 = 0;
            (void)dummy2;
            const int R0 = 16 * (T0 + c) + 4 * r;
            const int mi = R0 / DP, rg = (R0 % DP) / 4;     // compile-time after unrolling
#pragma unroll
            for (int f = 0; f < FT; ++f) yacc[f][rg] = fma(wg[f][mi], acc[c][f][r], yacc[f][rg]);
          }
      }
#pragma unroll
      for (int i = 0; i < NV; ++i) nxt[tid + i * NTHREADS] = pre[i];
      __syncthreads();
      ++gp;
      G4Piece<DP, FT, P + 1>::template run<WAVES>(packed, smem, npieces, gp, xb, yacc, runmax, den, q, wg, lc);
    }
  }
};

// ------------------------------------------------------------------------------------------------
// host: packing and launch
// ------------------------------------------------------------------------------------------------
bool gmmmap_has_g4(int DP) { return DP == 24 || DP == 40; }

// hU, hA: [M][DP][DP] row-major; hcz, hb: [M][DP]; hlc: [M]
int gmmmap_pack_g4(vcmi_gmmmap *g, const std::vector<double> &hU, const std::vector<double> &hA,
                   const std::vector<double> &hcz, const std::vector<double> &hb, const std::vector<double> &hlc) {
  const int DP = g->DP, M = g->M;
  if (!gmmmap_has_g4(DP)) return VCMI_OK;
  const G4Plan PL(DP);
  const int NG = (M + 3) / 4;
  const size_t pp = (size_t)DP * DP;
  std::vector<double> pk((size_t)NG * PL.np * PL.PB, 0.0);
  auto Uval = [&](int m, int row, int k) { return (m < M && k < DP) ? hU[pp * m + (size_t)row * DP + k] : 0.0; };
  auto Aval = [&](int m, int row, int k) { return (m < M && k < DP) ? hA[pp * m + (size_t)row * DP + k] : 0.0; };
  for (int G = 0; G < NG; ++G)
    for (int p = 0; p < PL.np; ++p) {
      double *blk = &pk[((size_t)G * PL.np + p) * PL.PB];
      int s = 0;
      if (PL.kind[p] == 0) {
        for (int c0 = 0; c0 < PL.nt[p]; c0 += 2)
          for (int ks = 0; ks < PL.KS; ++ks)
            for (int c = 0; c < 2; ++c) {
              const int gt = PL.t0[p] + c0 + c;     // tile = row group gt: rows 4 gt .. 4 gt + 3 of each of the 4 mixtures
              if (c0 + c >= PL.nt[p] || ks > gt) continue;
              for (int l = 0; l < 64; ++l) {
                const int i = l & 15, k = 4 * ks + (l >> 4);
                blk[(size_t)s * 64 + l] = Uval(4 * G + (i >> 2), 4 * gt + (i & 3), k);
              }
              ++s;
            }
        for (int c = 0; c < PL.nt[p]; ++c)
          for (int i = 0; i < 16; ++i) {
            const int m = 4 * G + (i >> 2), row = 4 * (PL.t0[p] + c) + (i & 3);
            blk[PL.CINIT_OFF + 16 * c + i] = (m < M) ? -hcz[(size_t)DP * m + row] : 0.0;
          }
      } else {
        for (int ks = 0; ks < PL.KS; ++ks)
          for (int c = 0; c < PL.nt[p]; ++c) {
            for (int l = 0; l < 64; ++l) {
              const int R = 16 * (PL.t0[p] + c) + (l & 15), k = 4 * ks + (l >> 4);
              blk[(size_t)s * 64 + l] = Aval(4 * G + R / DP, R % DP, k);
            }
            ++s;
          }
        for (int c = 0; c < PL.nt[p]; ++c)
          for (int i = 0; i < 16; ++i) {
            const int R = 16 * (PL.t0[p] + c) + i, m = 4 * G + R / DP;
            blk[PL.CINIT_OFF + 16 * c + i] = (m < M) ? hb[(size_t)DP * m + R % DP] : 0.0;
          }
      }
      if (p == 0)
        for (int r = 0; r < 4; ++r) blk[PL.LC_OFF + r] = (4 * G + r < M) ? hlc[4 * G + r] : -INFINITY;
    }
  VCMI_TRY(g->packed4.alloc(pk.size()));
  VCMI_HIP(hipMemcpy(g->packed4.p, pk.data(), pk.size() * 8, hipMemcpyHostToDevice));
  return VCMI_OK;
}

template <int DP, int FT, int WAVES>
static int launch_g4(const vcmi_gmmmap *g, const double *dX, int64_t ldx, int64_t T, double *dY, int64_t ldy, hipStream_t st) {
  constexpr G4Plan PL(DP);
  const size_t shmem = 2 * (size_t)PL.PB * sizeof(double);
  auto kern = gmmmap_g4_kernel<DP, FT, WAVES>;
  static bool attr_done = false;
  if (!attr_done) {
    VCMI_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    attr_done = true;
  }
  const int64_t per_wg = (int64_t)16 * FT * WAVES;
  const int64_t blocks = (T + per_wg - 1) / per_wg;
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(WAVES * 64), shmem, st, g->packed4.p, (g->M + 3) / 4, g->D, dX, ldx, T,
                     dY, ldy);
  VCMI_HIP(hipGetLastError());
  return VCMI_OK;
}

int gmmmap_convert_g4_device(vcmi_gmmmap *g, const double *dX, int64_t ldx, int64_t T, double *dY, int64_t ldy, hipStream_t st) {
  switch (g->DP) {
    case 24: return launch_g4<24, 1, 8>(g, dX, ldx, T, dY, ldy, st);
    case 40: return launch_g4<40, 1, 8>(g, dX, ldx, T, dY, ldy, st);
    default: return fail(VCMI_ERR_ARG, "no grouped-tiling instantiation for padded dimension %d", g->DP);
  }
}

}  // namespace vcmi
