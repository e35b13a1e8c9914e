// traj_solve_blk.hpp -- blocked banded Cholesky solve of the trajectory normal equations on v_mfma_f64_16x16x4.
// Included by traj.hip inside namespace vcmi (needs TrajUtt and traj_rsqrt).
//
// Solves (W' Dy^-1 W) y = W' Dy^-1 E of reference src/trajectory_gmmmap.jl:103-105 for one utterance per workgroup.
// The matrix P is block-pentadiagonal with D x D blocks (see the header of traj.hip).  Block step t works on the
// window   S00            r0        (block rows t, t+1, t+2; S20 and S22 are still the raw assembled blocks)
//          S10 S11        r1
//          S20 S21 S22    r2
//   1. pivot (ONE wave, no barriers): right-looking Cholesky of S00 on an 8 x 8 lane grid with the identity riding
//      along, so the same column operations leave U = L00^-1 (explicit, lower triangular) -- the only part of the
//      algorithm that is sequential in the scalar columns;
//   2. L10 = S10 U', L20 = S20 U'                                  (MFMA, in place; U' is triangular: k-steps skipped)
//   3. S11 -= L10 L10', S21 -= L20 L10', S22 -= L20 L20'          (MFMA, lower tiles only for the symmetric ones)
//   4. the panel [U; L10; z0; L20] goes to the HBM workspace, the window shifts by pointer rotation and block row
//      t+3 is assembled from the stencil.
// The right-hand side needs no code of its own: r_b lives in row D (a padding row of the 16-row tiles) of S_bb, a copy
// of r0 is put in row D of S10 and S20, and then row D of L10 / L20 is z0 = U r0 and the updates of step 3 apply
// r1 -= L10 z0, r2 -= L20 z0 to row D of S11 / S22.
// Back substitution: y_t = U' (z0 - L10' y_{t+1} - L20' y_{t+2}) -- two matrix-vector products, no sequential chain.
#pragma once

template <int D>
struct BlkCfg {
  static constexpr int DP = ((D + 1 + 15) / 16) * 16;   // rows / columns of a block buffer: 16-row tiles incl. the rhs row D
  static constexpr int NT = DP / 16;                    // tiles per dimension
  static constexpr int KS = (D + 3) / 4;                // k-steps of a product over the D columns
  static constexpr int LS = DP + 2;                     // row stride in doubles: LS/2 odd -> MFMA operand reads conflict-free
  static constexpr int BUF = DP * LS;                   // doubles per block buffer
  static constexpr int NB8 = (D + 7) / 8;               // 8 x 8 lane-grid tiles per dimension (pivot phase)
  static constexpr int CB = 2 * 8 * NB8 + 2;            // published pivot column: S part, U part, pivot
  static constexpr size_t PAN = (size_t)(3 * D + 1) * D;   // panel doubles per block step: U (D,D), L10 (D+1,D) incl. z0, L20 (D,D)
  static constexpr size_t lds_doubles = (size_t)6 * BUF + CB + 2 * D + 2 * 256;
};

typedef double blk_d4 __attribute__((ext_vector_type(4)));

#ifdef TRAJ_BLK_PROF
__device__ long long blk_prof[8];   // cycles of workgroup 0 per phase: pivot, trsm, update, panel, assemble, backsub
#define BLK_PROF_T0() long long pt_ = (long long)__builtin_readcyclecounter()
#define BLK_PROF(k)                                                     \
  do {                                                                  \
    const long long n_ = (long long)__builtin_readcyclecounter();       \
    if (blockIdx.x == 0 && threadIdx.x == 0) blk_prof[k] += n_ - pt_;   \
    pt_ = n_;                                                           \
  } while (0)
#else
#define BLK_PROF_T0()
#define BLK_PROF(k)
#endif

// blocks (a,a-2), (a,a-1), (a,a) of P and r_a (row D of the diagonal block) into three block buffers; every entry of the
// DP x DP area is written (zeros outside D x D, outside the band and beyond the utterance) so recycled buffers are clean.
template <int D>
__device__ void blk_assemble(double *Bm2, double *Bm1, double *Bd, int a, int T, const int64_t *__restrict__ mh,
                             const double *__restrict__ g, const double *__restrict__ Qall, int tid, int nthr) {
  using C = BlkCfg<D>;
  constexpr int D2 = 2 * D, DP = C::DP, LS = C::LS;
  const bool live = a < T;
  const double *Qa = live ? Qall + (size_t)(mh[a] - 1) * D2 * D2 : nullptr;
  const double *Qm = (live && a >= 1) ? Qall + (size_t)(mh[a - 1] - 1) * D2 * D2 : nullptr;
  const double *Qp = (live && a + 1 < T) ? Qall + (size_t)(mh[a + 1] - 1) * D2 * D2 : nullptr;
  for (int e = tid; e < DP * DP; e += nthr) {
    const int i = e / DP, j = e - i * DP;
    double vd = 0.0, v1 = 0.0, v2 = 0.0;
    if (live && j < D) {
      if (i < D) {
        vd = Qa[(size_t)i * D2 + j];                                             // Qss(a)
        if (Qm) {
          const double qdd = Qm[(size_t)(D + i) * D2 + (D + j)];
          vd += 0.25 * qdd;                                                      // + Qdd(a-1)/4
          v1 = 0.5 * Qm[(size_t)(D + i) * D2 + j] - 0.5 * Qa[(size_t)i * D2 + (D + j)];   // Qds(a-1)/2 - Qsd(a)/2
          if (a >= 2) v2 = -0.25 * qdd;                                          // -Qdd(a-1)/4
        }
        if (Qp) vd += 0.25 * Qp[(size_t)(D + i) * D2 + (D + j)];                 // + Qdd(a+1)/4
      } else if (i == D) {
        vd = g[(size_t)a * D2 + j];                                              // r_a = gs(a) + gd(a-1)/2 - gd(a+1)/2
        if (a >= 1) vd += 0.5 * g[(size_t)(a - 1) * D2 + D + j];
        if (a + 1 < T) vd -= 0.5 * g[(size_t)(a + 1) * D2 + D + j];
      }
    }
    Bd[i * LS + j] = vd;
    if (Bm1) Bm1[i * LS + j] = v1;
    if (Bm2) Bm2[i * LS + j] = v2;
  }
}

// Pivot phase, one wave.  B00 rows/cols < D hold S00 (lower triangle valid); on return they hold U = chol(S00)^-1
// (lower triangular, zeros above).  Lane (ti,tj) of the 8 x 8 grid owns elements i = ti + 8 ka, j = tj + 8 kb.
// s: S00 tiles kb <= ka; u: tiles of U' (= the identity rows of the augmented matrix), kb >= ka.
template <int D>
__device__ void blk_pivot(double *B00, double *cb, int lane, int *bad) {
  using C = BlkCfg<D>;
  constexpr int NB = C::NB8, LS = C::LS, DP = C::DP, UO = 8 * NB, PIV = 16 * NB;
  const int ti = lane >> 3, tj = lane & 7;
  double s[NB][NB], u[NB][NB];
#pragma unroll
  for (int ka = 0; ka < NB; ++ka)
#pragma unroll
    for (int kb = 0; kb < NB; ++kb) {
      const int i = ti + 8 * ka, j = tj + 8 * kb;
      s[ka][kb] = (kb <= ka && i < D && j < D) ? B00[i * LS + j] : 0.0;
      u[ka][kb] = (ka == kb && ti == tj && i < D) ? 1.0 : 0.0;
    }
  auto phase = [&](auto kc_tag, int c_lo, int c_hi) {
    constexpr int KC = decltype(kc_tag)::value;
    for (int c = c_lo; c < c_hi; ++c) {
      const int oc = c & 7;
      if (tj == oc) {       // owners publish column c: S rows i > c (finished rows as 0), U' rows i <= c
#pragma unroll
        for (int ka = KC; ka < NB; ++ka) {
          const int i = ti + 8 * ka;
          cb[i] = (i > c) ? s[ka][KC] : 0.0;
        }
        if (ti == oc) cb[PIV] = s[KC][KC];
#pragma unroll
        for (int ka = 0; ka <= KC; ++ka) {
          const int i = ti + 8 * ka;
          cb[UO + i] = (i <= c) ? u[ka][KC] : 0.0;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const double piv = cb[PIV];
      double lr_[NB], lc_[NB], ur_[NB];
#pragma unroll
      for (int k = KC; k < NB; ++k) {
        lr_[k] = cb[ti + 8 * k];
        lc_[k] = cb[tj + 8 * k];
      }
#pragma unroll
      for (int k = 0; k <= KC; ++k) ur_[k] = cb[UO + ti + 8 * k];
      const double urow = (lane <= c) ? cb[UO + lane] : 0.0;
      if (!(piv > 0.0) && lane == 0) *bad = 1;
      const double dinv = traj_rsqrt(piv), winv = dinv * dinv;
      // a_ij -= (a_ic / p) a_jc over the live tiles; finished rows / columns were published as zeros
#pragma unroll
      for (int ka = KC; ka < NB; ++ka) {
        const double f = lr_[ka] * winv;
#pragma unroll
        for (int kb = KC; kb <= ka; ++kb) s[ka][kb] = fma(-f, lc_[kb], s[ka][kb]);
      }
#pragma unroll
      for (int ka = 0; ka <= KC; ++ka) {
        const double f = ur_[ka] * winv;
#pragma unroll
        for (int kb = KC; kb < NB; ++kb) u[ka][kb] = fma(-f, lc_[kb], u[ka][kb]);
      }
      if (lane < DP) B00[c * LS + lane] = urow * dinv;     // row c of U = column c of U', final
      __builtin_amdgcn_wave_barrier();
    }
  };
  phase(std::integral_constant<int, 0>{}, 0, D < 8 ? D : 8);
  if constexpr (NB > 1) phase(std::integral_constant<int, 1>{}, 8, D < 16 ? D : 16);
  if constexpr (NB > 2) phase(std::integral_constant<int, 2>{}, 16, D < 24 ? D : 24);
  if constexpr (NB > 3) phase(std::integral_constant<int, 3>{}, 24, D < 32 ? D : 32);
  if constexpr (NB > 4) phase(std::integral_constant<int, 4>{}, 32, D < 40 ? D : 40);
  if constexpr (NB > 5) phase(std::integral_constant<int, 5>{}, 40, D < 48 ? D : 48);
}

// row tile `it` of  S <- S U'  in place (one wave): the row tile's A fragments are read first, every output tile
// (it, jt) needs only k < 16 (jt + 1) because U is lower triangular.  Columns >= D are stored as zeros.
template <int D>
__device__ __forceinline__ void blk_trsm_rowtile(double *S, const double *U, int it, int lane) {
  using C = BlkCfg<D>;
  constexpr int LS = C::LS, KS = C::KS, NT = C::NT;
  const int lrow = lane & 15, lq = lane >> 4;
  double a[KS];
  const double *xa = S + (16 * it + lrow) * LS + lq;
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) a[ks] = xa[4 * ks];
#pragma unroll
  for (int jt = 0; jt < NT; ++jt) {
    const double *yb = U + (16 * jt + lrow) * LS + lq;
    blk_d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
      if (ks < 4 * (jt + 1)) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ks], yb[4 * ks], acc, 0, 0, 0);
    const int col = 16 * jt + lrow;
#pragma unroll
    for (int r = 0; r < 4; ++r) S[(16 * it + 4 * r + lq) * LS + col] = (col < D) ? acc[r] : 0.0;
  }
}

// tile (it, jt) of  Cm -= X Y'  (k over the D columns).  Columns >= D are stored as zeros.
template <int D>
__device__ __forceinline__ void blk_update_tile(double *Cm, const double *X, const double *Y, int it, int jt, int lane) {
  using C = BlkCfg<D>;
  constexpr int LS = C::LS, KS = C::KS;
  const int lrow = lane & 15, lq = lane >> 4;
  double *cp = Cm + (16 * it + lq) * LS + 16 * jt + lrow;
  blk_d4 acc;
#pragma unroll
  for (int r = 0; r < 4; ++r) acc[r] = cp[4 * r * LS];
  const double *xa = X + (16 * it + lrow) * LS + lq, *yb = Y + (16 * jt + lrow) * LS + lq;
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-xa[4 * ks], yb[4 * ks], acc, 0, 0, 0);
  const bool keep = 16 * jt + lrow < D;
#pragma unroll
  for (int r = 0; r < 4; ++r) cp[4 * r * LS] = keep ? acc[r] : 0.0;
}

// Back substitution from the panels in the HBM workspace: y_t = U' (z0 - L10' y_{t+1} - L20' y_{t+2}).
// Panels are double-buffered in LDS (through registers, one panel ahead); both products are split over the four
// waves (lane = column, wave = quarter of the rows) and summed in fixed order.
template <int D>
__device__ void blk_backsub(const double *__restrict__ ws, int T, double *buf, double *yring, double *part,
                            double *__restrict__ Y) {
  using C = BlkCfg<D>;
  constexpr size_t PAN = C::PAN;
  constexpr int NPRE = (int)((PAN + 255) / 256);
  constexpr int OU = 0, OL1 = D * D, OZ = 2 * D * D, OL2 = (2 * D + 1) * D;
  const int tid = threadIdx.x, j = tid & 63, p = tid >> 6;
  for (int i = tid; i < 2 * D; i += 256) yring[i] = 0.0;
  {
    const double *pan = ws + (size_t)(T - 1) * PAN;
    for (size_t e = tid; e < PAN; e += 256) buf[((T - 1) & 1) * PAN + e] = pan[e];
  }
  __syncthreads();
  constexpr int R1 = (2 * D + 3) / 4;      // rows of [L10; L20] per wave
  constexpr int R2 = (D + 3) / 4;          // rows of U per wave
  for (int t = T - 1; t >= 0; --t) {
    const double *pb = buf + (size_t)(t & 1) * PAN;
    double *pn = buf + (size_t)((t + 1) & 1) * PAN;       // receives panel t-1
    double pre[NPRE];
    if (t > 0) {
      const double *pan = ws + (size_t)(t - 1) * PAN;
#pragma unroll
      for (int k = 0; k < NPRE; ++k) {
        const size_t e = tid + (size_t)k * 256;
        pre[k] = (e < PAN) ? pan[e] : 0.0;
      }
    }
    double *y1 = yring + ((t + 1) & 1) * D, *y2 = yring + (t & 1) * D;   // y_{t+1}, y_{t+2}
    if (j < D) {
      double sacc = (p == 0) ? pb[OZ + j] : 0.0;
      const int r_lo = p * R1, r_hi = (r_lo + R1 < 2 * D) ? r_lo + R1 : 2 * D;
      for (int r = r_lo; r < r_hi; ++r) {
        const double l = (r < D) ? pb[OL1 + r * D + j] : pb[OL2 + (r - D) * D + j];
        const double yv = (r < D) ? y1[r] : y2[r - D];
        sacc = fma(-l, yv, sacc);
      }
      part[p * 64 + j] = sacc;
    }
    __syncthreads();
    if (tid < D) part[256 + tid] = ((part[tid] + part[64 + tid]) + part[128 + tid]) + part[192 + tid];   // w
    __syncthreads();
    if (j < D) {
      double sacc = 0.0;
      const int r_lo = p * R2, r_hi = (r_lo + R2 < D) ? r_lo + R2 : D;
      for (int r = (r_lo > j ? r_lo : j); r < r_hi; ++r) sacc = fma(pb[OU + r * D + j], part[256 + r], sacc);
      part[p * 64 + j] = sacc;
    }
    __syncthreads();
    if (tid < D) {
      const double yv = ((part[tid] + part[64 + tid]) + part[128 + tid]) + part[192 + tid];
      y2[tid] = yv;                       // becomes y_t; the slot of y_{t+2} is free now
      Y[(size_t)t * D + tid] = yv;        // reshape(y, D, T), src/trajectory_gmmmap.jl:109
    }
    if (t > 0) {
#pragma unroll
      for (int k = 0; k < NPRE; ++k) {
        const size_t e = tid + (size_t)k * 256;
        if (e < PAN) pn[e] = pre[k];
      }
    }
    __syncthreads();
  }
}

template <int D>
__global__ void __launch_bounds__(256)
traj_solve_blk_kernel(const TrajUtt *__restrict__ utts, int n, const double *__restrict__ Qall,
                      const int64_t *__restrict__ mhat_all, const double *__restrict__ g_all, double *__restrict__ ws_all,
                      int64_t ws_stride, int *__restrict__ status) {
  using C = BlkCfg<D>;
  constexpr int D2 = 2 * D, DP = C::DP, LS = C::LS, NT = C::NT, BUF = C::BUF;
  constexpr size_t PAN = C::PAN;
  constexpr int NLOW = NT * (NT + 1) / 2, NJOB = 2 * NLOW + NT * NT;
  extern __shared__ double sm[];
  double *cb = sm + (size_t)6 * BUF;
  double *yring = cb + C::CB;
  double *part = yring + 2 * D;          // [2 * 256]
  __shared__ int bad;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

  for (int u = blockIdx.x; u < n; u += gridDim.x) {
    const TrajUtt U = utts[u];
    const int T = U.T;
    if (T == 0) continue;
    const int64_t *mh = mhat_all + U.frame0;
    const double *g = g_all + U.frame0 * D2;
    double *ws = ws_all + (size_t)blockIdx.x * ws_stride;
    double *b00 = sm, *b10 = sm + BUF, *b20 = sm + 2 * BUF, *b11 = sm + 3 * BUF, *b21 = sm + 4 * BUF, *b22 = sm + 5 * BUF;
    if (tid == 0) bad = 0;
    for (int e = tid; e < C::CB; e += 256) cb[e] = 0.0;
    blk_assemble<D>(nullptr, nullptr, b00, 0, T, mh, g, Qall, tid, 256);
    blk_assemble<D>(nullptr, b10, b11, 1, T, mh, g, Qall, tid, 256);
    blk_assemble<D>(b20, b21, b22, 2, T, mh, g, Qall, tid, 256);
    __syncthreads();

    BLK_PROF_T0();
    for (int t = 0; t < T; ++t) {
      if (wave == 0) {
        if (lane < DP) {               // r0 under S10 and S20: row D of L10 / L20 becomes z0 = U r0
          const double v = (lane < D) ? b00[D * LS + lane] : 0.0;
          b10[D * LS + lane] = v;
          b20[D * LS + lane] = v;
        }
        blk_pivot<D>(b00, cb, lane, &bad);
      }
      __syncthreads();
      BLK_PROF(0);
      for (int job = wave; job < 2 * NT; job += 4) blk_trsm_rowtile<D>(job < NT ? b10 : b20, b00, job < NT ? job : job - NT, lane);
      __syncthreads();
      BLK_PROF(1);
      for (int job = wave; job < NJOB; job += 4) {
        // S11 -= L10 L10' (lower tiles), S21 -= L20 L10' (all tiles), S22 -= L20 L20' (lower tiles)
        double *Cm;
        const double *X, *Y;
        int q = job, it, jt;
        if (q < NLOW) { Cm = b11; X = b10; Y = b10; }
        else if (q < NLOW + NT * NT) { q -= NLOW; Cm = b21; X = b20; Y = b10; }
        else { q -= NLOW + NT * NT; Cm = b22; X = b20; Y = b20; }
        if (Cm == b21) { it = q / NT; jt = q - it * NT; }
        else { it = 0; while (q > it) { q -= it + 1; ++it; } jt = q; }
        blk_update_tile<D>(Cm, X, Y, it, jt, lane);
      }
      __syncthreads();
      BLK_PROF(2);
      double *pan = ws + (size_t)t * PAN;
      for (int e = tid; e < D * D; e += 256) {
        const int i = e / D, j = e - i * D;
        pan[e] = b00[i * LS + j];
        pan[(size_t)(2 * D + 1) * D + e] = b20[i * LS + j];
      }
      for (int e = tid; e < (D + 1) * D; e += 256) {
        const int i = e / D, j = e - i * D;
        pan[(size_t)D * D + e] = b10[i * LS + j];
      }
      __syncthreads();
      BLK_PROF(3);
      {   // the window moves by one block: pointer rotation, the three freed buffers receive block row t+3
        double *f0 = b00, *f1 = b10, *f2 = b20;
        b00 = b11; b10 = b21; b11 = b22;
        b20 = f0; b21 = f1; b22 = f2;
      }
      blk_assemble<D>(b20, b21, b22, t + 3, T, mh, g, Qall, tid, 256);
      __syncthreads();
      BLK_PROF(4);
    }
    blk_backsub<D>(ws, T, sm, yring, part, U.Y);
    BLK_PROF(5);
    if (tid == 0 && bad) status[0] = 1;
    __syncthreads();
  }
}
