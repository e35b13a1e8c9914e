// traj_solve_blk.hpp -- blocked banded Cholesky solve of the trajectory normal equations on v_mfma_f64_16x16x4.
// Included by traj.hip inside namespace vcmi (needs TrajUtt and traj_rsqrt).
//
// Solves (W' Dy^-1 W) y = W' Dy^-1 E of reference src/trajectory_gmmmap.jl:103-105 for one utterance per workgroup.
// The matrix P is block-pentadiagonal with D x D blocks (see the header of traj.hip).  Block step t works on the
// window   S00            r0        (block rows t, t+1, t+2; S20 and S22 are still the raw assembled blocks)
//          S10 S11        r1
//          S20 S21 S22    r2
//   1. pivot (two waves, no workgroup barrier inside): right-looking Cholesky of S00 on an 8 x 8 lane grid (wave S) and
//      the identity rows of [S00; I] under the same column operations (wave U), which leave U = L00^-1 explicit and
//      lower triangular -- the only part of the algorithm that is sequential in the scalar columns;
//   2. L10 = S10 U', L20 = S20 U'                                  (MFMA, in place; U' is triangular: k-steps skipped)
//   3. S11 -= L10 L10', S21 -= L20 L10', S22 -= L20 L20'          (MFMA, lower tiles only for the symmetric ones)
//   4. the panel [M1 = L10 U (last row: h = U' z0); M2 = L20 U] goes to the HBM workspace, the window shifts by pointer
//      rotation and block row t+3 is assembled from the stencil.
// Only L10 and the S11 update feed the next pivot; everything else of step t runs one step late on the other two waves,
// beside pivot(t+1) (schedule: see the kernel).
// The right-hand side needs no code of its own: r_b lives in row D (a padding row of the 16-row tiles) of S_bb, a copy
// of r0 is put in row D of S10 and S20, and then row D of L10 / L20 is z0 = U r0 and the updates of step 3 apply
// r1 -= L10 z0, r2 -= L20 z0 to row D of S11 / S22.
// Back substitution: y_t = U' (z0 - L10' y_{t+1} - L20' y_{t+2}) = h - M1' y_{t+1} - M2' y_{t+2} -- one matrix-vector
// product per step, no sequential chain.
// Cycle counters per phase: build with -DTRAJ_BLK_PROF (make EXTRA=-DTRAJ_BLK_PROF), printed to stderr per call.
#pragma once

template <int D>
struct BlkCfg {
  static constexpr int DP = ((D + 1 + 15) / 16) * 16;   // rows / columns of a block buffer: 16-row tiles incl. the rhs row D
  static constexpr int NT = DP / 16;                    // tiles per dimension
  static constexpr int KS = (D + 3) / 4;                // k-steps of a product over the D columns
  static constexpr int LS = DP + 2;                     // row stride in doubles: LS/2 odd -> MFMA operand reads conflict-free
  static constexpr int BUF = DP * LS;                   // doubles per block buffer
  static constexpr int NB8 = (D + 7) / 8;               // 8 x 8 lane-grid tiles per dimension (pivot phase)
  static constexpr int CB = 64;                         // wave U's published column of U', permuted
  static constexpr int RING = D * 64;                   // wave S's published columns: one 64-double slot each (entries 6, 7 of a lane group: pivot, tag)
  static constexpr int JOINCOL = D < 11 ? D - 1 : 9;   // pivot column after which the pivot waves join the mid-phase barrier
  static constexpr size_t PAN = (size_t)(2 * D + 1) * D;   // panel doubles per block step: M1 = L10 U (D+1,D) incl. h = U' z0, M2 = L20 U (D,D)
  static constexpr size_t WORK = 6 * (size_t)BUF > 3 * PAN ? 6 * (size_t)BUF : 3 * PAN;   // six window buffers / three staged panels
  static constexpr size_t lds_doubles = WORK + CB + RING + 2 * D + 768 + 2;
};

typedef double blk_d4 __attribute__((ext_vector_type(4)));

#ifdef TRAJ_BLK_PROF
__device__ long long blk_prof[16];   // cycles of workgroup 0 per phase (printed by traj_check_status)
#define BLK_PROF_T0() long long pt_ = (long long)__builtin_readcyclecounter()
#define BLK_PROF(k)                                                     \
  do {                                                                  \
    const long long n_ = (long long)__builtin_readcyclecounter();       \
    if (blockIdx.x == 0 && threadIdx.x == 0) blk_prof[k] += n_ - pt_;   \
    pt_ = n_;                                                           \
  } while (0)
// time since the last BLK_PROF mark, seen by thread `thr` (does not move the mark)
#define BLK_PROF_AT(k, thr)                                                                                         \
  do {                                                                                                              \
    if (blockIdx.x == 0 && threadIdx.x == (thr)) blk_prof[k] += (long long)__builtin_readcyclecounter() - pt_;      \
  } while (0)
#else
#define BLK_PROF_T0()
#define BLK_PROF(k)
#define BLK_PROF_AT(k, thr)
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

// Pivot phase, TWO waves.  B00 rows/cols < D hold S00 (lower triangle valid); on return they hold U = chol(S00)^-1
// (lower triangular, zeros above).  Both waves use an 8 x 8 lane grid: lane (ti,tj) owns elements i = ti + 8 ka,
// j = tj + 8 kb of its matrix.
//   wave S (blk_pivot_s): right-looking Cholesky of S00, tiles kb <= ka.  Per column the owners publish it -- rows
//     i > c, finished rows as zeros -- into slot c of a ring in LDS, in a permuted order (row i at (i%8)*8 + i/8) so
//     that the values a lane needs are contiguous (16-byte LDS accesses); the pivot itself travels by v_readlane, so
//     its rsqrt chain runs while the column is on its way through LDS.  Then one lane bumps a counter in LDS.
//   wave U (blk_pivot_u): the identity rows of the augmented matrix [S00; I], tiles kb >= ka: the same column
//     operations turn them into U' = L00^-T.  It follows wave S through the ring (LDS operations of a wave execute in
//     order, so a column is visible before the counter that announces it) and leaves row c of U in B00 after column c.
// A lone wave issues one FP64 instruction per 5.6 cycles (10 if dependent) and sees ~84 cycles per LDS round trip
// (tools/microbench_wave.hip): a column costs ~670 cycles in one wave; split like this each wave carries half.
// Both waves join one workgroup barrier on the way (after column JOINCOL): the two other waves use it to order
// their deferred work of the previous block step.
typedef double pv_d2 __attribute__((ext_vector_type(2)));

template <int D>
__device__ void blk_pivot_s(const double *B00, double *ring, int fbase, int lane, int *bad) {
  using C = BlkCfg<D>;
  constexpr int NB = C::NB8, LS = C::LS, NQ = (NB + 1) / 2;
  const int ti = lane >> 3, tj = lane & 7;
  double s[NB][NB];
#pragma unroll
  for (int ka = 0; ka < NB; ++ka)
#pragma unroll
    for (int kb = 0; kb < NB; ++kb) {
      const int i = ti + 8 * ka, j = tj + 8 * kb;
      s[ka][kb] = (kb <= ka && i < D && j < D) ? B00[i * LS + j] : 0.0;
    }
  bool notpd = false;
  auto phase = [&](auto kc_tag, int c_lo, int c_hi) {
    constexpr int KC = decltype(kc_tag)::value;
    constexpr int Q0 = KC / 2;
#pragma nounroll
    for (int c = c_lo; c < c_hi; ++c) {
      const int oc = c & 7;
      const double pvl = s[KC][KC];
      const double piv = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(pvl), 9 * oc),
                                          __builtin_amdgcn_readlane(__double2loint(pvl), 9 * oc));
      pv_d2 *slot_w = reinterpret_cast<pv_d2 *>(ring + c * 64 + ti * 8);
      const pv_d2 *slot_c = reinterpret_cast<const pv_d2 *>(ring + c * 64 + tj * 8);
      if (tj == oc) {       // owners publish column c: rows i > c, finished rows as 0
#pragma unroll
        for (int q = Q0; q < NQ; ++q) {
          const int k0 = 2 * q, k1 = 2 * q + 1;
          pv_d2 v;
          v.x = (k0 >= KC && ti + 8 * k0 > c) ? s[k0][KC] : 0.0;
          v.y = (k1 < NB && ti + 8 * k1 > c) ? s[k1 < NB ? k1 : 0][KC] : 0.0;
          slot_w[q] = v;
        }
        // the pivot and the column's sequence tag, in a later instruction than the column itself (the LDS executes a
        // wave's operations in order): wave U polls the tag of lane group ti = 0
        slot_w[3] = pv_d2{piv, (double)(fbase + c + 1)};
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      double lr_[2 * NQ], lc_[2 * NQ];
#pragma unroll
      for (int q = Q0; q < NQ; ++q) {
        const pv_d2 x = slot_w[q], y = slot_c[q];
        lr_[2 * q] = x.x;
        lr_[2 * q + 1] = x.y;
        lc_[2 * q] = y.x;
        lc_[2 * q + 1] = y.y;
      }
      notpd |= !(piv > 0.0);
      const double dinv = traj_rsqrt(piv), winv = dinv * dinv;
      // a_ij -= (a_ic / p) a_jc over the live tiles; finished rows / columns were published as zeros
#pragma unroll
      for (int ka = KC; ka < NB; ++ka) {
        const double f = lr_[ka] * winv;
#pragma unroll
        for (int kb = KC; kb <= ka; ++kb) s[ka][kb] = fma(-f, lc_[kb], s[ka][kb]);
      }
      __builtin_amdgcn_wave_barrier();
      if (c == C::JOINCOL) __syncthreads();    // the other waves' L20 products are complete (see the kernel)
    }
  };
  phase(std::integral_constant<int, 0>{}, 0, D < 8 ? D : 8);
  if constexpr (NB > 1) phase(std::integral_constant<int, 1>{}, 8, D < 16 ? D : 16);
  if constexpr (NB > 2) phase(std::integral_constant<int, 2>{}, 16, D < 24 ? D : 24);
  if constexpr (NB > 3) phase(std::integral_constant<int, 3>{}, 24, D < 32 ? D : 32);
  if constexpr (NB > 4) phase(std::integral_constant<int, 4>{}, 32, D < 40 ? D : 40);
  if constexpr (NB > 5) phase(std::integral_constant<int, 5>{}, 40, D < 48 ? D : 48);
  if (notpd && lane == 0) *bad = 1;
}

template <int D>
__device__ void blk_pivot_u(double *B00, double *ring, int fbase, double *cbu,
                            int lane) {
  using C = BlkCfg<D>;
  constexpr int NB = C::NB8, LS = C::LS, DP = C::DP, NQ = (NB + 1) / 2;
  const int ti = lane >> 3, tj = lane & 7;
  double u[NB][NB];
#pragma unroll
  for (int ka = 0; ka < NB; ++ka)
#pragma unroll
    for (int kb = 0; kb < NB; ++kb) u[ka][kb] = (ka == kb && ti == tj && ti + 8 * ka < D) ? 1.0 : 0.0;
  pv_d2 *cbu_w = reinterpret_cast<pv_d2 *>(cbu + ti * 8);
  const double *cbu_row = cbu + (lane & 7) * 8 + (lane >> 3);
  auto phase = [&](auto kc_tag, int c_lo, int c_hi) {
    constexpr int KC = decltype(kc_tag)::value;
    constexpr int Q0 = KC / 2;
#pragma nounroll
    for (int c = c_lo; c < c_hi; ++c) {
      const int oc = c & 7;
      if (tj == oc) {       // owners publish column c of U': rows i <= c (needs nothing from wave S)
#pragma unroll
        for (int q = 0; q <= Q0; ++q) {
          const int k0 = 2 * q, k1 = 2 * q + 1;
          pv_d2 v;
          v.x = (ti + 8 * k0 <= c) ? u[k0][KC] : 0.0;
          v.y = (k1 <= KC && ti + 8 * k1 <= c) ? u[k1 <= KC ? k1 : 0][KC] : 0.0;
          cbu_w[q] = v;
        }
      }
      {                                          // column c of S00 published?  (its tag is unique per block step and column)
        unsigned long long *tagp = reinterpret_cast<unsigned long long *>(ring + c * 64 + 7);
        const unsigned long long want = (unsigned long long)__double_as_longlong((double)(fbase + c + 1));
        while (__hip_atomic_load(tagp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != want) __builtin_amdgcn_s_sleep(1);
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      const pv_d2 *slot_c = reinterpret_cast<const pv_d2 *>(ring + c * 64 + tj * 8);
      const double piv = ring[c * 64 + 6];
      double lc_[2 * NQ], ur_[2 * NQ];
#pragma unroll
      for (int q = Q0; q < NQ; ++q) {
        const pv_d2 y = slot_c[q];
        lc_[2 * q] = y.x;
        lc_[2 * q + 1] = y.y;
      }
#pragma unroll
      for (int q = 0; q <= Q0; ++q) {
        const pv_d2 x = cbu_w[q];
        ur_[2 * q] = x.x;
        ur_[2 * q + 1] = x.y;
      }
      const double urow = (lane <= c) ? *cbu_row : 0.0;
      const double dinv = traj_rsqrt(piv), winv = dinv * dinv;
#pragma unroll
      for (int ka = 0; ka <= KC; ++ka) {
        const double f = ur_[ka] * winv;
#pragma unroll
        for (int kb = KC; kb < NB; ++kb) u[ka][kb] = fma(-f, lc_[kb], u[ka][kb]);
      }
      if (lane < DP) B00[c * LS + lane] = urow * dinv;     // row c of U = column c of U', final
      __builtin_amdgcn_wave_barrier();
      if (c == C::JOINCOL) __syncthreads();
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

// tile (it, jt) of  Cm -= X Y'  (k over the D columns; even and odd k-steps in two accumulators: two independent
// MFMA chains).  Columns >= D are stored as zeros.
template <int D>
__device__ __forceinline__ void blk_update_tile(double *Cm, const double *X, const double *Y, int it, int jt, int lane) {
  using C = BlkCfg<D>;
  constexpr int LS = C::LS, KS = C::KS;
  const int lrow = lane & 15, lq = lane >> 4;
  double *cp = Cm + (16 * it + lq) * LS + 16 * jt + lrow;
  blk_d4 acc, acc2 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int r = 0; r < 4; ++r) acc[r] = cp[4 * r * LS];
  const double *xa = X + (16 * it + lrow) * LS + lq, *yb = Y + (16 * jt + lrow) * LS + lq;
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    if (ks & 1) acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(-xa[4 * ks], yb[4 * ks], acc2, 0, 0, 0);
    else acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-xa[4 * ks], yb[4 * ks], acc, 0, 0, 0);
  }
  const bool keep = 16 * jt + lrow < D;
#pragma unroll
  for (int r = 0; r < 4; ++r) cp[4 * r * LS] = keep ? acc[r] + acc2[r] : 0.0;
}

// tiles (it, 0 .. njt-1) of  Cm -= X Y'  with the A fragments of row tile `it` of X read once for the whole row group
// (a third fewer LDS reads than tile by tile; the column tiles are independent MFMA chains).
template <int D>
__device__ __forceinline__ void blk_update_rowgroup(double *Cm, const double *X, const double *Y, int it, int njt, int lane) {
  using C = BlkCfg<D>;
  constexpr int LS = C::LS, KS = C::KS, NT = C::NT;
  const int lrow = lane & 15, lq = lane >> 4;
  double a[KS];
  const double *xa = X + (16 * it + lrow) * LS + lq;
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) a[ks] = -xa[4 * ks];
#pragma unroll
  for (int jt = 0; jt < NT; ++jt) {
    if (jt < njt) {
      double *cp = Cm + (16 * it + lq) * LS + 16 * jt + lrow;
      const double *yb = Y + (16 * jt + lrow) * LS + lq;
      blk_d4 acc, acc2 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[r] = cp[4 * r * LS];
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        if (ks & 1) acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ks], yb[4 * ks], acc2, 0, 0, 0);
        else acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ks], yb[4 * ks], acc, 0, 0, 0);
      }
      const bool keep = 16 * jt + lrow < D;
#pragma unroll
      for (int r = 0; r < 4; ++r) cp[4 * r * LS] = keep ? acc[r] + acc2[r] : 0.0;
    }
  }
}

// row tile `it` of  L U  (U lower triangular: column tile jt needs only k >= 16 jt) straight from the accumulators to
// the HBM panel `out` (row-major, D columns): rows < nrows, columns < D.  The A fragments of the row tile are read
// once; the column tiles are independent MFMA chains.
template <int D>
__device__ __forceinline__ void blk_lu_rowtile_to_panel(const double *L, const double *U, int it, int nrows,
                                                        double *__restrict__ out, int lane) {
  using C = BlkCfg<D>;
  constexpr int LS = C::LS, KS = C::KS, NT = C::NT;
  const int lrow = lane & 15, lq = lane >> 4;
  double a[KS];
  const double *xa = L + (16 * it + lrow) * LS + lq;
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) a[ks] = xa[4 * ks];
  blk_d4 acc[NT];
#pragma unroll
  for (int jt = 0; jt < NT; ++jt) acc[jt] = blk_d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int ks = 0; ks < KS; ++ks)
#pragma unroll
    for (int jt = 0; jt < NT; ++jt)
      if (ks >= 4 * jt) acc[jt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ks], U[(4 * ks + lq) * LS + 16 * jt + lrow], acc[jt], 0, 0, 0);
#pragma unroll
  for (int jt = 0; jt < NT; ++jt) {
    const int col = 16 * jt + lrow;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 16 * it + 4 * r + lq;
      if (col < D && row < nrows) out[row * D + col] = acc[jt][r];
    }
  }
}

// Back substitution from the panels in the HBM workspace:
//   y_t = U' (z0 - L10' y_{t+1} - L20' y_{t+2}) = h - M1' y_{t+1} - M2' y_{t+2},   M1 = L10 U, M2 = L20 U, h = U' z0
// (formed by the deferred team of the factorisation): one product per step, split over the four waves (lane = column,
// wave = quarter of the rows) and summed in fixed order.  The loop is bound by the HBM read of the panels (25.9 KB per
// step at D = 40), so they are requested two steps ahead: global -> registers at the start of a step (two register sets
// alternate), registers -> one of three LDS slots at the end of the next one.  The results are collected in LDS and
// written out every 8 steps, followed by an explicit wait: a global store pending beside the panel loads would make
// every wait of the loop a wait for all of them (loads and stores share the vmcnt counter).
template <int D>
__device__ void blk_backsub(const double *__restrict__ ws, int T, double *buf, double *yring, double *part,
                            double *__restrict__ Y) {
  using C = BlkCfg<D>;
  constexpr int PAN = (int)C::PAN;
  constexpr int NPRE = (PAN + 255) / 256;
  constexpr int OH = D * D, OM2 = (D + 1) * D;
  constexpr int YB = 8;                    // steps per result flush
  const int tid = threadIdx.x, j = tid & 63, p = tid >> 6;
  double *ybuf = part + 320;               // [YB][D]
  auto fetch = [&](int t, double (&regs)[NPRE]) {
    const double *pan = ws + (size_t)t * PAN;
#pragma unroll
    for (int k = 0; k < NPRE; ++k) {
      const int e = tid + k * 256;
      regs[k] = pan[e < PAN ? e : 0];
    }
  };
  auto stage = [&](int t, const double (&regs)[NPRE]) {
    double *dst = buf + (size_t)(t % 3) * PAN;
#pragma unroll
    for (int k = 0; k < NPRE; ++k) {
      const int e = tid + k * 256;
      if (e < PAN) dst[e] = regs[k];
    }
  };
  double ra[NPRE], rb[NPRE];
  for (int i = tid; i < 2 * D; i += 256) yring[i] = 0.0;
  fetch(T - 1, ra);
  stage(T - 1, ra);
  if (T >= 2) fetch(T - 2, ra);
  __syncthreads();
  constexpr int R1 = (2 * D + 3) / 4;      // rows of [M1; M2] per wave
  auto step = [&](int t, double (&cur)[NPRE], double (&nxt)[NPRE]) {
    // `cur` holds panel t-1 (requested one step ago), `nxt` receives panel t-2
    if (t >= 2) fetch(t - 2, nxt);
    const double *pb = buf + (size_t)(t % 3) * PAN;
    double *y1 = yring + ((t + 1) & 1) * D, *y2 = yring + (t & 1) * D;   // y_{t+1}, y_{t+2}
    const int jc = j < D ? j : D - 1;
    {   // partial sums of h - [M1; M2]' [y_{t+1}; y_{t+2}]: compile-time trip count, all LDS reads issued up front
      double s0 = (p == 0) ? pb[OH + jc] : 0.0, s1 = 0.0;
      const int r_lo = p * R1;
#pragma unroll
      for (int q = 0; q < R1; ++q) {
        const int r = r_lo + q, rc = r < 2 * D ? r : 0;                  // wave-uniform
        const double *row = rc < D ? pb + rc * D : pb + OM2 + (rc - D) * D;
        const double l = row[jc], yv = rc < D ? y1[rc] : y2[rc - D];
        if (q & 1) s1 = fma(r < 2 * D ? -l : 0.0, yv, s1);
        else s0 = fma(r < 2 * D ? -l : 0.0, yv, s0);
      }
      part[p * 64 + j] = s0 + s1;
    }
    __syncthreads();
    if (tid < D) {
      const double yv = ((part[tid] + part[64 + tid]) + part[128 + tid]) + part[192 + tid];
      y2[tid] = yv;                       // becomes y_t; the slot of y_{t+2} is free now
      ybuf[(t & (YB - 1)) * D + tid] = yv;
    }
    if (t >= 1) stage(t - 1, cur);
    __syncthreads();
    if ((t & (YB - 1)) == 0) {            // rows t .. t+YB-1 of reshape(y, D, T), src/trajectory_gmmmap.jl:109
      const int nrow = (T - t < YB) ? T - t : YB;
      for (int e = tid; e < nrow * D; e += 256) Y[(size_t)t * D + e] = ybuf[e];
      __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): no store is pending when the loop goes on
    }
  };
  for (int t = T - 1; t >= 0; t -= 2) {
    step(t, ra, rb);
    if (t >= 1) step(t - 1, rb, ra);
  }
}

// Workgroup schedule of block step t (4 waves, one per SIMD; the scalar-column pivot is the critical path, so
// everything that does not feed the next pivot is deferred by one step and runs beside it):
//   phase 1   waves 0, 1: pivot(t) (Cholesky of S00 / its inverse U)
//             waves 2, 3: L20(t-1) | S21, S22 updates (t-1) | panel t-1 = [L10 U; L20 U] -> HBM, block row t+2 of the stencil
//                         fetched and, after the barrier, written into the three freed buffers
//   phase 2   all:        L10(t) = S10 U', then S11 -= L10 L10'   (-> S00 of step t+1)
// Buffers: b00, b10, b11 (window of step t) and p0, p1, p2 = U, L10, S20 -> L20 of step t-1, then block row t+2.
template <int D>
__global__ void __launch_bounds__(256)
traj_solve_blk_kernel(const TrajUtt *__restrict__ utts, int n, const double *__restrict__ Qall,
                      const int64_t *__restrict__ mhat_all, const double *__restrict__ g_all, double *__restrict__ ws_all,
                      int64_t ws_stride, int *__restrict__ status) {
  using C = BlkCfg<D>;
  constexpr int D2 = 2 * D, DP = C::DP, LS = C::LS, NT = C::NT, BUF = C::BUF;
  constexpr size_t PAN = C::PAN;
  constexpr int NLOW = NT * (NT + 1) / 2;
  constexpr int NDW = 2, NDT = 64 * NDW;   // deferred team
  constexpr int VW = (D % 2 == 0) ? 2 : 1;                 // elements per access: 16-byte LDS / global accesses when D is even
  constexpr int NIT = (D * D / VW + NDT - 1) / NDT;        // element groups per thread
  constexpr int RT = D / 16;             // row tile that holds the rhs row D
  extern __shared__ __attribute__((aligned(16))) double blk_sm[];
  double *const sm = blk_sm;             // 16-byte LDS accesses in the pivot phase
  // the pivot pair's exchange areas first: their DS offsets then fit the instructions' 16-bit immediates (behind the
  // 115 KB of window buffers every access needed its own v_add, ~8 per pivot column)
  double *cbu = sm;
  double *ring = cbu + C::CB;
  double *yring = ring + C::RING;
  double *part = yring + 2 * D;          // [768]
  int *flags = reinterpret_cast<int *>(part + 768);   // [0] not-PD flag
  int &bad = flags[0];
  double *const wk = part + 770;         // six window buffers / three staged panels
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

  // per-thread element tables of the deferred team (the same D x D elements every block step): LDS offset, offset in
  // a mixture's Q matrix, offset in a panel block -- a lone wave pays ~6 cycles per VALU instruction, so index
  // arithmetic is kept out of the step loop
  const int dt = tid - 64 * (4 - NDW);
  int eo[NIT], eq[NIT];
#pragma unroll
  for (int k = 0; k < NIT; ++k) {
    const int e = VW * (dt + NDT * k), ec = (e >= 0 && e < D * D) ? e : 0, i = ec / D, j = ec - i * D;
    eo[k] = (e >= 0 && e < D * D) ? i * LS + j : -1;
    eq[k] = i * D2 + j;
  }
  typedef double ev_t __attribute__((ext_vector_type(VW)));

  for (int u = blockIdx.x; u < n; u += gridDim.x) {
    const TrajUtt U = utts[u];
    const int T = U.T;
    if (T == 0) continue;
    const int64_t *mh = mhat_all + U.frame0;
    const double *g = g_all + U.frame0 * D2;
    double *ws = ws_all + (size_t)blockIdx.x * ws_stride;
    double *b00 = wk, *b10 = wk + BUF, *b11 = wk + 2 * BUF, *p0 = wk + 3 * BUF, *p1 = wk + 4 * BUF, *p2 = wk + 5 * BUF;
    if (tid == 0) {
      flags[0] = 0;
      flags[1] = 0;
    }
    for (int e = tid; e < C::CB + C::RING; e += 256) cbu[e] = 0.0;   // cbu and the ring (stale tags of the previous utterance)
    // p0, p1, p2 receive only D x D elements (+ the rhs row) per step: their padding stays zero from here on
    for (int e = tid; e < 3 * BUF; e += 256) p0[e] = 0.0;
    blk_assemble<D>(nullptr, nullptr, b00, 0, T, mh, g, Qall, tid, 256);
    blk_assemble<D>(nullptr, b10, b11, 1, T, mh, g, Qall, tid, 256);
    __syncthreads();
    BLK_PROF_T0();

    for (int t = 0; t <= T; ++t) {
      // ---------------- phase 1 ----------------
      ev_t vd[NIT], v1[NIT], v2[NIT];     // block row t+2 of the stencil, written to p0, p1, p2 during phase 2 (deferred team)
      double rv = 0.0;
      if (wave < 4 - NDW) {
        if (t < T) {
          if (wave == 0) blk_pivot_s<D>(b00, ring, t * D, lane, &bad);
          else blk_pivot_u<D>(b00, ring, t * D, cbu, lane);
          BLK_PROF_AT(3, 0);
          BLK_PROF_AT(4, 64);
        } else {
          __syncthreads();
        }
        __syncthreads();                  // end of phase 1
      } else {
        const int dw = wave - (4 - NDW);
        // mixtures of block rows t+1, t+2, t+3 (clamped), loaded before the products so that the stencil loads below
        // do not wait for them
        const int a = t + 2;
        const bool live = a < T, hasp = a + 1 < T;
        const int ac = live ? a : T - 1, am = ac >= 1 ? ac - 1 : 0, ap = hasp ? a + 1 : ac;
        const int64_t mxa = mh[ac], mxm = mh[am], mxp = mh[ap];
        const bool defer = t >= 1;
        if (defer) {                      // L20(t-1) = S20 U' in place; row D of S20 := r0(t-1) -> row D of L20 = z0
          for (int it = dw; it < NT; it += NDW) {
            if (it == RT) {
              if (lane < DP) p2[D * LS + lane] = (lane < D) ? p0[D * LS + lane] : 0.0;
              __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
              __builtin_amdgcn_wave_barrier();
              __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
            blk_trsm_rowtile<D>(p2, p0, it, lane);
          }
        }
        BLK_PROF_AT(8, 128);
        __syncthreads();
        BLK_PROF_AT(9, 128);
        if (defer) {                      // S21 -= L20 L10' (all tiles), S22 -= L20 L20' (lower tiles), row group by row group
          for (int it = 0; it < NT; ++it) {
            if ((it % NDW) == dw) blk_update_rowgroup<D>(b10, p2, p1, it, NT, lane);
            if (((it + 1) % NDW) == dw) blk_update_rowgroup<D>(b11, p2, p2, it, it + 1, lane);
          }
        }
        BLK_PROF_AT(6, 128);
        // Block row a = t+2 of the stencil.  Every load is unconditional on a clamped address (a select on a loaded
        // value would make the wave wait for each load in turn); masks are applied when the operands are combined.
        ev_t q0[NIT], q1[NIT], q2[NIT], q3[NIT], q4[NIT];
        const double *Qss = Qall + (size_t)(mxa - 1) * D2 * D2, *Qsd = Qss + D;
        const double *Qds = Qall + (size_t)(mxm - 1) * D2 * D2 + (size_t)D * D2, *Qdd = Qds + D;
        const double *Qpp = Qall + (size_t)(mxp - 1) * D2 * D2 + (size_t)D * D2 + D;
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
          q0[k] = *reinterpret_cast<const ev_t *>(Qss + eq[k]);
          q1[k] = *reinterpret_cast<const ev_t *>(Qsd + eq[k]);
          q2[k] = *reinterpret_cast<const ev_t *>(Qds + eq[k]);
          q3[k] = *reinterpret_cast<const ev_t *>(Qdd + eq[k]);
          q4[k] = *reinterpret_cast<const ev_t *>(Qpp + eq[k]);
        }
        // the right-hand side row (lanes 0..D-1 of the team): r_a = gs(a) + gd(a-1)/2 - gd(a+1)/2
        const int jr = dt < D ? dt : D - 1;
        const double g0 = g[(size_t)ac * D2 + jr], g1 = g[(size_t)am * D2 + D + jr], g2 = g[(size_t)ap * D2 + D + jr];
        BLK_PROF_AT(10, 128);
        if (defer) {   // panel t-1 -> HBM: M1 = L10 U (row D: h = U' z0) and M2 = L20 U, formed here so that the back
                       // substitution is one product per step on a 2/3-size panel (reads only: no barrier needed)
          double *pan = ws + (size_t)(t - 1) * PAN;
          for (int job = dw; job < 2 * NT; job += NDW) {
            if (job < NT) blk_lu_rowtile_to_panel<D>(p1, p0, job, D + 1, pan, lane);
            else blk_lu_rowtile_to_panel<D>(p2, p0, job - NT, D, pan + (D + 1) * D, lane);
          }
        }
        BLK_PROF_AT(11, 128);
        const double w4 = hasp ? 0.25 : 0.0, w2 = hasp ? 0.5 : 0.0, lv = live ? 1.0 : 0.0;
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
          vd[k] = lv * ((q0[k] + 0.25 * q3[k]) + w4 * q4[k]);     // Qss(a) + Qdd(a-1)/4 + Qdd(a+1)/4
          v1[k] = lv * (0.5 * q2[k] - 0.5 * q1[k]);               // Qds(a-1)/2 - Qsd(a)/2
          v2[k] = lv * (-0.25 * q3[k]);                           // -Qdd(a-1)/4
        }
        rv = lv * ((g0 + 0.5 * g1) - w2 * g2);
        BLK_PROF_AT(7, 128);
        __syncthreads();                  // end of phase 1: every read of p0, p1, p2 is done
      }
      BLK_PROF(0);
      if (t == T) break;
      // ---------------- phase 2 ----------------
      // The freed buffers p0, p1, p2 receive block row t+2 during this phase: wave 3 while the others form L10, wave 2
      // after its share of the S11 update (nothing reads them before phase 1 of the next step).
      auto write_row = [&](int k_lo, int k_hi) {
#pragma unroll
        for (int k = 0; k < NIT; ++k)
          if (k >= k_lo && k < k_hi && eo[k] >= 0) {
            *reinterpret_cast<ev_t *>(p0 + eo[k]) = vd[k];
            *reinterpret_cast<ev_t *>(p1 + eo[k]) = v1[k];
            *reinterpret_cast<ev_t *>(p2 + eo[k]) = v2[k];
          }
        if (k_lo == 0 && dt >= 0 && dt < D) p0[D * LS + dt] = rv;
      };
      if (wave < NT) {                    // L10 = S10 U' in place; row D of S10 := r0 -> row D of L10 = z0
        if (wave == RT) {
          if (lane < DP) b10[D * LS + lane] = (lane < D) ? b00[D * LS + lane] : 0.0;
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        blk_trsm_rowtile<D>(b10, b00, wave, lane);
      }
      if (wave == 3 || (wave == 2 && NT <= 2)) write_row(0, NIT);
      __syncthreads();
      BLK_PROF(1);
      for (int job = wave; job < NLOW; job += 4) {      // S11 -= L10 L10' (lower tiles) -> S00 of the next step
        int q = job, it = 0;
        while (q > it) { q -= it + 1; ++it; }
        blk_update_tile<D>(b11, b10, b10, it, q, lane);
      }
      if (wave == 2 && NT > 2) write_row(0, NIT);
      __syncthreads();
      BLK_PROF(2);
      {   // the window moves by one block
        double *f0 = p0, *f1 = p1;
        p0 = b00; p1 = b10;               // U(t), L10(t); p2 already holds S20(t)
        b00 = b11; b10 = f1; b11 = f0;    // S11 -> S00, block (t+2,t+1) -> S10, block (t+2,t+2) -> S11
      }
    }
    blk_backsub<D>(ws, T, wk, yring, part, U.Y);
    BLK_PROF(5);
    if (tid == 0 && bad) status[0] = 1;
    __syncthreads();
  }
}
