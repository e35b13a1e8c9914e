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
  static constexpr int CB = 64;                         // counters of the workgroup (ints): [0] pivot stage, [2] L20 arrivals, [3] S11 arrivals
  static constexpr int RING = D * 64;                   // pivot wave 1's scratch tiles (Pv2<D>::SCRATCH doubles are used)
  static constexpr size_t PAN = (size_t)(2 * D + 1) * D;   // panel doubles per block step: M1 = L10 U (D+1,D) incl. h = U' z0, M2 = L20 U (D,D)
  static constexpr size_t WORK = 6 * (size_t)BUF > 3 * PAN ? 6 * (size_t)BUF : 3 * PAN;   // six window buffers / three staged panels
  static constexpr size_t lds_doubles = WORK + CB + RING + 2 * D + 768 + 2;
};

typedef double blk_d4 __attribute__((ext_vector_type(4)));

#ifdef TRAJ_BLK_PROF
__device__ long long blk_prof[32];   // cycles of workgroup 0 per phase (printed by traj_check_status)
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

typedef double pv_d2 __attribute__((ext_vector_type(2)));

// ------------------------------------------------------------------------------------------------
// Pivot phase, TWO waves, blocked.  B00 rows/cols < D hold S00 (lower tiles valid); on return they hold
// U = chol(S00)^-1 (lower triangular, zeros above).  S00 is processed in 16 x 16 diagonal blocks:
//   wave 0, scalar chain per diagonal block: ONE LANE OWNS A WHOLE ROW of the block and a pivot column is applied with
//     v_fmac_f64_dpp row_newbcast -- the multiply-add reads row c's value straight out of lane c's register, so the
//     chain never touches LDS: pivot broadcast -> reciprocal (hardware estimate + two Newton steps) -> multiplier ->
//     DPP multiply-adds on the columns to its right and on the columns of the unit-lower inverse (each of the four
//     16-lane groups keeps all of the block but only a quarter of the inverse's columns).  ~190 counts of s_memtime
//     per column (tools/microbench_pivot.hip; one FP64 instruction issues per ~5 counts, tools/microbench_f64lat.hip);
//     the column-by-column scheme of round 1, which published every column through LDS, paid ~600.
//   wave 1, MFMA: the panel below a finished diagonal block  L_rk = A_rk U_kk'  and the rank-16 trailing update
//     A_rc -= L_rk L_ck'  (the next diagonal tile first, so that wave 0 can go on), then U = L00^-1 block row by block
//     row (U_kc = -U_kk W_kc, W_kc = sum_j L_kj U_jc formed while wave 0 is busy with block k), 16 x 16 x 16 products.
// The two waves hand over through counters in LDS.  Row D of the last tile row (the right-hand side living in the
// tile padding) is never modified: L rows >= D are stored as zeros and U is written for rows < D only.
// ------------------------------------------------------------------------------------------------
template <int D>
struct Pv2 {
  static constexpr int NTD = (D + 15) / 16;      // diagonal blocks
  static constexpr int TS = 18;                  // row stride of the scratch tiles (TS/2 odd: conflict-free operand reads)
  static constexpr int TILE = 16 * TS;
  // scratch (in the ring area of the column-by-column scheme): L tiles (r,k), r > k, then the W tiles of one block row
  static constexpr int NL = NTD * (NTD - 1) / 2;
  static constexpr int SCRATCH = (NL + (NTD > 1 ? NTD - 1 : 1)) * TILE;
  __device__ static constexpr int lidx(int r, int k) { return r * (r - 1) / 2 + k; }
};

__device__ __forceinline__ void pv2_signal(int *flag, int v, int lane) {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  if (lane == 0) __hip_atomic_store(flag, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void pv2_wait(int *flag, int v) {
  while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < v) __builtin_amdgcn_s_sleep(1);
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// acc += A B'  /  acc += A B  over one 16 x 16 x 16 product (4 k-steps); tile origins and row strides
__device__ __forceinline__ blk_d4 pv2_mm_nt(const double *A, int lda, const double *B, int ldb, int lane, blk_d4 acc) {
  const int lrow = lane & 15, lq = lane >> 4;
#pragma unroll
  for (int ks = 0; ks < 4; ++ks)
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A[lrow * lda + 4 * ks + lq], B[lrow * ldb + 4 * ks + lq], acc, 0, 0, 0);
  return acc;
}
__device__ __forceinline__ blk_d4 pv2_mm_nn(const double *A, int lda, const double *B, int ldb, int lane, blk_d4 acc) {
  const int lrow = lane & 15, lq = lane >> 4;
#pragma unroll
  for (int ks = 0; ks < 4; ++ks)
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A[lrow * lda + 4 * ks + lq], B[(4 * ks + lq) * ldb + lrow], acc, 0, 0, 0);
  return acc;
}
// store acc (sign applied) into a tile: rows [0, nrows) get the values, the others zeros (zero_rest) or stay untouched
__device__ __forceinline__ void pv2_store(double *Tl, int ld, blk_d4 acc, double sign, int nrows, bool zero_rest, int lane) {
  const int lrow = lane & 15, lq = lane >> 4;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = 4 * r + lq;
    if (row < nrows) Tl[row * ld + lrow] = sign * acc[r];
    else if (zero_rest) Tl[row * ld + lrow] = 0.0;
  }
}

// wave 0: the scalar chains of the diagonal blocks.  flags[0] counts finished stages of this block step (see pv2_wave1).
template <int D, bool kHandOver = true>      // kHandOver = false: no waiting for wave 1 (tools/microbench_pivot.hip)
__device__ void pv2_wave0(double *B00, int *flags, int fbase, int lane, int *bad) {
  using C = BlkCfg<D>;
  using P = Pv2<D>;
  constexpr int LS = C::LS;
  const int i = lane & 15, g = lane >> 4;
  bool notpd = false;
#pragma unroll 1
  for (int k = 0; k < P::NTD; ++k) {
    if (kHandOver && k > 0) pv2_wait(flags, fbase + 2 * k);     // tile (k,k) has the updates of the blocks before it
    const int r0 = 16 * k, nb = (D - r0 < 16) ? D - r0 : 16;
    // a[j]: row i of the block (all four 16-lane groups hold the same copy: the DPP broadcasts stay inside a group);
    // e[q]: entry (i, 4q + g) of the unit-lower inverse -- its columns are independent, so each group keeps a quarter
    double a[16], e[4];
    // (a diagonal tile is updated as a whole by the MFMA products -- X X' is symmetric bit for bit -- so its upper
    // triangle is as good as the lower one and row i is read as it lies: eight 16-byte loads)
    {
      const pv_d2 *row = reinterpret_cast<const pv_d2 *>(B00 + (r0 + i) * LS + r0);
#pragma unroll
      for (int j = 0; j < 16; j += 2) {
        const pv_d2 v = row[j / 2];
        a[j] = (i < nb && j < nb) ? v.x : ((i == j) ? 1.0 : 0.0);
        a[j + 1] = (i < nb && j + 1 < nb) ? v.y : ((i == j + 1) ? 1.0 : 0.0);
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) e[q] = (i == 4 * q + g) ? 1.0 : 0.0;
    double pvi = 1.0;                                       // the pivot of row i
#ifdef TRAJ_BLK_FREE_CHAIN
    // Experiment (wrong results, timing only): the scalar chain costs nothing -- U_kk = diag(S_kk)^-1/2, hand-overs kept.
    // The step time of this build is the bound of what ANY faster pivot scheme could reach.
#pragma unroll
    for (int j = 0; j < 16; ++j) pvi = (i == j) ? fabs(a[j]) + 1.0 : pvi;
#else
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      if (c < nb) {                                        // wave-uniform
        double pv;
        asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(pv) : "v"(a[c]), "n"(c));
        pvi = (i == c) ? pv : pvi;
        // row i -= (a_ic / p) row c for the rows below c: 1/p by the hardware reciprocal and two Newton steps (the only
        // arithmetic on the chain from one pivot to the next; 1/sqrt(p) scales the finished rows after the last column)
        double winv = __builtin_amdgcn_rcp(pv);
        winv = winv * fma(-pv, winv, 2.0);
        winv = winv * fma(-pv, winv, 2.0);
        const double m = ((i > c) ? -a[c] : 0.0) * winv;
#pragma unroll
        for (int j = c + 1; j < 16; ++j)
          asm volatile("v_fmac_f64_dpp %0, %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "+v"(a[j]) : "v"(m), "n"(c));
        // columns 4q + g <= c of the inverse (for the others row c holds an exact zero: the product changes nothing)
#pragma unroll
        for (int q = 0; q <= c / 4; ++q)
          asm volatile("v_fmac_f64_dpp %0, %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "+v"(e[q]) : "v"(m), "n"(c));
      }
    }
#endif
    // U_kk = D^-1/2 (unit-lower inverse): row i scaled by 1/sqrt(p_i); zeros above the diagonal
    notpd |= !(pvi > 0.0);
    if (i < nb) {
      const double sc = traj_rsqrt(pvi);
#pragma unroll
      for (int q = 0; q < 4; ++q) B00[(r0 + i) * LS + r0 + 4 * q + g] = (4 * q + g <= i) ? e[q] * sc : 0.0;
    }
    pv2_signal(flags, fbase + 2 * k + 1, lane);
  }
  if (__builtin_amdgcn_ballot_w64(notpd) != 0 && lane == 0) *bad = 1;      // any row's pivot
}

// wave 1: panels, trailing updates and the off-diagonal blocks of U on MFMA.  Stage counter flags[0] of block step
// `fbase / 16`:  fbase + 2k + 1 = diagonal block k inverted (wave 0),  fbase + 2k + 2 = tile (k+1,k+1) updated (wave 1).
template <int D>
__device__ void pv2_wave1(double *B00, double *scratch, int *flags, int fbase, int lane) {
  using C = BlkCfg<D>;
  using P = Pv2<D>;
  constexpr int LS = C::LS, NTD = P::NTD, TS = P::TS;
  const blk_d4 zero = {0.0, 0.0, 0.0, 0.0};
  double *Ls = scratch, *Wt = scratch + P::NL * P::TILE;      // L tiles (r,k), r > k; W tiles of one block row
  auto tile = [&](int r, int c) { return B00 + 16 * r * LS + 16 * c; };
  auto rows_of = [&](int r) { return (D - 16 * r < 16) ? D - 16 * r : 16; };
  auto wave_sync = [&]() {      // LDS stores of this wave -> loads by its other lanes
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  auto ltile = [&](int r, int k) { return Ls + P::lidx(r, k) * P::TILE; };
  auto trailing = [&](int r, int c, int k) {      // A_rc -= L_rk L_ck'
    const int lrow = lane & 15, lq = lane >> 4;
    double *cp = tile(r, c) + lq * LS + lrow;
    blk_d4 acc;
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = -cp[4 * q * LS];
    acc = pv2_mm_nt(ltile(r, k), TS, ltile(c, k), TS, lane, acc);
#pragma unroll
    for (int q = 0; q < 4; ++q) cp[4 * q * LS] = -acc[q];
  };
  for (int k = 0; k < NTD; ++k) {
    pv2_wait(flags, fbase + 2 * k + 1);
    if (k + 1 < NTD) {
      // what wave 0 waits for: the panel tile below the block, L = A U_kk' (rows >= D stored as zeros), and the update
      // of the next diagonal tile
      pv2_store(ltile(k + 1, k), TS, pv2_mm_nt(tile(k + 1, k), LS, tile(k, k), LS, lane, zero), 1.0, rows_of(k + 1), true, lane);
      wave_sync();
      trailing(k + 1, k + 1, k);
      pv2_signal(flags, fbase + 2 * k + 2, lane);
      // the rest of the panel and of the rank-16 update, beside wave 0's next chain
      for (int r = k + 2; r < NTD; ++r)
        pv2_store(ltile(r, k), TS, pv2_mm_nt(tile(r, k), LS, tile(k, k), LS, lane, zero), 1.0, rows_of(r), true, lane);
      wave_sync();
      for (int c = k + 1; c < NTD; ++c)
        for (int r = (c == k + 1) ? c + 1 : c; r < NTD; ++r) trailing(r, c, k);
    }
    // U = L00^-1, row k of the off-diagonal blocks: U_kc = -U_kk W_kc with W_kc = sum_{j=c}^{k-1} L_kj U_jc formed while
    // wave 0 was busy with block k (below): after the last diagonal block only these independent products remain
    if (k > 0) {
      blk_d4 acc[NTD > 1 ? NTD - 1 : 1];
#pragma unroll
      for (int c = 0; c < NTD - 1; ++c)
        if (c < k) acc[c] = pv2_mm_nn(tile(k, k), LS, Wt + c * P::TILE, TS, lane, zero);
#pragma unroll
      for (int c = 0; c < NTD - 1; ++c)
        if (c < k) pv2_store(tile(k, c), LS, acc[c], -1.0, rows_of(k), false, lane);
      wave_sync();
    }
    if (k + 1 < NTD) {
      for (int c = 0; c <= k; ++c) {
        blk_d4 acc = zero;
        for (int j = c; j <= k; ++j) acc = pv2_mm_nn(ltile(k + 1, j), TS, tile(j, c), LS, lane, acc);
        pv2_store(Wt + c * P::TILE, TS, acc, 1.0, 16, false, lane);
      }
      wave_sync();
    }
  }
}

// The MFMA building blocks below share one shape: every LDS operand of the call is requested first (a few dozen
// ds_reads in flight), then the products run back to back with the column tiles as independent accumulator chains, then
// the results are stored.  A tile-by-tile order (load C, 10 products, store C, next tile) exposes an LDS round trip
// and the 16-pass drain of the accumulators per tile: 50-59 % of the FP64 MFMA rate when measured alone
// (tools/microbench_blkops.hip), against 80-90 % in this form.

// row tile `it` of  S <- S U'  in place (one wave): output tile (it, jt) needs only k < 16 (jt + 1) because U is lower
// triangular.  Columns >= D are stored as zeros.
template <int D>
__device__ __forceinline__ void blk_trsm_rowtile(double *S, const double *U, int it, int lane) {
  using C = BlkCfg<D>;
  constexpr int LS = C::LS, KS = C::KS, NT = C::NT;
  const int lrow = lane & 15, lq = lane >> 4;
  double a[KS], b[NT][KS];
  const double *xa = S + (16 * it + lrow) * LS + lq;
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) a[ks] = xa[4 * ks];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks)
#pragma unroll
    for (int jt = 0; jt < NT; ++jt)
      if (ks < 4 * (jt + 1)) b[jt][ks] = U[(16 * jt + lrow) * LS + lq + 4 * ks];
  blk_d4 acc[NT];
#pragma unroll
  for (int jt = 0; jt < NT; ++jt) acc[jt] = blk_d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int ks = 0; ks < KS; ++ks)
#pragma unroll
    for (int jt = 0; jt < NT; ++jt)
      if (ks < 4 * (jt + 1)) acc[jt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ks], b[jt][ks], acc[jt], 0, 0, 0);
#pragma unroll
  for (int jt = 0; jt < NT; ++jt) {
    const int col = 16 * jt + lrow;
#pragma unroll
    for (int r = 0; r < 4; ++r) S[(16 * it + 4 * r + lq) * LS + col] = (col < D) ? acc[jt][r] : 0.0;
  }
}

// tile (it, jt) of  Cm -= X Y'  (k over the D columns; even and odd k-steps in two accumulators).  Columns >= D are
// stored as zeros.
template <int D>
__device__ __forceinline__ void blk_update_tile(double *Cm, const double *X, const double *Y, int it, int jt, int lane) {
  using C = BlkCfg<D>;
  constexpr int LS = C::LS, KS = C::KS;
  const int lrow = lane & 15, lq = lane >> 4;
  double *cp = Cm + (16 * it + lq) * LS + 16 * jt + lrow;
  const double *xa = X + (16 * it + lrow) * LS + lq, *yb = Y + (16 * jt + lrow) * LS + lq;
  double a[KS], b[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    a[ks] = xa[4 * ks];
    b[ks] = yb[4 * ks];
  }
  blk_d4 acc, acc2 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int r = 0; r < 4; ++r) acc[r] = -cp[4 * r * LS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    if (ks & 1) acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ks], b[ks], acc2, 0, 0, 0);
    else acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ks], b[ks], acc, 0, 0, 0);
  }
  const bool keep = 16 * jt + lrow < D;
#pragma unroll
  for (int r = 0; r < 4; ++r) cp[4 * r * LS] = keep ? -(acc[r] + acc2[r]) : 0.0;
}

// tiles (it, 0 .. NJ-1) of  Cm -= X Y': the A fragments of row tile `it` of X are read once for the row group and the
// column tiles are independent MFMA chains (accumulating  X Y' - Cm, negated when stored).
template <int D, int NJ>
__device__ __forceinline__ void blk_update_rowgroup_n(double *Cm, const double *X, const double *Y, int it, int lane) {
  using C = BlkCfg<D>;
  constexpr int LS = C::LS, KS = C::KS;
  const int lrow = lane & 15, lq = lane >> 4;
  double a[KS], b[NJ][KS];
  blk_d4 acc[NJ];
  const double *xa = X + (16 * it + lrow) * LS + lq;
  double *cp = Cm + (16 * it + lq) * LS + lrow;
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) a[ks] = xa[4 * ks];
#pragma unroll
  for (int jt = 0; jt < NJ; ++jt)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[jt][r] = -cp[4 * r * LS + 16 * jt];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks)
#pragma unroll
    for (int jt = 0; jt < NJ; ++jt) b[jt][ks] = Y[(16 * jt + lrow) * LS + lq + 4 * ks];
  if constexpr (NJ == 1) {      // one tile: two chains (even and odd k-steps)
    blk_d4 acc2 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      if (ks & 1) acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ks], b[0][ks], acc2, 0, 0, 0);
      else acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ks], b[0][ks], acc[0], 0, 0, 0);
    }
    acc[0] += acc2;
  } else {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int jt = 0; jt < NJ; ++jt) acc[jt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ks], b[jt][ks], acc[jt], 0, 0, 0);
  }
#pragma unroll
  for (int jt = 0; jt < NJ; ++jt) {
    const bool keep = 16 * jt + lrow < D;
#pragma unroll
    for (int r = 0; r < 4; ++r) cp[4 * r * LS + 16 * jt] = keep ? -acc[jt][r] : 0.0;
  }
}
template <int D>
__device__ __forceinline__ void blk_update_rowgroup(double *Cm, const double *X, const double *Y, int it, int njt, int lane) {
  constexpr int NT = BlkCfg<D>::NT;
  static_assert(NT <= 4, "row groups of up to four column tiles");
  if (njt == 1) blk_update_rowgroup_n<D, 1>(Cm, X, Y, it, lane);
  else if (njt == 2) blk_update_rowgroup_n<D, NT >= 2 ? 2 : 1>(Cm, X, Y, it, lane);
  else if (njt == 3) blk_update_rowgroup_n<D, NT >= 3 ? 3 : 1>(Cm, X, Y, it, lane);
  else if (njt == 4) blk_update_rowgroup_n<D, NT >= 4 ? 4 : 1>(Cm, X, Y, it, lane);
}

// row tile `it` of  L U  (U lower triangular: column tile jt needs only k >= 16 jt) straight from the accumulators to
// the HBM panel `out` (row-major, D columns): rows < nrows, columns < D.
template <int D>
__device__ __forceinline__ void blk_lu_rowtile_to_panel(const double *L, const double *U, int it, int nrows,
                                                        double *__restrict__ out, int lane) {
  using C = BlkCfg<D>;
  constexpr int LS = C::LS, KS = C::KS, NT = C::NT;
  const int lrow = lane & 15, lq = lane >> 4;
  double a[KS], b[NT][KS];
  const double *xa = L + (16 * it + lrow) * LS + lq;
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) a[ks] = xa[4 * ks];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks)
#pragma unroll
    for (int jt = 0; jt < NT; ++jt)
      if (ks >= 4 * jt) b[jt][ks] = U[(4 * ks + lq) * LS + 16 * jt + lrow];
  blk_d4 acc[NT];
#pragma unroll
  for (int jt = 0; jt < NT; ++jt) acc[jt] = blk_d4{0.0, 0.0, 0.0, 0.0};
  // the short chains first in every k-step, so that the last products of the call belong to three different chains
#pragma unroll
  for (int ks = 0; ks < KS; ++ks)
#pragma unroll
    for (int jt = NT - 1; jt >= 0; --jt)
      if (ks >= 4 * jt) acc[jt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ks], b[jt][ks], acc[jt], 0, 0, 0);
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
  const bool act = tid < 256;              // (eight-wave workgroups: the other waves only keep the barriers' count)
  auto fetch = [&](int t, double (&regs)[NPRE]) {
    if (!act) return;
    const double *pan = ws + (size_t)t * PAN;
#pragma unroll
    for (int k = 0; k < NPRE; ++k) {
      const int e = tid + k * 256;
      regs[k] = pan[e < PAN ? e : 0];
    }
  };
  auto stage = [&](int t, const double (&regs)[NPRE]) {
    if (!act) return;
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
    if (act) {   // partial sums of h - [M1; M2]' [y_{t+1}; y_{t+2}]: compile-time trip count, all LDS reads issued up front
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
      if (p < 4) part[p * 64 + j] = s0 + s1;      // the product is split over four waves; further waves only stage panels
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
      if (act)
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
#ifndef TRAJ_W1_L20
#define TRAJ_W1_L20 1             // L20 row tiles formed by pivot wave 1 (when there are three)
#endif
#define TRAJ_DEFERRED_WAVES 2     // waves of the deferred team (beside the two pivot waves); 4 was measured: the kernel
                                  // then has two waves per SIMD, spills 68 VGPRs and the scalar chain shares its FP64 pipe: 40.6 ms
#ifndef TRAJ_DEFERRED_WAVES_NT3
#define TRAJ_DEFERRED_WAVES_NT3 6 // the same at three tiles per dimension (D = 32..46): 6 = eight waves, two per SIMD, see blk_job_owner (2: round 4's kernel)
#endif
template <int D>
__host__ __device__ constexpr int blk_deferred_waves() { return BlkCfg<D>::NT == 3 ? TRAJ_DEFERRED_WAVES_NT3 : TRAJ_DEFERRED_WAVES; }
#ifndef TRAJ_SEPARATE_BACKSUB_ALL
#define TRAJ_SEPARATE_BACKSUB_ALL 0
#endif
// The back substitution inside the factorisation kernel or as traj_backsub_blk_kernel behind it (eight-wave kernels, and D = 30:
// its four-wave kernel runs two workgroups per CU under a 256-register cap and spilled 55 registers with the back substitution
// inside -- 256 / 1024 utterances of 2000 frames 17.1 / 45.8 -> 13.3 / 39.3 ms; D = 16 the same either way, D = 24 7 % slower at
// 1024 utterances with the separate kernel, which runs one workgroup per CU: TRAJ_SEPARATE_BACKSUB_ALL=1 for the A/B).
template <int D>
__host__ __device__ constexpr bool blk_fused_backsub() { return !TRAJ_SEPARATE_BACKSUB_ALL && blk_deferred_waves<D>() <= 2 && D < 30; }
template <int D>
__host__ __device__ constexpr int blk_threads() { return 64 * (2 + blk_deferred_waves<D>()); }
// Which wave runs job j of the deferred list (see the kernel).  At three tiles per dimension (D = 32..46) by measured
// job lengths (tools/microbench_blkops.hip, counts of s_memtime: S21 row group 2.9k, S22 row groups 2.9k / 2.0k / 1.4k,
// panel row tile 2.2k alone, ~3k beside the other waves' stores) and by when a wave becomes free: the deferred waves
// carry the row groups and one panel tile each, pivot wave 1 two M1 tiles after its last product, pivot wave 0 two M2
// tiles after its last chain.  Other splits of the panel tiles between the four waves measured within 0.1 ms of this
// one; without wave 1's L20 tile 0.7 ms slower; wave 0 leaving the barrier after the L10 row tiles out: no gain.
template <int NT, int NPW, int NDW>
__device__ constexpr int blk_job_owner(int j) {
  if (NT == 3 && NDW == 2) {
    //               S21 r0 r1 r2 | S22 r2 r1 r0 | M1 t0 t1 t2 | M2 t0 t1 t2
    constexpr int own[12] = {2, 3, 2, 3, 2, 3, 1, 1, 2, 3, 0, 0};
    return own[j];
  }
  if (NT == 3 && NDW == 6) {
    // Eight waves, two per SIMD (wave w on SIMD w % 4): wave 4 shares the scalar chain's SIMD and gets no products (a
    // dependent FP64 chain advances one instruction per MFMA slot beside a wave that streams MFMAs); wave 5 shares the
    // SIMD of pivot wave 1 and gets the two lightest jobs.
    //               S21 r0 r1 r2 | S22 r2 r1 r0 | M1 t0 t1 t2 | M2 t0 t1 t2
#ifndef TRAJ_W8_OWNERS
#define TRAJ_W8_OWNERS {2, 3, 6, 7, 6, 7, 6, 7, 5, 0, 3, 5}
#endif
    constexpr int own[12] = TRAJ_W8_OWNERS;
    return own[j];
  }
  return NPW + j % NDW;
}


// Two workgroups per CU where they fit (static D <= 25: 16-row tiles up to 32 rows -> <= 70 KB of LDS; the second launch
// bound caps the registers at 256): two independent scalar chains then share a CU's SIMDs, and a batch of more utterances
// (or vc() chunks) than CUs runs 1.5-1.6x faster (tools/traj_occupancy_probe.py).  At D = 30..40 the window alone is
// 115 KB: one workgroup per CU.
template <int D>
__global__ void __launch_bounds__(blk_threads<D>(), (BlkCfg<D>::lds_doubles * 8 <= 80 * 1024) ? 2 : 1)
traj_solve_blk_kernel(const TrajUtt *__restrict__ utts, int n, const double *__restrict__ Qall,
                      const int64_t *__restrict__ mhat_all, const double *__restrict__ g_all, double *__restrict__ ws_all,
                      int64_t ws_stride, int *__restrict__ status) {
  using C = BlkCfg<D>;
  constexpr int D2 = 2 * D, DP = C::DP, LS = C::LS, NT = C::NT, BUF = C::BUF;
  constexpr size_t PAN = C::PAN;
  constexpr int NLOW = NT * (NT + 1) / 2;
  constexpr int NPW = 2, NDW = blk_deferred_waves<D>(), NDT = 64 * NDW;      // pivot pair, deferred team
  constexpr int W1L = BlkCfg<D>::NT >= 3 ? TRAJ_W1_L20 : 0;
  static_assert(W1L == 0 || D / 16 >= W1L, "pivot wave 1 does not form the L20 row tile that holds the rhs row");
  constexpr int NW = NPW + NDW, NTHR = 64 * NW;
  constexpr int VW = (D % 2 == 0) ? 2 : 1;                 // elements per access: 16-byte LDS / global accesses when D is even
  constexpr int NIT = (D * D / VW + NDT - 1) / NDT;        // element groups per thread
  static_assert(Pv2<D>::SCRATCH <= BlkCfg<D>::RING, "pivot wave 1's scratch tiles live in the ring area");
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
  const int dt = tid - 64 * NPW;
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
    for (int e = tid; e < C::CB + C::RING; e += NTHR) cbu[e] = 0.0;   // the counters and wave 1's scratch tiles
    // p0, p1, p2 receive only D x D elements (+ the rhs row) per step: their padding stays zero from here on
    for (int e = tid; e < 3 * BUF; e += NTHR) p0[e] = 0.0;
    blk_assemble<D>(nullptr, nullptr, b00, 0, T, mh, g, Qall, tid, NTHR);
    blk_assemble<D>(nullptr, b10, b11, 1, T, mh, g, Qall, tid, NTHR);
    __syncthreads();
    BLK_PROF_T0();

    for (int t = 0; t <= T; ++t) {
      // ---------------- phase 1 ----------------
      ev_t vd[NIT], v1[NIT], v2[NIT];     // block row t+2 of the stencil, written to p0, p1, p2 during phase 2 (deferred team)
      double rv = 0.0;
      // The deferred work on step t-1.  First L20(t-1) = S20 U' in place, one row tile per wave where there are three
      // (W1L: pivot wave 1 forms tile 0 while it waits for wave 0's first chain); the waves that formed tiles meet on the
      // arrival counter `mid` (the pivot pair is not held up by it).  Everything after that -- the S21 / S22 row groups
      // and the row tiles of the panel -- is a list of independent jobs shared out by blk_job_owner: most to the
      // deferred waves, the rest to wave 0 after the last chain of the step and to wave 1 after its last product (one
      // wave alone runs these at ~90 % of the FP64 MFMA rate, so what matters is that no SIMD idles).  A queue in LDS
      // from which the waves took jobs as they became free was measured too: 29.0 ms against 26.6 ms, the loop over
      // run-time job numbers costs more than the balance gains.
      const bool defer = t >= 1;
      int *const mid = reinterpret_cast<int *>(cbu) + 2;
      auto mid_arrive = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) __hip_atomic_fetch_add(mid, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      };
      auto l20_rowtile = [&](int it) {    // row D of S20 := r0(t-1) -> row D of L20 = z0
        if (it == RT) {
          if (lane < DP) p2[D * LS + lane] = (lane < D) ? p0[D * LS + lane] : 0.0;
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        blk_trsm_rowtile<D>(p2, p0, it, lane);
      };
      // jobs, the long ones first: S21 -= L20 L10' by row group (NT tiles each), S22 -= L20 L20' by row group (lower
      // tiles), then the panel of step t-1 -> HBM: M1 = L10 U (row D: h = U' z0) and M2 = L20 U by row tile, formed here
      // so that the back substitution is one product per step on a 2/3-size panel
      auto run_job = [&](int j) {
        double *pan = ws + (size_t)(t - 1) * PAN;
        if (j < NT) blk_update_rowgroup<D>(b10, p2, p1, j, NT, lane);
        else if (j < 2 * NT) blk_update_rowgroup<D>(b11, p2, p2, 2 * NT - 1 - j, 2 * NT - j, lane);
        else if (j < 3 * NT) blk_lu_rowtile_to_panel<D>(p1, p0, j - 2 * NT, D + 1, pan, lane);
        else blk_lu_rowtile_to_panel<D>(p2, p0, j - 3 * NT, D, pan + (D + 1) * D, lane);
      };
      if (wave < NPW) {
        if (t < T) {
          if (wave == 0) {
            pv2_wave0<D>(b00, reinterpret_cast<int *>(cbu), 16 * t, lane, &bad);
          } else {
            if (defer && W1L) {
              l20_rowtile(0);
              mid_arrive();
            }
            pv2_wave1<D>(b00, ring, reinterpret_cast<int *>(cbu), 16 * t, lane);
          }
          BLK_PROF_AT(3, 0);
          BLK_PROF_AT(4, 64);
        } else if (wave == 1 && W1L) {
          l20_rowtile(0);
          mid_arrive();
        }
        if (defer) {                      // the pivot pair's share of the jobs
          pv2_wait(mid, (NDW + W1L) * t);
#pragma unroll
          for (int j = 0; j < 4 * NT; ++j)
            if (blk_job_owner<NT, NPW, NDW>(j) < NPW && blk_job_owner<NT, NPW, NDW>(j) == wave) run_job(j);
        }
        BLK_PROF_AT(9, 0);
        BLK_PROF_AT(12, 64);
      } else {
        const int dw = wave - NPW;
        // mixtures of block rows t+1, t+2, t+3 (clamped), loaded before the products so that the stencil loads below
        // do not wait for them
        const int a = t + 2;
        const bool live = a < T, hasp = a + 1 < T;
        const int ac = live ? a : T - 1, am = ac >= 1 ? ac - 1 : 0, ap = hasp ? a + 1 : ac;
        const int64_t mxa = mh[ac], mxm = mh[am], mxp = mh[ap];
        if (defer) {
          for (int it = W1L + dw; it < NT; it += NDW) l20_rowtile(it);
          mid_arrive();
          BLK_PROF_AT(8, 64 * NPW);
          pv2_wait(mid, (NDW + W1L) * t);
#pragma unroll
          for (int j = 0; j < 2 * NT; ++j)
            if (blk_job_owner<NT, NPW, NDW>(j) >= NPW && blk_job_owner<NT, NPW, NDW>(j) == wave) run_job(j);
        }
        BLK_PROF_AT(6, 64 * NPW);
        BLK_PROF_AT(14, 64 * NPW + 64);
        BLK_PROF_AT(24 + wave, 64 * wave);      // ... and the end of its S21 / S22 jobs
        // (two waves per SIMD: the stencil values are fetched after the panel jobs -- the SIMD's other wave covers the wait,
        // and 5 x NIT values in flight beside a job's 52 operand doubles do not fit 256 registers)
#ifndef TRAJ_W8_STENCIL_MID
#define TRAJ_W8_STENCIL_MID 0
#endif
        constexpr bool kLateStencil = NDW > 2 && !TRAJ_W8_STENCIL_MID;
        // (measured and dropped: M1 tiles -- which need nothing of this step -- run by the waves without an L20 tile while the
        // others form L20: 23.9 against 23.45 ms, the products beside them delay L20, which everything else waits for)
        auto late_panel_jobs = [&]() {
#pragma unroll
          for (int j = 2 * NT; j < 4 * NT; ++j)
            if (blk_job_owner<NT, NPW, NDW>(j) >= NPW && blk_job_owner<NT, NPW, NDW>(j) == wave) run_job(j);
        };
        // Wave 4 of eight shares the scalar chain's SIMD and gets no job at all.  (Measured: a panel job for it after wave 0's
        // last diagonal block -- even the polling for that moment alone, one LDS read per s_sleep -- costs the chain 1.4 ms
        // per 2000 steps: 24.8 against 23.3 ms.)
        constexpr bool chainmate = false;
        if (kLateStencil && defer && !chainmate) late_panel_jobs();
        // Block row a = t+2 of the stencil.  Every load is unconditional on a clamped address (a select on a loaded
        // value would make the wave wait for each load in turn); masks are applied when the operands are combined.
        // Issued between the row-group jobs and the panel jobs: early enough to be back in time, late enough that the
        // 5 x NIT values in flight do not crowd the registers of the long jobs.
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
        BLK_PROF_AT(10, 64 * NPW);
        if (!kLateStencil && defer && !chainmate) late_panel_jobs();
        BLK_PROF_AT(11, 64 * NPW);
        const double w4 = hasp ? 0.25 : 0.0, w2 = hasp ? 0.5 : 0.0, lv = live ? 1.0 : 0.0;
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
          vd[k] = lv * ((q0[k] + 0.25 * q3[k]) + w4 * q4[k]);     // Qss(a) + Qdd(a-1)/4 + Qdd(a+1)/4
          v1[k] = lv * (0.5 * q2[k] - 0.5 * q1[k]);               // Qds(a-1)/2 - Qsd(a)/2
          v2[k] = lv * (-0.25 * q3[k]);                           // -Qdd(a-1)/4
        }
        rv = lv * ((g0 + 0.5 * g1) - w2 * g2);
        BLK_PROF_AT(7, 64 * NPW);
        BLK_PROF_AT(13, 64 * NPW + 64);
        BLK_PROF_AT(16 + wave, 64 * wave);      // every deferred wave's arrival at the barrier (its own clock since the S11 update)
      }
      __syncthreads();                    // end of phase 1: every read of p0, p1, p2 is done
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
      // deferred waves that form no L10 row tile write their stencil values now, the others after the S11 update
      if (wave >= NPW && wave >= NT) write_row(0, NIT);
      __syncthreads();
      BLK_PROF(1);
      // S11 -= L10 L10' (lower tiles) -> S00 of the next step.  Wave 0 forms tile (0,0) and goes straight on to the scalar
      // chain of the next step's first diagonal block (all it needs); the other waves share the remaining tiles and meet on
      // an arrival counter: wave 1's trailing updates and the deferred team need them (and the stencil rows) complete,
      // the chain does not.  This takes the S11 update off the critical path of the factorisation.
      if (wave == 0) {
        blk_update_tile<D>(b11, b10, b10, 0, 0, lane);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      } else {
        // (eight waves: wave 4 sits on the chain's SIMD, where wave 0 has just gone on to the next chain -- no tile for it)
        const int first = (NW == 8) ? (wave == 4 ? NLOW : (wave < 4 ? wave : wave - 1)) : wave;
        for (int job = first; job < NLOW; job += (NW == 8 ? NW - 2 : NW - 1)) {
          int q = job, it = 0;
          while (q > it) { q -= it + 1; ++it; }
          blk_update_tile<D>(b11, b10, b10, it, q, lane);
        }
        if (wave >= NPW && wave < NT) write_row(0, NIT);
        int *s11 = reinterpret_cast<int *>(cbu) + 3;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) __hip_atomic_fetch_add(s11, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        pv2_wait(s11, (NW - 1) * (t + 1));
      }
      BLK_PROF(2);
      {   // the window moves by one block
        double *f0 = p0, *f1 = p1;
        p0 = b00; p1 = b10;               // U(t), L10(t); p2 already holds S20(t)
        b00 = b11; b10 = f1; b11 = f0;    // S11 -> S00, block (t+2,t+1) -> S10, block (t+2,t+2) -> S11
      }
    }
    // (eight waves: the back substitution is a kernel of its own, traj_backsub_blk_kernel -- inside this one it would be
    // compiled for 256 registers beside everything else and spill in its loop: 5.0k instead of 2.8k counts per step)
    if constexpr (blk_fused_backsub<D>()) blk_backsub<D>(ws, T, wk, yring, part, U.Y);
    BLK_PROF(5);
    if (tid == 0 && bad) status[0] = 1;
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------
// The back substitution alone, as a STREAM (launched behind traj_solve_blk_kernel where that kernel has eight waves): workgroup
// b reads the panels workgroup b of the factorisation left in its workspace -- the host launches the pair per batch of at
// most gridDim.x utterances, so that a workspace holds one utterance's panels.
//   y_t = h_t - M1_t' y_{t+1} - M2_t' y_{t+2}:  3240 multiply-adds per step at D = 40, but 25.9 KB of panel -- 13.3 GB per 256
// utterances of 2000 frames: the loop is bound by HBM, so it is built around the read:
//   loader waves (NLOAD of them, panel i = l, l + NLOAD, ...; i counts from the LAST block step): LDS-DMA of whole panels
//     (global_load_lds_dwordx4, 1 KB per wave instruction, no registers) into a ring of RING slots, PF panels in flight
//     per wave (vmcnt is a 6-bit counter: PF x NI <= 60); a slot is reused once the compute wave has gone past it (`done`),
//     a panel is published (`ready[slot]`) when the wave's vmcnt says its last kilobyte has landed;
//   ONE compute wave: lane j < D owns column j -- 2 D + 1 conflict-free LDS reads of the panel, the 2 D values of y_{t+1},
//     y_{t+2} as broadcast reads (16 bytes each) from a three-deep LDS ring it writes itself, four accumulators; no barrier
//     in the loop, no other wave on its critical path.
// Panels are read in whole kilobytes: the last read of a panel runs up to 1008 bytes into the next one (the workspace has that
// much slack behind its last panel).
// ------------------------------------------------------------------------------------------------
template <int D>
struct BacksubCfg {
  static constexpr int PAN = (int)BlkCfg<D>::PAN;
  static constexpr int NI = (PAN * 8 + 1023) / 1024;                 // 1 KB wave instructions per panel
  static constexpr int SLOT = NI * 128;                              // doubles per ring slot
  static constexpr int PF = (60 / NI) < 3 ? (60 / NI) : 3;           // panels in flight per loader wave
  static constexpr int RING_MAX = (150 * 1024) / (NI * 1024);
  static constexpr int NLOAD = ((RING_MAX - 1) / PF) < 3 ? ((RING_MAX - 1) / PF) : 3;
  static constexpr int RING = NLOAD * PF + 1;
  static constexpr int THREADS = 64 * (1 + NLOAD);
  static constexpr size_t lds_bytes = (size_t)RING * SLOT * 8 + 3 * 64 * 8 + 64 * 4;      // ring | y ring (3 x 64) | flags
  static_assert(PF >= 1 && NLOAD >= 1 && RING >= NLOAD + 1, "ring too small");
};

template <int D>
__global__ void __launch_bounds__(BacksubCfg<D>::THREADS)
traj_backsub_blk_kernel(const TrajUtt *__restrict__ utts, int n, const double *__restrict__ ws_all, int64_t ws_stride) {
  using B = BacksubCfg<D>;
  constexpr int PAN = B::PAN, NI = B::NI, SLOT = B::SLOT, PF = B::PF, NLOAD = B::NLOAD, RING = B::RING;
  constexpr int OH = D * D, OM2 = (D + 1) * D;
  extern __shared__ __attribute__((aligned(16))) double blk_sm[];
  double *ring = blk_sm;                              // [RING][SLOT]
  double *ysh = ring + (size_t)RING * SLOT;           // [3][64]: y_t ring (slot t % 3)
  int *ready = reinterpret_cast<int *>(ysh + 3 * 64); // [RING] = index + 1 of the panel the slot holds; [RING] = done
  int *done = ready + RING;
  const int u = blockIdx.x;
  if (u >= n) return;
  const TrajUtt U = utts[u];
  const int T = U.T;
  if (T == 0) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const double *ws = ws_all + (size_t)blockIdx.x * ws_stride;
  if (tid <= RING) ready[tid] = 0;                    // (ready[RING] is `done`)
  for (int e = tid; e < 3 * 64; e += B::THREADS) ysh[e] = 0.0;
  __syncthreads();
  if (wave > 0) {
    // ---- loader wave l: panels l, l + NLOAD, ... (counted from the last block step)
    const int l = wave - 1;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)(reinterpret_cast<char *>(ring));
    const unsigned lane_off = 16u * (unsigned)lane;
    int issued = l, signalled = l, inflight = 0;
    while (signalled < T) {
      while (inflight < PF && issued < T) {
        pv2_wait(done, issued - RING + 1);            // the slot's previous panel has been consumed
        const char *gp = reinterpret_cast<const char *>(ws + (size_t)(T - 1 - issued) * PAN);
        const unsigned la = lds0 + (unsigned)(issued % RING) * (unsigned)(SLOT * 8);
#pragma unroll
        for (int k = 0; k < NI; ++k) dma_1k(gp + 1024 * k, la + 1024u * k, lane_off);
        issued += NLOAD;
        ++inflight;
      }
      // the oldest panel in flight has landed when at most (inflight - 1) x NI of this wave's loads are outstanding
      if (inflight >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NI < 63 ? 2 * NI : 63) : "memory");
      else if (inflight == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NI < 63 ? NI : 63) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      pv2_signal(ready + signalled % RING, signalled + 1, lane);
      signalled += NLOAD;
      --inflight;
    }
    return;
  }
  // ---- the compute wave
  const int j = lane < D ? lane : D - 1;
  double *Y = U.Y;
  for (int i = 0; i < T; ++i) {
    const int t = T - 1 - i;
    pv2_wait(ready + i % RING, i + 1);
    const double *pb = ring + (size_t)(i % RING) * SLOT;
    const double *y1 = ysh + ((t + 1) % 3) * 64, *y2 = ysh + ((t + 2) % 3) * 64;      // y_{t+1}, y_{t+2} (zeros beyond the end)
    // the lane's column of the panel into registers, then the slot is handed back at once: the loaders run ahead while the
    // sums are formed (a slot held for the whole step left the read 10 % below what the loaders reach alone)
    double mh = pb[OH + j], m1v[D], m2v[D];
#pragma unroll
    for (int r = 0; r < D; ++r) m1v[r] = pb[r * D + j];
#pragma unroll
    for (int r = 0; r < D; ++r) m2v[r] = pb[OM2 + r * D + j];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    pv2_signal(done, i + 1, lane);
    typedef double d2 __attribute__((ext_vector_type(2)));
    double a0 = mh, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#ifdef TRAJ_BACKSUB_NO_COMPUTE
    if (false)
#endif
    {
#pragma unroll
      for (int r = 0; r < D; r += 2) {
        const d2 yy = *reinterpret_cast<const d2 *>(y1 + r);                            // (uniform address: a broadcast read)
        const d2 zz = *reinterpret_cast<const d2 *>(y2 + r);
        if (r & 2) {
          a2 = fma(-m1v[r], yy.x, a2);
          a3 = fma(-m2v[r], zz.x, a3);
          if (r + 1 < D) {
            a2 = fma(-m1v[r + 1], yy.y, a2);
            a3 = fma(-m2v[r + 1], zz.y, a3);
          }
        } else {
          a0 = fma(-m1v[r], yy.x, a0);
          a1 = fma(-m2v[r], zz.x, a1);
          if (r + 1 < D) {
            a0 = fma(-m1v[r + 1], yy.y, a0);
            a1 = fma(-m2v[r + 1], zz.y, a1);
          }
        }
      }
    }
    const double yv = (a0 + a1) + (a2 + a3);
    if (lane < D) {
      ysh[(t % 3) * 64 + lane] = yv;                  // the slot of y_{t+3}, read for the last time one step ago
      Y[(size_t)t * D + lane] = yv;                   // row t of reshape(y, D, T), src/trajectory_gmmmap.jl:109
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
}
template <int D>
constexpr size_t blk_backsub_lds_bytes() { return BacksubCfg<D>::lds_bytes; }
