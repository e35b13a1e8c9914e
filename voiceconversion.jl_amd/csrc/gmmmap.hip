// gmmmap.hip -- GMM joint-density mapping on MI355X (gfx950): model preparation, the batched
// fvconvert kernels (FP64 MFMA tile kernel + generic FP64 VALU kernel), posterior and argmax.
//
// Replaces, for whole (D,T) matrices at once:
//   GMMMap ctor / GMMMapParam / split_joint_gmm          reference src/gmmmap.jl:23-90
//   GaussianMixtureModel (Hermitian + Cholesky)          reference src/gmm.jl:8-20
//   fvconvert(g::GMMMap, x)                              reference src/gmmmap.jl:101-118
//   predict_proba / predict                              reference src/gmm.jl:24-58
//   vc(c::FrameByFrameConverter, fm)                     reference src/common.jl:7-26
//
// Math.  Per mixture m, prepared once on the host:
//   A_m = Syx_m inv(Sxx_m)              b_m  = muy_m - A_m mux_m
//   L_m L_m' = Hermitian(Sxx_m)         U_m  = inv(L_m)   (lower triangular)     cz_m = U_m mux_m
//   lc_m = log w_m - (D log 2pi + 2 sum_i log L_m[i,i]) / 2
// Per frame x:   z_m = U_m x - cz_m,  l_m = lc_m - |z_m|^2 / 2,  p = softmax(l),  y = sum_m p_m (A_m x + b_m).
// The softmax is evaluated online (running max + rescale), so no (M,T) posterior is ever stored by convert.
//
// MFMA kernel data layout: the 2*Dp rows [U_m ; A_m] (Dp = D rounded up to 4) are cut into 16-row tiles;
// D[16 rows x 16 frames] += W[16 rows x 4 k] * X[4 k x 16 frames] is one v_mfma_f64_16x16x4_f64.  U-only
// tiles skip the k-steps that are entirely above the diagonal.  Each mixture's operand fragments are stored
// in HBM in issue order, 64 lanes x 8 B per step, so the global->LDS stage is a straight copy and every
// ds_read_b64 is conflict-free.
#include "vcmi_common.hpp"
#include "host_linalg.hpp"
#include "gmmmap_handle.hpp"
#include "devgroup.hpp"
#include "hostpipe.hpp"
#include "fp64_exp.hpp"
#include "grouping.hpp"
#ifndef VCMI_CONVERT_PRIO
#define VCMI_CONVERT_PRIO 1        // s_setprio: 1 = a wave's stretches WITHOUT MFMAs (|z|^2, the test, the y update, the barrier) at high
                                   // priority, so that it is back in an MFMA stream sooner (+1..2 % on one box); 2 = the reverse; 0 = none
#endif

#include <atomic>
#include <cmath>
#include <cstdlib>
#include <limits>

namespace vcmi {

// ------------------------------------------------------------------------------------------------
// compile-time description of the row tiling for a padded dimension DP (multiple of 4)
// ------------------------------------------------------------------------------------------------
// UONLY: only the whitening rows U_m (log-density / posterior / argmax path: the regression rows are not staged)
template <int DP, bool UONLY = false>
struct Tiling {
  static constexpr int KS = DP / 4;                       // k-steps covering all of x
  static constexpr int NT = UONLY ? (DP + 15) / 16 : (2 * DP + 15) / 16;   // 16-row tiles over [U rows ; A rows]
  static constexpr int NU = (DP + 15) / 16;               // tiles that contain at least one U row
  __host__ __device__ static constexpr int steps(int t) {  // k-steps tile t needs
    return (16 * t + 15 < DP) ? ((4 * (t + 1) < KS) ? 4 * (t + 1) : KS) : KS;
  }
  // MODE 3 (predict with early exit) stores the whitening tiles one after the other in REVERSE order (tile NU-1 first):
  // rtile_off(i) = first fragment of the i-th tile in that order
  __host__ __device__ static constexpr int tile_off(int t) {      // fragments before tile t in the k-step-major order of the U tiles
    int n = 0;
    for (int ks = 0; ks < KS; ++ks)
      for (int u = 0; u < NU && u < NT; ++u)
        if (ks < steps(u)) ++n;
    return t >= NU ? n : 0;
  }
  // position of fragment (k-step ks, U tile t) in the k-step-major order of the U tiles (the order the packer writes them in)
  __host__ __device__ static constexpr int ufrag_pos(int ks, int t) {
    int n = 0;
    for (int k = 0; k < ks; ++k)
      for (int u = 0; u < NU && u < NT; ++u)
        if (k < steps(u)) ++n;
    for (int u = 0; u < t; ++u)
      if (ks < steps(u)) ++n;
    return n;
  }
  __host__ __device__ static constexpr int rtile_off(int i) {
    int n = 0;
    for (int u = 0; u < i; ++u) n += steps(NU - 1 - u);
    return n;
  }
  __host__ __device__ static constexpr int nsteps() {
    int n = 0;
    for (int t = 0; t < NT; ++t) n += steps(t);
    return n;
  }
  static constexpr int NSTEPS = nsteps();
  // per-mixture block in doubles: [fragments NSTEPS*64 | cinit NT*16 | lc | pad], multiple of 32 doubles
  static constexpr int CINIT_OFF = NSTEPS * 64;
  static constexpr int LC_OFF = CINIT_OFF + NT * 16;
  static constexpr int BLK = ((LC_OFF + 1 + 1023) / 1024) * 1024;   // whole double2 per thread for 256- and 512-thread groups
};

// runtime mirror used by the host-side packer (same formulas, any DP)
struct TilingRT {
  int DP, KS, NT, NU, NSTEPS, CINIT_OFF, LC_OFF, BLK;
  explicit TilingRT(int dp, bool uonly = false) : DP(dp) {
    KS = DP / 4;
    NT = uonly ? (DP + 15) / 16 : (2 * DP + 15) / 16;
    NU = (DP + 15) / 16;
    NSTEPS = 0;
    for (int t = 0; t < NT; ++t) NSTEPS += steps(t);
    CINIT_OFF = NSTEPS * 64;
    LC_OFF = CINIT_OFF + NT * 16;
    BLK = ((LC_OFF + 1 + 1023) / 1024) * 1024;
  }
  int steps(int t) const { return (16 * t + 15 < DP) ? std::min(4 * (t + 1), KS) : KS; }
};

typedef double d4 __attribute__((ext_vector_type(4)));

// q[lane] -> the sum over the four lanes {lane % 16 + 16 g} (the lane groups of an MFMA result column), in every lane.
// v_permlane16_swap / v_permlane32_swap (gfx950) exchange 16-lane rows / wave halves between two registers, so each level is
// two moves, two swaps per 32-bit half and one add -- no LDS round trip (__shfl_xor compiles to ds_bpermute_b32 pairs with an
// s_waitcnt each).  Same pairs added in the same association as q += shfl_xor(q, 16); q += shfl_xor(q, 32): bit-identical.
__device__ __forceinline__ double sum_lane_groups(double q) {
  unsigned lo = __double2loint(q), hi = __double2hiint(q);
  auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  const double x = __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
  lo = __double2loint(x);
  hi = __double2hiint(x);
  auto c = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  auto d = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  return __hiloint2double(d[0], c[0]) + __hiloint2double(d[1], c[1]);
}

// Two columns' worth at once: q0[lane], q1[lane] -> in the EVEN lane groups the lane-group sum of q0, in the ODD ones that of
// q1.  The first exchange swaps 16-lane rows between the two registers themselves (no copies): six moves / swaps and two adds
// instead of the sixteen and four of two sum_lane_groups calls -- and every VALU instruction of a wave, FP64 or not, waits
// for the FP64 MFMAs of the SIMD's other waves.  Association: (q_g + q_g^1) + (the other pair), as in sum_lane_groups.
__device__ __forceinline__ double sum_lane_groups_pair(double q0, double q1) {
  const unsigned l0 = __double2loint(q0), h0 = __double2hiint(q0), l1 = __double2loint(q1), h1 = __double2hiint(q1);
  auto a = __builtin_amdgcn_permlane16_swap(l0, l1, false, false);     // {rows (q0 r0, q1 r0, q0 r2, q1 r2), rows (q0 r1, q1 r1, q0 r3, q1 r3)}
  auto b = __builtin_amdgcn_permlane16_swap(h0, h1, false, false);
  const double x = __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
  const unsigned lo = __double2loint(x), hi = __double2hiint(x);
  auto c = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  auto d = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  return __hiloint2double(d[0], c[0]) + __hiloint2double(d[1], c[1]);
}
// v (selected layout: tile 0's value in the even lane groups, tile 1's in the odd ones) -> {tile 0's, tile 1's} in every lane
__device__ __forceinline__ void unpair_lane_groups(double v, double &v0, double &v1) {
  const unsigned lo = __double2loint(v), hi = __double2hiint(v);
  auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  v0 = __hiloint2double(b[0], a[0]);
  v1 = __hiloint2double(b[1], a[1]);
}

// ------------------------------------------------------------------------------------------------
// Rows of x and y as whole 128-byte lines (round 6).  The MFMA operand / result layout gives lane group g of a frame's column the
// features 4 j + g (slot j = k-step j): a load or store instruction then touches 32 bytes of each of its 16 frames, and the
// pieces cost ~8 % of the screened conversion's step (profiles/r06_ab).  Where the rows are 16-byte aligned and unpadded, lane
// group g moves features 16 b + 4 g .. + 3 instead -- 32 contiguous bytes per lane, a line per frame and block of four slots --
// and a 4 x 4 transpose across the lane groups (two exchange stages: wave halves with v_permlane32_swap, then 16-lane rows with
// v_permlane16_swap; an involution) converts between the two: (slot j, group g) <-> (slot g, group j).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void transpose_lane_groups4(double &e0, double &e1, double &e2, double &e3) {
  unsigned lo[4] = {(unsigned)__double2loint(e0), (unsigned)__double2loint(e1), (unsigned)__double2loint(e2), (unsigned)__double2loint(e3)};
  unsigned hi[4] = {(unsigned)__double2hiint(e0), (unsigned)__double2hiint(e1), (unsigned)__double2hiint(e2), (unsigned)__double2hiint(e3)};
#pragma unroll
  for (int j = 0; j < 2; ++j) {                              // lane groups {2, 3} of slot j <-> lane groups {0, 1} of slot j + 2
    auto a = __builtin_amdgcn_permlane32_swap(lo[j], lo[j + 2], false, false);
    auto c = __builtin_amdgcn_permlane32_swap(hi[j], hi[j + 2], false, false);
    lo[j] = a[0];
    lo[j + 2] = a[1];
    hi[j] = c[0];
    hi[j + 2] = c[1];
  }
#pragma unroll
  for (int j = 0; j < 4; j += 2) {                           // odd lane groups of slot j <-> even lane groups of slot j + 1
    auto a = __builtin_amdgcn_permlane16_swap(lo[j], lo[j + 1], false, false);
    auto c = __builtin_amdgcn_permlane16_swap(hi[j], hi[j + 1], false, false);
    lo[j] = a[0];
    lo[j + 1] = a[1];
    hi[j] = c[0];
    hi[j + 1] = c[1];
  }
  e0 = __hiloint2double(hi[0], lo[0]);
  e1 = __hiloint2double(hi[1], lo[1]);
  e2 = __hiloint2double(hi[2], lo[2]);
  e3 = __hiloint2double(hi[3], lo[3]);
}
__device__ __forceinline__ bool rows_as_lines(const void *base, int64_t ld, int D, int DP) {
  return D == DP && DP >= 16 && (ld & 1) == 0 && (reinterpret_cast<uintptr_t>(base) & 15u) == 0;
}
// xo[ks] = row[4 ks + lgrp] for the lane's frame (`row` = its first feature; zero when !live or beyond D).  ALL lanes of the
// wave must call it (the transpose exchanges registers across lanes); a frame that is not live passes any valid row.
template <int KS>
__device__ __forceinline__ void load_frame_row(const double *row, bool live, bool lines, int D, int lgrp, double (&xo)[KS]) {
  constexpr int NB = KS / 4;
  if (lines) {
    typedef double xd2 __attribute__((ext_vector_type(2)));
    xd2 v[NB > 0 ? NB : 1][2];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      v[b][0] = *reinterpret_cast<const xd2 *>(row + 16 * b + 4 * lgrp);
      v[b][1] = *reinterpret_cast<const xd2 *>(row + 16 * b + 4 * lgrp + 2);
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      double e0 = v[b][0].x, e1 = v[b][0].y, e2 = v[b][1].x, e3 = v[b][1].y;
      transpose_lane_groups4(e0, e1, e2, e3);
      xo[4 * b] = live ? e0 : 0.0;
      xo[4 * b + 1] = live ? e1 : 0.0;
      xo[4 * b + 2] = live ? e2 : 0.0;
      xo[4 * b + 3] = live ? e3 : 0.0;
    }
#pragma unroll
    for (int ks = 4 * NB; ks < KS; ++ks) xo[ks] = live ? row[4 * ks + lgrp] : 0.0;
  } else {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int k = 4 * ks + lgrp;
      xo[ks] = (live && k < D) ? row[k] : 0.0;
    }
  }
}
// row[4 j + lgrp] = yo[j] for the lane's frame; ALL lanes of the wave must call it
template <int KS>
__device__ __forceinline__ void store_frame_row(double *row, bool live, bool lines, int D, int lgrp, const double (&yo)[KS]) {
  constexpr int NB = KS / 4;
  if (lines) {
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      double e0 = yo[4 * b], e1 = yo[4 * b + 1], e2 = yo[4 * b + 2], e3 = yo[4 * b + 3];
      transpose_lane_groups4(e0, e1, e2, e3);
      if (live) {
        typedef double yd2 __attribute__((ext_vector_type(2)));
        *reinterpret_cast<yd2 *>(row + 16 * b + 4 * lgrp) = yd2{e0, e1};
        *reinterpret_cast<yd2 *>(row + 16 * b + 4 * lgrp + 2) = yd2{e2, e3};
      }
    }
  }
  if (live) {
#pragma unroll
    for (int j = 0; j < KS; ++j) {
      const int r = 4 * j + lgrp;
      if ((!lines || j >= 4 * NB) && r < D) row[r] = yo[j];
    }
  }
}

// ------------------------------------------------------------------------------------------------
// MFMA tile kernel.  One wave owns FT tiles of 16 frames; a workgroup of WAVES waves shares the
// per-mixture operand block, double-buffered in LDS.
// MODE 0: convert (writes Y).  MODE 1: log-weighted densities l_m (writes LP (M,T)), no A tiles used.
// MODE 2: predict -- the 1-based index of the first maximum of l_m over m (src/gmm.jl:44-47) written as int64 to Y[frame];
// the (M,T) matrix never exists.  MODE 3: the same result from fragments stored tile by tile, last tile first, with an
// EXACT early exit: |z|^2 only grows as tiles are added, so once lc - |z partial|^2 / 2 <= runmax on all 16 frames of a
// frame tile, mixture m cannot be its first maximum and the remaining tiles are skipped.  The LAST rows of the Cholesky
// whitening carry the small conditional variances, i.e. most of a wrong mixture's distance: they go first.
// ------------------------------------------------------------------------------------------------
// PRUNE (MODE 0 only) -- three code shapes of the same arithmetic:
//   2  "peaked": everything the round-3 loop does (last whitening tile first, tests on the lane groups' shares, per-tile
//      branches around idle MFMAs) -- pays when almost every (tile, mixture) pair is decided out early;
//   1  "broad": every whitening tile of every mixture, ONE wave-uniform branch around the regression + softmax update of a
//      mixture that no frame of the wave's tiles gives a posterior above e^-prune -- the loop body is two straight blocks;
//   0  dense (prune = +inf): no test at all.
// In shapes 0 and 1 the weight e^(l - max) is formed BEFORE the regression tiles, so that its dependent chain issues in
// the shadow of their MFMAs instead of after them.
template <int DP, int FT, int WAVES, int MODE, int NBUF, int PRUNE = 2>
__global__ void __launch_bounds__(WAVES * 64) __attribute__((amdgpu_waves_per_eu(DP <= 40 ? (FT == 2 ? 3 : 4) : 2)))
gmmmap_mfma_kernel(const double *__restrict__ packed, int M, int D, const double *__restrict__ X, int64_t ldx,
                   int64_t T, double *__restrict__ Y, int64_t ldy, double prune, unsigned long long *__restrict__ nreg,
                   const int *__restrict__ perm, const int *__restrict__ gkey) {
  // perm / gkey (MODE 0, optional): frames GROUPED by (approximate) mixture -- position fr of the launch is frame perm[fr],
  // gkey[frame] its group -- see gmmmap_group_* below: a 16-frame tile then holds frames of one mixture, the workgroup starts
  // its mixture loop at that mixture, the running maximum is tight from the first iteration and the pruning removes (almost)
  // every other regression.  Frames are independent of each other, so the results do not depend on the grouping.
  using TL = Tiling<DP, MODE >= 1>;
  constexpr int KS = TL::KS, NT = TL::NT, NU = TL::NU, BLK = TL::BLK;
  constexpr int NTHREADS = WAVES * 64;
  constexpr int NV = BLK / 2 / NTHREADS;   // double2 copies per thread per block (BLK is a multiple of 1024)
  static_assert(BLK % (2 * NTHREADS) == 0, "block size must be a whole number of double2 per thread");
  extern __shared__ double smem[];                          // NBUF * BLK doubles
  // MODE 1: write-combining stage for the (T,M) log-density output.  A lane owns a frame and produces one value per
  // mixture iteration; written directly these are 8-byte stores 8*M bytes apart (PMC: WRITE_SIZE 4x the bytes of the
  // matrix).  Values of 8 consecutive mixtures are parked here and then leave as 64-byte rows.
  constexpr int LROW = 10;                                  // 8 values + pad: 16-byte aligned, conflict-free for 16 lanes
  __shared__ __attribute__((aligned(16))) double lstage[(MODE == 1) ? WAVES * FT * 16 * LROW : 2];
  __shared__ double etab[(MODE == 0 && PRUNE < 2) ? 64 : 1];   // 2^(j/64) for vc_exp_tab
  if constexpr (MODE == 0 && PRUNE < 2) {
    if (threadIdx.x < 64) etab[threadIdx.x] = kExp2Tab[threadIdx.x];   // (visible after the barrier that follows the first stage)
  }

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);     // the same number in a scalar register (uniform address arithmetic)
  const unsigned lane_off = 16u * (unsigned)lane;               // this lane's 16 bytes of a 1 KB LDS-DMA wave instruction
  const int lcol = lane & 15;   // frame within a 16-frame tile (MFMA column)
  const int lgrp = lane >> 4;   // k within a k-step (operands) / row offset within a register group (results)
  const int64_t frame0 = ((int64_t)blockIdx.x * WAVES + wave) * (16 * FT);

  // B operands: xb[f][ks] = X[k = 4 ks + lgrp][frame = frame0 + 16 f + lcol], zero outside (D, T)
  double xb[FT][KS];
  int64_t frow[FT];              // the frame a tile column stands for (perm: grouped launch)
#pragma unroll
  for (int f = 0; f < FT; ++f) {
    const int64_t fr = frame0 + 16 * f + lcol;
    frow[f] = (MODE == 0 && perm != nullptr && fr < T) ? (int64_t)perm[fr] : fr;
    load_frame_row<KS>(X + (fr < T ? frow[f] : (int64_t)0) * ldx, fr < T, rows_as_lines(X, ldx, D, DP), D, lgrp, xb[f]);
  }
  // first mixture of the loop: the group of the workgroup's first frame (grouped launch), else 0
  int mfirst = 0;
  if (MODE == 0 && perm != nullptr) {
    const int64_t f0 = (int64_t)blockIdx.x * WAVES * (16 * FT);
    mfirst = (f0 < T) ? gkey[perm[f0]] : 0;
    mfirst = (mfirst >= 0 && mfirst < M) ? mfirst : 0;
  }

  int nreg_wave = 0;              // MODE 0: (tile, mixture) regressions this wave evaluated (diagnostic counter, see nreg)
  int nmfma_wave = 0;             // MODE 0: v_mfma_f64_16x16x4 instructions this wave issued (nreg[1]; wave-uniform scalar adds)
  unsigned tiles_in_range = 0;    // the wave's tiles that hold at least one frame < T
#pragma unroll
  for (int f = 0; f < FT; ++f)
    if (frame0 + 16 * f < T) tiles_in_range |= 1u << f;
  double yacc[FT][KS];
  double runmax[FT], den[FT];
  int bestm[FT];                  // MODE 2: runmax = the largest l_m so far, bestm = its (first) mixture
#pragma unroll
  for (int f = 0; f < FT; ++f) {
    bestm[f] = 0;
    runmax[f] = -INFINITY;
    den[f] = 0.0;
#pragma unroll
    for (int j = 0; j < KS; ++j) yacc[f][j] = 0.0;
  }

  // stage the first block
  {
    const double2 *src = reinterpret_cast<const double2 *>(packed + (size_t)mfirst * BLK);
    double2 *dst = reinterpret_cast<double2 *>(smem);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int e = tid + i * NTHREADS;
      dst[e] = src[e];
    }
  }
  __syncthreads();

  // (the constant address space makes a uniform-address load a scalar load; the blocks are read-only for the kernel)
  const __attribute__((address_space(4))) double *packed_c = (const __attribute__((address_space(4))) double *)packed;
  double lc_next = packed_c[(size_t)mfirst * BLK + TL::LC_OFF];      // (every mode; mfirst = 0 outside MODE 0)
#ifdef VCMI_CONVERT_PROF
  unsigned long long prof_barrier_ = 0;
  const unsigned long long prof_t0_ = __builtin_readcyclecounter();
#endif
#ifdef VCMI_CONVERT_PROF2
  // phase probe (s_memtime counts; reading it waits for the wave's outstanding LDS / scalar operations, which the phase
  // boundaries do anyway): [0] loop top -> lc known, [1] first part of the mixture (peaked: last whitening tile + test;
  // dense / broad: whitening + |z|^2), [2] the rest up to the barrier, [3] vmcnt + barrier
  unsigned long long pp_[4] = {0, 0, 0, 0};
  unsigned long long pt_ = __builtin_readcyclecounter();
#define VCMI_PP(i) { const unsigned long long n_ = __builtin_readcyclecounter(); pp_[i] += n_ - pt_; pt_ = n_; }
#else
#define VCMI_PP(i)
#endif
  // Two copies of the loop body, one per staging buffer (the inner loop is unrolled): the buffer's offset is then a
  // compile-time constant that goes into the immediate offsets of the LDS reads, instead of a run-time base that costs two or
  // three VALU address instructions at the start of every MFMA stream.
  // (Measured, one box: dense 5.10 -> 5.08 ms, peaked 1.700 -> 1.674; the broad shape spills four registers in this form and
  // loses 2 %: it keeps the single copy.)
  constexpr int UNR = (MODE == 0 && PRUNE == 1) ? 1 : 2;
  for (int mo = 0; mo < M; mo += UNR) {
#pragma unroll
  for (int u = 0; u < UNR; ++u) {
    const int mi = mo + u;
    if (UNR > 1 && mi >= M) break;                                         // (odd M: wave-uniform)
    const int par = (UNR > 1) ? u : (mi & 1);
    const int m = (mfirst + mi < M) ? mfirst + mi : mfirst + mi - M;       // mixtures in rotated order (mfirst = 0: index order)
    const double *cur = smem + (NBUF == 2 ? par * BLK : 0);
    double2 *nxt = reinterpret_cast<double2 *>(smem + (NBUF == 2 ? (par ^ 1) * BLK : 0));
    // How block m+1 gets into LDS (unconditionally: the last iteration re-reads its own block, so the loop body has no
    // exec-mask branch).  Rounds 1-3 requested the whole block into registers here and stored it to LDS after the MFMA work:
    // the compiler sinks those loads down to the LDS stores (24 registers it does not have at 3 waves per SIMD), so the full
    // L2 latency stood in front of every barrier.  STAGE_DMA (two buffers): global_load_lds_dwordx4 -- 16 bytes per lane
    // straight into the other buffer, no registers; a wave instruction fills 1 KB; issued here, waited for (vmcnt) just
    // before the barrier.  The target buffer was last read in the previous iteration, before that iteration's barrier.
    // (A single buffer, NBUF = 1 -- blocks beyond 40 KB -- keeps the register path.)
#ifndef VCMI_CONVERT_STAGE
#define VCMI_CONVERT_STAGE 2
#endif
    constexpr bool STAGE_DMA = (VCMI_CONVERT_STAGE == 2 && NBUF == 2);
    double2 pre[STAGE_DMA ? 1 : NV];
    const int mn = (mi + 1 < M) ? ((m + 1 < M) ? m + 1 : 0) : m;
    if constexpr (STAGE_DMA) {
      // Addresses: a UNIFORM base (scalar registers, scalar arithmetic) plus one loop-invariant 32-bit lane offset.  Written
      // with per-lane pointers (base + 16 tid), every one of the NV instructions cost a 64-bit VALU add, a VALU add for the
      // LDS address and a v_readfirstlane to get it into M0 -- and a VALU instruction of any kind waits for the FP64 MFMAs of
      // the SIMD's other waves: the phase probe (-DVCMI_CONVERT_PROF2) showed loop top -> first MFMA taking as long as the 20
      // MFMAs of the peaked loop's first stage.
      // (The builtin does not select the scalar-base form -- it folds the lane offset into a per-lane 64-bit pointer again --
      // so the instruction is written out: M0 = LDS byte address of the wave's 1 KB, saddr = uniform global address,
      // vaddr = the lane's 16-byte offset.)
      const char *gbase = reinterpret_cast<const char *>(packed + (size_t)mn * BLK) + 1024 * wave_u;
      const unsigned lbase = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)(reinterpret_cast<char *>(nxt)) + 1024u * wave_u;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const char *ga = gbase + 16 * NTHREADS * i;
        const unsigned la = lbase + 16u * NTHREADS * i;
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(lane_off), "s"(ga), "s"(la) : "memory", "m0");
      }
      __builtin_amdgcn_sched_barrier(0);
    } else {
      const double2 *nsrc = reinterpret_cast<const double2 *>(packed + (size_t)mn * BLK);
#pragma unroll
      for (int i = 0; i < NV; ++i) pre[i] = nsrc[tid + i * NTHREADS];
    }

    // lc_m: a scalar load from the block's copy in global memory, requested one iteration ahead (MODE 0; a uniform address:
    // s_load, no VALU, no LDS round trip at the top of the iteration), compared as an integer (-inf = zero weight)
    double lc;
    // Where the NEXT lc is requested matters: an outstanding scalar load turns the compiler's s_waitcnt on LDS reads into
    // lgkmcnt(0) (scalar loads return out of order), and the barrier waits for it too.  So: right after the first stage's
    // |z|^2 arithmetic -- its LDS reads are done, ~30 VALU instructions and a branch follow -- (VCMI_LC_NEXT below).
    lc = lc_next;
    bool lcn_done = false;
#define VCMI_LC_NEXT                                             \
  {                                                              \
    lc_next = packed_c[(size_t)mn * BLK + TL::LC_OFF];           \
    lcn_done = true;                                             \
  }
    // the high word of -inf (lc is never NaN): a scalar compare
    const bool has_weight = (unsigned)((unsigned long long)__double_as_longlong(lc) >> 32) != 0xFFF00000u;
#ifdef VCMI_CONVERT_PROF2
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    VCMI_PP(0)
#endif
    if constexpr (MODE == 0 && PRUNE < 2) {
      if (has_weight) {
        d4 acc[FT][NT];
#pragma unroll
        for (int t = 0; t < NU; ++t) {
          d4 c;
#pragma unroll
          for (int r = 0; r < 4; ++r) c[r] = cur[TL::CINIT_OFF + 16 * t + 4 * r + lgrp];
#pragma unroll
          for (int f = 0; f < FT; ++f) acc[f][t] = c;
        }
        // ---------------- phase U: whitening tiles, z = U x - cz (k-step major: NU x FT independent chains) ----------------
        // the fragment of step s+1 is requested before the MFMAs of step s
        constexpr int NUS = TL::tile_off(NU);                 // fragments of the whitening phase
#if VCMI_CONVERT_PRIO == 1
        __builtin_amdgcn_s_setprio(0);
#elif VCMI_CONVERT_PRIO == 2
        __builtin_amdgcn_s_setprio(3);
#endif
        int s = 0;
        double a_cur = cur[lane];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
          for (int t = 0; t < NU; ++t) {
            if (ks < TL::steps(t)) {
              const double a = a_cur;
              ++s;
              if (s < NUS) a_cur = cur[s * 64 + lane];
#pragma unroll
              for (int f = 0; f < FT; ++f) acc[f][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, xb[f][ks], acc[f][t], 0, 0, 0);
            }
          }
        }
        nmfma_wave += FT * NUS;
#if VCMI_CONVERT_PRIO == 1
        __builtin_amdgcn_s_setprio(3);      // A/B: the non-MFMA stretches of a wave at high priority
#elif VCMI_CONVERT_PRIO == 2
        __builtin_amdgcn_s_setprio(0);
#endif
        // |z|^2 per frame tile, then everything scalar about the softmax in the SELECTED layout when the wave has two tiles:
        // the even lane groups carry tile 0's l, running maximum and denominator, the odd ones tile 1's (psel picks) -- one
        // reduction, one test, one exp for both tiles
        double qv[FT];
#pragma unroll
        for (int f = 0; f < FT; ++f) {
          double qq = 0.0;
#pragma unroll
          for (int t = 0; t < NU; ++t) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (16 * t + 4 * r < DP) qq = fma(acc[f][t][r], acc[f][t][r], qq);
          }
          qv[f] = qq;
        }
        VCMI_LC_NEXT
        constexpr bool PAIRED = (FT == 2);
        double lsel;                                   // PAIRED: l of this lane's tile; else l of tile 0
        if constexpr (PAIRED) lsel = lc - 0.5 * sum_lane_groups_pair(qv[0], qv[1]);
        else lsel = lc - 0.5 * sum_lane_groups(qv[0]);
        VCMI_PP(1)
        bool go = true;
        if constexpr (PRUNE == 1) {
          // p_m <= e^(l_m - runmax): below e^-prune on every frame of the wave's tiles -> neither the regression nor the softmax
          // update can change y (the term is under the rounding error of those that are kept)
          go = __builtin_amdgcn_ballot_w64(lsel > runmax[0] - prune) != 0;
        }
        if (go) {
          nreg_wave += __builtin_popcount(tiles_in_range);
          nmfma_wave += FT * (TL::NSTEPS - NUS);
          // lazy rescale: only when some frame of the wave has a new maximum (wave-uniform; sc = 1 exactly for the others)
          if (__builtin_amdgcn_ballot_w64(lsel > runmax[0]) != 0) {
            const double nm = fmax(runmax[0], lsel);
            const double scs = vc_exp(runmax[0] - nm);
            den[0] *= scs;
            runmax[0] = nm;
            double sc[FT];
            if constexpr (PAIRED) unpair_lane_groups(scs, sc[0], sc[1]);
            else sc[0] = scs;
#pragma unroll
            for (int f = 0; f < FT; ++f) {
#pragma unroll
              for (int j = 0; j < KS; ++j) yacc[f][j] *= sc[f];
            }
          }
          double wg[FT];
          {
            const double e = vc_exp_tab(lsel - runmax[0], etab);
            den[0] += e;
            if constexpr (PAIRED) unpair_lane_groups(e, wg[0], wg[1]);
            else wg[0] = e;
          }
          // ---------------- phase A: regression tiles, E = A x + b ----------------
#if VCMI_CONVERT_PRIO == 1
          __builtin_amdgcn_s_setprio(0);
#elif VCMI_CONVERT_PRIO == 2
          __builtin_amdgcn_s_setprio(3);
#endif
#pragma unroll
          for (int t = NU; t < NT; ++t) {
            d4 c;
#pragma unroll
            for (int r = 0; r < 4; ++r) c[r] = cur[TL::CINIT_OFF + 16 * t + 4 * r + lgrp];
#pragma unroll
            for (int f = 0; f < FT; ++f) acc[f][t] = c;
          }
          int sa = NUS;
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
            for (int t = NU; t < NT; ++t) {
              const double a = cur[sa * 64 + lane];
              ++sa;
#pragma unroll
              for (int f = 0; f < FT; ++f) acc[f][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, xb[f][ks], acc[f][t], 0, 0, 0);
            }
          }
#if VCMI_CONVERT_PRIO == 1
          __builtin_amdgcn_s_setprio(3);
#elif VCMI_CONVERT_PRIO == 2
          __builtin_amdgcn_s_setprio(0);
#endif
#pragma unroll
          for (int f = 0; f < FT; ++f) {
#pragma unroll
            for (int t = NU - 1; t < NT; ++t) {
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const int p0 = 16 * t + 4 * r;
                if (p0 >= DP && p0 < 2 * DP) {
                  const int j = (p0 - DP) / 4;
                  yacc[f][j] = fma(wg[f], acc[f][t][r], yacc[f][j]);
                }
              }
            }
          }
        }
      }
    } else
    if (has_weight) {   // zero-weight mixtures have posterior exactly 0 (wave-uniform branch)
      // ---------------- phase U: whitening tiles, z = U x - cz ----------------
      d4 acc[FT][NT];
      double q[FT];
      unsigned live = (1u << FT) - 1u;      // MODE 3: frame tiles of this wave that mixture m can still be the maximum of
      unsigned ulive = (1u << FT) - 1u;     // MODE 0: frame tiles whose whitening was completed (the others are decided: out)
      if constexpr (MODE == 3) {
        double qp[FT];
#pragma unroll
        for (int f = 0; f < FT; ++f) qp[f] = 0.0;
#pragma unroll
        for (int i = 0; i < NU; ++i) {
          const int t = NU - 1 - i;
          if (!live) break;
          d4 c;
#pragma unroll
          for (int r = 0; r < 4; ++r) c[r] = cur[TL::CINIT_OFF + 16 * t + 4 * r + lgrp];
#pragma unroll
          for (int f = 0; f < FT; ++f) acc[f][t] = c;
#pragma unroll
          for (int ks = 0; ks < TL::steps(t); ++ks) {
            const double a = cur[(TL::rtile_off(i) + ks) * 64 + lane];
#pragma unroll
            for (int f = 0; f < FT; ++f)
              if (live >> f & 1u) acc[f][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, xb[f][ks], acc[f][t], 0, 0, 0);
          }
#pragma unroll
          for (int f = 0; f < FT; ++f) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (16 * t + 4 * r < DP) qp[f] = fma(acc[f][t][r], acc[f][t][r], qp[f]);
          }
          if (i + 1 < NU) {
#pragma unroll
            for (int f = 0; f < FT; ++f) {
              if (!(live >> f & 1u)) continue;
              double qq = qp[f];
              qq = sum_lane_groups(qq);
              if (__builtin_amdgcn_ballot_w64(lc - 0.5 * qq > runmax[f]) == 0) live &= ~(1u << f);
            }
          }
        }
        VCMI_LC_NEXT
#pragma unroll
        for (int f = 0; f < FT; ++f) {
          double qq = qp[f];
          qq = sum_lane_groups(qq);
          q[f] = qq;
        }
      } else if constexpr (MODE == 0 && (TL::NU > 1)) {
        // fvconvert: the LAST whitening tile first (it spans all k-steps and carries the rows with the small conditional
        // variances, i.e. most of a wrong mixture's distance).  Its share of |z|^2 alone usually puts the mixture e^-prune
        // under every frame's running maximum -- certainly on grouped frames, where the maximum is the winner's from the
        // first iteration -- and then the other whitening tiles of this mixture are not computed at all (tested on the lane
        // groups' shares, no cross-lane sum; wave-uniform per frame tile).  Every tile keeps its own accumulation order.
        constexpr int TLAST = NU - 1;
        {
          // (round 4) the last tile's KS operand fragments are all requested before its first MFMA -- the compiler's order was
          // read, wait, FT MFMAs, read, wait, ...: an LDS round trip per k-step pair in a stage of only FT * KS MFMAs -- and
          // the other tiles' initial values are read only when they are needed (below)
          d4 c;
#pragma unroll
          for (int r = 0; r < 4; ++r) c[r] = cur[TL::CINIT_OFF + 16 * TLAST + 4 * r + lgrp];
#pragma unroll
          for (int f = 0; f < FT; ++f) acc[f][TLAST] = c;
          double afr[KS];
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) afr[ks] = cur[TL::ufrag_pos(ks, TLAST) * 64 + lane];
          __builtin_amdgcn_sched_barrier(0);       // (the scheduler sinks the reads back to their MFMAs otherwise)
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
            for (int f = 0; f < FT; ++f) acc[f][TLAST] = __builtin_amdgcn_mfma_f64_16x16x4f64(afr[ks], xb[f][ks], acc[f][TLAST], 0, 0, 0);
          }
        }
        nmfma_wave += FT * KS;
        VCMI_LC_NEXT
        unsigned und = (1u << FT) - 1u;        // frame tiles on which the mixture is still undecided
#pragma unroll
        for (int f = 0; f < FT; ++f) {
          double qq = 0.0;
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (16 * TLAST + 4 * r < DP) qq = fma(acc[f][TLAST][r], acc[f][TLAST][r], qq);
          q[f] = qq;
          if (prune < 1e300) {
            const unsigned long long u = __builtin_amdgcn_ballot_w64(lc - 0.5 * qq > runmax[f] - prune);
            if ((u & (u >> 16) & (u >> 32) & (u >> 48) & 0xffffull) == 0) und &= ~(1u << f);
          }
        }
        ulive = und;
        VCMI_PP(1)
        if (und) {
          nmfma_wave += __builtin_popcount(und) * (TL::tile_off(NU) - KS);
#pragma unroll
          for (int t = 0; t < TLAST; ++t) {
            d4 c;
#pragma unroll
            for (int r = 0; r < 4; ++r) c[r] = cur[TL::CINIT_OFF + 16 * t + 4 * r + lgrp];
#pragma unroll
            for (int f = 0; f < FT; ++f) acc[f][t] = c;
          }
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
            for (int t = 0; t < TLAST; ++t) {
              if (ks < TL::steps(t)) {
                const double a = cur[TL::ufrag_pos(ks, t) * 64 + lane];
#pragma unroll
                for (int f = 0; f < FT; ++f)
                  if (FT == 1 || (und >> f & 1u)) acc[f][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, xb[f][ks], acc[f][t], 0, 0, 0);
              }
            }
          }
#pragma unroll
          for (int f = 0; f < FT; ++f) {
            double qq = q[f];
#pragma unroll
            for (int t = 0; t < TLAST; ++t) {
#pragma unroll
              for (int r = 0; r < 4; ++r) qq = fma(acc[f][t][r], acc[f][t][r], qq);
            }
            q[f] = qq;                         // still the lane group's share (see the pruning test below)
          }
        }
      } else {
#pragma unroll
      for (int t = 0; t < NU; ++t) {
        d4 c;
#pragma unroll
        for (int r = 0; r < 4; ++r) c[r] = cur[TL::CINIT_OFF + 16 * t + 4 * r + lgrp];
#pragma unroll
        for (int f = 0; f < FT; ++f) acc[f][t] = c;
      }
      int s = 0;
      if (MODE == 0) nmfma_wave += FT * TL::tile_off(NU);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
        for (int t = 0; t < NU; ++t) {
          if (ks < TL::steps(t)) {
            const double a = cur[s * 64 + lane];
            ++s;
#pragma unroll
            for (int f = 0; f < FT; ++f) acc[f][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, xb[f][ks], acc[f][t], 0, 0, 0);
          }
        }
      }
      VCMI_LC_NEXT
#pragma unroll
      for (int f = 0; f < FT; ++f) {
        double qq = 0.0;
#pragma unroll
        for (int t = 0; t < NU; ++t) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (16 * t + 4 * r < DP) qq = fma(acc[f][t][r], acc[f][t][r], qq);
          }
        }
        if (MODE == 2 || (MODE == 1 && FT != 2)) {   // (MODE 0 reduces over the four lane groups only when it has to: see below;
          qq = sum_lane_groups(qq);                  //  MODE 1 with two tiles: both in one exchange, below)
        }
        q[f] = qq;
      }
      }

      if (MODE == 1) {
        // log-weighted density of mixture m for the wave's frames.  Two tiles: ONE exchange sequence sums both (the even lane
        // groups end with tile 0's |z|^2, the odd ones with tile 1's) and lane groups 0 and 1 write their tile's value
        if constexpr (FT == 2) {
          const double lsel = lc - 0.5 * sum_lane_groups_pair(q[0], q[1]);
          if (lgrp < 2) lstage[((wave * FT + lgrp) * 16 + lcol) * LROW + (m & 7)] = lsel;
        } else if (lgrp == 0) {
#pragma unroll
          for (int f = 0; f < FT; ++f) lstage[((wave * FT + f) * 16 + lcol) * LROW + (m & 7)] = lc - 0.5 * q[f];
        }
      } else if (MODE == 2 || MODE == 3) {
#pragma unroll
        for (int f = 0; f < FT; ++f) {
          if (MODE == 3 && !(live >> f & 1u)) continue;     // l_m <= runmax on the whole tile: not its first maximum
          const double l = lc - 0.5 * q[f];
          if (l > runmax[f]) {          // strict: the first maximum wins, as in posterior_finish_kernel<1>
            runmax[f] = l;
            bestm[f] = m;
          }
        }
      } else {
        // Which of the wave's frame tiles can mixture m still contribute to?  p_m <= e^(l_m - runmax) for every frame, so
        // when l_m < runmax - prune for all 16 frames of a tile, its posterior there is below e^-prune (default e^-46 =
        // 1e-20, under the rounding error of the other terms) and neither the regression tiles A_m x + b_m nor the
        // softmax update can change y: both are skipped for that tile (wave-uniform).  prune = +inf keeps the dense loop.
        // q[f] still holds each lane GROUP's share of |z|^2 (lane = 16 group + frame).  |z|^2 is at least any one share, so a
        // frame is already out when one of its four lanes says so: only tiles with a frame that no single share rules out pay
        // for the cross-lane sum (two LDS round trips) -- for a wrong mixture every share is enormous.
        unsigned active = 0;
#pragma unroll
        for (int f = 0; f < FT; ++f) {
          if (!(ulive >> f & 1u)) continue;     // decided on the last tile's share alone
          if (prune < 1e300) {                  // (the dense loop, prune = +inf, has nothing to decide)
            const unsigned long long undecided = __builtin_amdgcn_ballot_w64(lc - 0.5 * q[f] > runmax[f] - prune);
            if ((undecided & (undecided >> 16) & (undecided >> 32) & (undecided >> 48) & 0xffffull) == 0) continue;
          }
          double qq = q[f];
          qq = sum_lane_groups(qq);
          q[f] = qq;
          if (__builtin_amdgcn_ballot_w64(lc - 0.5 * qq > runmax[f] - prune) != 0) active |= 1u << f;
        }
        if (MODE == 0) nreg_wave += __builtin_popcount(active & tiles_in_range);
        if (active) {
          if (MODE == 0) nmfma_wave += __builtin_popcount(active) * (TL::NSTEPS - TL::tile_off(NU));
          // ---------------- phase A: regression tiles, E = A x + b (wave-uniform branches around an idle tile's MFMAs) ----------------
          int sa = TL::tile_off(NU);
#pragma unroll
          for (int t = NU; t < NT; ++t) {
            d4 c;
#pragma unroll
            for (int r = 0; r < 4; ++r) c[r] = cur[TL::CINIT_OFF + 16 * t + 4 * r + lgrp];
#pragma unroll
            for (int f = 0; f < FT; ++f) acc[f][t] = c;
          }
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
            for (int t = NU; t < NT; ++t) {
              const double a = cur[sa * 64 + lane];
              ++sa;
#pragma unroll
              for (int f = 0; f < FT; ++f)
                if (FT == 1 || (active >> f & 1u)) acc[f][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, xb[f][ks], acc[f][t], 0, 0, 0);
            }
          }
          // online softmax update:  y <- y * e^(old-new) + e^(l-new) * E
#pragma unroll
          for (int f = 0; f < FT; ++f) {
            if (FT > 1 && !(active >> f & 1u)) continue;
            const double l = lc - 0.5 * q[f];
#ifndef VCMI_LAZY
#define VCMI_LAZY 1
#endif
            if (!VCMI_LAZY || __builtin_amdgcn_ballot_w64(l > runmax[f]) != 0) {   // wave-uniform: some frame has a new maximum
              const double nm = fmax(runmax[f], l);
              const double sc = vc_exp(runmax[f] - nm);
              den[f] *= sc;
              runmax[f] = nm;
#pragma unroll
              for (int j = 0; j < KS; ++j) yacc[f][j] *= sc;
            }
            const double wg = vc_exp(l - runmax[f]);
            den[f] += wg;
#pragma unroll
            for (int t = NU - 1; t < NT; ++t) {
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const int p0 = 16 * t + 4 * r;
                if (p0 >= DP && p0 < 2 * DP) {
                  const int j = (p0 - DP) / 4;
                  yacc[f][j] = fma(wg, acc[f][t][r], yacc[f][j]);
                }
              }
            }
          }
        }
      }
    }

    else if (MODE == 1 && lgrp == 0) {
#pragma unroll
      for (int f = 0; f < FT; ++f) lstage[((wave * FT + f) * 16 + lcol) * LROW + (m & 7)] = -INFINITY;
    }
    if (MODE == 1 && ((m & 7) == 7 || m == M - 1)) {
      // flush mixtures m0..m of the wave's FT*16 frames: lane = (frame, half row) -> 32 contiguous bytes each
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the wave's own ds_writes above have landed
      const int m0 = m & ~7, nv = m - m0 + 1;
#pragma unroll
      for (int i = 0; i < (FT * 16 * 2 + 63) / 64; ++i) {
        const int slot = lane + 64 * i, fl = slot >> 1, h = slot & 1;
        if (fl < FT * 16) {
          const int64_t fr = frame0 + fl;
          const double *row = &lstage[(wave * FT * 16 + fl) * LROW + 4 * h];
          double *dst = Y + fr * ldy + m0 + 4 * h;
          if (fr < T) {
            if (nv == 8 && ((ldy | m0) & 1) == 0) {
              const double2 a = *reinterpret_cast<const double2 *>(row), b2 = *reinterpret_cast<const double2 *>(row + 2);
              *reinterpret_cast<double2 *>(dst) = a;
              *reinterpret_cast<double2 *>(dst + 2) = b2;
            } else {
#pragma unroll
              for (int j = 0; j < 4; ++j)
                if (4 * h + j < nv) dst[j] = row[j];
            }
          }
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // reads done before the next group's ds_writes reuse the rows
    }

    if (!lcn_done) lc_next = packed_c[(size_t)mn * BLK + TL::LC_OFF];       // (e.g. a mixture without weight: nothing ran above)
    if (NBUF == 1) __syncthreads();   // single buffer: everyone is done reading before it is overwritten
    VCMI_PP(2)
#ifdef VCMI_CONVERT_PROF
    const unsigned long long tb0_ = __builtin_readcyclecounter();
#endif
    if constexpr (STAGE_DMA) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
#pragma unroll
      for (int i = 0; i < NV; ++i) nxt[tid + i * NTHREADS] = pre[i];
    }
    __syncthreads();
#ifdef VCMI_CONVERT_PROF
    prof_barrier_ += __builtin_readcyclecounter() - tb0_;      // probe build: cycles between arriving at the barrier and leaving it
#endif
    VCMI_PP(3)
  }
  }
#undef VCMI_LC_NEXT
#ifdef VCMI_CONVERT_PROF2
  if (MODE == 0 && blockIdx.x == 1000 && lane == 0)
    printf("convert prof shape %d wave %d: top %llu | first %llu | rest %llu | barrier %llu (s_memtime counts over %d mixtures)\n", PRUNE, wave, pp_[0],
           pp_[1], pp_[2], pp_[3], M);
#endif
#undef VCMI_PP

  if (MODE == 2 || MODE == 3) {
    if (lgrp == 0) {
#pragma unroll
      for (int f = 0; f < FT; ++f) {
        const int64_t fr = frame0 + 16 * f + lcol;
        if (fr < T) reinterpret_cast<int64_t *>(Y)[fr] = bestm[f] + 1;
      }
    }
  }
  if (MODE == 0) {
    if (nreg && lane == 0) {
#ifdef VCMI_CONVERT_PROF
      // probe build: the counters carry s_memtime counts instead -- [0] waiting at the loop's barrier, [1] the whole loop
      atomicAdd(nreg, prof_barrier_);
      atomicAdd(nreg + 1, __builtin_readcyclecounter() - prof_t0_);
#else
      atomicAdd(nreg, (unsigned long long)nreg_wave);
      atomicAdd(nreg + 1, (unsigned long long)nmfma_wave);
#endif
    }
    if constexpr (PRUNE < 2 && FT == 2) {        // the dense / broad loops kept the denominators in the selected layout
      const double ds = den[0];
      unpair_lane_groups(ds, den[0], den[1]);
    }
#pragma unroll
    for (int f = 0; f < FT; ++f) {
      const int64_t fr = frame0 + 16 * f + lcol;
      const double inv = 1.0 / den[f];
      double yo[KS];
#pragma unroll
      for (int j = 0; j < KS; ++j) yo[j] = yacc[f][j] * inv;
      store_frame_row<KS>(Y + (fr < T ? frow[f] : (int64_t)0) * ldy, fr < T, rows_as_lines(Y, ldy, D, DP), D, lgrp, yo);
    }
  }
}

}  // namespace vcmi
#include "gmmmap_screen.hpp"
namespace vcmi {

// ------------------------------------------------------------------------------------------------
// Generic VALU kernel (any D): one lane per frame, x and the y accumulator in LDS ([d][lane] layout,
// conflict-free), parameters read through wave-uniform (scalar) loads.
// MODE 0 convert, 1 log-weighted densities (M,T).
// ------------------------------------------------------------------------------------------------
template <int MODE>
__global__ void __launch_bounds__(64)
gmmmap_generic_kernel(const double *__restrict__ U, const double *__restrict__ A, const double *__restrict__ cz,
                      const double *__restrict__ b, const double *__restrict__ lcv, int M, int D, int DP,
                      const double *__restrict__ X, int64_t ldx, int64_t T, double *__restrict__ Y, int64_t ldy) {
  extern __shared__ double smem[];
  double *xs = smem;              // [D][64]
  double *ys = smem + (size_t)D * 64;   // [D][64] (MODE 0)
  const int lane = threadIdx.x;
  const int64_t fr = (int64_t)blockIdx.x * 64 + lane;
  const bool live = fr < T;
  for (int d = 0; d < D; ++d) {
    xs[d * 64 + lane] = live ? X[fr * ldx + d] : 0.0;
    if (MODE == 0) ys[d * 64 + lane] = 0.0;
  }
  double runmax = -INFINITY, den = 0.0;
  for (int m = 0; m < M; ++m) {
    const double lc = lcv[m];
    if (lc == -INFINITY) {
      if (MODE == 1 && live) Y[fr * ldy + m] = -INFINITY;
      continue;
    }
    const double *Um = U + (size_t)m * DP * DP;
    double q = 0.0;
    for (int i = 0; i < D; ++i) {
      double z = -cz[(size_t)m * DP + i];
      for (int j = 0; j <= i; ++j) z = fma(Um[i * DP + j], xs[j * 64 + lane], z);
      q = fma(z, z, q);
    }
    const double l = lc - 0.5 * q;
    if (MODE == 1) {
      if (live) Y[fr * ldy + m] = l;
    } else {
      const double nm = fmax(runmax, l);
      const double sc = vc_exp(runmax - nm);
      const double wg = vc_exp(l - nm);
      den = fma(den, sc, wg);
      runmax = nm;
      const double *Am = A + (size_t)m * DP * DP;
      for (int i = 0; i < D; ++i) {
        double e = b[(size_t)m * DP + i];
        for (int j = 0; j < D; ++j) e = fma(Am[i * DP + j], xs[j * 64 + lane], e);
        ys[i * 64 + lane] = fma(wg, e, ys[i * 64 + lane] * sc);
      }
    }
  }
  if (MODE == 0 && live) {
    const double inv = 1.0 / den;
    for (int d = 0; d < D; ++d) Y[fr * ldy + d] = ys[d * 64 + lane] * inv;
  }
}

// ------------------------------------------------------------------------------------------------
// Log-weighted densities for the dimensions the MFMA kernel above is not instantiated for, up to D = 16 NTMAX -- above all
// 80 < D <= 160 (e.g. the 160-dimensional joint GMM of delta-augmented features that TrajectoryGMMMap is trained from):
// the x operands of that kernel (D/4 k-steps x FT tiles) and a mixture's whitening block (103 KB at D = 160) no longer
// fit registers / a double-buffered LDS block, so here
//   * a workgroup owns 128 frames: wave w keeps the B-operand fragments of its 16 frames (D/4 doubles per lane) for the
//     whole kernel;
//   * the whitening blocks stream through LDS one 16-row tile at a time ("piece": rows 16r .. 16r+15, the 16(r+1)
//     columns up to the diagonal, straight from the row-major U of the generic layout, zeros above the diagonal),
//     double-buffered: the next piece travels global -> registers during the products of the current one;
//   * z = U x - cz per row tile: accumulators start from -cz, 4(r+1) v_mfma_f64_16x16x4 per tile and wave, |z|^2 summed
//     per frame across the row tiles and the four lane groups.
// Row stride of a piece == 18 (mod 32) doubles: the 16-row fragment reads are conflict-free (as in estep.hip).
// ------------------------------------------------------------------------------------------------
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F &f) {     // f(integral_constant<int, I>) for I .. N-1, unrolled at compile time
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

// Stages of a mixture's whitening block: consecutive row tiles staged and multiplied together.  The first row tiles are
// short (4, 8, 12 ... k-steps): taken one at a time their products do not cover the latency of the next piece's loads,
// so the stages are cut to at least ~28 k-steps each (at NTMAX = 10: tiles 0-3 | 4-5 | 6 | 7 | 8 | 9).
template <int NTMAX>
struct TiledStages {
  static constexpr int kMinSteps = 28;
  int first[NTMAX], last[NTMAX], n;
  constexpr TiledStages() : first{}, last{}, n(0) {
    int r = 0;
    while (r < NTMAX) {
      int e = r, steps = 4 * (r + 1);
      while (steps < kMinSteps && e + 1 < NTMAX) {
        ++e;
        steps += 4 * (e + 1);
      }
      first[n] = r;
      last[n] = e;
      ++n;
      r = e + 1;
    }
  }
  constexpr int rows(int s) const { return 16 * (last[s] - first[s] + 1); }
  constexpr int cols(int s) const { return 16 * (last[s] + 1); }
  constexpr int rs(int s) const { return (cols(s) + 13) / 32 * 32 + 18; }        // == 18 (mod 32): conflict-free fragment reads
  constexpr int piece(int s) const { return rows(s) * rs(s) + rows(s); }         // + the rows' cz values
  constexpr int max_piece() const {
    int v = 0;
    for (int s = 0; s < n; ++s) v = piece(s) > v ? piece(s) : v;
    return v;
  }
};

template <int NTMAX>
__global__ void __launch_bounds__(512)
logdens_tiled_kernel(const double *__restrict__ U, const double *__restrict__ cz, const double *__restrict__ lcv, int M, int D,
                     int DP, const double *__restrict__ X, int64_t ldx, int64_t T, double *__restrict__ LP, int64_t ldy) {
  constexpr int KSMAX = 4 * NTMAX;
  constexpr TiledStages<NTMAX> ST{};
  constexpr int PIECE = ST.max_piece();
  constexpr int NPRE = 4;                               // staged pairs per thread: 512 x 4 x 2 doubles >= the largest stage
  extern __shared__ double tsm[];                       // [2][PIECE]
  typedef double d2 __attribute__((ext_vector_type(2)));
  typedef double d4 __attribute__((ext_vector_type(4)));
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lcol = lane & 15, lgrp = lane >> 4;
  const int NT = (D + 15) / 16;
  const int64_t f = (int64_t)blockIdx.x * 128 + 16 * wave + lcol;
  const int64_t fc = f < T ? f : T - 1;               // frames beyond T read a valid row and are not written

  double xfrag[KSMAX];
#pragma unroll
  for (int ks = 0; ks < KSMAX; ++ks) {
    const int k = 4 * ks + lgrp;
    xfrag[ks] = (k < D) ? X[fc * ldx + k] : 0.0;
  }

  // staging of stage s of mixture m: element pairs (row, 2 c2) of the rows x cols block, zeros above the diagonal and
  // beyond D
  d2 pre[NPRE];
  double precz = 0.0;
  auto fetch = [&](int m, auto sc) {
    constexpr int s = decltype(sc)::value;
    constexpr int W2 = ST.cols(s) / 2, NP = (ST.rows(s) * W2 + 511) / 512, R0 = 16 * ST.first[s];
    static_assert(NP <= NPRE, "stage larger than the staging registers");
    const double *Um = U + (size_t)m * DP * DP;
    // every load is unconditional on a clamped address (a branch or a select on the loaded value would make the wave
    // wait for each load in turn); the zeros above the diagonal and beyond D are applied when the values go to LDS
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int e0 = tid + 512 * i, e = e0 < ST.rows(s) * W2 ? e0 : 0;
      const int row = e / W2, c = 2 * (e - row * W2), gr = R0 + row;
      const int grc = gr < D ? gr : D - 1, cc = c + 1 < DP ? c : DP - 2;
      pre[i] = *reinterpret_cast<const d2 *>(Um + (size_t)grc * DP + cc);
    }
    const int zr = (tid < ST.rows(s) && R0 + tid < D) ? R0 + tid : 0;
    precz = cz[(size_t)m * DP + zr];
  };
  auto stash = [&](double *dst, auto sc) {
    constexpr int s = decltype(sc)::value;
    constexpr int W2 = ST.cols(s) / 2, NP = (ST.rows(s) * W2 + 511) / 512, RS = ST.rs(s);
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int e = tid + 512 * i;
      if (e < ST.rows(s) * W2) {
        const int row = e / W2, c = 2 * (e - row * W2), gr = 16 * ST.first[s] + row;
        d2 v = pre[i];
        v.x = (gr < D && c <= gr) ? v.x : 0.0;
        v.y = (gr < D && c + 1 <= gr) ? v.y : 0.0;
        *reinterpret_cast<d2 *>(dst + row * RS + c) = v;
      }
    }
    if (tid < ST.rows(s)) dst[ST.rows(s) * RS + tid] = (16 * ST.first[s] + tid < D) ? precz : 0.0;
  };
  // one stage: the next one -- (m, s+1), or (m+1, 0) -- travels global -> registers during the products, registers -> the
  // other LDS buffer after them
  auto step = [&](int m, auto sc, double *cur, double *nxt, double &q) {
    constexpr int s = decltype(sc)::value;
    constexpr int RS = ST.rs(s), SN = (s + 1 < ST.n) ? s + 1 : 0;
    const bool more_here = (s + 1 < ST.n) && (ST.first[SN] < NT);
    if (more_here) fetch(m, std::integral_constant<int, SN>{});
    else if (m + 1 < M) fetch(m + 1, std::integral_constant<int, 0>{});
#pragma unroll
    for (int r = ST.first[s]; r <= ST.last[s]; ++r) {
      if (r < NT) {                                     // workgroup-uniform
        const int lr = 16 * (r - ST.first[s]);
        // two accumulator chains (even / odd k-steps), the A fragments read four k-steps ahead of their products
        d4 acc, acc2 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = -cur[ST.rows(s) * RS + lr + 4 * i + lgrp];
        const double *ap = cur + (lr + lcol) * RS + lgrp;
        constexpr int NK = 4, AHEAD = 4;
        const int nks = NK * (r + 1);
        double af[AHEAD];
#pragma unroll
        for (int i = 0; i < AHEAD; ++i) af[i] = ap[4 * i];
#pragma unroll
        for (int ks = 0; ks < nks; ks += AHEAD) {
          double an[AHEAD];
#pragma unroll
          for (int i = 0; i < AHEAD; ++i) an[i] = (ks + AHEAD + i < nks) ? ap[4 * (ks + AHEAD + i)] : 0.0;
#pragma unroll
          for (int i = 0; i < AHEAD; i += 2) {
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(af[i], xfrag[ks + i], acc, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(af[i + 1], xfrag[ks + i + 1], acc2, 0, 0, 0);
          }
#pragma unroll
          for (int i = 0; i < AHEAD; ++i) af[i] = an[i];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const double z = acc[i] + acc2[i];
          q = fma(z, z, q);
        }
      }
    }
    if (more_here) stash(nxt, std::integral_constant<int, SN>{});
    else if (m + 1 < M) stash(nxt, std::integral_constant<int, 0>{});
    __syncthreads();
  };

  fetch(0, std::integral_constant<int, 0>{});
  stash(tsm, std::integral_constant<int, 0>{});
  __syncthreads();
  int pc = 0;                                           // stages done: buffer parity
  for (int m = 0; m < M; ++m) {
    double q = 0.0;
    auto run = [&](auto sc) {
      constexpr int s = decltype(sc)::value;
      if (s < ST.n && ST.first[s < ST.n ? s : 0] < NT) {     // workgroup-uniform
        step(m, std::integral_constant<int, (s < ST.n ? s : 0)>{}, tsm + (pc & 1) * PIECE, tsm + ((pc & 1) ^ 1) * PIECE, q);
        ++pc;
      }
    };
    static_for<0, ST.n>(run);
    q += __shfl_xor(q, 16);
    q += __shfl_xor(q, 32);
    const double lc = lcv[m];
    if (lgrp == 0 && f < T) LP[f * ldy + m] = (lc == -INFINITY) ? -INFINITY : lc - 0.5 * q;
  }
}

// ------------------------------------------------------------------------------------------------
// (M,T) log-weighted densities -> posterior in place (MODE 0) or 1-based argmax (MODE 1).
// Lanes ACROSS the mixtures of a frame (8, 16, 32 or 64 of them: several frames per wave where M is small), so that a wave
// reads whole rows of the (T,M) matrix.  (Round 5: one lane per frame walked its row three times with a stride of M doubles
// between the lanes -- 1.47 ms for 5e5 x 64, fifteen times what the bytes take; tools/efficiency_sweep.py posterior.)
// Follows src/gmm.jl:28-29 (max-shifted log-sum-exp, exp(l - lse)) and :46 (the FIRST maximum: ties go to the smaller index).
// ------------------------------------------------------------------------------------------------
template <int MODE>
__global__ void __launch_bounds__(256)
posterior_finish_kernel(double *__restrict__ LP, int M, int64_t T, int64_t *__restrict__ idx) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lpf = M <= 8 ? 8 : (M <= 16 ? 16 : (M <= 32 ? 32 : 64)), fpw = 64 / lpf, sub = lane / lpf, sl = lane % lpf;
  for (int64_t f0 = ((int64_t)blockIdx.x * 4 + wave) * fpw; f0 < T; f0 += (int64_t)gridDim.x * 4 * fpw) {
    const int64_t fr = f0 + sub;
    const bool live = fr < T;
    double *l = LP + (live ? fr : T - 1) * M;
    // the lane's first maximum among m = sl, sl + lpf, ... (increasing m, strict >), then across the lanes
    double u = -INFINITY;
    int best = 0x7fffffff;
    if (sl < M) {
      u = l[sl];
      best = sl;
      for (int m = sl + lpf; m < M; m += lpf) {
        const double v = l[m];
        if (v > u) {
          u = v;
          best = m;
        }
      }
    }
    for (int o = lpf / 2; o >= 1; o >>= 1) {
      const double ou = __shfl_xor(u, o);
      const int ob = __shfl_xor(best, o);
      const bool take = ou > u || (ou == u && ob < best);
      u = take ? ou : u;
      best = take ? ob : best;
    }
    if (MODE == 1) {
      if (live && sl == 0) idx[fr] = best + 1;
      continue;
    }
    double s = 0.0;
    for (int m = sl; m < M; m += lpf) s += vc_exp(l[m] - u);
    for (int o = lpf / 2; o >= 1; o >>= 1) s += __shfl_xor(s, o);
    const double lse = u + log(s);
    if (live)
      for (int m = sl; m < M; m += lpf) l[m] = vc_exp(l[m] - lse);
  }
}
static inline unsigned posterior_finish_grid(int M, int64_t T) {
  const int lpf = M <= 8 ? 8 : (M <= 16 ? 16 : (M <= 32 ? 32 : 64)), fpw = 64 / lpf;
  return (unsigned)std::min<int64_t>((T + 4 * fpw - 1) / (4 * fpw), 8192);
}

// ------------------------------------------------------------------------------------------------
// fvconvert for dimensions the MFMA tile kernel has no instantiation for (80 < padded D <= 160, e.g. the 82..96-dimensional
// static + delta vectors of 41..48 coefficients): log-weighted densities by logdens_tiled_kernel (MFMA), then this kernel --
// one wave per frame: softmax over the mixtures, and  y = sum_m p_m (A_m x + b_m)  over the mixtures whose posterior
// reaches e^-prune (the same rule as the tile kernel's pruning; prune = +inf: all of them), in mixture order.  A_m is read
// TRANSPOSED (At[m][k][d]: lanes along d are contiguous); x sits in LDS and is broadcast per k.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
convert_from_logdens_kernel(const double *__restrict__ LP, int M, int D, int DP, const double *__restrict__ At,
                            const double *__restrict__ b, const double *__restrict__ X, int64_t ldx, int64_t T,
                            double *__restrict__ Y, int64_t ldy, double prune) {
  __shared__ double xs_all[4][160];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double *xs = xs_all[wave];
  for (int64_t fr = (int64_t)blockIdx.x * 4 + wave; fr < T; fr += (int64_t)gridDim.x * 4) {
    const double *l = LP + fr * M;
    double u = -INFINITY;
    for (int m = lane; m < M; m += 64) u = fmax(u, l[m]);
#pragma unroll
    for (int sh = 1; sh < 64; sh <<= 1) u = fmax(u, __shfl_xor(u, sh));
    double sden = 0.0;
    for (int m0 = 0; m0 < M; m0 += 64) {        // fixed order: lanes' terms of a 64-mixture group summed by a butterfly
      double e = (m0 + lane < M) ? vc_exp(l[m0 + lane] - u) : 0.0;
#pragma unroll
      for (int sh = 1; sh < 64; sh <<= 1) e += __shfl_xor(e, sh);
      sden += e;
    }
    for (int k = lane; k < D; k += 64) xs[k] = X[fr * ldx + k];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    double y0 = 0.0, y1 = 0.0, y2 = 0.0;
    for (int m = 0; m < M; ++m) {
      const double lm = l[m];                                   // wave-uniform (same address on every lane)
      if (!(lm > u - prune)) continue;
      const double p = vc_exp(lm - u) / sden;
      const double *am = At + (size_t)m * DP * DP + lane;
      double a0 = (lane < D) ? b[(size_t)m * DP + lane] : 0.0, a1 = (lane + 64 < D) ? b[(size_t)m * DP + lane + 64] : 0.0,
             a2 = (lane + 128 < D) ? b[(size_t)m * DP + lane + 128] : 0.0;
      for (int k = 0; k < D; ++k) {
        const double xk = xs[k];
        const double *row = am + (size_t)k * DP;
        if (lane < DP) a0 = fma(row[0], xk, a0);
        if (lane + 64 < DP) a1 = fma(row[64], xk, a1);
        if (lane + 128 < DP) a2 = fma(row[128], xk, a2);
      }
      y0 = fma(p, a0, y0);
      y1 = fma(p, a1, y1);
      y2 = fma(p, a2, y2);
    }
    if (lane < D) Y[fr * ldy + lane] = y0;
    if (lane + 64 < D) Y[fr * ldy + lane + 64] = y1;
    if (lane + 128 < D) Y[fr * ldy + lane + 128] = y2;
    __builtin_amdgcn_wave_barrier();                            // xs is rewritten for the wave's next frame
  }
}

// ------------------------------------------------------------------------------------------------
// Grouping of the frames before fvconvert (gmmmap_mfma_kernel MODE 0 with perm / gkey).
// The pruning of the convert kernel works per 16-frame tile: a regression is skipped when NONE of the tile's frames gives the
// mixture a posterior above e^-prune.  Frames that arrive in an order unrelated to their mixtures (the benchmark draws them
// independently: ~14 of 64 mixtures own a frame of a tile, and the loop meets ~24 before its running maximum is tight) make
// that union large; the same frames grouped by mixture leave one or two.  The grouping does not have to be right -- any
// permutation gives the same y, frame by frame -- only cheap and mostly right: the NEAREST SOURCE MEAN (Euclidean), one
// small MFMA product per tile ([-2 mu | |mu|^2] x [x ; 1], 44 MFMAs per 16 frames at D = 40, M = 64 against 2400 for the
// conversion).  Three kernels: keys + a histogram per chunk of 1024 frames, a prefix over (group, chunk), and a STABLE
// counting-sort scatter (round 4; the first version ranked with LDS / global atomics, which left the order inside a group --
// hence the tiles, the rotated mixture order of a boundary workgroup and the last bit of a frame shared by several
// mixtures -- to the scheduler).
// gfrag[mt][ks][lane]: A-operand fragments, rows = mixtures 16 mt + (lane & 15), k = 4 ks + (lane >> 4); the last k-step
// carries |mu|^2 (rows >= M: 1e300, never the minimum).
// ------------------------------------------------------------------------------------------------
constexpr int kGroupKeyDims = 24;     // dimensions the nearest-mean key is taken over (a multiple of 4)

// keys + one histogram per CHUNK of 1024 consecutive frames (chunkhist[c][m]); a workgroup walks chunks blockIdx.x,
// blockIdx.x + gridDim.x, ... with the operand fragments staged once.  The counts are integers: whatever order the LDS
// atomics arrive in, the histogram is the same.
template <int DP>
__global__ void __launch_bounds__(256)
gmmmap_group_key_kernel(const double *__restrict__ gfrag, int M, int D, const double *__restrict__ X, int64_t ldx, int64_t T,
                        int *__restrict__ key, int *__restrict__ chunkhist) {
  // KS: the k-steps the key LOOKS AT -- the first 24 dimensions at most (kGroupKeyDims).  The key only has to be cheap and
  // mostly right, and the distance over 24 of 40 dimensions picks the same groups (CPU simulation of the kernel's rule, 6e4
  // frames: regressions evaluated 0.0168 / 0.0168 on the SURVEY 8d model, 0.6765 / 0.6760 on the reference's trained model,
  // 0.0351 / 0.0341 on the broad synthetic one; 16 dimensions: 0.0168 / 0.680 / 0.081) for 28 instead of 44 MFMAs per tile.
  constexpr int KS = (DP / 4 < kGroupKeyDims / 4) ? DP / 4 : kGroupKeyDims / 4, KS1 = KS + 1;
  extern __shared__ double gsm[];
  const int MT = (M + 15) / 16, nfrag = MT * KS1 * 64;
  int *hist = reinterpret_cast<int *>(gsm + nfrag);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lcol = lane & 15, lgrp = lane >> 4;
  for (int e = tid; e < nfrag; e += 256) gsm[e] = gfrag[e];
  const double one = (lgrp == 0) ? 1.0 : 0.0;
  const int64_t nchunks = (T + kGroupChunk - 1) / kGroupChunk;
  for (int64_t c = blockIdx.x; c < nchunks; c += gridDim.x) {
    for (int m = tid; m < M; m += 256) hist[m] = 0;
    __syncthreads();
    for (int i = 0; i < kGroupChunk / 64; ++i) {
      const int64_t fr = c * kGroupChunk + 16 * (4 * i + wave) + lcol;
      if (fr - lcol >= T) break;                                    // (wave-uniform)
      double xb[KS];
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int k = 4 * ks + lgrp;
        xb[ks] = (fr < T && k < D) ? X[fr * ldx + k] : 0.0;
      }
      double best = INFINITY;
      int bm = 0;
      for (int mt = 0; mt < MT; ++mt) {
        const double *A = gsm + (size_t)mt * KS1 * 64 + lane;
        d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A[ks * 64], xb[ks], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A[KS * 64], one, acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (acc[r] < best) {
            best = acc[r];
            bm = 16 * mt + 4 * r + lgrp;
          }
      }
#pragma unroll
      for (int sh = 16; sh < 64; sh <<= 1) {
        const double ov = __shfl_xor(best, sh);
        const int om = __shfl_xor(bm, sh);
        if (ov < best || (ov == best && om < bm)) {
          best = ov;
          bm = om;
        }
      }
      if (lgrp == 0 && fr < T) {
        bm = bm < M ? bm : 0;
        key[fr] = bm;
        atomicAdd(&hist[bm], 1);
      }
    }
    __syncthreads();
    for (int m = tid; m < M; m += 256) chunkhist[c * M + m] = hist[m];
    __syncthreads();
  }
}

// The same keys on the BF16 matrix pipe (round 5).  The key only has to be mostly right, and the FP64 products were half of the
// kernel's time (28 MFMAs of 64 cycles per tile beside a read of x that is bound by HBM): -2 mu and x are split into bf16
// hi + lo (gmmmap_screen.hpp: split_bf16; products to ~2^-16 relative) and the distances over the first 24 dimensions are three
// v_mfma_f32_16x16x32_bf16 per 16 mixtures (slot j of lane group g <-> feature 4 j + g: the lane's own FP64 operand), |mu|^2
// in the accumulator's initial value.  gfrag16[mt]: 64 x 16 bytes of hi, 64 x 16 bytes of lo, 4 x 4 floats of |mu|^2 (lane
// group g of the result holds rows 4 g .. 4 g + 3; rows >= M: 1e30).  Deterministic like the FP64 kernel; a frame between
// two means may get the other key, which costs a regression, never a result.
constexpr int kKey16TileBytes = 2 * 1024 + 64;
template <int DP>
__global__ void __launch_bounds__(256)
gmmmap_group_key16_kernel(const double *__restrict__ gfrag16, int M, int D, const double *__restrict__ X, int64_t ldx, int64_t T,
                          int *__restrict__ key, int *__restrict__ chunkhist) {
  constexpr int KS = (DP / 4 < kGroupKeyDims / 4) ? DP / 4 : kGroupKeyDims / 4;
  static_assert(KS <= 8, "one K = 32 instruction per term");
  extern __shared__ double gsm[];
  const int MT = (M + 15) / 16, nd = MT * (kKey16TileBytes / 8);
  int *hist = reinterpret_cast<int *>(gsm + nd);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lcol = lane & 15, lgrp = lane >> 4;
  for (int e = tid; e < nd; e += 256) gsm[e] = gfrag16[e];
  const int64_t nchunks = (T + kGroupChunk - 1) / kGroupChunk;
  for (int64_t c = blockIdx.x; c < nchunks; c += gridDim.x) {
    for (int m = tid; m < M; m += 256) hist[m] = 0;
    __syncthreads();
    // two frame tiles per pass: both tiles' rows are requested before either is used (the kernel is a chain of exposed load
    // latencies: 16 passes per wave and chunk, each waiting for its rows -- round 6)
    for (int i = 0; i < kGroupChunk / 64; i += 2) {
      const int64_t fr0 = c * kGroupChunk + 16 * (4 * i + wave) + lcol;
      if (fr0 - lcol >= T) break;                                   // (wave-uniform)
      double xv[2][KS];
      const bool klines = rows_as_lines(X, ldx, D, DP);             // (whole lines for the first 16 features: load_frame_row)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int64_t fr = fr0 + 64 * u;
        load_frame_row<KS>(X + (fr < T ? fr : T - 1) * ldx, true, klines, D, lgrp, xv[u]);
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int64_t fr = fr0 + 64 * u;
        if (fr - lcol >= T) break;                                  // (wave-uniform: the second tile of the pass lies beyond T)
        u32x4_t bh = {0u, 0u, 0u, 0u}, bl = {0u, 0u, 0u, 0u};
        {
          unsigned short h[8] = {0, 0, 0, 0, 0, 0, 0, 0}, l[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) {
            const int k = 4 * ks + lgrp;
            const double x = (fr < T && k < D) ? xv[u][ks] : 0.0;
            split_bf16(x, h[ks], l[ks]);
          }
#pragma unroll
          for (int w2 = 0; w2 < 4; ++w2) {
            bh[w2] = (unsigned)h[2 * w2] | ((unsigned)h[2 * w2 + 1] << 16);
            bl[w2] = (unsigned)l[2 * w2] | ((unsigned)l[2 * w2 + 1] << 16);
          }
        }
        float best = INFINITY;
        int bm = 0;
        for (int mt = 0; mt < MT; ++mt) {
          const char *tb = reinterpret_cast<const char *>(gsm) + (size_t)mt * kKey16TileBytes;
          const u32x4_t aph = *reinterpret_cast<const u32x4_t *>(tb + 16 * lane), apl = *reinterpret_cast<const u32x4_t *>(tb + 1024 + 16 * lane);
          f32x4_t acc = *reinterpret_cast<const f32x4_t *>(tb + 2048 + 16 * lgrp);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, aph), __builtin_bit_cast(bf16x8_t, bh), acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, aph), __builtin_bit_cast(bf16x8_t, bl), acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, apl), __builtin_bit_cast(bf16x8_t, bh), acc, 0, 0, 0);
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (acc[r] < best) {
              best = acc[r];
              bm = 16 * mt + 4 * lgrp + r;
            }
        }
#pragma unroll
        for (int sh = 16; sh < 64; sh <<= 1) {
          const float ov = __shfl_xor(best, sh);
          const int om = __shfl_xor(bm, sh);
          if (ov < best || (ov == best && om < bm)) {
            best = ov;
            bm = om;
          }
        }
        if (lgrp == 0 && fr < T) {
          bm = bm < M ? bm : 0;
          key[fr] = bm;
          atomicAdd(&hist[bm], 1);
        }
      }
    }
    __syncthreads();
    for (int m = tid; m < M; m += 256) chunkhist[c * M + m] = hist[m];
    __syncthreads();
  }
}

// One workgroup per group m: chunkhist[c][m] -> its exclusive prefix over the chunks (in place) and total[m].
__global__ void __launch_bounds__(256)
gmmmap_group_scan_kernel(int *__restrict__ chunkhist, int64_t nchunks, int M, int *__restrict__ total, const int64_t *__restrict__ gate) {
  if (gate && *gate == 0) return;
  __shared__ int wtot[4];
  const int m = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t per = (nchunks + 255) / 256, lo = std::min<int64_t>(nchunks, tid * per), hi = std::min<int64_t>(nchunks, lo + per);
  int sum = 0;
  for (int64_t c = lo; c < hi; ++c) sum += chunkhist[c * M + m];
  // exclusive prefix of the 256 stretch sums: shuffles within a wave, the four wave totals through LDS (integers: any order
  // gives the same numbers; one thread walking 256 LDS entries took 7 of the kernel's 10 us)
  int inc = sum;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int t = __shfl_up(inc, d);
    if (lane >= d) inc += t;
  }
  if (lane == 63) wtot[wave] = inc;
  __syncthreads();
  int base = 0;
  for (int w = 0; w < wave; ++w) base += wtot[w];
  if (tid == 255) total[m] = base + inc;
  int run = base + inc - sum;
  for (int64_t c = lo; c < hi; ++c) {
    const int v = chunkhist[c * M + m];
    chunkhist[c * M + m] = run;
    run += v;
  }
}

// perm: frames in group order, and inside a group in FRAME order (a stable counting sort: the permutation, hence every tile
// of the convert kernel and every sum it forms, is a function of the data alone -- repeat runs are bit-identical).  One
// workgroup per chunk; position = sum of the smaller groups' totals + the group's frames in earlier chunks (chunkhist after
// the scan) + those in earlier 64-frame rows of this chunk + those on lower lanes of the row.
__global__ void __launch_bounds__(256)
gmmmap_group_scatter_kernel(const int *__restrict__ key, int64_t T, int M, const int *__restrict__ chunkhist,
                            const int *__restrict__ total, int *__restrict__ perm, const int64_t *__restrict__ gate) {
  if (gate && *gate == 0) return;
  extern __shared__ int lsm[];             // [M] group bases, then [16][M] row counts -> row bases
  int *base = lsm, *rowcnt = lsm + M;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t c = blockIdx.x, f0 = c * kGroupChunk;
  // group bases: exclusive prefix of the totals (wave 0: a lane sums its stretch, the lanes' sums are scanned by shuffles) + the
  // group's frames in earlier chunks.  (One thread walking total[] and chunkhist[] through global memory took 10 us per
  // workgroup at 128 groups: most of this kernel's time.)
  for (int m = tid; m < M; m += 256) base[m] = total[m];
  for (int e = tid; e < 16 * M; e += 256) rowcnt[e] = 0;
  __syncthreads();
  if (wave == 0) {
    const int per = (M + 63) / 64, lo = lane * per, hi = (lo + per < M) ? lo + per : M;
    int s = 0;
    for (int m = lo; m < hi; ++m) s += base[m];
    int incl = s;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int v = __shfl_up(incl, o);
      if (lane >= o) incl += v;
    }
    int run = incl - s;
    for (int m = lo; m < hi; ++m) {
      const int v = base[m];
      base[m] = run;
      run += v;
    }
  }
  __syncthreads();
  for (int m = tid; m < M; m += 256) base[m] += chunkhist[c * M + m];
  int k[4], rank[4];
  const int nbits = 32 - __builtin_clz((unsigned)(M > 1 ? M - 1 : 1));
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = wave + 4 * i;                                   // row of 64 consecutive frames
    const int64_t fr = f0 + 64 * r + lane;
    k[i] = fr < T ? key[fr] : -1;
    // the lanes of the row that hold the same key, by one ballot per key BIT (log2 M turns whatever the number of distinct
    // keys: unsorted keys of 128 groups put ~40 distinct ones into a row, one turn each in the first version)
    unsigned long long eq = __builtin_amdgcn_ballot_w64(k[i] >= 0);
    for (int b = 0; b < nbits; ++b) {
      const bool bit = (k[i] >> b) & 1;
      const unsigned long long bal = __builtin_amdgcn_ballot_w64(bit);
      eq &= bit ? bal : ~bal;
    }
    rank[i] = __builtin_amdgcn_mbcnt_hi((unsigned)(eq >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)eq, 0));
    if (k[i] >= 0 && rank[i] == 0) rowcnt[r * M + k[i]] = __builtin_popcountll(eq);
  }
  __syncthreads();
  for (int m = tid; m < M; m += 256) {
    int run = 0;
    for (int r = 0; r < 16; ++r) {
      const int v = rowcnt[r * M + m];
      rowcnt[r * M + m] = run;
      run += v;
    }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = wave + 4 * i;
    const int64_t fr = f0 + 64 * r + lane;
    if (k[i] >= 0) perm[base[k[i]] + rowcnt[r * M + k[i]] + rank[i]] = (int)fr;
  }
}

// ------------------------------------------------------------------------------------------------
// launch helpers
// ------------------------------------------------------------------------------------------------
template <int DP, int MODE, int FTV, int WV, int PRUNE = 2>
static int launch_mfma(const vcmi_gmmmap *g, const double *dX, int64_t ldx, int64_t T, double *dY, int64_t ldy,
                       hipStream_t st, const int *perm = nullptr, const int *gkey = nullptr) {
  constexpr int FT = (MODE >= 1) ? 2 : FTV;
  constexpr int WAVES = WV;
  using TL = Tiling<DP, MODE >= 1>;
  // double-buffer the per-mixture block when two copies fit in half of the CU's 160 KiB LDS
  // (eight-wave workgroups -- one per CU -- take two buffers whenever they fit the CU's LDS at all)
  constexpr int NBUF = (2 * (size_t)TL::BLK * sizeof(double) <= (WAVES == 8 ? 156 : 80) * 1024) ? 2 : 1;
  const size_t shmem = NBUF * (size_t)TL::BLK * sizeof(double);
  auto kern = gmmmap_mfma_kernel<DP, FT, WAVES, MODE, NBUF, PRUNE>;
  // the attribute is per DEVICE: one flag per device, so a host that drives several GPUs sets it on each of them
  static std::atomic<bool> attr_done[64];
  int dev = 0;
  VCMI_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64 || !attr_done[dev].load(std::memory_order_acquire)) {
    VCMI_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)shmem));
    if (dev >= 0 && dev < 64) attr_done[dev].store(true, std::memory_order_release);
  }
  const int64_t per_wg = (int64_t)16 * FT * WAVES;
  const int64_t blocks = (T + per_wg - 1) / per_wg;
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(WAVES * 64), shmem, st, MODE == 3 ? g->packedU2.p : (MODE >= 1 ? g->packedU.p : g->packed.p), g->M,
                     g->D, dX, ldx, T, dY, ldy, g->prune, MODE == 0 ? g->prune_count.p : nullptr, MODE == 0 ? perm : nullptr,
                     MODE == 0 ? gkey : nullptr);
  VCMI_HIP(hipGetLastError());
  return VCMI_OK;
}

// Calls of up to this many frames do not fill the chip with 128-frame workgroups (2000 frames: 16 of 256 CUs, each walking
// all the mixtures for two frame tiles per wave): they run ONE frame tile per wave -- twice the workgroups, half the loop
// each.  Device-resident call of 2000 frames, D = 40, M = 64: 224 -> 127 us (the reference's 32-mixture model: 118 -> 79;
// one frame: 129 -> 83).  Grouping such calls (three more launches) does not pay: tools/small_T_sweep.py.
static constexpr int64_t kSmallCallFrames = 32768;
static constexpr int64_t kSortMinFrames = 8192;

#ifndef VCMI_CONVERT_FT
#define VCMI_CONVERT_FT 2          // frame tiles per wave and waves per workgroup of the D <= 48 convert kernels (A/B builds)
#endif
#ifndef VCMI_CONVERT_WAVES
#define VCMI_CONVERT_WAVES 4
#endif
// (the arg-max kernel, MODE 3, takes the eight waves as well: 3.48 -> 3.25 ms per 512k frames at D = 80; the log-density kernel,
// MODE 1, gains nothing from them -- its two four-wave workgroups per CU already fill the register file: 3.87 / 3.85 ms)
#ifndef VCMI_WIDE_ALL_MODES
#define VCMI_WIDE_ALL_MODES 0
#endif
#ifndef VCMI_CONVERT_WAVES_WIDE
// Waves per workgroup of the fvconvert kernels beyond D = 48 (one frame tile per wave).  A mixture's block is 46 KB (DP = 52) to
// 102 KB (DP = 80) there, so a CU holds ONE or two workgroups: with four waves that was four to eight waves per CU, each
// workgroup streaming the whole model through LDS for its 64 frames.  Eight waves share a block (and two buffers where they
// fit: DP <= 72): 5e5 frames, M = 64: D = 56 8.06 -> 3.16 ms, D = 64 8.50 -> 3.34, D = 72 15.1 -> 4.54, D = 80 20.7 -> 10.1
// (tools/efficiency_sweep.py; A/B: -DVCMI_CONVERT_WAVES_WIDE=4).
#define VCMI_CONVERT_WAVES_WIDE 8
#endif
template <int MODE, int PRUNE = 2>
static int dispatch_mfma(const vcmi_gmmmap *g, const double *dX, int64_t ldx, int64_t T, double *dY, int64_t ldy,
                         hipStream_t st, const int *perm = nullptr, const int *gkey = nullptr) {
  if (MODE == 0 && g->DP <= 48 && T <= kSmallCallFrames && !debug_flag(kDbgConvertWideTiles)) {     // one frame tile per wave: twice the workgroups, half the loop each
    switch (g->DP) {
#define VCMI_CASE(DPV) \
  case DPV: return launch_mfma<DPV, MODE, 1, 4, PRUNE>(g, dX, ldx, T, dY, ldy, st, perm, gkey);
      VCMI_CASE(16) VCMI_CASE(20) VCMI_CASE(24) VCMI_CASE(28) VCMI_CASE(32) VCMI_CASE(36) VCMI_CASE(40) VCMI_CASE(44) VCMI_CASE(48)
#undef VCMI_CASE
      default: break;
    }
  }
  switch (g->DP) {
#define VCMI_CASE(DPV) \
  case DPV: return launch_mfma<DPV, MODE, ((DPV) <= 48 ? VCMI_CONVERT_FT : 1), ((DPV) == 40 && MODE == 0 ? VCMI_CONVERT_WAVES : ((DPV) > 48 && (MODE == 0 || MODE == 3 || VCMI_WIDE_ALL_MODES) ? VCMI_CONVERT_WAVES_WIDE : 4)), PRUNE>(g, dX, ldx, T, dY, ldy, st, perm, gkey);
    VCMI_CASE(16) VCMI_CASE(20) VCMI_CASE(24) VCMI_CASE(28) VCMI_CASE(32) VCMI_CASE(36) VCMI_CASE(40) VCMI_CASE(44)
    VCMI_CASE(48) VCMI_CASE(52) VCMI_CASE(56) VCMI_CASE(60) VCMI_CASE(64) VCMI_CASE(68) VCMI_CASE(72) VCMI_CASE(76)
    VCMI_CASE(80)
#undef VCMI_CASE
    default: return fail(VCMI_ERR_ARG, "no MFMA instantiation for padded dimension %d", g->DP);
  }
}

bool gmmmap_has_mfma(int DP) {
  switch (DP) {
    case 16: case 20: case 24: case 28: case 32: case 36: case 40: case 44: case 48: case 52: case 56: case 60: case 64:
    case 68: case 72: case 76: case 80: return true;
    default: return false;
  }
}

template <int MODE>
static int launch_generic(const vcmi_gmmmap *g, const double *dX, int64_t ldx, int64_t T, double *dY, int64_t ldy,
                          hipStream_t st) {
  const size_t shmem = (size_t)g->D * 64 * sizeof(double) * (MODE == 0 ? 2 : 1);
  if (shmem > 160 * 1024) return fail(VCMI_ERR_ARG, "dimension %d too large for the generic kernel", g->D);
  auto kern = gmmmap_generic_kernel<MODE>;
  if (shmem > 64 * 1024)
    VCMI_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)shmem));
  const int64_t blocks = (T + 63) / 64;
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(64), shmem, st, g->U.p, g->A.p, g->cz.p, g->b.p, g->lc.p, g->M,
                     g->D, g->DP, dX, ldx, T, dY, ldy);
  VCMI_HIP(hipGetLastError());
  return VCMI_OK;
}

static bool use_mfma(const vcmi_gmmmap *g) {
  if (g->kernel_choice == 1) return false;
  return gmmmap_has_mfma(g->DP);
}

// shape 3 (gmmmap_screen.hpp): DP = 16..48, any M up to 1024 (the survivors' bitmap)
static bool screen_has_kernel(int DP) { return DP >= 16 && DP <= 48 && DP % 4 == 0; }
template <int DP, int FT, bool B16 = false>
static int launch_screen(const vcmi_gmmmap *g, const double *dX, int64_t ldx, int64_t T, double *dY, int64_t ldy, hipStream_t st,
                         const int *perm, const int *gkey) {
  constexpr int WAVES = 4;
  using TL = Tiling<DP, false>;
  constexpr int STG = B16 ? screen16_stage_doubles(DP) : screen_stage_doubles(DP);
  constexpr int BUF = (TL::BLK > STG) ? TL::BLK : STG;
  const size_t shmem = 2 * (size_t)BUF * sizeof(double);
  auto kern = gmmmap_screen_kernel<DP, FT, WAVES, B16>;
  static std::atomic<bool> attr_done[64];
  int dev = 0;
  VCMI_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64 || !attr_done[dev].load(std::memory_order_acquire)) {
    VCMI_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    if (dev >= 0 && dev < 64) attr_done[dev].store(true, std::memory_order_release);
  }
  const int64_t per_wg = (int64_t)16 * FT * WAVES;
  hipLaunchKernelGGL(kern, dim3((unsigned)((T + per_wg - 1) / per_wg)), dim3(WAVES * 64), shmem, st, g->packed.p, B16 ? g->packedQ16.p : g->packedQ.p, g->screen_rpm,
                     g->M, g->D, dX, ldx, T, dY, ldy, g->prune, g->prune_count.p, perm, gkey);
  VCMI_HIP(hipGetLastError());
  return VCMI_OK;
}
static int dispatch_screen(const vcmi_gmmmap *g, const double *dX, int64_t ldx, int64_t T, double *dY, int64_t ldy, hipStream_t st,
                           const int *perm, const int *gkey) {
  const bool narrow = T <= kSmallCallFrames && !debug_flag(kDbgConvertWideTiles);     // one frame tile per wave, as dispatch_mfma
  // the screen on the BF16 matrix pipe (certified bound from split operands) where it exists: four rows per mixture, DP <= 40
  if (g->packedQ16.p && g->screen_rpm == 4 && screen16_has(g->DP) && !debug_flag(kDbgScreenFp64)) {
    switch (g->DP) {
#define VCMI_CASE(DPV) \
  case DPV: return narrow ? launch_screen<DPV, 1, true>(g, dX, ldx, T, dY, ldy, st, perm, gkey) : launch_screen<DPV, 2, true>(g, dX, ldx, T, dY, ldy, st, perm, gkey);
      VCMI_CASE(16) VCMI_CASE(20) VCMI_CASE(24) VCMI_CASE(28) VCMI_CASE(32) VCMI_CASE(36) VCMI_CASE(40)
#undef VCMI_CASE
      default: break;
    }
  }
  switch (g->DP) {
#define VCMI_CASE(DPV) \
  case DPV: return narrow ? launch_screen<DPV, 1>(g, dX, ldx, T, dY, ldy, st, perm, gkey) : launch_screen<DPV, 2>(g, dX, ldx, T, dY, ldy, st, perm, gkey);
    VCMI_CASE(16) VCMI_CASE(20) VCMI_CASE(24) VCMI_CASE(28) VCMI_CASE(32) VCMI_CASE(36) VCMI_CASE(40) VCMI_CASE(44) VCMI_CASE(48)
#undef VCMI_CASE
    default: return fail(VCMI_ERR_ARG, "no screening kernel for padded dimension %d", g->DP);
  }
}

// predict on grouped frames with the four-row screen (gmmmap_screen_argmax_kernel): every padded dimension of the tile kernel
template <int DP>
static int launch_screen_argmax(const vcmi_gmmmap *g, const double *dX, int64_t ldx, int64_t T, int64_t *didx, hipStream_t st,
                                const int *perm, const int *gkey) {
  constexpr int WAVES = 4;
  using TL = Tiling<DP, true>;
  constexpr int BUF = (TL::BLK > screen_stage_doubles(DP)) ? TL::BLK : screen_stage_doubles(DP);
  const size_t shmem = 2 * (size_t)BUF * sizeof(double);
  auto kern = gmmmap_screen_argmax_kernel<DP, WAVES>;
  static std::atomic<bool> attr_done[64];
  int dev = 0;
  VCMI_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64 || !attr_done[dev].load(std::memory_order_acquire)) {
    VCMI_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    if (dev >= 0 && dev < 64) attr_done[dev].store(true, std::memory_order_release);
  }
  const int64_t per_wg = (int64_t)32 * WAVES;
  hipLaunchKernelGGL(kern, dim3((unsigned)((T + per_wg - 1) / per_wg)), dim3(WAVES * 64), shmem, st, g->packedU.p, g->packedQA.p, g->M, g->D,
                     dX, ldx, T, didx, perm, gkey);
  VCMI_HIP(hipGetLastError());
  return VCMI_OK;
}
static int dispatch_screen_argmax(const vcmi_gmmmap *g, const double *dX, int64_t ldx, int64_t T, int64_t *didx, hipStream_t st,
                                  const int *perm, const int *gkey) {
  switch (g->DP) {
#define VCMI_CASE(DPV) \
  case DPV: return launch_screen_argmax<DPV>(g, dX, ldx, T, didx, st, perm, gkey);
    VCMI_CASE(16) VCMI_CASE(20) VCMI_CASE(24) VCMI_CASE(28) VCMI_CASE(32) VCMI_CASE(36) VCMI_CASE(40) VCMI_CASE(44)
    VCMI_CASE(48) VCMI_CASE(52) VCMI_CASE(56) VCMI_CASE(60) VCMI_CASE(64) VCMI_CASE(68) VCMI_CASE(72) VCMI_CASE(76)
    VCMI_CASE(80)
#undef VCMI_CASE
    default: return fail(VCMI_ERR_ARG, "no screening kernel for padded dimension %d", g->DP);
  }
}

static constexpr double kScreenArgmaxFrac = 0.10;      // predict: screened arg-max when at most this fraction of the mixtures survives the screen
static constexpr double kBroadModelFrac = 0.35;
// shape 3 pays while the survivors of the four-row screen stay few: a survivor costs a whole mixture (42 MFMA steps at D = 40)
// for every wave of its workgroup, a screened mixture 2.5 -- against the 10 of shape 2's last-tile test
static constexpr double kScreenModelFrac = 0.05;
// Which loop shape converts with this handle (gmmmap_mfma_kernel's PRUNE): 2 "peaked" when, for the model's own frames, the
// last whitening tile's share of |z|^2 alone puts most mixtures e^-prune under the best one (model_undecided_frac, estimated
// once by prepare(): 0.02 on the SURVEY 8d synthetic models) -- the loop that looks at that tile first then skips the other
// whitening tiles; 1 "broad" otherwise (every whitening tile is needed anyway: the straight loop is faster).
static int convert_shape(const vcmi_gmmmap *g) {
  const bool can_screen = screen_has_kernel(g->DP) && g->M <= 1024 && g->packedQ.p != nullptr;
  if (debug_flag(kDbgConvertShapeBroad)) return 1;
  if (debug_flag(kDbgConvertShapePeaked)) return 2;
  if (debug_flag(kDbgConvertShapeScreened) && can_screen) return 3;
  if (can_screen && g->model_undecided4_frac <= kScreenModelFrac) return 3;     // (grouped calls only: see gmmmap_convert_device)
  return g->model_undecided_frac > kBroadModelFrac ? 1 : 2;
}

// The three grouping kernels (keys + chunk histograms, prefix, stable scatter) on g's scratch: *key = group of every frame,
// *perm = frames in group order.  The caller brackets its use of them with g->grp_order.enter / leave.
static size_t group_key_shmem(const vcmi_gmmmap *g, bool fp64) {
  const int MT = (g->M + 15) / 16;
  return (fp64 ? (size_t)MT * (std::min(g->DP / 4, kGroupKeyDims / 4) + 1) * 64 * sizeof(double) : (size_t)MT * kKey16TileBytes) +
         (size_t)g->M * sizeof(int);
}
static bool group_key_fp64(const vcmi_gmmmap *g) { return debug_flag(kDbgGroupKeyFp64) || !g->gfrag16.p; }
static bool can_group(const vcmi_gmmmap *g, int64_t T) {
  return T >= kSortMinFrames && T < ((int64_t)1 << 31) && g->M >= 4 && g->gfrag.p && group_key_shmem(g, group_key_fp64(g)) <= 64 * 1024;
}
static int launch_grouping(vcmi_gmmmap *g, const double *dX, int64_t ldx, int64_t T, hipStream_t st, int **key_out, int **perm_out) {
  const bool fp64 = group_key_fp64(g);
  const size_t gshmem = group_key_shmem(g, fp64);
  const int64_t nchunks = (T + kGroupChunk - 1) / kGroupChunk;
  VCMI_TRY(g->grp.reserve((size_t)2 * T + (size_t)(nchunks + 1) * g->M));
  VCMI_TRY(g->grp_order.enter(st));
  int *key = g->grp.p, *perm = key + T, *chunkhist = perm + T, *total = chunkhist + nchunks * g->M;
  int cus = 256;
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, g->device);
  const unsigned kgrid = (unsigned)std::min<int64_t>(nchunks, (int64_t)cus * 4);
  switch (g->DP) {
#define VCMI_CASE(DPV) \
  case DPV: \
    if (fp64) hipLaunchKernelGGL(gmmmap_group_key_kernel<DPV>, dim3(kgrid), dim3(256), gshmem, st, g->gfrag.p, g->M, g->D, dX, ldx, T, key, chunkhist); \
    else hipLaunchKernelGGL(gmmmap_group_key16_kernel<DPV>, dim3(kgrid), dim3(256), gshmem, st, g->gfrag16.p, g->M, g->D, dX, ldx, T, key, chunkhist); \
    break;
    VCMI_CASE(16) VCMI_CASE(20) VCMI_CASE(24) VCMI_CASE(28) VCMI_CASE(32) VCMI_CASE(36) VCMI_CASE(40) VCMI_CASE(44)
    VCMI_CASE(48) VCMI_CASE(52) VCMI_CASE(56) VCMI_CASE(60) VCMI_CASE(64) VCMI_CASE(68) VCMI_CASE(72) VCMI_CASE(76)
    VCMI_CASE(80)
#undef VCMI_CASE
    default: return fail(VCMI_ERR_ARG, "no MFMA instantiation for padded dimension %d", g->DP);
  }
  hipLaunchKernelGGL(gmmmap_group_scan_kernel, dim3((unsigned)g->M), dim3(256), 0, st, chunkhist, nchunks, g->M, total, (const int64_t *)nullptr);
  hipLaunchKernelGGL(gmmmap_group_scatter_kernel, dim3((unsigned)nchunks), dim3(256), (size_t)17 * g->M * sizeof(int), st,
                     key, T, g->M, chunkhist, total, perm, (const int64_t *)nullptr);
  VCMI_HIP(hipGetLastError());
  *key_out = key;
  *perm_out = perm;
  return VCMI_OK;
}

// convert on device pointers
int gmmmap_convert_device(vcmi_gmmmap *g, const double *dX, int64_t ldx, int64_t T, double *dY, int64_t ldy,
                          hipStream_t st) {
  if (T == 0) return VCMI_OK;
  if (g->kernel_choice == 2 && !gmmmap_has_mfma(g->DP))
    return fail(VCMI_ERR_ARG, "MFMA kernel forced but dimension %d has no instantiation", g->D);
  if (use_mfma(g)) {
    // frames grouped by their nearest source mean first (see gmmmap_group_key_kernel): worth its three small kernels from a
    // few thousand frames on; the prune = +inf (dense) setting has nothing to gain from it
    if (can_group(g, T) && g->prune < 1e300 && !debug_flag(kDbgConvertNoGrouping)) {
      int *key = nullptr, *perm = nullptr;
      VCMI_TRY(launch_grouping(g, dX, ldx, T, st, &key, &perm));
      const int shape = convert_shape(g);
      const int rc = shape == 3 ? dispatch_screen(g, dX, ldx, T, dY, ldy, st, perm, key)
                   : shape == 1 ? dispatch_mfma<0, 1>(g, dX, ldx, T, dY, ldy, st, perm, key)
                                : dispatch_mfma<0, 2>(g, dX, ldx, T, dY, ldy, st, perm, key);
      (void)g->grp_order.leave(st);
      return rc;
    }
    if (!(g->prune < 1e300) && !debug_flag(kDbgConvertShapePeaked)) return dispatch_mfma<0, 0>(g, dX, ldx, T, dY, ldy, st);
    // (frames in the caller's order: the screen of shape 3 needs the tight running maximum of grouped frames -> shape 2)
    return convert_shape(g) == 1 ? dispatch_mfma<0, 1>(g, dX, ldx, T, dY, ldy, st) : dispatch_mfma<0, 2>(g, dX, ldx, T, dY, ldy, st);
  }
  if (g->kernel_choice != 1 && g->At.p && g->D > 16 && g->D <= 160) {
    // no tile-kernel instantiation (80 < padded D <= 160, or a padded dimension outside its list): MFMA log-densities
    // (logdens_tiled_kernel) + softmax / regression over the mixtures that matter, in chunks that bound the (T,M) scratch
    const int64_t chunk = std::max<int64_t>(4096, ((int64_t)1 << 24) / std::max(g->M, 1));
    for (int64_t t0 = 0; t0 < T; t0 += chunk) {
      const int64_t n = std::min(chunk, T - t0);
      VCMI_TRY(g->scratch_lp.reserve((size_t)std::min(chunk, T) * g->M));
      VCMI_TRY(gmmmap_logdens_device(g, dX + t0 * ldx, ldx, n, g->scratch_lp.p, st));
      const unsigned blocks = (unsigned)std::min<int64_t>((n + 3) / 4, 16384);
      hipLaunchKernelGGL(convert_from_logdens_kernel, dim3(blocks), dim3(256), 0, st, g->scratch_lp.p, g->M, g->D, g->DP, g->At.p,
                         g->b.p, dX + t0 * ldx, ldx, n, dY + t0 * ldy, ldy, g->prune);
      VCMI_HIP(hipGetLastError());
    }
    return VCMI_OK;
  }
  return launch_generic<0>(g, dX, ldx, T, dY, ldy, st);
}

// log-weighted densities (M,T) on device pointers
int gmmmap_logdens_device(vcmi_gmmmap *g, const double *dX, int64_t ldx, int64_t T, double *dLP, hipStream_t st) {
  if (T == 0) return VCMI_OK;
  if (use_mfma(g)) return dispatch_mfma<1>(g, dX, ldx, T, dLP, g->M, st);
  if (g->kernel_choice != 1 && g->D > 16 && g->D <= 160) {      // no instantiation of the kernel above (D > 80, or a padded
                                                                // dimension outside its list): tiles streamed through LDS
    constexpr int NTMAX = 10;
    constexpr TiledStages<NTMAX> ST{};
    const size_t shmem = 2 * (size_t)ST.max_piece() * sizeof(double);
    static std::atomic<bool> attr_done[64];
    if (!attr_done[g->device & 63].load(std::memory_order_acquire)) {
      VCMI_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(logdens_tiled_kernel<NTMAX>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
      attr_done[g->device & 63].store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL(logdens_tiled_kernel<NTMAX>, dim3((unsigned)((T + 127) / 128)), dim3(512), shmem, st, g->U.p, g->cz.p,
                       g->lc.p, g->M, g->D, g->DP, dX, ldx, T, dLP, (int64_t)g->M);
    VCMI_HIP(hipGetLastError());
    return VCMI_OK;
  }
  return launch_generic<1>(g, dX, ldx, T, dLP, g->M, st);
}

int gmmmap_posterior_device(vcmi_gmmmap *g, const double *dX, int64_t ldx, int64_t T, double *dP, hipStream_t st) {
  if (T == 0) return VCMI_OK;
  VCMI_TRY(gmmmap_logdens_device(g, dX, ldx, T, dP, st));
  hipLaunchKernelGGL(posterior_finish_kernel<0>, dim3(posterior_finish_grid(g->M, T)), dim3(256), 0, st, dP, g->M, T,
                     (int64_t *)nullptr);
  VCMI_HIP(hipGetLastError());
  return VCMI_OK;
}

int gmmmap_predict_device(vcmi_gmmmap *g, const double *dX, int64_t ldx, int64_t T, int64_t *didx, hipStream_t st, bool allow_screen) {
  if (T == 0) return VCMI_OK;
  if (use_mfma(g) && !debug_flag(kDbgPredictTwoPass)) {    // argmax inside the MFMA kernel: no (M,T) matrix, one launch
    // long inputs: frames grouped by nearest source mean, then the screened arg-max (exact; gmmmap_screen.hpp)
    // -- where the model's own frames leave it few survivors (model_argmax_survivors_frac, estimated by prepare(): 3.7 x on the
    // SURVEY 8d model's frames; on a broad model the bound from four directions never reaches the 40 nats that the full
    // 80-dimensional distance of the best mixture costs, every mixture survives and the early-exit kernel is as fast) and the
    // caller allows it (the trajectory conversion does not: its static + delta vectors are not draws from p(x) -- measured
    // 4.1 against 3.3 ms per 512,000 frames there, tools/predict_screen_ab.py)
    const bool screen_pays = (g->model_argmax_survivors_frac <= kScreenArgmaxFrac || debug_flag(kDbgPredictScreen)) && allow_screen;
    if (screen_pays && g->packedQA.p && g->packedU.p && g->M <= 1024 && can_group(g, T) && !debug_flag(kDbgPredictNoScreen) &&
        !debug_flag(kDbgPredictNoEarlyExit)) {
      int *key = nullptr, *perm = nullptr;
      VCMI_TRY(launch_grouping(g, dX, ldx, T, st, &key, &perm));
      const int rc = dispatch_screen_argmax(g, dX, ldx, T, didx, st, perm, key);
      (void)g->grp_order.leave(st);
      return rc;
    }
    // host-prepared handles carry the reversed, tile-by-tile fragments: predict with the exact early exit (MODE 3)
    if (g->packedU2.p && !debug_flag(kDbgPredictNoEarlyExit))
      return dispatch_mfma<3>(g, dX, ldx, T, reinterpret_cast<double *>(didx), 0, st);
    return dispatch_mfma<2>(g, dX, ldx, T, reinterpret_cast<double *>(didx), 0, st);
  }
  VCMI_TRY(g->scratch_lp.reserve((size_t)T * g->M));
  VCMI_TRY(gmmmap_logdens_device(g, dX, ldx, T, g->scratch_lp.p, st));
  hipLaunchKernelGGL(posterior_finish_kernel<1>, dim3(posterior_finish_grid(g->M, T)), dim3(256), 0, st, g->scratch_lp.p,
                     g->M, T, didx);
  VCMI_HIP(hipGetLastError());
  return VCMI_OK;
}

// ------------------------------------------------------------------------------------------------
// model preparation (host) and upload
// ------------------------------------------------------------------------------------------------
// How broad is the model?  256 frames are drawn from p(x) itself (stratified over the weights, fixed-seed normal deviates:
// x = mu_m + L_m z, i.e. U_m (x - mu_m) = z solved by forward substitution) and for each the mixtures within e^-46 of the best
// one are counted.  Returns the mean fraction of the M mixtures.  A property of the model only; it selects the loop SHAPE of
// fvconvert (convert_shape), never a result.
static double model_active_fraction(const std::vector<double> &hU, const std::vector<double> &hcz, const std::vector<double> &hlc,
                                    const std::vector<double> &hP, const std::vector<double> &hcP,
                                    const std::vector<double> &hmux, const double *w, int D, int DP, int M,
                                    double *undecided_frac, double *undecided_rows, double *argmax_survivors) {
  constexpr int S = 256;
  const size_t pp = (size_t)DP * DP;
  std::vector<double> cdf(M);
  double tot = 0.0;
  for (int m = 0; m < M; ++m) cdf[m] = (tot += (w[m] > 0.0 ? w[m] : 0.0));
  if (undecided_frac) *undecided_frac = 0.0;
  if (undecided_rows) undecided_rows[0] = undecided_rows[1] = undecided_rows[2] = 1.0;   // on the 4 / 2 / 1 strongest screening rows
  if (!(tot > 0.0) || M < 2) return 0.0;
  std::vector<int> counts(S, 0), undecided(S, 0), undecided4(3 * S, 0), asurv(S, 0);
  if (argmax_survivors) *argmax_survivors = 1.0;
  const int r_last = 16 * ((DP + 15) / 16 - 1);                       // first row of the last whitening tile
  host_parallel_for(S, 8, [&](int64_t lo, int64_t hi) {
    std::vector<double> x(D), z(D);
    for (int s = (int)lo; s < (int)hi; ++s) {
      const double u = (s + 0.5) / S * tot;
      int m = 0;
      while (m + 1 < M && cdf[m] < u) ++m;
      uint64_t st = 0x9E3779B97F4A7C15ull * (uint64_t)(s + 1);          // splitmix64 stream per frame
      auto rnd = [&]() {
        st += 0x9E3779B97F4A7C15ull;
        uint64_t v = st;
        v = (v ^ (v >> 30)) * 0xBF58476D1CE4E5B9ull;
        v = (v ^ (v >> 27)) * 0x94D049BB133111EBull;
        v ^= v >> 31;
        return ((double)(v >> 11) + 0.5) * (1.0 / 9007199254740992.0);
      };
      for (int d = 0; d < D; d += 2) {                                  // Box-Muller
        const double r = std::sqrt(-2.0 * std::log(rnd())), a = 6.283185307179586 * rnd();
        z[d] = r * std::cos(a);
        if (d + 1 < D) z[d + 1] = r * std::sin(a);
      }
      const double *Um = &hU[pp * m];
      for (int r = 0; r < D; ++r) {                                     // U_m (x - mu_m) = z, U_m lower triangular
        double acc = z[r];
        for (int c = 0; c < r; ++c) acc -= Um[(size_t)r * DP + c] * x[c];
        x[r] = acc / Um[(size_t)r * DP + r];
      }
      for (int d = 0; d < D; ++d) x[d] += hmux[(size_t)D * m + d];
      double best = -INFINITY;
      std::vector<double> l(M), qlast(M), qlast4(3 * (size_t)M);
      for (int n = 0; n < M; ++n) {
        const double *Un = &hU[pp * n];
        double q = 0.0, ql = 0.0, ql4[3] = {0.0, 0.0, 0.0};
        for (int r = 0; r < D; ++r) {
          double zz = -hcz[(size_t)DP * n + r];
          for (int c = 0; c <= r; ++c) zz += Un[(size_t)r * DP + c] * x[c];
          q += zz * zz;
          if (r >= r_last) ql += zz * zz;
        }
        if (!hP.empty())
          for (int i = 0; i < 4; ++i) {                                  // the screen's rows: strongest first
            double pz = -hcP[(size_t)n * 4 + i];
            for (int c = 0; c < D; ++c) pz += hP[((size_t)n * 4 + i) * DP + c] * x[c];
            for (int c = 0; c < 3; ++c)
              if (i < (4 >> c)) ql4[c] += pz * pz;
          }
        l[n] = hlc[n] - 0.5 * q;
        qlast[n] = ql;
        for (int c = 0; c < 3; ++c) qlast4[3 * (size_t)n + c] = ql4[c];
        best = std::max(best, l[n]);
      }
      int cnt = 0;
      for (int n = 0; n < M; ++n) cnt += (l[n] > best - 46.0);
      counts[s] = cnt;
      int und = 0;                                                     // ... and on the last 16-row whitening tile's share alone
      for (int n = 0; n < M; ++n) und += (hlc[n] - 0.5 * qlast[n] > best - 46.0);
      undecided[s] = und;
      {                                                                // predict's screen: bound from four rows >= the best log-density
        int sv = 0;
        for (int n = 0; n < M; ++n) sv += (hlc[n] - 0.5 * qlast4[3 * (size_t)n] >= best);
        asurv[s] = sv;
      }
      for (int c = 0; c < 3; ++c) {                                    // ... and on the 4 / 2 / 1 strongest screening rows alone (shape 3)
        int und4 = 0;
        for (int n = 0; n < M; ++n) und4 += (hlc[n] - 0.5 * qlast4[3 * (size_t)n + c] > best - 46.0);
        undecided4[3 * s + c] = und4;
      }
    }
  });
  double sum = 0.0, sumu = 0.0, sumu4[3] = {0.0, 0.0, 0.0};
  for (int s = 0; s < S; ++s) {
    sum += counts[s];
    sumu += undecided[s];
    for (int c = 0; c < 3; ++c) sumu4[c] += undecided4[3 * s + c];
  }
  if (undecided_frac) *undecided_frac = sumu / ((double)S * M);
  if (undecided_rows)
    for (int c = 0; c < 3; ++c) undecided_rows[c] = sumu4[c] / ((double)S * M);
  if (argmax_survivors && !hP.empty()) {
    double sa = 0.0;
    for (int s = 0; s < S; ++s) sa += asurv[s];
    *argmax_survivors = sa / ((double)S * M);
  }
  return sum / ((double)S * M);
}

// px_only: (mu, sigma) describe a plain GMM p(x) of dimension Dj (no target half): only the whitening side is prepared
// (used by the full-covariance E-step, estep.hip); the regression blocks stay zero and the convert layouts are skipped.
static int prepare(vcmi_gmmmap *g, const double *w, const double *mu, const double *sigma, int Dj, int M, int swap,
                   bool px_only = false) {
  const int D = px_only ? Dj : Dj >> 1;   // src/gmmmap.jl:70
  const int DP = (D + 3) / 4 * 4;
  g->D = D;
  g->M = M;
  g->DP = DP;
  const size_t dd = (size_t)D * D, pp = (size_t)DP * DP;
  const size_t reg = px_only ? 0 : 1;   // the regression side (A, b, Sxy, Syy) does not exist for a p(x)-only handle
  g->h_A_julia.assign(reg * dd * M, 0.0);
  g->h_Sxy.assign(reg * dd * M, 0.0);
  g->h_Syy.assign(reg * dd * M, 0.0);
  g->h_A.assign(reg * dd * M, 0.0);
  g->h_mux.assign((size_t)D * M, 0.0);
  g->h_muy.assign((size_t)D * M, 0.0);
  std::vector<double> hU(pp * M, 0.0), hA(reg * pp * M, 0.0), hcz((size_t)DP * M, 0.0), hb(reg * DP * M, 0.0), hlc(M);
  const bool want_screen = !px_only && D >= 4 && gmmmap_has_mfma(DP) && M <= 1024;       // (fvconvert's screen: DP <= 48; predict's: every tile-kernel dimension)
  std::vector<double> hP(want_screen ? (size_t)M * 4 * DP : 0, 0.0), hcP(want_screen ? (size_t)M * 4 : 0, 0.0);
  const int xo = (swap && !px_only) ? D : 0, yo = px_only ? 0 : (swap ? 0 : D);   // src/gmmmap.jl:74-78
  const double LOG2PI = 1.8378770664093454835606594728112;
  // The mixtures are independent (an inverse, a Cholesky factorisation and a triangular inverse each: 64 x 160^3 flop for
  // the joint model of delta features): they are shared out over the library's host threads.  The first failing mixture
  // (lowest index) is reported, as the sequential loop would.
  std::atomic<int> bad_singular{M}, bad_notpd{M};
  auto lower_to = [](std::atomic<int> &a, int v) {
    int cur = a.load();
    while (v < cur && !a.compare_exchange_weak(cur, v)) {}
  };
  host_parallel_for(M, 1, [&](int64_t m_lo, int64_t m_hi) {
  std::vector<double> Sxx(dd), Syx(dd), inv(dd), L(dd), Ui(dd);
  for (int m = (int)m_lo; m < (int)m_hi; ++m) {
    const double *S = sigma + (size_t)Dj * Dj * m;   // column-major (Dj,Dj)
    double *mux = &g->h_mux[(size_t)D * m], *muy = &g->h_muy[(size_t)D * m];
    for (int d = 0; d < D; ++d) {
      mux[d] = mu[xo + d + (size_t)Dj * m];
      muy[d] = px_only ? 0.0 : mu[yo + d + (size_t)Dj * m];
    }
    // row-major copies of the four blocks, src/gmmmap.jl:41-52
    for (int r = 0; r < D; ++r)
      for (int c = 0; c < D; ++c) {
        Sxx[(size_t)r * D + c] = S[(xo + r) + (size_t)Dj * (xo + c)];
        if (px_only) continue;
        Syx[(size_t)r * D + c] = S[(yo + r) + (size_t)Dj * (xo + c)];
        g->h_Sxy[dd * m + (size_t)r * D + c] = S[(xo + r) + (size_t)Dj * (yo + c)];
        g->h_Syy[dd * m + (size_t)r * D + c] = S[(yo + r) + (size_t)Dj * (yo + c)];
      }
    // A_m = Syx inv(Sxx) on the raw block, src/gmmmap.jl:35
    double *Am = px_only ? nullptr : &g->h_A[dd * m];
    if (!px_only) {
      if (!la::inverse(Sxx.data(), D, inv.data())) {
        lower_to(bad_singular, m);
        continue;
      }
      la::matmul(Syx.data(), inv.data(), D, Am);
      for (int r = 0; r < D; ++r)
        for (int c = 0; c < D; ++c) {
          g->h_A_julia[dd * m + r + (size_t)D * c] = Am[(size_t)r * D + c];
          hA[pp * m + (size_t)r * DP + c] = Am[(size_t)r * D + c];
        }
    }
    // p(x): Hermitian(Sxx) (upper triangle mirrored, src/gmm.jl:16) -> Cholesky -> U = inv(L)
    if (!la::cholesky_from_upper(Sxx.data(), D, L.data())) {
      lower_to(bad_notpd, m);
      continue;
    }
    la::lower_inverse(L.data(), D, Ui.data());
    double logdiag = 0.0;
    for (int d = 0; d < D; ++d) logdiag += std::log(L[(size_t)d * D + d]);
    hlc[m] = (w[m] > 0.0) ? std::log(w[m]) - 0.5 * (D * LOG2PI + 2.0 * logdiag)
                          : -std::numeric_limits<double>::infinity();   // zero-weight: posterior 0
    for (int r = 0; r < D; ++r) {
      double cz = 0.0, ba = 0.0;
      for (int c = 0; c < D; ++c) {
        hU[pp * m + (size_t)r * DP + c] = Ui[(size_t)r * D + c];
        cz += Ui[(size_t)r * D + c] * mux[c];
        if (!px_only) ba += Am[(size_t)r * D + c] * mux[c];
      }
      hcz[(size_t)DP * m + r] = cz;
      if (!px_only) hb[(size_t)DP * m + r] = muy[r] - ba;
    }
    // screening rows of shape 3 (gmmmap_screen.hpp): P_m = diag(sqrt(kappa_i)) v_i' over the FOUR LARGEST eigenpairs of
    // inv(Sxx_m) = U'U -- the directions of smallest variance.  |P_m (x - mu_m)|^2 is a partial sum of the eigen-expansion of
    // (x - mu_m)' inv(Sxx_m) (x - mu_m) = |z_m|^2: a lower bound of it, and per row the largest one any direction can give.
    if (want_screen) {
      std::vector<double> G(dd), V(dd);
      for (int r = 0; r < D; ++r)
        for (int c = 0; c < D; ++c) {
          double sacc = 0.0;
          for (int k = std::max(r, c); k < D; ++k) sacc += Ui[(size_t)k * D + r] * Ui[(size_t)k * D + c];
          G[(size_t)r * D + c] = sacc;
        }
      la::sym_eigen_jacobi(G.data(), D, V.data());
      int top[4] = {-1, -1, -1, -1};
      for (int i = 0; i < 4; ++i) {
        double bestv = -1.0;
        for (int j = 0; j < D; ++j) {
          bool used = false;
          for (int u = 0; u < i; ++u) used = used || top[u] == j;
          if (!used && G[(size_t)j * D + j] > bestv) {
            bestv = G[(size_t)j * D + j];
            top[i] = j;
          }
        }
        const double sk = std::sqrt(std::max(bestv, 0.0));
        double cp = 0.0;
        for (int k = 0; k < D; ++k) {
          const double v = sk * V[(size_t)k * D + top[i]];
          hP[((size_t)m * 4 + i) * DP + k] = v;
          cp += v * mux[k];
        }
        hcP[(size_t)m * 4 + i] = cp;
      }
    }
  }
  });
  {
    const int bs = bad_singular.load(), bp = bad_notpd.load();
    if (bs < M && bs <= bp) return fail(VCMI_ERR_NOT_PD, "Sigma^xx of mixture %d is singular", bs + 1);
    if (bp < M) return fail(VCMI_ERR_NOT_PD, "Sigma^xx of mixture %d is not positive definite", bp + 1);
  }
  if (!px_only) g->model_active_frac = model_active_fraction(hU, hcz, hlc, hP, hcP, g->h_mux, w, D, DP, M, &g->model_undecided_frac, g->model_undecided_rows, &g->model_argmax_survivors_frac);
  // row-major blocks for the generic kernels; a p(x)-only handle that takes the MFMA path needs only its packed blocks
  // (device buffers are grow-only so that a handle re-prepared every EM iteration does not re-allocate)
  if (!(px_only && gmmmap_has_mfma(DP))) {
    VCMI_TRY(g->U.reserve(hU.size()));
    VCMI_TRY(g->cz.reserve(hcz.size()));
    VCMI_TRY(g->lc.reserve(hlc.size()));
    VCMI_TRY(upload_now(g->U.p, hU.data(), hU.size() * 8));
    VCMI_TRY(upload_now(g->cz.p, hcz.data(), hcz.size() * 8));
    VCMI_TRY(upload_now(g->lc.p, hlc.data(), hlc.size() * 8));
  }
  if (!px_only) {
    VCMI_TRY(g->A.reserve(hA.size()));
    VCMI_TRY(g->b.reserve(hb.size()));
    VCMI_TRY(upload_now(g->A.p, hA.data(), hA.size() * 8));
    VCMI_TRY(upload_now(g->b.p, hb.data(), hb.size() * 8));
    if (!gmmmap_has_mfma(DP) && D > 16 && D <= 160) {      // convert_from_logdens_kernel reads A transposed
      std::vector<double> hAt(hA.size());
      for (int m = 0; m < M; ++m)
        for (int r = 0; r < DP; ++r)
          for (int k = 0; k < DP; ++k) hAt[pp * m + (size_t)k * DP + r] = hA[pp * m + (size_t)r * DP + k];
      VCMI_TRY(g->At.reserve(hAt.size()));
      VCMI_TRY(upload_now(g->At.p, hAt.data(), hAt.size() * 8));
    }
  }

  // packed operand blocks for the MFMA kernel (issue order: phase U k-major over U tiles, then phase A)
  // variant 0: [U ; A] (convert), 1: U only (log-densities, predict), 2: U only, tile by tile with the LAST tile first
  // (predict with early exit, MODE 3)
  for (int uonly = px_only ? 1 : 0; uonly < 3 && gmmmap_has_mfma(DP); ++uonly) {
    TilingRT tl(DP, uonly != 0);
    std::vector<double> pk((size_t)tl.BLK * M, 0.0);
    auto wrow = [&](int m, int p, int k) -> double {   // row p of [U_m ; A_m], column k
      if (k >= DP) return 0.0;
      if (p < DP) return hU[pp * m + (size_t)p * DP + k];
      if (p < 2 * DP && !uonly) return hA[pp * m + (size_t)(p - DP) * DP + k];
      return 0.0;
    };
    for (int m = 0; m < M; ++m) {
      double *blk = &pk[(size_t)tl.BLK * m];
      int s = 0;
      if (uonly == 2) {
        for (int t = std::min(tl.NU, tl.NT) - 1; t >= 0; --t)
          for (int ks = 0; ks < tl.steps(t); ++ks) {
            for (int l = 0; l < 64; ++l) blk[(size_t)s * 64 + l] = wrow(m, 16 * t + (l & 15), 4 * ks + (l >> 4));
            ++s;
          }
      } else {
        for (int phase = 0; phase < 2; ++phase) {
          const int t0 = phase == 0 ? 0 : tl.NU, t1 = phase == 0 ? std::min(tl.NU, tl.NT) : tl.NT;
          for (int ks = 0; ks < tl.KS; ++ks)
            for (int t = t0; t < t1; ++t) {
              if (ks >= tl.steps(t)) continue;
              for (int l = 0; l < 64; ++l) blk[(size_t)s * 64 + l] = wrow(m, 16 * t + (l & 15), 4 * ks + (l >> 4));
              ++s;
            }
        }
      }
      for (int p = 0; p < tl.NT * 16; ++p) {
        double c = 0.0;
        if (p < DP) c = -hcz[(size_t)DP * m + p];
        else if (p < 2 * DP && !uonly) c = hb[(size_t)DP * m + (p - DP)];
        blk[tl.CINIT_OFF + p] = c;
      }
      blk[tl.LC_OFF] = hlc[m];
    }
    DevBuf<double> &dst = uonly == 2 ? g->packedU2 : (uonly ? g->packedU : g->packed);
    VCMI_TRY(dst.reserve(pk.size()));
    VCMI_TRY(upload_now(dst.p, pk.data(), pk.size() * 8));
  }
  // stages of the screen of shape 3 (gmmmap_screen.hpp) on the first rpm rows of every mixture's P_m: rpm = the row count
  // with the smallest estimated cost per 16-frame tile -- KS MFMAs screen 16 / rpm mixtures; a mixture the screen does not
  // rule out costs its whole whitening (and usually its regression) for the four waves that share it
  if (want_screen && screen_has_kernel(DP)) {
    const int KSQ = DP / 4, QFR = screen_frag_doubles(DP), STG = screen_stage_doubles(DP), NQ = screen_quads(DP);
    int best = 4;
    double best_cost = 1e300;
    for (int c = 0; c < 3; ++c) {
      const int rpm = 4 >> c;
      const double extra = std::max(0.0, g->model_undecided_rows[c] - 1.0 / M);      // wrong mixtures let through, per frame
      const double cost = (double)KSQ * M * rpm / 16.0 + 4.0 * extra * M * (2 * KSQ + 2);
      if (cost < best_cost) {
        best_cost = cost;
        best = rpm;
      }
    }
    if (debug_flag(kDbgScreenRows4)) best = 4;        // (test hooks, read when the converter is CREATED)
    if (debug_flag(kDbgScreenRows2)) best = 2;
    if (debug_flag(kDbgScreenRows1)) best = 1;
    g->screen_rpm = best;
    g->model_undecided4_frac = g->model_undecided_rows[best == 4 ? 0 : best == 2 ? 1 : 2];     // what the chosen screen lets through
    const int rpm = best, mpt = 16 / rpm, nst = (M + mpt * NQ - 1) / (mpt * NQ);
    std::vector<double> pq((size_t)nst * STG, 0.0);
    for (int st = 0; st < nst; ++st)
      for (int q = 0; q < NQ; ++q) {
        const int m0 = (NQ * st + q) * mpt;
        double *fr = &pq[(size_t)st * STG + (size_t)q * KSQ * 64], *cl = &pq[(size_t)st * STG + QFR + (size_t)q * 32];
        for (int ks = 0; ks < KSQ; ++ks)
          for (int l = 0; l < 64; ++l) {
            const int i = l & 15, k = 4 * ks + (l >> 4), m = m0 + screen_row_mixture(i, rpm), row = screen_row_index(i, rpm);
            fr[(size_t)ks * 64 + l] = (m < M && k < DP) ? hP[((size_t)m * 4 + row) * DP + k] : 0.0;
          }
        for (int j = 0; j < 4; ++j) {
          for (int r = 0; r < 4; ++r) {                        // register r of lane group j holds tile row 4 r + j
            const int i = 4 * r + j, m = m0 + screen_row_mixture(i, rpm), row = screen_row_index(i, rpm);
            cl[j * 8 + r] = (m < M) ? -hcP[(size_t)m * 4 + row] : 0.0;
          }
          for (int u = 0; u < 4; ++u) {                        // sub-mixture u of lane group j (u < 4 / rpm)
            const int m = m0 + (4 / rpm) * j + u;
            cl[j * 8 + 4 + u] = (u < 4 / rpm && m < M) ? hlc[m] : -std::numeric_limits<double>::infinity();
          }
        }
      }
    VCMI_TRY(g->packedQ.reserve(pq.size()));
    VCMI_TRY(upload_now(g->packedQ.p, pq.data(), pq.size() * 8));
    // the same four rows split into bf16 hi + lo for the screen on the BF16 matrix pipe (gmmmap_screen.hpp, B16)
    if (rpm == 4 && screen16_has(DP)) {
      const int KS8 = std::min(KSQ, 8), NTL = KSQ - KS8, STG16 = screen16_stage_doubles(DP), nst16 = (M + 4 * NQ - 1) / (4 * NQ);
      std::vector<double> p16((size_t)nst16 * STG16, 0.0);
      const double kEps = 1.0 / 4096.0;                            // 2^-12: see the error bound in gmmmap_screen.hpp
      for (int st = 0; st < nst16; ++st)
        for (int q = 0; q < NQ; ++q) {
          const int m0 = (NQ * st + q) * 4;
          unsigned short *fr = reinterpret_cast<unsigned short *>(&p16[(size_t)st * STG16 + (size_t)q * screen16_tile_doubles()]);
          double *cl = &p16[(size_t)st * STG16 + (size_t)NQ * screen16_tile_doubles() + (size_t)q * 32];
          for (int l = 0; l < 64; ++l) {
            const int r = l & 15, gq = l >> 4, m = m0 + (r >> 2), row = r & 3;     // tile row r <-> mixture r >> 2, screening row r & 3
            auto pv = [&](int ks) -> double {                      // P[row][feature 4 ks + lane group]
              const int k = 4 * ks + gq;
              return (m < M && ks < KSQ && k < DP) ? hP[((size_t)m * 4 + row) * DP + k] : 0.0;
            };
            unsigned short ph[10] = {0}, pl[10] = {0};
            for (int ks = 0; ks < KSQ; ++ks) split_bf16(pv(ks), ph[ks], pl[ks]);
            for (int j = 0; j < 8; ++j) {
              fr[(size_t)l * 8 + j] = j < KS8 ? ph[j] : 0;                       // Ph, k-steps 0..7
              fr[512 + (size_t)l * 8 + j] = j < KS8 ? pl[j] : 0;                 // Pl, k-steps 0..7
            }
            const unsigned short t0h = NTL > 0 ? ph[KS8] : 0, t1h = NTL > 1 ? ph[KS8 + 1] : 0;
            const unsigned short t0l = NTL > 0 ? pl[KS8] : 0, t1l = NTL > 1 ? pl[KS8 + 1] : 0;
            const unsigned short tail[8] = {t0h, t1h, t0h, t1h, t0l, t1l, 0, 0};  // against {xh8, xh9, xl8, xl9, xh8, xh9, 0, 0}
            for (int j = 0; j < 8; ++j) fr[1024 + (size_t)l * 8 + j] = tail[j];
          }
          for (int j = 0; j < 4; ++j) {
            const int m = m0 + j;
            for (int r = 0; r < 4; ++r) {
              double nrm = 0.0;
              if (m < M)
                for (int k = 0; k < D; ++k) nrm += hP[((size_t)m * 4 + r) * DP + k] * hP[((size_t)m * 4 + r) * DP + k];
              const double c = (m < M) ? hcP[(size_t)m * 4 + r] : 0.0;
              float *cf = reinterpret_cast<float *>(cl + j * 8);          // {c (4 floats) | 2^-12 |P| (4) | 2^-12 |c| (4)}, margins rounded UP
              auto up = [](double v) { return std::nextafterf((float)(v * (1.0 + 0x1p-20)), INFINITY); };
              cf[r] = (float)c;                                           // (its FP32 rounding is inside the 2^-12 |c| margin)
              cf[4 + r] = up(kEps * std::sqrt(nrm));
              cf[8 + r] = up(kEps * std::fabs(c));
            }
            cl[j * 8 + 6] = (m < M) ? hlc[m] : -std::numeric_limits<double>::infinity();
          }
        }
      VCMI_TRY(g->packedQ16.reserve(p16.size()));
      VCMI_TRY(upload_now(g->packedQ16.p, p16.data(), p16.size() * 8));
    }
  }
  // ... and of predict's screen (gmmmap_screen_argmax_kernel): always four rows per mixture, every tile-kernel dimension
  if (want_screen) {
    const int KSQ = DP / 4, QFR = screen_frag_doubles(DP), STG = screen_stage_doubles(DP), NQ = screen_quads(DP);
    const int nst = (M + 4 * NQ - 1) / (4 * NQ);
    std::vector<double> pq((size_t)nst * STG, 0.0);
    for (int st = 0; st < nst; ++st)
      for (int q = 0; q < NQ; ++q) {
        const int m0 = (NQ * st + q) * 4;
        double *fr = &pq[(size_t)st * STG + (size_t)q * KSQ * 64], *cl = &pq[(size_t)st * STG + QFR + (size_t)q * 32];
        for (int ks = 0; ks < KSQ; ++ks)
          for (int l = 0; l < 64; ++l) {
            const int i = l & 15, k = 4 * ks + (l >> 4), m = m0 + (i & 3), row = i >> 2;
            fr[(size_t)ks * 64 + l] = (m < M && k < DP) ? hP[((size_t)m * 4 + row) * DP + k] : 0.0;
          }
        for (int j = 0; j < 4; ++j) {
          const int m = m0 + j;
          for (int r = 0; r < 4; ++r) cl[j * 8 + r] = (m < M) ? -hcP[(size_t)m * 4 + r] : 0.0;
          for (int u = 0; u < 4; ++u) cl[j * 8 + 4 + u] = (u == 0 && m < M) ? hlc[m] : -std::numeric_limits<double>::infinity();
        }
      }
    VCMI_TRY(g->packedQA.reserve(pq.size()));
    VCMI_TRY(upload_now(g->packedQA.p, pq.data(), pq.size() * 8));
  }
  if (!g->h_mux.empty()) {     // operand of the frame grouping (gmmmap_group_key_kernel): [-2 mu^x | |mu^x|^2] over its first dimensions, fragment order
    const int KSK = std::min(DP / 4, kGroupKeyDims / 4), KS1 = KSK + 1, MT = (M + 15) / 16, DK = std::min(D, 4 * KSK);
    std::vector<double> gf((size_t)MT * KS1 * 64, 0.0);
    for (int mt = 0; mt < MT; ++mt)
      for (int ks = 0; ks < KS1; ++ks)
        for (int l = 0; l < 64; ++l) {
          const int m = 16 * mt + (l & 15), k = 4 * ks + (l >> 4);
          double v = 0.0;
          if (ks < KS1 - 1) {
            if (m < M && k < DK) v = -2.0 * g->h_mux[(size_t)D * m + k];
          } else if ((l >> 4) == 0) {
            v = 1e300;
            if (m < M) {
              v = 0.0;
              for (int d = 0; d < DK; ++d) v += g->h_mux[(size_t)D * m + d] * g->h_mux[(size_t)D * m + d];
            }
          }
          gf[((size_t)mt * KS1 + ks) * 64 + l] = v;
        }
    VCMI_TRY(g->gfrag.reserve(gf.size()));
    VCMI_TRY(upload_now(g->gfrag.p, gf.data(), gf.size() * 8));
    // ... and for the BF16 matrix pipe (gmmmap_group_key16_kernel): -2 mu split into bf16 hi + lo, |mu|^2 as floats
    {
      std::vector<double> g16((size_t)MT * (kKey16TileBytes / 8), 0.0);
      for (int mt = 0; mt < MT; ++mt) {
        unsigned short *hi = reinterpret_cast<unsigned short *>(&g16[(size_t)mt * (kKey16TileBytes / 8)]), *lo = hi + 512;
        float *msq = reinterpret_cast<float *>(hi + 1024);
        for (int l = 0; l < 64; ++l) {
          const int m = 16 * mt + (l & 15), gq = l >> 4;
          for (int j = 0; j < 8; ++j) {
            const int k = 4 * j + gq;
            const double v = (m < M && j < KSK && k < DK) ? -2.0 * g->h_mux[(size_t)D * m + k] : 0.0;
            split_bf16(v, hi[(size_t)l * 8 + j], lo[(size_t)l * 8 + j]);
          }
        }
        for (int r = 0; r < 16; ++r) {
          const int m = 16 * mt + r;
          double v = 1e30;
          if (m < M) {
            v = 0.0;
            for (int d = 0; d < DK; ++d) v += g->h_mux[(size_t)D * m + d] * g->h_mux[(size_t)D * m + d];
          }
          msq[r] = (float)v;                                  // lane group r >> 2 reads floats 4 (r >> 2) .. + 3
        }
      }
      VCMI_TRY(g->gfrag16.reserve(g16.size()));
      VCMI_TRY(upload_now(g->gfrag16.p, g16.data(), g16.size() * 8));
    }
  }
  return VCMI_OK;
}

// p(x)-only handle over a plain GMM of dimension D (weights (M), mu (D,M), sigma (D,D,M)); used by estep.hip.
// *inout == nullptr creates a handle; otherwise the existing handle (same device) is re-prepared in place, reusing its
// device buffers -- the caller must have drained every stream that still reads them.
int gmm_px_create(const double *w, const double *mu, const double *sigma, int D, int M, vcmi_gmmmap **inout) {
  VCMI_TRY(check_device());
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (*inout && (*inout)->device != dev) {
    delete *inout;
    *inout = nullptr;
  }
  vcmi_gmmmap *g = *inout ? *inout : new (std::nothrow) vcmi_gmmmap();
  if (!g) return fail(VCMI_ERR_OOM, "out of host memory");
  g->device = dev;
  int rc = prepare(g, w, mu, sigma, D, M, 0, /*px_only=*/true);
  if (rc != VCMI_OK) {
    delete g;
    *inout = nullptr;
    return rc;
  }
  *inout = g;
  return VCMI_OK;
}

// ------------------------------------------------------------------------------------------------
// p(x) preparation on the device (used by the EM loop: the parameters never leave HBM between iterations).
// One workgroup per mixture: Hermitian(Sigma) (upper triangle mirrored, src/gmm.jl:16) -> right-looking Cholesky in
// LDS -> U = L^-1 by forward substitution (thread = column) -> cz = U mu, lc = log w - (D log 2pi + logdet)/2 ->
// row-major U/cz/lc for the generic kernels and the U-only MFMA operand blocks in issue order.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
px_prep_kernel(const double *__restrict__ w, const double *__restrict__ mu, const double *__restrict__ sigma, int D, int DP,
               const int *__restrict__ table, int nsteps, int cinit_off, int ncinit, int lc_off, int blk_len,
               double *__restrict__ U, double *__restrict__ cz, double *__restrict__ lc, double *__restrict__ packedU,
               int *__restrict__ flag) {
  extern __shared__ double smem_px[];
  const int LD = D + 1, tid = threadIdx.x, m = blockIdx.x;
  double *S = smem_px;                    // [D][LD] covariance, then its lower Cholesky factor
  double *V = S + (size_t)D * LD;         // [D][LD] U = L^-1 (lower)
  __shared__ double czs[256];
  __shared__ double lcs;
  __shared__ int bad;
  const double *Sg = sigma + (size_t)m * D * D;
  for (int e = tid; e < D * D; e += 256) {
    const int r = e / D, c = e - r * D;
    S[r * LD + c] = (r <= c) ? Sg[r + (size_t)D * c] : Sg[c + (size_t)D * r];
    V[r * LD + c] = (r == c) ? 1.0 : 0.0;
  }
  if (tid == 0) bad = 0;
  __syncthreads();
  // Cholesky and U = L^-1 in ONE loop over the columns: the row operations that eliminate column j of L (row j scaled by
  // 1 / l_jj, row i -= l_ij row j) are applied to the identity beside it -- row j of V is final when its step comes, and the
  // updates of a step are independent of each other (threads as a 16 x 16 grid over (row, column)).  Round 5: the inverse used
  // to be a forward substitution AFTER the loop, one thread per column walking 3200 dependent multiply-adds through LDS:
  // two thirds of this kernel's 181 us.
  for (int j = 0; j < D; ++j) {
    const double d = S[j * LD + j];
    if (!(d > 0.0) && tid == 0) bad = 1;
    const double sd = sqrt(d), isd = 1.0 / sd;
    __syncthreads();                      // every thread has read the pivot
    for (int i = j + tid; i < D; i += 256) S[i * LD + j] = (i == j) ? sd : S[i * LD + j] / sd;
    for (int c = tid; c <= j; c += 256) V[j * LD + c] *= isd;
    __syncthreads();
    for (int i = j + 1 + (tid >> 4); i < D; i += 16) {
      const double lij = S[i * LD + j];
      // trailing update of the lower triangle: no division per element
      for (int k = j + 1 + (tid & 15); k <= i; k += 16) S[i * LD + k] = fma(-lij, S[k * LD + j], S[i * LD + k]);
      // the same row operation on the identity's rows: columns 0 .. j of row i
      for (int c = (tid & 15); c <= j; c += 16) V[i * LD + c] = fma(-lij, V[j * LD + c], V[i * LD + c]);
    }
    // (all operands first, then the products, then the stores -- 7 x 7 guarded elements per thread, unrolled -- was tried
    // against the dependent read-modify-writes of these loops: 435 us instead of 145, most of its slots are empty at D = 80)
    __syncthreads();
  }
  if (tid < D) {
    double s = 0.0;
    for (int c = 0; c <= tid; ++c) s = fma(V[tid * LD + c], mu[c + (size_t)D * m], s);
    czs[tid] = s;
  } else if (tid == 255) {
    double ld = 0.0;
    for (int d = 0; d < D; ++d) ld += log(S[d * LD + d]);
    const double LOG2PI = 1.8378770664093454835606594728112;
    lcs = (w[m] > 0.0) ? log(w[m]) - 0.5 * (D * LOG2PI + 2.0 * ld) : -INFINITY;   // zero weight: posterior 0
  }
  __syncthreads();
  for (int e = tid; e < DP * DP; e += 256) {
    const int r = e / DP, c = e - r * DP;
    U[(size_t)m * DP * DP + e] = (r < D && c <= r) ? V[r * LD + c] : 0.0;
  }
  for (int e = tid; e < DP; e += 256) cz[(size_t)m * DP + e] = (e < D) ? czs[e] : 0.0;
  if (tid == 0) {
    lc[m] = lcs;
    if (bad) atomicMax(flag, m + 1);
  }
  if (packedU) {
    double *blk = packedU + (size_t)m * blk_len;
    for (int e = tid; e < blk_len; e += 256) {
      double v = 0.0;
      const int s = e >> 6, l = e & 63;
      if (s < nsteps) {
        const int t = table[s] >> 16, ks = table[s] & 0xffff;
        const int p = 16 * t + (l & 15), k = 4 * ks + (l >> 4);
        if (p < D && k <= p) v = V[p * LD + k];
      } else if (e >= cinit_off && e < cinit_off + ncinit) {
        const int p = e - cinit_off;
        if (p < D) v = -czs[p];
      } else if (e == lc_off) {
        v = lcs;
      }
      blk[e] = v;
    }
  }
}

// The same preparation for dimensions whose two D x (D+1) images do not fit LDS (D > 98, e.g. the 160-dimensional joint
// model of delta features): ONE lower triangle in packed storage (D(D+1)/2 doubles: 103 KB at D = 160) holds the
// covariance, then its Cholesky factor, then -- inverted in place, column by column from the last, with the column
// being replaced kept in a D-vector -- U = L^-1.  Outputs: the generic layout only (U, cz, lc: what
// logdens_tiled_kernel reads).
__global__ void __launch_bounds__(512)
px_prep_packed_kernel(const double *__restrict__ w, const double *__restrict__ mu, const double *__restrict__ sigma, int D,
                      int DP, double *__restrict__ U, double *__restrict__ cz, double *__restrict__ lc,
                      int *__restrict__ flag) {
  extern __shared__ double smem_pp[];
  const int tid = threadIdx.x, m = blockIdx.x, NTH = 512;
  double *P = smem_pp;                                   // packed lower triangle: (i, j <= i) at i (i + 1) / 2 + j
  double *col = P + (size_t)D * (D + 1) / 2;             // [D] the column being replaced (inverse), then cz
  __shared__ double lcs;
  __shared__ int bad;
  auto at = [](int i, int j) { return i * (i + 1) / 2 + j; };
  const double *Sg = sigma + (size_t)m * D * D;
  // Hermitian(Sigma): the upper triangle of the column-major block is the matrix (src/gmm.jl:16): (i, j <= i) = Sg(j, i)
  for (int e = tid; e < D * D; e += NTH) {
    const int i = e / D, j = e - i * D;
    if (j <= i) P[at(i, j)] = Sg[j + (size_t)D * i];
  }
  if (tid == 0) bad = 0;
  __syncthreads();
  // right-looking Cholesky
  for (int j = 0; j < D; ++j) {
    const double d = P[at(j, j)];
    if (!(d > 0.0) && tid == 0) bad = 1;
    const double sd = sqrt(d);
    __syncthreads();                                     // every thread has read the pivot
    for (int i = j + tid; i < D; i += NTH) P[at(i, j)] = (i == j) ? sd : P[at(i, j)] / sd;
    __syncthreads();
    const int n = D - 1 - j;
    for (int e = tid; e < n * n; e += NTH) {
      const int ii = e / n, kk = e - ii * n;
      const int i = j + 1 + ii, k = j + 1 + kk;
      if (i >= k) P[at(i, k)] = fma(-P[at(i, j)], P[at(k, j)], P[at(i, k)]);
    }
    __syncthreads();
  }
  if (tid == 0) {
    double ld = 0.0;
    for (int d = 0; d < D; ++d) ld += log(P[at(d, d)]);
    const double LOG2PI = 1.8378770664093454835606594728112;
    lcs = (w[m] > 0.0) ? log(w[m]) - 0.5 * (D * LOG2PI + 2.0 * ld) : -INFINITY;   // zero weight: posterior 0
  }
  // U = L^-1 in place: column j from the columns to its right (already inverse) and the original column j of L:
  //   U_jj = 1 / L_jj,   U_ij = -(sum_{k=j+1..i} U_ik L_kj) U_jj   (i > j)
  for (int j = D - 1; j >= 0; --j) {
    for (int i = j + tid; i < D; i += NTH) col[i] = P[at(i, j)];
    __syncthreads();
    const double ujj = 1.0 / col[j];
    for (int i = j + tid; i < D; i += NTH) {
      double v = ujj;
      if (i > j) {
        double sacc = 0.0;
        for (int k = j + 1; k <= i; ++k) sacc = fma(P[at(i, k)], col[k], sacc);
        v = -sacc * ujj;
      }
      P[at(i, j)] = v;
    }
    __syncthreads();
  }
  // cz = U mu
  for (int i = tid; i < D; i += NTH) {
    double sacc = 0.0;
    for (int c = 0; c <= i; ++c) sacc = fma(P[at(i, c)], mu[c + (size_t)D * m], sacc);
    col[i] = sacc;
  }
  __syncthreads();
  for (int e = tid; e < DP * DP; e += NTH) {
    const int r = e / DP, c = e - r * DP;
    U[(size_t)m * DP * DP + e] = (r < D && c <= r) ? P[at(r, c)] : 0.0;
  }
  for (int e = tid; e < DP; e += NTH) cz[(size_t)m * DP + e] = (e < D) ? col[e] : 0.0;
  if (tid == 0) {
    lc[m] = lcs;
    if (bad) atomicMax(flag, m + 1);
  }
}

static bool px_prep_fits_full(int D) {      // px_prep_kernel: two D x (D+1) images
  return (size_t)2 * D * (D + 1) * sizeof(double) + 4096 <= (size_t)160 * 1024;
}
bool gmm_px_device_prepare_supported(int D) {
  if (D < 1 || D > 256) return false;
  if (px_prep_fits_full(D)) return true;
  // packed variant: only for dimensions whose log-densities do not need the MFMA operand blocks (logdens_tiled_kernel)
  const int DP = (D + 3) / 4 * 4;
  return !gmmmap_has_mfma(DP) && ((size_t)D * (D + 1) / 2 + D) * sizeof(double) + 4096 <= (size_t)160 * 1024;
}

int gmm_px_prepare_device(vcmi_gmmmap **inout, const double *d_w, const double *d_mu, const double *d_sigma, int D, int M,
                          int *d_flag, hipStream_t st) {
  VCMI_TRY(check_device());
  if (!gmm_px_device_prepare_supported(D)) return fail(VCMI_ERR_ARG, "on-device p(x) preparation: dimension %d too large", D);
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (*inout && (*inout)->device != dev) {
    delete *inout;
    *inout = nullptr;
  }
  vcmi_gmmmap *g = *inout ? *inout : new (std::nothrow) vcmi_gmmmap();
  if (!g) return fail(VCMI_ERR_OOM, "out of host memory");
  *inout = g;
  g->device = dev;
  const int DP = (D + 3) / 4 * 4;
  g->D = D;
  g->DP = DP;
  g->M = M;
  VCMI_TRY(g->U.reserve((size_t)M * DP * DP));
  VCMI_TRY(g->cz.reserve((size_t)M * DP));
  VCMI_TRY(g->lc.reserve((size_t)M));
  const bool mfma = gmmmap_has_mfma(DP);
  TilingRT tl(DP, true);
  if (mfma) {
    VCMI_TRY(g->packedU.reserve((size_t)tl.BLK * M));
    if (g->px_table_dp != DP) {           // issue order: k-major over the U tiles (the host packer's phase 0)
      std::vector<int> tab;
      for (int ks = 0; ks < tl.KS; ++ks)
        for (int t = 0; t < std::min(tl.NU, tl.NT); ++t)
          if (ks < tl.steps(t)) tab.push_back((t << 16) | ks);
      if ((int)tab.size() != tl.NSTEPS) return fail(VCMI_ERR_ARG, "internal: tiling table mismatch");
      VCMI_TRY(g->px_table.reserve(tab.size()));
      VCMI_TRY(upload_now(g->px_table.p, tab.data(), tab.size() * sizeof(int)));
      g->px_table_dp = DP;
    }
  }
  if (!px_prep_fits_full(D)) {
    const size_t shp = ((size_t)D * (D + 1) / 2 + D) * sizeof(double);
    VCMI_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(px_prep_packed_kernel),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)shp));
    hipLaunchKernelGGL(px_prep_packed_kernel, dim3(M), dim3(512), shp, st, d_w, d_mu, d_sigma, D, DP, g->U.p, g->cz.p, g->lc.p,
                       d_flag);
    VCMI_HIP(hipGetLastError());
    return VCMI_OK;
  }
  const size_t shmem = (size_t)2 * D * (D + 1) * sizeof(double);
  VCMI_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(px_prep_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)shmem));
  hipLaunchKernelGGL(px_prep_kernel, dim3(M), dim3(256), shmem, st, d_w, d_mu, d_sigma, D, DP, mfma ? g->px_table.p : nullptr,
                     mfma ? tl.NSTEPS : 0, tl.CINIT_OFF, tl.NT * 16, tl.LC_OFF, tl.BLK, g->U.p, g->cz.p, g->lc.p,
                     mfma ? g->packedU.p : nullptr, d_flag);
  VCMI_HIP(hipGetLastError());
  return VCMI_OK;
}

}  // namespace vcmi

// ------------------------------------------------------------------------------------------------
// device-group replicas (devgroup.hpp)
// ------------------------------------------------------------------------------------------------
namespace vcmi {

void gmmmap_sync_replicas(vcmi_gmmmap *g) {
  const uint64_t ep = group_epoch();
  if (g->replicas_epoch == ep && (int)g->replicas.size() == group_size()) return;
  for (vcmi_gmmmap *r : g->replicas) delete r;
  g->replicas.assign((size_t)group_size(), nullptr);
  g->replicas_epoch = ep;
}

int gmmmap_member(vcmi_gmmmap *g, int member, vcmi_gmmmap **out) {
  // g itself serves the first member that sits on g's device; every other member gets its own converter
  bool first_on_home = group_device(member) == g->device;
  for (int k = 0; k < member && first_on_home; ++k)
    if (group_device(k) == g->device) first_on_home = false;
  if (first_on_home) {
    *out = g;
    return VCMI_OK;
  }
  vcmi_gmmmap *&r = g->replicas[(size_t)member];
  if (!r) {
    if (g->in_w.empty()) return fail(VCMI_ERR_ARG, "this converter cannot be replicated on another device");
    vcmi_gmmmap *n = new (std::nothrow) vcmi_gmmmap();
    if (!n) return fail(VCMI_ERR_OOM, "out of host memory");
    (void)hipGetDevice(&n->device);
    const int rc = prepare(n, g->in_w.data(), g->in_mu.data(), g->in_sigma.data(), g->in_Dj, g->M, g->in_swap);
    if (rc != VCMI_OK) {
      delete n;
      return rc;
    }
    r = n;
  }
  r->kernel_choice = g->kernel_choice;
  r->prune = g->prune;
  *out = r;
  return VCMI_OK;
}

// frames below which a call is not worth spreading over the group
static constexpr int64_t kGroupMinFrames = 4096;
// chunks of the host pipeline hold at least this many frames: one full round of workgroups on 256 CUs
static constexpr int64_t kMinChunkFrames = 98304;

// Frames are independent: member i of the group converts the contiguous block [lo, hi).
template <class PerShard>
static int over_frames(vcmi_gmmmap *g, int64_t T, const PerShard &fn) {
  const int m = group_size();
  if (m == 0 || T < kGroupMinFrames) return fn(g, (int64_t)0, T);
  gmmmap_sync_replicas(g);
  return group_run(m, [&](int i) -> int {
    int64_t lo, hi;
    shard_range(T, i, m, &lo, &hi);
    if (hi == lo) return VCMI_OK;
    vcmi_gmmmap *r = nullptr;
    VCMI_TRY(gmmmap_member(g, i, &r));
    return fn(r, lo, hi);
  });
}

}  // namespace vcmi

// ------------------------------------------------------------------------------------------------
// C-ABI
// ------------------------------------------------------------------------------------------------
using namespace vcmi;

extern "C" int vcmi_gmmmap_create(const double *weights, const double *mu, const double *sigma, int Dj, int M, int swap,
                                  vcmi_gmmmap **out) {
  if (!weights || !mu || !sigma || !out) return fail(VCMI_ERR_ARG, "vcmi_gmmmap_create: NULL argument");
  if (Dj < 2 || (Dj & 1) || M < 1) return fail(VCMI_ERR_DIM, "vcmi_gmmmap_create: joint dimension %d / mixtures %d invalid", Dj, M);
  *out = nullptr;
  VCMI_TRY(check_device());
  vcmi_gmmmap *g = new (std::nothrow) vcmi_gmmmap();
  if (!g) return fail(VCMI_ERR_OOM, "out of host memory");
  (void)hipGetDevice(&g->device);
  int rc = prepare(g, weights, mu, sigma, Dj, M, swap);
  if (rc != VCMI_OK) {
    delete g;
    return rc;
  }
  g->in_w.assign(weights, weights + M);
  g->in_mu.assign(mu, mu + (size_t)Dj * M);
  g->in_sigma.assign(sigma, sigma + (size_t)Dj * Dj * M);
  g->in_Dj = Dj;
  g->in_swap = swap;
  *out = g;
  return VCMI_OK;
}

extern "C" int vcmi_gmmmap_destroy(vcmi_gmmmap *g) {
  delete g;
  return VCMI_OK;
}
extern "C" int vcmi_gmmmap_dim(const vcmi_gmmmap *g) { return g ? g->D : -1; }
extern "C" int vcmi_gmmmap_ncomponents(const vcmi_gmmmap *g) { return g ? g->M : -1; }
extern "C" int vcmi_gmmmap_get_A(const vcmi_gmmmap *g, double *A) {
  if (!g || !A) return fail(VCMI_ERR_ARG, "vcmi_gmmmap_get_A: NULL argument");
  memcpy(A, g->h_A_julia.data(), g->h_A_julia.size() * sizeof(double));
  return VCMI_OK;
}
extern "C" int vcmi_gmmmap_set_kernel(vcmi_gmmmap *g, int which) {
  if (!g || which < 0 || which > 2) return fail(VCMI_ERR_ARG, "vcmi_gmmmap_set_kernel: bad argument");
  if (which == 2 && !gmmmap_has_mfma(g->DP)) return fail(VCMI_ERR_ARG, "no MFMA instantiation for dimension %d", g->D);
  g->kernel_choice = which;
  return VCMI_OK;
}

extern "C" int vcmi_gmmmap_prune_stats(vcmi_gmmmap *g, int enable, int64_t *evaluated) {
  if (!g) return fail(VCMI_ERR_ARG, "vcmi_gmmmap_prune_stats: NULL handle");
  if (evaluated) {
    *evaluated = 0;
    if (g->prune_count.p) {
      unsigned long long h = 0;
      VCMI_HIP(hipMemcpy(&h, g->prune_count.p, sizeof(h), hipMemcpyDeviceToHost));     // (synchronises with the null stream)
      *evaluated = (int64_t)h;
    }
  }
  if (enable) {
    if (!g->prune_count.p) VCMI_TRY(g->prune_count.alloc(3));
    VCMI_HIP(hipMemset(g->prune_count.p, 0, 3 * sizeof(unsigned long long)));
  } else {
    g->prune_count.release();
  }
  return VCMI_OK;
}

extern "C" int vcmi_gmmmap_convert_plan(vcmi_gmmmap *g, int64_t *mfma_issued, int *shape, double *model_active_frac,
                                        double *model_undecided_frac) {
  if (!g) return fail(VCMI_ERR_ARG, "vcmi_gmmmap_convert_plan: NULL handle");
  if (mfma_issued) {
    *mfma_issued = 0;
    if (g->prune_count.p) {
      unsigned long long h = 0;
      VCMI_HIP(hipMemcpy(&h, g->prune_count.p + 1, sizeof(h), hipMemcpyDeviceToHost));
      *mfma_issued = (int64_t)h;
    }
  }
  if (shape) *shape = !use_mfma(g) ? -1 : (!(g->prune < 1e300) ? 0 : convert_shape(g));
  if (model_active_frac) *model_active_frac = g->model_active_frac;
  if (model_undecided_frac) *model_undecided_frac = g->model_undecided_frac;
  return VCMI_OK;
}

extern "C" int vcmi_gmmmap_set_prune(vcmi_gmmmap *g, double nats) {
  if (!g) return fail(VCMI_ERR_ARG, "vcmi_gmmmap_set_prune: NULL handle");
  if (!(nats >= 40.0)) return fail(VCMI_ERR_ARG, "vcmi_gmmmap_set_prune: threshold %g nats would change y at double precision (>= 40)", nats);
  g->prune = nats >= 1e300 ? INFINITY : nats;
  return VCMI_OK;
}

static int check_xy(const vcmi_gmmmap *g, const void *X, int64_t ldx, int64_t T, const void *Y, int64_t ldy, const char *who) {
  if (!g) return fail(VCMI_ERR_ARG, "%s: NULL handle", who);
  if (T < 0) return fail(VCMI_ERR_ARG, "%s: negative frame count", who);
  if (T > 0 && (!X || !Y)) return fail(VCMI_ERR_ARG, "%s: NULL buffer", who);
  if (ldx < g->D || ldy < 1) return fail(VCMI_ERR_DIM, "%s: Inconsistent dimensions (leading dimension %lld < dim %d)", who, (long long)ldx, g->D);
  return VCMI_OK;
}

extern "C" int vcmi_gmmmap_convert_dev(vcmi_gmmmap *g, const double *dX, int64_t ldx, int64_t T, double *dY, int64_t ldy,
                                       void *stream) {
  VCMI_TRY(check_xy(g, dX, ldx, T, dY, ldy, "vcmi_gmmmap_convert_dev"));
  if (ldy < g->D) return fail(VCMI_ERR_DIM, "vcmi_gmmmap_convert_dev: ldy %lld < dim %d", (long long)ldy, g->D);
  return gmmmap_convert_device(g, dX, ldx, T, dY, ldy, as_stream(stream));
}

// Host pointers: chunks of frames travel pageable -> pinned -> HBM -> kernel -> pinned -> pageable with the three
// stages of consecutive chunks overlapping (hostpipe.hpp); with a device group the frame range is split first.
extern "C" int vcmi_gmmmap_convert(vcmi_gmmmap *g, const double *X, int64_t ldx, int64_t T, double *Y, int64_t ldy) {
  VCMI_TRY(check_xy(g, X, ldx, T, Y, ldy, "vcmi_gmmmap_convert"));
  if (ldy < g->D) return fail(VCMI_ERR_DIM, "vcmi_gmmmap_convert: ldy %lld < dim %d", (long long)ldy, g->D);
  if (T == 0) return VCMI_OK;
  const size_t row = (size_t)g->D * 8;
  return over_frames(g, T, [&](vcmi_gmmmap *r, int64_t lo, int64_t hi) -> int {
    return staged_pipeline(X + lo * ldx, row, (size_t)ldx * 8, Y + lo * ldy, row, (size_t)ldy * 8, hi - lo, kMinChunkFrames,
                           [&](const void *dIn, void *dOut, int64_t, int64_t n, hipStream_t st) -> int {
                             return gmmmap_convert_device(r, (const double *)dIn, r->D, n, (double *)dOut, r->D, st);
                           });
  });
}

extern "C" int vcmi_vc_frames(vcmi_gmmmap *g, const double *fm, int64_t T, double *out) {
  if (!g) return fail(VCMI_ERR_ARG, "vcmi_vc_frames: NULL handle");
  if (T < 0 || (T > 0 && (!fm || !out))) return fail(VCMI_ERR_ARG, "vcmi_vc_frames: bad argument");
  if (T == 0) return VCMI_OK;
  const int64_t ld = g->D + 1;   // row 1 is the power coefficient, src/common.jl:11
  const size_t row = (size_t)ld * 8;
  return over_frames(g, T, [&](vcmi_gmmmap *r, int64_t lo, int64_t hi) -> int {
    return staged_pipeline(fm + lo * ld, row, row, out + lo * ld, row, row, hi - lo, kMinChunkFrames,
                           [&](const void *dIn, void *dOut, int64_t, int64_t n, hipStream_t st) -> int {
                             // the copy keeps row 1 (src/common.jl:23); the kernel then overwrites rows 2..D+1
                             // (hipMemcpyDefault: a small call hands the pinned slots themselves to this function)
                             VCMI_HIP(hipMemcpyAsync(dOut, dIn, (size_t)n * row, hipMemcpyDefault, st));
                             return gmmmap_convert_device(r, (const double *)dIn + 1, ld, n, (double *)dOut + 1, ld, st);
                           });
  });
}

extern "C" int vcmi_gmmmap_posterior_dev(vcmi_gmmmap *g, const double *dX, int64_t ldx, int64_t T, double *dP, void *stream) {
  VCMI_TRY(check_xy(g, dX, ldx, T, dP, 1, "vcmi_gmmmap_posterior_dev"));
  return gmmmap_posterior_device(g, dX, ldx, T, dP, as_stream(stream));
}

extern "C" int vcmi_gmmmap_posterior(vcmi_gmmmap *g, const double *X, int64_t ldx, int64_t T, double *P) {
  VCMI_TRY(check_xy(g, X, ldx, T, P, 1, "vcmi_gmmmap_posterior"));
  if (T == 0) return VCMI_OK;
  const int M = g->M;
  return over_frames(g, T, [&](vcmi_gmmmap *r, int64_t lo, int64_t hi) -> int {
    return staged_pipeline(X + lo * ldx, (size_t)r->D * 8, (size_t)ldx * 8, P + lo * M, (size_t)M * 8, (size_t)M * 8, hi - lo,
                           kMinChunkFrames, [&](const void *dIn, void *dOut, int64_t, int64_t n, hipStream_t st) -> int {
                             return gmmmap_posterior_device(r, (const double *)dIn, r->D, n, (double *)dOut, st);
                           });
  });
}

extern "C" int vcmi_gmmmap_predict_dev(vcmi_gmmmap *g, const double *dX, int64_t ldx, int64_t T, int64_t *didx, void *stream) {
  VCMI_TRY(check_xy(g, dX, ldx, T, didx, 1, "vcmi_gmmmap_predict_dev"));
  return gmmmap_predict_device(g, dX, ldx, T, didx, as_stream(stream));
}

extern "C" int vcmi_gmmmap_predict(vcmi_gmmmap *g, const double *X, int64_t ldx, int64_t T, int64_t *idx) {
  VCMI_TRY(check_xy(g, X, ldx, T, idx, 1, "vcmi_gmmmap_predict"));
  if (T == 0) return VCMI_OK;
  return over_frames(g, T, [&](vcmi_gmmmap *r, int64_t lo, int64_t hi) -> int {
    return staged_pipeline(X + lo * ldx, (size_t)r->D * 8, (size_t)ldx * 8, idx + lo, 8, 8, hi - lo, kMinChunkFrames,
                           [&](const void *dIn, void *dOut, int64_t, int64_t n, hipStream_t st) -> int {
                             return gmmmap_predict_device(r, (const double *)dIn, r->D, n, (int64_t *)dOut, st);
                           });
  });
}
