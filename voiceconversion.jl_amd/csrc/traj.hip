// traj.hip -- trajectory (MLPG) conversion with a GMM on static+delta features, MI355X (gfx950).
//
// Replaces   TrajectoryGMMMap ctor (Dy_m = inv(Syy_m - A_m Sxy_m))      reference src/trajectory_gmmmap.jl:11-31
//            constructW / compute_wt (never materialised: W is a stencil) reference src/trajectory_gmmmap.jl:39-61
//            fvconvert(tgmm, X)                                          reference src/trajectory_gmmmap.jl:65-110
//            vc(c::TrajectoryConverter, fm)                              reference src/common.jl:31-63
//            push_delta                                                  reference src/datasets.jl:6-13
//
// Math.  With mhat_t = argmax_m p(m | X_t), E_t = mu^y_m + A_m (X_t - mu^x_m), Q_t = Dy_mhat_t cut
// into DxD blocks [[Qss,Qsd],[Qds,Qdd]] and g_t = Q_t E_t = [gs; gd], the normal equations
// (W' Dy^-1 W) y = W' Dy^-1 E of :103-105 are block-pentadiagonal:
//   P[t,t]   = Qss(t) + Qdd(t-1)/4 + Qdd(t+1)/4        r[t] = gs(t) + gd(t-1)/2 - gd(t+1)/2
//   P[t,t-1] = Qds(t-1)/2 - Qsd(t)/2                   (terms with t-1 < 1 or t+1 > T dropped, as W drops them)
//   P[t,t-2] = -Qdd(t-1)/4
// One workgroup per utterance factorises P = L L' with a right-looking Cholesky on a sliding 3D x 3D window held
// in LDS (the right-hand side rides along as an extra row, so z = L^-1 r falls out of the same updates), streams
// the D-column panels of L to an HBM workspace, and back-substitutes L' y = z reading the panels in reverse.
#include "vcmi_common.hpp"
#include "host_linalg.hpp"
#include "gmmmap_handle.hpp"
#include "devgroup.hpp"
#include "hostpipe.hpp"
#include "lds_dma.hpp"
#include "postf.hpp"

#include <algorithm>
#include <cstdlib>
#include <type_traits>

struct vcmi_traj {
  vcmi_gmmmap *g = nullptr;
  int D2 = 0;          // dim(t) = 2D (static + delta), src/trajectory_gmmmap.jl:35
  int D = 0;           // static dimension
  int M = 0;
  int64_t length = 0;  // length(t), src/trajectory_gmmmap.jl:34
  vcmi::DevBuf<double> AT, QT, bvec, Q;   // [M][k][r] transposed A and Q (coalesced gemv), b [M][2D], Q row-major [M][2D][2D]
  vcmi::DevBuf<double> Qfrag, Afrag;      // Q and A in v_mfma_f64_16x16x4 A-operand order [M][row tile][k-step][lane]
  vcmi::DevBuf<int> gperm;                // frames of every utterance grouped by mixture (traj_g_mfma_kernel)
  int NT = 0, KS = 0;                     // row tiles / k-steps of Qfrag
  vcmi::DevBuf<double> gbuf, ws, xbuf, ybuf;
  // Static dimensions without an instantiation of the blocked solver run in the next larger one (Dpad): Qpad is Q with
  // the extra static dimensions decoupled (unit diagonal in Qss, zeros elsewhere), gpad / ypad the padded right-hand
  // sides and solutions of a call
  int Dpad = 0;
  vcmi::DevBuf<double> Qpad, gpad, ypad;
  bool big = false;                       // static D beyond the LDS-window solvers (D >= 47): traj_solve_big_kernel, window in HBM
  vcmi::DevBuf<double> gwin;
  vcmi::DevBuf<unsigned char> uttpad;
  vcmi::DevBuf<int64_t> mhat;
  vcmi::DevBuf<int> status;
  vcmi::DevBuf<unsigned char> uttbuf;
  // converters on the other devices of a device group (vcmi_set_devices), made lazily by the members' worker threads
  std::vector<vcmi_traj *> replicas;
  uint64_t replicas_epoch = 0;
  ~vcmi_traj() {
    for (vcmi_traj *r : replicas) delete r;
  }
};

// TrajectoryGVGMMMap(tgmm, mu^v, Sigma^vv), src/trajectory_gmmmap.jl:114-130
struct vcmi_trajgv {
  vcmi_traj *t = nullptr;
  vcmi::DevBuf<double> muv, pv;   // (D), (D,D) = inv(Sigma^vv) in the Julia memory image
  std::vector<double> h_muv, h_pv;             // host copies for the device-group replicas
  std::vector<vcmi_trajgv *> replicas;
  uint64_t replicas_epoch = 0;
  ~vcmi_trajgv() {
    for (vcmi_trajgv *r : replicas) delete r;
  }
};

namespace vcmi {

struct TrajGV {          // per-call parameters of the GV ascent
  const double *muv, *pv;
  int epochs;
  double alpha;
};

struct TrajUtt {
  const double *X;   // (2D,T) dense
  double *Y;         // (D,T) dense
  int64_t frame0;    // offset of this utterance in the packed per-frame scratch (mhat, g)
  int32_t T;
  int32_t idx;       // position in the caller's batch (the list is sorted by length before the launch)
};

// A pointer read from a descriptor in memory has no known address space: every access through it is a FLAT
// instruction (both wait counters, no saddr form).  The utterance matrices are global memory: say so.
typedef double __attribute__((address_space(1))) gdouble;

// ------------------------------------------------------------------------------------------------
// g_t = Q_mhat (A_mhat x_t + b_mhat): one workgroup per frame, thread r owns output row r
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(128)
traj_g_kernel(const double *__restrict__ X, int64_t nframes, int D2, const int64_t *__restrict__ mhat,
              const double *__restrict__ AT, const double *__restrict__ QT, const double *__restrict__ bvec,
              double *__restrict__ G) {
  extern __shared__ double sm[];   // x[D2], e[D2]
  double *xs = sm, *es = sm + D2;
  const int64_t fr = blockIdx.x;
  const int m = (int)mhat[fr] - 1;
  for (int r = threadIdx.x; r < D2; r += blockDim.x) xs[r] = X[fr * D2 + r];
  __syncthreads();
  const double *A = AT + (size_t)m * D2 * D2, *Q = QT + (size_t)m * D2 * D2;
  for (int r = threadIdx.x; r < D2; r += blockDim.x) {
    double e = bvec[(size_t)m * D2 + r];                       // E = mu^y + A (x - mu^x), src/trajectory_gmmmap.jl:88
    for (int k = 0; k < D2; ++k) e = fma(A[(size_t)k * D2 + r], xs[k], e);
    es[r] = e;
  }
  __syncthreads();
  for (int r = threadIdx.x; r < D2; r += blockDim.x) {
    double s = 0.0;
    for (int k = 0; k < D2; ++k) s = fma(Q[(size_t)k * D2 + r], es[k], s);
    G[fr * D2 + r] = s;
  }
}

// ------------------------------------------------------------------------------------------------
// banded Cholesky solve, one workgroup (256 threads) per utterance
// ------------------------------------------------------------------------------------------------
// 1/sqrt(x) in FP64: hardware v_rsq_f64 seed + two Newton steps (each roughly doubles the correct bits; the seed
// has >= 26) -- a short dependent chain instead of the IEEE sqrt + divide expansion on the per-column critical path.
__device__ __forceinline__ double traj_rsqrt(double x) {
  double y = __builtin_amdgcn_rsq(x);
  y = y * fma(-0.5 * x * y, y, 1.5);
  y = y * fma(-0.5 * x * y, y, 1.5);
  return y;
}

// back substitution  L' y = z  from the panels in the HBM workspace (shared by both solve kernels).
// Panels are double-buffered in LDS (`buf`, 2 x PAN doubles): panel t-1 is fetched (coalesced, through registers)
// while step t computes.  The sequential part -- the D-step triangular solve -- runs in one wave with the needed
// row entries and reciprocal diagonal preloaded, so its chain is one shuffle + one FMA per step.
template <int NPRE>
__device__ void traj_backsub(const double *__restrict__ ws, size_t PAN, int D, int T, double *buf, double *yring, double *wv,
                             double *rdiag, double *__restrict__ Y) {
  const int tid = threadIdx.x, W3 = 3 * D;
#ifdef TRAJ_NO_BACKSUB
  return;
#endif
  for (int i = tid; i < 2 * D; i += 256) yring[i] = 0.0;
  {
    const double *pan = ws + (size_t)(T - 1) * PAN;
    for (size_t e = tid; e < PAN; e += 256) buf[((T - 1) & 1) * PAN + e] = pan[e];
  }
  __syncthreads();
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
    // w = z - E' y1 - F' y2  (thread j owns column j; panel rows D..3D-1 hold E then F)
    if (tid < D) {
      double s = pb[(size_t)W3 * D + tid];
      for (int i = 0; i < D; ++i) s = fma(-pb[(size_t)(D + i) * D + tid], y1[i], s);
      for (int i = 0; i < D; ++i) s = fma(-pb[(size_t)(2 * D + i) * D + tid], y2[i], s);
      wv[tid] = s;
    } else if (tid >= 64 && tid < 64 + D) {
      rdiag[tid - 64] = 1.0 / pb[(size_t)(tid - 64) * D + (tid - 64)];
    }
    __syncthreads();
    // Dg' y = w : sequential in k, one wave
    if (tid < 64) {
      double w = (tid < D) ? wv[tid] : 0.0;
      const double rd = (tid < D) ? rdiag[tid] : 0.0;
      for (int k = D - 1; k >= 0; --k) {
        const double lk = (tid < k) ? pb[(size_t)k * D + tid] : 0.0;   // independent of the chain: issued ahead
        const double yk = __shfl(w * rd, k);
        w = (tid == k) ? yk : fma(-lk, yk, w);
      }
      if (tid < D) {
        y2[tid] = w;                       // becomes y_t; the slot of y_{t+2} is free now
        Y[(size_t)t * D + tid] = w;        // reshape(y, D, T), src/trajectory_gmmmap.jl:109
      }
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

// assemble global block row a of P (and r) into local block row la of the LDS window: blocks (a,a-2), (a,a-1), (a,a)
__device__ void traj_add_block_row(double *Wd, double *rr, int LD, int D, int a, int la, int T,
                                   const int64_t *__restrict__ mh, const double *__restrict__ g,
                                   const double *__restrict__ Qall) {
  const int tid = threadIdx.x, D2 = 2 * D;
  const double *Qa = Qall + (size_t)(mh[a] - 1) * D2 * D2;
  const double *Qm = (a >= 1) ? Qall + (size_t)(mh[a - 1] - 1) * D2 * D2 : nullptr;
  const double *Qp = (a + 1 < T) ? Qall + (size_t)(mh[a + 1] - 1) * D2 * D2 : nullptr;
  int i = tid / D, j = tid - i * D;                      // one division per call, then incremental
  const int di = 256 / D, dj = 256 - di * D;
  for (int e = tid; e < D * D; e += 256) {
    double *row = Wd + (size_t)(la * D + i) * LD;
    double v = Qa[(size_t)i * D2 + j];                                        // Qss(a)
    if (Qm) v += 0.25 * Qm[(size_t)(D + i) * D2 + (D + j)];                   // + Qdd(a-1)/4
    if (Qp) v += 0.25 * Qp[(size_t)(D + i) * D2 + (D + j)];                   // + Qdd(a+1)/4
    row[la * D + j] = v;
    if (la >= 1 && a >= 1)
      row[(la - 1) * D + j] = 0.5 * Qm[(size_t)(D + i) * D2 + j] - 0.5 * Qa[(size_t)i * D2 + (D + j)];   // Qds(a-1)/2 - Qsd(a)/2
    if (la >= 2 && a >= 2)
      row[(la - 2) * D + j] = -0.25 * Qm[(size_t)(D + i) * D2 + (D + j)];     // -Qdd(a-1)/4
    i += di;
    j += dj;
    if (j >= D) { j -= D; ++i; }
  }
  for (int k = tid; k < D; k += 256) {
    double v = g[(size_t)a * D2 + k];
    if (a >= 1) v += 0.5 * g[(size_t)(a - 1) * D2 + D + k];
    if (a + 1 < T) v -= 0.5 * g[(size_t)(a + 1) * D2 + D + k];
    rr[la * D + k] = v;
  }
}

// LDS window: rows/cols 0..3D-1 = global rows t*D .. t*D+3D-1 of the band (row stride LD); after block column t is
// finished the lower-right 2D x 2D part is shifted up-left by D (through registers) and block row t+3 is assembled
// into the freed rows.  No index arithmetic beyond adds in the hot loops.
static constexpr int kBackPre = (139 * 46 + 255) / 256;   // panel doubles per thread for the largest supported D
static constexpr int kShiftRegs = 36;   // ceil(4*46^2 / 256) doubles per thread for the window shift

__global__ void __launch_bounds__(256)
traj_solve_kernel(const TrajUtt *__restrict__ utts, int n, int D, const double *__restrict__ Qall,
                  const int64_t *__restrict__ mhat_all, const double *__restrict__ g_all, double *__restrict__ ws_all,
                  int64_t ws_stride, int *__restrict__ status) {
  const int D2 = 2 * D, W3 = 3 * D, LD = W3 + 1;
  extern __shared__ double sm[];
  double *Wd = sm;                       // [W3][LD]
  double *rr = Wd + (size_t)W3 * LD;     // [W3] right-hand side riding along as an extra row
  double *lcol = rr + W3;                // [W3] scaled pivot column of the current elimination step
  double *yring = lcol + W3;             // [2][D]  y_{t+1}, y_{t+2} during back-substitution
  double *wv = yring + 2 * D;            // [D]
  __shared__ int bad;
  __shared__ double zc_s;
  const int tid = threadIdx.x;
  const int ti = tid >> 4, tj = tid & 15;
  const size_t PAN = (size_t)(W3 + 1) * D;   // panel: rows 0..3D-1 of L[:, block t] (relative to t) + z row

  for (int u = blockIdx.x; u < n; u += gridDim.x) {
    const TrajUtt U = utts[u];
    const int T = U.T;
    if (T == 0) continue;
    const int64_t *mh = mhat_all + U.frame0;
    const double *g = g_all + U.frame0 * D2;
    double *ws = ws_all + (size_t)blockIdx.x * ws_stride;
    if (tid == 0) bad = 0;

    auto add_block_row = [&](int a, int la) { traj_add_block_row(Wd, rr, LD, D, a, la, T, mh, g, Qall); };

    for (int a = 0; a < 3 && a < T; ++a) add_block_row(a, a);
    __syncthreads();

    // ---------------- factorisation + forward substitution ----------------
    for (int t = 0; t < T; ++t) {
      const int nb = (T - t < 3) ? T - t : 3;      // block rows alive in the window
      const int nrows = nb * D;
      for (int c = 0; c < D; ++c) {
        // (a) pivot and scaled column into lcol (the window column itself is left untouched until (b))
        const double piv = Wd[(size_t)c * LD + c];
        if (!(piv > 0.0) && tid == 0) bad = 1;
        const double dinv = traj_rsqrt(piv);
        for (int lr = c + tid; lr < nrows; lr += 256) lcol[lr] = (lr == c) ? piv * dinv : Wd[(size_t)lr * LD + c] * dinv;
        if (tid == 255) zc_s = rr[c] * dinv;
        __syncthreads();
        // (b) rank-1 update of the trailing lower triangle and of the rhs; the finished column goes back to the window
        const int rem = nrows - c - 1;
        for (int a = ti; a < rem; a += 16) {
          const int ri = c + 1 + a;
          const double lic = lcol[ri];
          double *row = Wd + (size_t)ri * LD + c + 1;
          for (int b = tj; b <= a; b += 16) row[b] = fma(-lic, lcol[c + 1 + b], row[b]);
        }
        const double zc = zc_s;
        for (int lr = c + tid; lr < nrows; lr += 256) {
          Wd[(size_t)lr * LD + c] = lcol[lr];
          if (lr > c) rr[lr] = fma(-zc, lcol[lr], rr[lr]);
          else rr[lr] = zc;
        }
        __syncthreads();
      }
      // stream the finished panel: rows 0..3D-1 (zero beyond nrows), columns of block t; then the z row
      double *pan = ws + (size_t)t * PAN;
      {
        int lr = tid / D, cc = tid - lr * D;
        const int dl = 256 / D, dc = 256 - dl * D;
        for (int e = tid; e < W3 * D; e += 256) {
          pan[e] = (lr < nrows) ? Wd[(size_t)lr * LD + cc] : 0.0;
          lr += dl;
          cc += dc;
          if (cc >= D) { cc -= D; ++lr; }
        }
      }
      for (int cc = tid; cc < D; cc += 256) pan[(size_t)W3 * D + cc] = rr[cc];
      // shift the window up-left by D (through registers), then assemble block row t+3
      double sh[kShiftRegs];
      double rsh = 0.0;
      {
        int i = tid / D2, j = tid - i * D2;
        const int di = 256 / D2, dj = 256 - di * D2;
#pragma unroll
        for (int k = 0; k < kShiftRegs; ++k) {
          sh[k] = (i < D2) ? Wd[(size_t)(i + D) * LD + (j + D)] : 0.0;
          i += di;
          j += dj;
          if (j >= D2) { j -= D2; ++i; }
        }
        if (tid < D2) rsh = rr[tid + D];
      }
      __syncthreads();
      {
        int i = tid / D2, j = tid - i * D2;
        const int di = 256 / D2, dj = 256 - di * D2;
#pragma unroll
        for (int k = 0; k < kShiftRegs; ++k) {
          if (i < D2) Wd[(size_t)i * LD + j] = sh[k];
          i += di;
          j += dj;
          if (j >= D2) { j -= D2; ++i; }
        }
        if (tid < D2) rr[tid] = rsh;
      }
      if (t + 3 < T) add_block_row(t + 3, 2);
      __syncthreads();
    }

    traj_backsub<kBackPre>(ws, PAN, D, T, Wd, yring, wv, lcol, U.Y);
    if (tid == 0 && bad) status[0] = 1;
    __syncthreads();
  }
}

#include "traj_solve_blk.hpp"

// ------------------------------------------------------------------------------------------------
// g_t = Q_mhat (A_mhat x_t + b_mhat) on v_mfma_f64_16x16x4 (replaces one workgroup per frame streaming both 2D x 2D
// matrices from L2: 52 GB of L2 traffic per 512k frames).  One workgroup per utterance: frames grouped by mixture
// (counting sort, segments padded to 16), so a tile of 16 frames has ONE mixture; wave i owns row tile i of both
// products.  The two products chain without a layout change: register r of wave i's E accumulator (row 16i+4r+lane/16,
// frame lane%16) IS the B operand of k-step 4i+r of the second product for the same lane -- it only has to be shared
// with the other waves, through LDS.
// ------------------------------------------------------------------------------------------------
static constexpr int kGMaxKS = 24;   // k-steps of the widest supported feature vector (2D <= 96)
typedef double g_d4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(384)
traj_g_mfma_kernel(const TrajUtt *__restrict__ utts, int n, int D2, int M, int KS, const double *__restrict__ Afrag,
                   const double *__restrict__ Qfrag, const double *__restrict__ bvec, const int64_t *__restrict__ mhat_all,
                   int *__restrict__ perm_all, double *__restrict__ G_all, int split) {
  extern __shared__ double gsm2[];
  const int nthr = blockDim.x, NT = nthr >> 6;
  double *Xt = gsm2;                              // [4*KS][16] x of the tile's frames, k-major
  double *Ef = Xt + (size_t)4 * KS * 16;          // [4*NT][64] E accumulators in B-operand order (k-step major)
  int *cnt = reinterpret_cast<int *>(Ef + (size_t)4 * NT * 64);   // [M]
  int *start = cnt + M;                           // [M]
  __shared__ int tidx[16];
  __shared__ int ntiles_s;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lcol = lane & 15, lgrp = lane >> 4;
  // `split` (a power of two) workgroups share an utterance: a tile takes ~6 us of dependent loads and barriers and an
  // utterance has T / 16 and more of them -- with one workgroup per utterance and 256 utterances the kernel is a latency
  // chain on every CU.  Member `part` takes the tiles part, part + split, ... of the utterance's grouped order; every member
  // builds that order itself, so it has to be a function of the data: a STABLE counting sort (ranks by ballot / mbcnt per
  // wave, waves and chunks of frames in order), not one by atomics.  (Splitting by MIXTURE instead -- whole groups per
  // member, atomics allowed -- was measured first: a smooth trajectory stays in one or two mixtures, one member gets all the
  // tiles.)  The member index is the HIGH part of blockIdx: the members of an utterance sit on different XCDs.
  int *wcnt = start + M;                          // [NT][M] per-wave counts of the current chunk
  const int nutt = (int)(gridDim.x / (unsigned)split);
  const int part = (int)(blockIdx.x / (unsigned)nutt);
  for (int u = (int)(blockIdx.x % (unsigned)nutt); u < n; u += nutt) {
    const TrajUtt U = utts[u];
    const int T = U.T;
    if (T == 0) continue;
    const int64_t *mh = mhat_all + U.frame0;
    double *G = G_all + U.frame0 * D2;
    // regions indexed by the utterance's position in the batch (frame0 grows with it): disjoint whatever the launch order
    int *perm = perm_all + U.frame0 + (int64_t)16 * M * U.idx;
    // frames grouped by mixture, in frame order inside a group
    for (int m = tid; m < M; m += nthr) cnt[m] = 0;
    __syncthreads();
    for (int t = tid; t < T; t += nthr) atomicAdd(&cnt[(int)mh[t] - 1], 1);
    __syncthreads();
    if (tid == 0) {
      int pos = 0;
      for (int m = 0; m < M; ++m) {
        start[m] = pos;
        pos += (cnt[m] + 15) / 16 * 16;
        cnt[m] = 0;                                     // from here on: frames of mixture m placed so far
      }
      ntiles_s = pos / 16;
    }
    __syncthreads();
    const int ntiles = ntiles_s;
    for (int e = tid; e < ntiles * 16; e += nthr)
      if (((e >> 4) & (split - 1)) == part) perm[e] = -1;          // own tiles only
    for (int c0 = 0; c0 < T; c0 += nthr) {
      for (int e = tid; e < NT * M; e += nthr) wcnt[e] = 0;
      __syncthreads();
      const int t = c0 + tid;
      const int m = t < T ? (int)mh[t] - 1 : -1;
      int rank = 0;
      unsigned long long rem = __builtin_amdgcn_ballot_w64(m >= 0);
      while (rem) {                                     // one turn per distinct mixture among the wave's 64 frames
        const int lead = __builtin_ctzll(rem);
        const int k0 = __builtin_amdgcn_readlane(m, lead);
        const unsigned long long mask = __builtin_amdgcn_ballot_w64(m == k0);
        if (m == k0) rank = __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
        if (lane == lead) wcnt[wave * M + k0] = __builtin_popcountll(mask);
        rem &= ~mask;
      }
      __syncthreads();
      if (m >= 0) {
        int off = cnt[m] + rank;
        for (int w = 0; w < wave; ++w) off += wcnt[w * M + m];
        const int slot = start[m] + off;
        if (((slot >> 4) & (split - 1)) == part) perm[slot] = t;
      }
      __syncthreads();
      for (int m2 = tid; m2 < M; m2 += nthr) {
        int sum = 0;
        for (int w = 0; w < NT; ++w) sum += wcnt[w * M + m2];
        cnt[m2] += sum;
      }
      __syncthreads();
    }
    double afr[kGMaxKS], qfr[kGMaxKS];
    int mcur = -1;
    for (int tile = part; tile < ntiles; tile += split) {
      if (tid < 16) tidx[tid] = perm[tile * 16 + tid];
      __syncthreads();
      for (int e = tid; e < 4 * KS * 16; e += nthr) {
        const int k = e >> 4, t = tidx[e & 15];
        Xt[e] = (t >= 0 && k < D2) ? U.X[(size_t)t * D2 + k] : 0.0;
      }
      const int m = (int)mh[tidx[0]] - 1;               // slot 0 of a tile is never padding
      if (m != mcur) {
        mcur = m;
        const double *A = Afrag + (((size_t)m * NT + wave) * KS) * 64 + lane;
        const double *Q = Qfrag + (((size_t)m * NT + wave) * KS) * 64 + lane;
#pragma unroll
        for (int ks = 0; ks < kGMaxKS; ++ks) {
          afr[ks] = (ks < KS) ? A[(size_t)ks * 64] : 0.0;
          qfr[ks] = (ks < KS) ? Q[(size_t)ks * 64] : 0.0;
        }
      }
      g_d4 acc;                                         // E = A x + b, src/trajectory_gmmmap.jl:88
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * wave + lgrp + 4 * r;
        acc[r] = (row < D2) ? bvec[(size_t)m * D2 + row] : 0.0;
      }
      __syncthreads();
#pragma unroll
      for (int ks = 0; ks < kGMaxKS; ++ks)
        if (ks < KS) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(afr[ks], Xt[(4 * ks + lgrp) * 16 + lcol], acc, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) Ef[(size_t)(4 * wave + r) * 64 + lane] = acc[r];
      __syncthreads();
      g_d4 gac = {0.0, 0.0, 0.0, 0.0};                  // g = Q E
#pragma unroll
      for (int ks = 0; ks < kGMaxKS; ++ks)
        if (ks < KS) gac = __builtin_amdgcn_mfma_f64_16x16x4f64(qfr[ks], Ef[(size_t)ks * 64 + lane], gac, 0, 0, 0);
      const int t = tidx[lcol];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * wave + lgrp + 4 * r;
        if (row < D2 && t >= 0) G[(size_t)t * D2 + row] = gac[r];
      }
      __syncthreads();
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Global-variance ascent, fvconvert(tgv::TrajectoryGVGMMMap, X), src/trajectory_gmmmap.jl:139-189.
// One workgroup per utterance runs all epochs:  y <- y + alpha * ( omega (r - P y) + gvgrad(y) ),  omega = 1/(2T),
// with P y = W' D^-1 W y applied as  u_t = [y_t ; (y_{t+1} - y_{t-1})/2]  ->  v_t = Q_mhat_t u_t  ->
// (P y)_t = vs_t + vd_{t-1}/2 - vd_{t+1}/2  (the stencil of W; W is never built) and r = W' D^-1 E from the g_t the
// solve already has.  v = Q u runs on v_mfma_f64_16x16x4 over tiles of 16 consecutive frames: wave i owns row tile i
// of Q; the tile's frames usually share one or two mixtures, so the product is accumulated over the DISTINCT
// mixtures of the tile with the B operand masked to that mixture's frames (exact: the other frames add 0).
// ------------------------------------------------------------------------------------------------
template <typename YP>
__device__ void gv_moments(YP y, int D, int T, int nthr, double *red, double *mean, double *var) {
  const int tid = threadIdx.x;
  const int NG = nthr / D;                 // frame groups per dimension
  const int d = tid % D, g = tid / D;
  double s = 0.0;
  if (g < NG) {
    // 8 loads in flight per thread (a plain loop keeps one: with one workgroup per CU the passes of this kernel are
    // bound by memory-level parallelism, not by HBM bandwidth); the additions stay in frame order
    int t = g;
    for (; t + 7 * NG < T; t += 8 * NG) {
      double v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = y[(size_t)(t + q * NG) * D + d];
#pragma unroll
      for (int q = 0; q < 8; ++q) s += v[q];
    }
    for (; t < T; t += NG) s += y[(size_t)t * D + d];
  }
  if (g < NG) red[g * D + d] = s;
  __syncthreads();
  if (tid < D) {
    double m = 0.0;
    for (int k = 0; k < NG; ++k) m += red[k * D + tid];
    mean[tid] = m / (double)T;
  }
  __syncthreads();
  s = 0.0;
  if (g < NG) {
    const double m = mean[d];
    int t = g;
    for (; t + 7 * NG < T; t += 8 * NG) {
      double v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = y[(size_t)(t + q * NG) * D + d];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const double e = v[q] - m;
        s = fma(e, e, s);
      }
    }
    for (; t < T; t += NG) {
      const double e = y[(size_t)t * D + d] - m;
      s = fma(e, e, s);
    }
    red[g * D + d] = s;
  }
  __syncthreads();
  if (tid < D) {
    double v = 0.0;
    for (int k = 0; k < NG; ++k) v += red[k * D + tid];
    var[tid] = v / (double)(T - 1);        // Julia's var: corrected
  }
  __syncthreads();
}

typedef double gv_d4 __attribute__((ext_vector_type(4)));
static constexpr int kGvNB = 8;   // 16-frame tiles per round of the GV product
static constexpr int kGvMaxKS = 24;   // k-steps of the widest supported feature vector (2D <= 96)

__global__ void __launch_bounds__(384)
traj_gv_kernel(const TrajUtt *__restrict__ utts, int n, int D, int M, int KS, const double *__restrict__ Qfrag,
               const int64_t *__restrict__ mhat_all, const double *__restrict__ g_all, double *__restrict__ ws_all,
               int64_t ws_stride, TrajGV gv) {
  extern __shared__ double gsm[];
  const int D2 = 2 * D, nthr = blockDim.x, NT = nthr >> 6;
  double *Ut = gsm;                        // [kGvNB][4*KS][16]  u of the tiles' frames, k-major
  double *red = Ut + (size_t)kGvNB * 4 * KS * 16;  // [2][nthr]
  double *mean = red + 2 * nthr;           // [D]
  double *var = mean + D;                  // [D]
  double *coef = var + D;                  // [D]
  int *cnt = reinterpret_cast<int *>(coef + D);   // [M] frames per mixture, then the fill cursor
  int *start = cnt + M;                           // [M] first slot of the mixture's (16-padded) segment
  __shared__ int tidx[16 * kGvNB];
  __shared__ int ntiles_s;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lcol = lane & 15, lgrp = lane >> 4;

  for (int u = blockIdx.x; u < n; u += gridDim.x) {
    const TrajUtt U = utts[u];
    const int T = U.T;
    if (T < 2) continue;                   // var() of one frame is undefined; the host rejects such calls
    const int64_t *mh = mhat_all + U.frame0;
    const double *g = g_all + U.frame0 * D2;
    gdouble *y = (gdouble *)U.Y;
    double *V = ws_all + (size_t)blockIdx.x * ws_stride;   // [T][2D]
    double *R = V + (size_t)T * D2;                        // [T][D]   r = W' D^-1 E
    int *perm = reinterpret_cast<int *>(R + (size_t)T * D); // frames grouped by mixture, segments padded to 16 with -1
    const double omega = 1.0 / (2.0 * (double)T);

    // frames grouped by mixture (the selection mhat is fixed over the epochs): every 16-frame tile of the product
    // below then has ONE mixture.  The order inside a group comes from atomics and is irrelevant: each frame's
    // product is computed independently.
    for (int m = tid; m < M; m += nthr) cnt[m] = 0;
    __syncthreads();
    for (int t = tid; t < T; t += nthr) atomicAdd(&cnt[(int)mh[t] - 1], 1);
    __syncthreads();
    if (tid == 0) {
      int pos = 0;
      for (int m = 0; m < M; ++m) {
        start[m] = pos;
        pos += (cnt[m] + 15) / 16 * 16;
        cnt[m] = 0;
      }
      ntiles_s = pos / 16;
    }
    __syncthreads();
    const int ntiles = ntiles_s;
    for (int e = tid; e < ntiles * 16; e += nthr) perm[e] = -1;
    __syncthreads();
    for (int t = tid; t < T; t += nthr) {
      const int m = (int)mh[t] - 1;
      perm[start[m] + atomicAdd(&cnt[m], 1)] = t;
    }

    // eq. (58): y <- sqrt(mu^v / var(y)) (y - mean) + mean, src/trajectory_gmmmap.jl:152; and r.
    // Every pass that writes y also accumulates the moments of what it writes (thread = (dimension d, frame group g),
    // the mapping of gv_moments): sum (y - c) and sum (y - c)^2 with the shift c = mean before the pass, so that
    // mean = c + S1/T, var = (S2 - S1^2/T)/(T-1) lose nothing to cancellation and the next epoch needs no pass of
    // its own over y for them.
    gv_moments(y, D, T, nthr, red, mean, var);
    const int NGm = nthr / D, dm = tid % D, gm = tid / D;
    auto finish_moments = [&](double s1, double s2) {
      if (gm < NGm) {
        red[gm * D + dm] = s1;
        red[nthr + gm * D + dm] = s2;
      }
      __syncthreads();
      if (tid < D) {
        double a1 = 0.0, a2 = 0.0;
        for (int k = 0; k < NGm; ++k) {
          a1 += red[k * D + tid];
          a2 += red[nthr + k * D + tid];
        }
        mean[tid] += a1 / (double)T;
        var[tid] = (a2 - a1 * a1 / (double)T) / (double)(T - 1);       // Julia's var: corrected
      }
      __syncthreads();
    };
    {
      double s1 = 0.0, s2 = 0.0;
      if (gm < NGm) {
        const double mu = mean[dm], sc = sqrt(gv.muv[dm] / var[dm]);
#pragma unroll 8
        for (int t = gm; t < T; t += NGm) {
          const size_t e = (size_t)t * D + dm;
          const int tm = t >= 1 ? t - 1 : t, tp = t + 1 < T ? t + 1 : t;
          const double yo = y[e], g0 = g[(size_t)t * D2 + dm], g1 = g[(size_t)tm * D2 + D + dm], g2 = g[(size_t)tp * D2 + D + dm];
          const double yn = sc * (yo - mu) + mu;
          y[e] = yn;
          R[e] = (g0 + (t >= 1 ? 0.5 : 0.0) * g1) - (t + 1 < T ? 0.5 : 0.0) * g2;
          const double dv = yn - mu;
          s1 += dv;
          s2 = fma(dv, dv, s2);
        }
      }
      __syncthreads();
      finish_moments(s1, s2);
    }

    double afr[kGvMaxKS];
    int mcur = -1;
    for (int ep = 0; ep < gv.epochs; ++ep) {
      // gvgrad coefficients, src/trajectory_gmmmap.jl:171-189: -2/T (pv' (var(y) - mu^v)), times (y - mean) below;
      // mean and var of the current y come from the pass that wrote it
      if (tid < D) {
        double s = 0.0;
        for (int j = 0; j < D; ++j) s = fma(gv.pv[j + (size_t)D * tid], var[j] - gv.muv[j], s);
        coef[tid] = -2.0 / (double)T * s;
      }
      // v_t = Q_mhat_t u_t for every frame: kGvNB single-mixture tiles of 16 frames per round, so that the gathers of
      // y, the loads of the Q fragments and the barriers are paid once per 16*kGvNB frames
      for (int tile0 = 0; tile0 < ntiles; tile0 += kGvNB) {
        const int nb = (ntiles - tile0 < kGvNB) ? ntiles - tile0 : kGvNB;
        if (tid < 16 * nb) tidx[tid] = perm[tile0 * 16 + tid];
        __syncthreads();
        // u_t = [y_t ; (y_{t+1} - y_{t-1})/2]: two unconditional loads per element on clamped addresses with 0 / 1 / +-1/2
        // weights (a branch or a select on the loaded value would serialise the loads), eight elements in flight
#pragma unroll 8
        for (int e = tid; e < nb * 4 * KS * 16; e += nthr) {
          const int b = e / (4 * KS * 16), q = e - b * (4 * KS * 16);
          const int k = q >> 4, t = tidx[b * 16 + (q & 15)];
          const bool ok = t >= 0 && k < D2, st = k < D;
          const int tc = t >= 0 ? t : 0, kd = st ? (k < D ? k : 0) : (k < D2 ? k - D : 0);
          const int tp = tc + 1 < T ? tc + 1 : tc, tm = tc >= 1 ? tc - 1 : tc;
          const double a = y[(size_t)(st ? tc : tp) * D + kd], c = y[(size_t)(st ? tc : tm) * D + kd];
          const double wa = !ok ? 0.0 : (st ? 1.0 : (tc + 1 < T ? 0.5 : 0.0)), wc = (!ok || st) ? 0.0 : (tc >= 1 ? -0.5 : 0.0);
          Ut[e] = wa * a + wc * c;
        }
        __syncthreads();
        for (int b = 0; b < nb; ++b) {
          const int m = (int)mh[tidx[b * 16]] - 1;         // the tile's mixture (slot 0 of a tile is never padding)
          if (m != mcur) {                                 // Q fragments of this wave's row tile: all k-steps in flight at
            mcur = m;                                      // once, kept in registers while consecutive tiles share m
            const double *A = Qfrag + (((size_t)m * NT + wave) * KS) * 64 + lane;
#pragma unroll
            for (int ks = 0; ks < kGvMaxKS; ++ks) afr[ks] = (ks < KS) ? A[(size_t)ks * 64] : 0.0;
          }
          const double *Ub = Ut + (size_t)b * 4 * KS * 16;
          gv_d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int ks = 0; ks < kGvMaxKS; ++ks)
            if (ks < KS) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(afr[ks], Ub[(4 * ks + lgrp) * 16 + lcol], acc, 0, 0, 0);
          const int t = tidx[b * 16 + lcol];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = 16 * wave + lgrp + 4 * r;
            if (row < D2 && t >= 0) V[(size_t)t * D2 + row] = acc[r];
          }
        }
        __syncthreads();
      }
      // y <- y + alpha * ( omega (r - P y) + coef (y - mean) ), eq. (52), src/trajectory_gmmmap.jl:163-166
      double s1 = 0.0, s2 = 0.0;
      if (gm < NGm) {
        const double mu = mean[dm], cf = coef[dm];
#pragma unroll 8
        for (int t = gm; t < T; t += NGm) {
          const size_t e = (size_t)t * D + dm;
          const int tm = t >= 1 ? t - 1 : t, tp = t + 1 < T ? t + 1 : t;
          const double vs = V[(size_t)t * D2 + dm], vm = V[(size_t)tm * D2 + D + dm], vp = V[(size_t)tp * D2 + D + dm];
          const double yy = y[e], rr = R[e];
          const double py = (vs + (t >= 1 ? 0.5 : 0.0) * vm) - (t + 1 < T ? 0.5 : 0.0) * vp;
          const double dy = omega * (rr - py) + cf * (yy - mu);
          const double yn = fma(gv.alpha, dy, yy);
          y[e] = yn;
          const double dv = yn - mu;
          s1 += dv;
          s2 = fma(dv, dv, s2);
        }
      }
      __syncthreads();
      finish_moments(s1, s2);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// traj_gv2_kernel: the same ascent with the product phase run by TWO teams of waves.  With one workgroup per CU the
// product rounds of traj_gv_kernel are a chain  gather (memory latency) -> barrier -> MFMA + store -> barrier;  here
// waves NT..2NT-1 gather the rows of y for round r+1 into the other of two LDS images of u while waves 0..NT-1 run the
// MFMAs of round r -- one barrier per round, and the streaming passes (scaling, update, moments) run on twice the
// threads.  The frame permutation and the tile mixtures live in LDS (no dependent global load in the rounds), which
// bounds the utterance length; longer utterances take traj_gv_kernel.
// ------------------------------------------------------------------------------------------------
static constexpr int kGv2NB = 4;     // 16-frame tiles per round
#ifndef VCMI_GV2_GB
#define VCMI_GV2_GB 12
#endif
static constexpr int kGv2GB = VCMI_GV2_GB;   // elements of u a gather thread has in flight
static constexpr int kGv2Threads = 768;   // 12 waves: NT = ceil(2D/16) MFMA waves, the rest gather (three waves per SIMD: 168 VGPRs)

__global__ void __launch_bounds__(kGv2Threads)
traj_gv2_kernel(const TrajUtt *__restrict__ utts, int n, int D, int M, int KS, int pcap, const double *__restrict__ Qfrag,
                const int64_t *__restrict__ mhat_all, const double *__restrict__ g_all, double *__restrict__ ws_all,
                int64_t ws_stride, TrajGV gv) {
  extern __shared__ double gsm[];
  const int D2 = 2 * D, nthr = blockDim.x, NT = (D2 + 15) / 16, nmf = 64 * NT, ngth = nthr - nmf;
  const int UTS = 4 * KS * 17, UTR = kGv2NB * UTS;
  double *Ut = gsm;                        // [2][kGv2NB][4*KS][17]  u of the tiles' frames, k-major, row stride 17
  double *red = Ut + 2 * (size_t)UTR;      // [2][nthr]
  double *mean = red + 2 * nthr;           // [D]
  double *var = mean + D;                  // [D]
  double *coef = var + D;                  // [D]
  int *cnt = reinterpret_cast<int *>(coef + D);   // [M] frames per mixture, then the fill cursor
  int *start = cnt + M;                           // [M] first slot of the mixture's (16-padded) segment
  int *perm = start + M;                          // [pcap] frames grouped by mixture, segments padded to 16 with -1
  int *tilem = perm + pcap;                       // [pcap / 16 + 1] mixture of every tile
  __shared__ int ntiles_s;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lcol = lane & 15, lgrp = lane >> 4;
  const bool gatherer = wave >= NT;
  const int gtid = tid - nmf;

  for (int u = blockIdx.x; u < n; u += gridDim.x) {
    const TrajUtt U = utts[u];
    const int T = U.T;
    if (T < 2) continue;                   // var() of one frame is undefined; the host rejects such calls
    const int64_t *mh = mhat_all + U.frame0;
    const double *g = g_all + U.frame0 * D2;
    gdouble *y = (gdouble *)U.Y;
    double *V = ws_all + (size_t)blockIdx.x * ws_stride;   // [T][2D]
    double *R = V + (size_t)T * D2;                        // [T][D]   r = W' D^-1 E
    const double omega = 1.0 / (2.0 * (double)T);

    // frames grouped by mixture (see traj_gv_kernel)
    for (int m = tid; m < M; m += nthr) cnt[m] = 0;
    __syncthreads();
    for (int t = tid; t < T; t += nthr) atomicAdd(&cnt[(int)mh[t] - 1], 1);
    __syncthreads();
    if (tid == 0) {
      int pos = 0;
      for (int m = 0; m < M; ++m) {
        start[m] = pos;
        pos += (cnt[m] + 15) / 16 * 16;
        cnt[m] = 0;
      }
      ntiles_s = pos / 16;
    }
    __syncthreads();
    const int ntiles = ntiles_s;
    for (int e = tid; e < ntiles * 16; e += nthr) perm[e] = -1;
    __syncthreads();
    for (int t = tid; t < T; t += nthr) {
      const int m = (int)mh[t] - 1;
      perm[start[m] + atomicAdd(&cnt[m], 1)] = t;
    }
    __syncthreads();
    for (int i = tid; i < ntiles; i += nthr) tilem[i] = (int)mh[perm[i * 16]] - 1;
    for (int e = tid; e < 2 * UTR; e += nthr) Ut[e] = 0.0;      // rows k >= 2D of the k-padding stay zero
    // elements of u a gather thread fetches in a round: e = gtid + i * ngth -> (frame slot e / 2D, k = e % 2D), k fastest
    // across threads; one division here, increments in the rounds
    const int gsl0 = gatherer ? gtid / D2 : 0, gk0 = gatherer ? gtid % D2 : 0, dsl = ngth / D2, dk = ngth % D2;

    // eq. (58) and r; every pass that writes y accumulates the moments of what it writes (see traj_gv_kernel)
    gv_moments(y, D, T, nthr, red, mean, var);
    const int NGm = nthr / D, dm = tid % D, gm = tid / D;
    auto finish_moments = [&](double s1, double s2) {
      if (gm < NGm) {
        red[gm * D + dm] = s1;
        red[nthr + gm * D + dm] = s2;
      }
      __syncthreads();
      if (tid < D) {
        double a1 = 0.0, a2 = 0.0;
        for (int k = 0; k < NGm; ++k) {
          a1 += red[k * D + tid];
          a2 += red[nthr + k * D + tid];
        }
        mean[tid] += a1 / (double)T;
        var[tid] = (a2 - a1 * a1 / (double)T) / (double)(T - 1);       // Julia's var: corrected
      }
      __syncthreads();
    };
    {
      double s1 = 0.0, s2 = 0.0;
      if (gm < NGm) {
        const double mu = mean[dm], sc = sqrt(gv.muv[dm] / var[dm]);
#pragma unroll 4
        for (int t = gm; t < T; t += NGm) {
          const size_t e = (size_t)t * D + dm;
          const int tm = t >= 1 ? t - 1 : t, tp = t + 1 < T ? t + 1 : t;
          const double yo = y[e], g0 = g[(size_t)t * D2 + dm], g1 = g[(size_t)tm * D2 + D + dm], g2 = g[(size_t)tp * D2 + D + dm];
          const double yn = sc * (yo - mu) + mu;
          y[e] = yn;
          R[e] = (g0 + (t >= 1 ? 0.5 : 0.0) * g1) - (t + 1 < T ? 0.5 : 0.0) * g2;
          const double dv = yn - mu;
          s1 += dv;
          s2 = fma(dv, dv, s2);
        }
      }
      __syncthreads();
      finish_moments(s1, s2);
    }

    // gather of one round into an LDS image of u: two unconditional loads per element on clamped addresses, eight
    // elements in flight, 0 / 1 / +-1/2 weights applied on the way into LDS
    auto gather = [&](int r, double *Ub0) {
      const int tile0 = r * kGv2NB, total = 16 * kGv2NB * D2;
      int sl = gsl0, k = gk0;
      for (int e = gtid; e < total; e += kGv2GB * ngth) {
        double ya[kGv2GB], yc[kGv2GB];
        int tt[kGv2GB], oo[kGv2GB];
#pragma unroll
        for (int i = 0; i < kGv2GB; ++i) {
          const bool in = e + i * ngth < total;
          const int idx = tile0 * 16 + sl;
          const int t = (in && idx < ntiles * 16) ? perm[idx] : -1;
          tt[i] = (t < 0 ? -1 : t) | (k < D ? 0 : 1 << 30);         // frame and static / delta half
          oo[i] = in ? ((sl >> 4) * 4 * KS + k) * 17 + (sl & 15) : -1;
          const bool st = k < D;
          const int tc = t >= 0 ? t : 0, kd = st ? k : k - D;
          const int tp = tc + 1 < T ? tc + 1 : tc, tm = tc >= 1 ? tc - 1 : tc;
          ya[i] = y[(size_t)(st ? tc : tp) * D + kd];
          yc[i] = y[(size_t)(st ? tc : tm) * D + kd];
          sl += dsl;
          k += dk;
          if (k >= D2) {
            k -= D2;
            ++sl;
          }
        }
#pragma unroll
        for (int i = 0; i < kGv2GB; ++i)
          if (oo[i] >= 0) {
            const bool ok = tt[i] >= 0, st = (tt[i] & (1 << 30)) == 0;
            const int t = tt[i] & ~(1 << 30);
            const double wa = !ok ? 0.0 : (st ? 1.0 : (t + 1 < T ? 0.5 : 0.0)), wc = (!ok || st) ? 0.0 : (t >= 1 ? -0.5 : 0.0);
            Ub0[oo[i]] = wa * ya[i] + wc * yc[i];
          }
      }
    };

    double afr[kGvMaxKS];
    int mcur = -1;
    const int nrounds = (ntiles + kGv2NB - 1) / kGv2NB;
    BLK_PROF_T0();
    for (int ep = 0; ep < gv.epochs; ++ep) {
      // gvgrad coefficients, src/trajectory_gmmmap.jl:171-189: -2/T (pv' (var(y) - mu^v)), times (y - mean) below
      if (tid < D) {
        double s = 0.0;
        for (int j = 0; j < D; ++j) s = fma(gv.pv[j + (size_t)D * tid], var[j] - gv.muv[j], s);
        coef[tid] = -2.0 / (double)T * s;
      }
      // v_t = Q_mhat_t u_t for every frame
      if (gatherer) gather(0, Ut);
      __syncthreads();
      for (int r = 0; r < nrounds; ++r) {
        if (gatherer) {
          if (r + 1 < nrounds) gather(r + 1, Ut + (size_t)((r + 1) & 1) * UTR);
        } else {
          const double *Ub0 = Ut + (size_t)(r & 1) * UTR;
          const int tile0 = r * kGv2NB, nb = (ntiles - tile0 < kGv2NB) ? ntiles - tile0 : kGv2NB;
          for (int b = 0; b < nb; ++b) {
            const int m = tilem[tile0 + b];
            if (m != mcur) {                               // Q fragments of this wave's row tile, kept while tiles share m
              mcur = m;
              const double *A = Qfrag + (((size_t)m * NT + wave) * KS) * 64 + lane;
#pragma unroll
              for (int ks = 0; ks < kGvMaxKS; ++ks) afr[ks] = (ks < KS) ? A[(size_t)ks * 64] : 0.0;
            }
            const double *Ub = Ub0 + (size_t)b * UTS;
            gv_d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int ks = 0; ks < kGvMaxKS; ++ks)
              if (ks < KS) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(afr[ks], Ub[(4 * ks + lgrp) * 17 + lcol], acc, 0, 0, 0);
            const int t = perm[(tile0 + b) * 16 + lcol];
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
              const int row = 16 * wave + lgrp + 4 * r4;
              if (row < D2 && t >= 0) V[(size_t)t * D2 + row] = acc[r4];
            }
          }
        }
        __syncthreads();
      }
      BLK_PROF(13);
      // y <- y + alpha * ( omega (r - P y) + coef (y - mean) ), eq. (52), src/trajectory_gmmmap.jl:163-166
      double s1 = 0.0, s2 = 0.0;
      if (gm < NGm) {
        const double mu = mean[dm], cf = coef[dm];
#pragma unroll 4
        for (int t = gm; t < T; t += NGm) {
          const size_t e = (size_t)t * D + dm;
          const int tm = t >= 1 ? t - 1 : t, tp = t + 1 < T ? t + 1 : t;
          const double vs = V[(size_t)t * D2 + dm], vm = V[(size_t)tm * D2 + D + dm], vp = V[(size_t)tp * D2 + D + dm];
          const double yy = y[e], rr = R[e];
          const double py = (vs + (t >= 1 ? 0.5 : 0.0) * vm) - (t + 1 < T ? 0.5 : 0.0) * vp;
          const double dy = omega * (rr - py) + cf * (yy - mu);
          const double yn = fma(gv.alpha, dy, yy);
          y[e] = yn;
          const double dv = yn - mu;
          s1 += dv;
          s2 = fma(dv, dv, s2);
        }
      }
      __syncthreads();
      BLK_PROF(14);
      finish_moments(s1, s2);
      BLK_PROF(15);
    }
  }
}

// g (frames x [gs (D); gd (D)]) -> gpad (frames x [gs (Dp); gd (Dp)]), zeros in the padding
__global__ void __launch_bounds__(256)
traj_pad_g_kernel(const double *__restrict__ g, int64_t nframes, int D, int Dp, double *__restrict__ gpad) {
  const int64_t n = nframes * 2 * Dp;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
    const int64_t f = e / (2 * Dp);
    const int c = (int)(e - f * 2 * Dp), half = c / Dp, d = c - half * Dp;
    gpad[e] = (d < D) ? g[f * 2 * D + half * D + d] : 0.0;
  }
}
// ypad (frames x Dp) -> the utterances' own (T, D) outputs
__global__ void __launch_bounds__(256)
traj_unpad_y_kernel(const TrajUtt *__restrict__ utts, const double *__restrict__ ypad, int D, int Dp) {
  const TrajUtt u = utts[blockIdx.x];
  const int64_t n = (int64_t)u.T * D;
  for (int64_t e = (int64_t)blockIdx.y * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.y * 256) {
    const int64_t tt = e / D;
    u.Y[e] = ypad[(u.frame0 + tt) * Dp + (e - tt * D)];
  }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
// the blocked solver's instantiations (traj_run) and the one a static dimension without its own runs in (0: none)
static bool traj_blk_has(int D) {
  switch (D) {
    case 12: case 16: case 20: case 24: case 25: case 30: case 32: case 40: case 46: return true;
    default: return false;
  }
}
static int traj_blk_padded_dim(int D) {
  if (traj_blk_has(D) || D > 46) return 0;      // (46: the largest static dimension whose window -- 47 rows in three 16-row
                                                // tiles -- fits the LDS; D = 47 would put the rhs row into a fourth tile)
  for (int d = D + 1; d <= 46; ++d)
    if (traj_blk_has(d)) return d;
  return 0;
}

// ------------------------------------------------------------------------------------------------
// Any static dimension (D >= 47: the reference has no limit, src/trajectory_gmmmap.jl:65-110): the algorithm of
// traj_solve_kernel with its 3D x 3D window, right-hand side, pivot column and the vectors of the back substitution in HBM
// (a per-workgroup scratch, L2-resident) instead of LDS -- a fallback for completeness, not a fast path: every
// __syncthreads also orders the workgroup's global accesses.  Window shift through a second buffer; the triangular solve of
// the back substitution column by column across the workgroup.
// ------------------------------------------------------------------------------------------------
static size_t traj_big_win_doubles(int D) {
  const size_t W3 = 3 * (size_t)D, D2 = 2 * (size_t)D;
  return W3 * (W3 + 1) + 2 * W3 + 2 * D + 2 * D + D2 * D2 + D2 + 64;
}

// PK = true (47 <= D <= 64): the window's LOWER TRIANGLE in packed storage, (i, j <= i) at i (i + 1) / 2 + j, fits LDS
// (148 KB at D = 64) together with the small vectors; only the shift buffer and the panels stay in HBM.  PK = false: everything
// in the per-workgroup HBM scratch.
static size_t traj_big_lds_bytes(int D) {
  const size_t W3 = 3 * (size_t)D;
  return (W3 * (W3 + 1) / 2 + 2 * W3 + 2 * D + 2 * D + 8) * sizeof(double);
}

template <bool PK>
__global__ void __launch_bounds__(256)
traj_solve_big_kernel(const TrajUtt *__restrict__ utts, int n, int D, const double *__restrict__ Qall,
                      const int64_t *__restrict__ mhat_all, const double *__restrict__ g_all, double *__restrict__ ws_all,
                      int64_t ws_stride, int *__restrict__ status, double *__restrict__ gwin_all, int64_t gwin_stride) {
  const int D2 = 2 * D, W3 = 3 * D, LD = W3 + 1;
  extern __shared__ double sm_big[];
  double *gw = gwin_all + (size_t)blockIdx.x * gwin_stride;
  double *Wd = PK ? sm_big : gw;                                   // the window: packed lower triangle (LDS) or [W3][LD] (HBM)
  double *vec = PK ? sm_big + (size_t)W3 * (W3 + 1) / 2 : gw + (size_t)W3 * LD;
  double *rr = vec;                      // [W3]
  double *lcol = rr + W3;                // [W3]
  double *yring = lcol + W3;             // [2][D]
  double *wv = yring + 2 * D;            // [D]
  double *rdiag = wv + D;                // [D]
  double *tmp = PK ? gw : rdiag + D;     // [D2][D2] + [D2] in HBM: the shifted part of the window on its way up-left
  auto W = [&](int i, int j) -> double & { return PK ? Wd[(size_t)i * (i + 1) / 2 + j] : Wd[(size_t)i * LD + j]; };
  __shared__ int bad;
  __shared__ double zc_s;
  const int tid = threadIdx.x;
  const int ti = tid >> 4, tj = tid & 15;
  const size_t PAN = (size_t)(W3 + 1) * D;

  for (int u = blockIdx.x; u < n; u += gridDim.x) {
    const TrajUtt U = utts[u];
    const int T = U.T;
    if (T == 0) continue;
    const int64_t *mh = mhat_all + U.frame0;
    const double *g = g_all + U.frame0 * D2;
    double *ws = ws_all + (size_t)blockIdx.x * ws_stride;
    if (tid == 0) bad = 0;
    // block row a of P (blocks (a,a-2), (a,a-1), (a,a): traj_add_block_row's terms) into window block row la; lower triangle only
    auto add_block_row = [&](int a, int la) {
      const double *Qa = Qall + (size_t)(mh[a] - 1) * D2 * D2;
      const double *Qm = (a >= 1) ? Qall + (size_t)(mh[a - 1] - 1) * D2 * D2 : nullptr;
      const double *Qp = (a + 1 < T) ? Qall + (size_t)(mh[a + 1] - 1) * D2 * D2 : nullptr;
      for (int e = tid; e < D * D; e += 256) {
        const int i = e / D, j = e - i * D;
        if (j <= i) {
          double v = Qa[(size_t)i * D2 + j];
          if (Qm) v += 0.25 * Qm[(size_t)(D + i) * D2 + (D + j)];
          if (Qp) v += 0.25 * Qp[(size_t)(D + i) * D2 + (D + j)];
          W(la * D + i, la * D + j) = v;
        }
        if (la >= 1 && a >= 1) W(la * D + i, (la - 1) * D + j) = 0.5 * Qm[(size_t)(D + i) * D2 + j] - 0.5 * Qa[(size_t)i * D2 + (D + j)];
        if (la >= 2 && a >= 2) W(la * D + i, (la - 2) * D + j) = -0.25 * Qm[(size_t)(D + i) * D2 + (D + j)];
      }
      for (int k = tid; k < D; k += 256) {
        double v = g[(size_t)a * D2 + k];
        if (a >= 1) v += 0.5 * g[(size_t)(a - 1) * D2 + D + k];
        if (a + 1 < T) v -= 0.5 * g[(size_t)(a + 1) * D2 + D + k];
        rr[la * D + k] = v;
      }
    };
    for (int a = 0; a < 3 && a < T; ++a) add_block_row(a, a);
    __syncthreads();
    for (int t = 0; t < T; ++t) {
      const int nb = (T - t < 3) ? T - t : 3;
      const int nrows = nb * D;
      for (int c = 0; c < D; ++c) {
        const double piv = W(c, c);
        if (!(piv > 0.0) && tid == 0) bad = 1;
        const double dinv = traj_rsqrt(piv);
        for (int lr = c + tid; lr < nrows; lr += 256) lcol[lr] = (lr == c) ? piv * dinv : W(lr, c) * dinv;
        if (tid == 255) zc_s = rr[c] * dinv;
        __syncthreads();
        const int rem = nrows - c - 1;
        for (int a = ti; a < rem; a += 16) {
          const int ri = c + 1 + a;
          const double lic = lcol[ri];
          double *row = &W(ri, c + 1);
          for (int b = tj; b <= a; b += 16) row[b] = fma(-lic, lcol[c + 1 + b], row[b]);
        }
        const double zc = zc_s;
        for (int lr = c + tid; lr < nrows; lr += 256) {
          W(lr, c) = lcol[lr];
          if (lr > c) rr[lr] = fma(-zc, lcol[lr], rr[lr]);
          else rr[lr] = zc;
        }
        __syncthreads();
      }
      double *pan = ws + (size_t)t * PAN;
      for (int e = tid; e < W3 * D; e += 256) {
        const int lr = e / D, cc = e - lr * D;
        pan[e] = (lr < nrows && cc <= lr) ? W(lr, cc) : 0.0;
      }
      for (int cc = tid; cc < D; cc += 256) pan[(size_t)W3 * D + cc] = rr[cc];
      // the lower-right 2D x 2D part (its lower triangle) moves up-left by D
      for (int e = tid; e < D2 * D2; e += 256) {
        const int i = e / D2, j = e - i * D2;
        if (j <= i) tmp[e] = W(i + D, j + D);
      }
      for (int k = tid; k < D2; k += 256) tmp[(size_t)D2 * D2 + k] = rr[k + D];
      __syncthreads();
      for (int e = tid; e < D2 * D2; e += 256) {
        const int i = e / D2, j = e - i * D2;
        if (j <= i) W(i, j) = tmp[e];
      }
      for (int k = tid; k < D2; k += 256) rr[k] = tmp[(size_t)D2 * D2 + k];
      __syncthreads();
      if (t + 3 < T) add_block_row(t + 3, 2);
      __syncthreads();
    }
    // ---------------- back substitution: y_t = Dg'^-1 (z - E' y_{t+1} - F' y_{t+2}) from the panels ----------------
    for (int i = tid; i < 2 * D; i += 256) yring[i] = 0.0;
    __syncthreads();
    for (int t = T - 1; t >= 0; --t) {
      const double *pb = ws + (size_t)t * PAN;
      if (PK) {
        // the panel into the (now free) window area first, by all threads: the sums and the column-by-column solve below then
        // read LDS instead of walking HBM with one dependent load after the other (PAN <= W3 (W3 + 1) / 2 for every D)
        for (size_t e = tid; e < PAN; e += 256) sm_big[e] = pb[e];
        __syncthreads();
        pb = sm_big;
      }
      double *y1 = yring + ((t + 1) & 1) * D, *y2 = yring + (t & 1) * D;
      for (int j = tid; j < D; j += 256) {
        double sacc = pb[(size_t)W3 * D + j];
        for (int i = 0; i < D; ++i) sacc = fma(-pb[(size_t)(D + i) * D + j], y1[i], sacc);
        for (int i = 0; i < D; ++i) sacc = fma(-pb[(size_t)(2 * D + i) * D + j], y2[i], sacc);
        wv[j] = sacc;
        rdiag[j] = 1.0 / pb[(size_t)j * D + j];
      }
      for (int k = D - 1; k >= 0; --k) {
        __syncthreads();
        const double yk = wv[k] * rdiag[k];
        for (int j = tid; j < k; j += 256) wv[j] = fma(-pb[(size_t)k * D + j], yk, wv[j]);
        if (tid == 0) {
          lcol[k] = yk;                      // (lcol is free during the back substitution)
          U.Y[(size_t)t * D + k] = yk;
        }
      }
      __syncthreads();
      for (int j = tid; j < D; j += 256) y2[j] = lcol[j];
      __syncthreads();
    }
    if (tid == 0 && bad) status[0] = 1;
    __syncthreads();
  }
}

static size_t solve_lds_bytes(int D) {
  const size_t W3 = 3 * (size_t)D;
  const size_t NK = (W3 + 15) / 16;
  return (W3 * (W3 + 1) + 2 * W3 + 2 * (NK * 16 + 2) + 2 * D + D) * sizeof(double);   // covers both solve kernels
}

static int traj_run(vcmi_traj *t, std::vector<TrajUtt> &utts, int64_t nframes, bool contiguous, const double *dX0,
                    hipStream_t st, const TrajGV *gv = nullptr) {
  const int n = (int)utts.size();
  if (n == 0 || nframes == 0) return VCMI_OK;
  const int D = t->D, D2 = t->D2;
  VCMI_TRY(t->mhat.reserve((size_t)nframes));
  VCMI_TRY(t->gbuf.reserve((size_t)nframes * D2));
  // (1) mhat = predict(g.px, X), src/trajectory_gmmmap.jl:82
  const bool g_mfma = t->NT <= 6 && !debug_flag(kDbgTrajGScalar);   // g_t on MFMA tiles (one workgroup per utterance)
  if (contiguous) {
    VCMI_TRY(gmmmap_predict_device(t->g, dX0, D2, nframes, t->mhat.p, st, /*allow_screen=*/false));
    if (!g_mfma)
      hipLaunchKernelGGL(traj_g_kernel, dim3((unsigned)nframes), dim3(128), 2 * D2 * sizeof(double), st, dX0, nframes, D2,
                         t->mhat.p, t->AT.p, t->QT.p, t->bvec.p, t->gbuf.p);
  } else {
    for (auto &u : utts) {
      if (u.T == 0) continue;
      VCMI_TRY(gmmmap_predict_device(t->g, u.X, D2, u.T, t->mhat.p + u.frame0, st, /*allow_screen=*/false));
      if (!g_mfma)
        hipLaunchKernelGGL(traj_g_kernel, dim3((unsigned)u.T), dim3(128), 2 * D2 * sizeof(double), st, u.X, (int64_t)u.T, D2,
                           t->mhat.p + u.frame0, t->AT.p, t->QT.p, t->bvec.p, t->gbuf.p + (size_t)u.frame0 * D2);
    }
  }
  VCMI_HIP(hipGetLastError());
  // (2) banded solve
  int Tmax = 0;
  for (auto &u : utts) Tmax = std::max(Tmax, (int)u.T);
  int dev = 0, cus = 256;
  (void)hipGetDevice(&dev);
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  const int grid = std::min(n, cus);
  int64_t ws_stride = (int64_t)Tmax * (3 * D + 1) * D;
  if (gv) ws_stride = std::max<int64_t>(ws_stride, (int64_t)Tmax * 3 * D + (17 * (int64_t)Tmax + 1) / 2 + 16);   // V, r, perm
  VCMI_TRY(t->ws.reserve((size_t)grid * ws_stride));
  VCMI_TRY(t->status.reserve(1));
  VCMI_TRY(t->uttbuf.reserve(sizeof(TrajUtt) * n));
  VCMI_HIP(hipMemsetAsync(t->status.p, 0, sizeof(int), st));
  // longest utterances first
  std::stable_sort(utts.begin(), utts.end(), [](const TrajUtt &a, const TrajUtt &b) { return a.T > b.T; });
  VCMI_TRY(upload_now(t->uttbuf.p, utts.data(), sizeof(TrajUtt) * n));
  const size_t shmem = solve_lds_bytes(D);
  const TrajUtt *du = reinterpret_cast<const TrajUtt *>(t->uttbuf.p);
  if (g_mfma) {
    VCMI_TRY(t->gperm.reserve((size_t)nframes + (size_t)16 * t->M * n));
    const int nthr = 64 * t->NT;
    const size_t shg = ((size_t)4 * t->KS * 16 + (size_t)4 * t->NT * 64) * sizeof(double) + (2 + (size_t)t->NT) * (size_t)t->M * sizeof(int);
    VCMI_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(traj_g_mfma_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)shg));
    // workgroups per utterance: only while the batch leaves CUs idle (one utterance, or the twenty 100-frame chunks of one:
    // 328 -> ~70 us for 100 frames).  A batch that fills the chip gains nothing -- 256 x 2000 frames: 0.93 / 0.96 / 0.98 /
    // 1.05 ms with 1 / 2 / 4 / 8 members: there the kernel is bound by the A / Q fragments every change of mixture reloads
    // from L2 (102 KB each), not by the latency of a tile.
    int split = 1;
    while (split < 8 && (int64_t)n * 2 * split <= (int64_t)cus) split *= 2;
    hipLaunchKernelGGL(traj_g_mfma_kernel, dim3((unsigned)std::min<int64_t>((int64_t)n * split, (int64_t)4 * cus / split * split)),
                       dim3(nthr), shg, st, du, n, D2, t->M, t->KS, t->Afrag.p, t->Qfrag.p, t->bvec.p, t->mhat.p, t->gperm.p, t->gbuf.p, split);
    VCMI_HIP(hipGetLastError());
  }
  bool launched = false;
  // the dimension the blocked solver runs in, and its operands: the utterances' own, or the padded copies
  const int Ds = t->Dpad ? t->Dpad : D;
  const double *Qs = t->Q.p, *gs = t->gbuf.p;
  const TrajUtt *dus = du;
  int64_t ws_stride_s = ws_stride;
  if (t->Dpad && !debug_flag(kDbgTrajGeneric)) {
    const int Dp = t->Dpad;
    VCMI_TRY(t->gpad.reserve((size_t)nframes * 2 * Dp));
    VCMI_TRY(t->ypad.reserve((size_t)nframes * Dp));
    VCMI_TRY(t->uttpad.reserve(sizeof(TrajUtt) * n));
    std::vector<TrajUtt> up(utts);
    for (auto &u : up) u.Y = t->ypad.p + (size_t)u.frame0 * Dp;
    VCMI_TRY(upload_now(t->uttpad.p, up.data(), sizeof(TrajUtt) * n));
    ws_stride_s = (int64_t)Tmax * (3 * Dp + 1) * Dp;
    VCMI_TRY(t->ws.reserve((size_t)grid * std::max(ws_stride, ws_stride_s)));
    hipLaunchKernelGGL(traj_pad_g_kernel, dim3((unsigned)std::min<int64_t>((nframes * 2 * Dp + 255) / 256, 4096)), dim3(256), 0, st,
                       t->gbuf.p, nframes, D, Dp, t->gpad.p);
    VCMI_HIP(hipGetLastError());
    Qs = t->Qpad.p;
    gs = t->gpad.p;
    dus = reinterpret_cast<const TrajUtt *>(t->uttpad.p);
  }
  if (!debug_flag(kDbgTrajGeneric)) {
    switch (Ds) {
#define VCMI_TRAJ_BLK_CASE(DV)                                                                                      \
  case DV: {                                                                                                        \
    auto kern = traj_solve_blk_kernel<DV>;                                                                          \
    const size_t shb = BlkCfg<DV>::lds_doubles * sizeof(double);                                                    \
    VCMI_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,  \
                                 (int)shb));                                                                        \
    /* as many workgroups as the device holds at once: two per CU where LDS and registers allow (static D <= 30) */ \
    int occ = 1;                                                                                                    \
    if (debug_flag(kDbgTrajOneWgPerCu) ||                                                                           \
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, blk_threads<DV>(), shb) != hipSuccess || occ < 1)        \
      occ = 1;                                                                                                      \
    const int grid_blk = (int)std::min<int64_t>(n, (int64_t)cus * occ);                                             \
    VCMI_TRY(t->ws.reserve((size_t)grid_blk * std::max(ws_stride, ws_stride_s) + 256)); /* + slack: whole-KB reads */ \
    if (blk_fused_backsub<DV>()) {                                                                                  \
      hipLaunchKernelGGL(kern, dim3(grid_blk), dim3(blk_threads<DV>()), shb, st, dus, n, Qs, t->mhat.p, gs, t->ws.p,  \
                         ws_stride_s, t->status.p);                                                                 \
    } else {                                                                                                        \
      /* eight waves: factorisation and back substitution are two kernels, per batch of grid_blk utterances */       \
      auto kb = traj_backsub_blk_kernel<DV>;                                                                        \
      const size_t shs = blk_backsub_lds_bytes<DV>();                                                               \
      VCMI_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kb), hipFuncAttributeMaxDynamicSharedMemorySize,  \
                                   (int)shs));                                                                      \
      for (int b0 = 0; b0 < n; b0 += grid_blk) {                                                                    \
        const int nb = std::min(grid_blk, n - b0);                                                                  \
        hipLaunchKernelGGL(kern, dim3(nb), dim3(blk_threads<DV>()), shb, st, dus + b0, nb, Qs, t->mhat.p, gs,        \
                           t->ws.p, ws_stride_s, t->status.p);                                                      \
        hipLaunchKernelGGL(kb, dim3(nb), dim3(BacksubCfg<DV>::THREADS), shs, st, dus + b0, nb, t->ws.p, ws_stride_s);                    \
      }                                                                                                             \
    }                                                                                                               \
    launched = true;                                                                                                \
  } break;
      VCMI_TRAJ_BLK_CASE(12) VCMI_TRAJ_BLK_CASE(16) VCMI_TRAJ_BLK_CASE(20) VCMI_TRAJ_BLK_CASE(24) VCMI_TRAJ_BLK_CASE(25)
      VCMI_TRAJ_BLK_CASE(30) VCMI_TRAJ_BLK_CASE(32) VCMI_TRAJ_BLK_CASE(40) VCMI_TRAJ_BLK_CASE(46)
#undef VCMI_TRAJ_BLK_CASE
      default: break;
    }
  }
  if (launched && t->Dpad) {
    hipLaunchKernelGGL(traj_unpad_y_kernel, dim3(n, 8), dim3(256), 0, st, du, t->ypad.p, D, t->Dpad);
    VCMI_HIP(hipGetLastError());
  }
  if (!launched && t->big) {
    const int64_t gstride = (int64_t)traj_big_win_doubles(D);
    VCMI_TRY(t->gwin.reserve((size_t)grid * gstride));
    const size_t lds_pk = traj_big_lds_bytes(D);
    if (lds_pk <= 160 * 1024 - 256) {       // D <= 64: the window's lower triangle in LDS
      VCMI_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(traj_solve_big_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)lds_pk));
      hipLaunchKernelGGL(traj_solve_big_kernel<true>, dim3(grid), dim3(256), lds_pk, st, du, n, D, t->Q.p, t->mhat.p, t->gbuf.p, t->ws.p,
                         ws_stride, t->status.p, t->gwin.p, gstride);
    } else {
      hipLaunchKernelGGL(traj_solve_big_kernel<false>, dim3(grid), dim3(256), 0, st, du, n, D, t->Q.p, t->mhat.p, t->gbuf.p, t->ws.p,
                         ws_stride, t->status.p, t->gwin.p, gstride);
    }
  } else if (!launched) {
    VCMI_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(traj_solve_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)shmem));
    hipLaunchKernelGGL(traj_solve_kernel, dim3(grid), dim3(256), shmem, st, du, n, D, t->Q.p, t->mhat.p, t->gbuf.p, t->ws.p,
                       ws_stride, t->status.p);
  }
  VCMI_HIP(hipGetLastError());
  if (gv && gv->epochs >= 0) {
    // (3) global-variance ascent on the solved trajectories, in place; workspace: V (2D,T) + r (D,T) <= the panel area
    // two-team kernel when the frame permutation of the longest utterance fits in LDS beside the two images of u
    const int pcap = ((Tmax + 15) / 16 + t->M) * 16;
    const int nthr2 = kGv2Threads;
    const size_t shmem2 = ((size_t)2 * kGv2NB * 4 * t->KS * 17 + 2 * nthr2 + 3 * (size_t)D) * sizeof(double) +
                          (2 * (size_t)t->M + (size_t)pcap + (size_t)pcap / 16 + 2) * sizeof(int);
    if (shmem2 <= 160 * 1024 - 256 && t->NT <= 6 && !debug_flag(kDbgGvOneTeam)) {
      VCMI_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(traj_gv2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)shmem2));
      hipLaunchKernelGGL(traj_gv2_kernel, dim3(grid), dim3(nthr2), shmem2, st, du, n, D, t->M, t->KS, pcap, t->Qfrag.p, t->mhat.p,
                         t->gbuf.p, t->ws.p, ws_stride, *gv);
    } else {
      const int nthr = 64 * t->NT;
      const size_t shmem = ((size_t)kGvNB * 4 * t->KS * 16 + 2 * nthr + 3 * (size_t)D) * sizeof(double) + 2 * (size_t)t->M * sizeof(int);
      VCMI_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(traj_gv_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)shmem));
      hipLaunchKernelGGL(traj_gv_kernel, dim3(grid), dim3(nthr), shmem, st, du, n, D, t->M, t->KS, t->Qfrag.p, t->mhat.p, t->gbuf.p,
                         t->ws.p, ws_stride, *gv);
    }
    VCMI_HIP(hipGetLastError());
  }
  return VCMI_OK;
}

static int traj_check_status(vcmi_traj *t, hipStream_t st) {
#ifdef TRAJ_BLK_PROF
  {
    long long h[32], z[32] = {0};
    (void)hipStreamSynchronize(st);
    (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(blk_prof), sizeof(h));
    (void)hipMemcpyToSymbol(HIP_SYMBOL(blk_prof), z, sizeof(z));
    fprintf(stderr, "blk_prof cycles: phase1 %lld trsm %lld update %lld backsub %lld | pivot S done at %lld, pivot U at %lld, deferred: S21/S22 at %lld, panel+assembly at %lld\n",
            h[0], h[1], h[2], h[5], h[3], h[4], h[6], h[7]);
    fprintf(stderr, "   gv kernel: product phase %lld, update %lld, moments %lld\n", h[13], h[14], h[15]);
    fprintf(stderr, "   deferred wave 2: L20 done %lld, S21/S22 done %lld, loads issued %lld, panel stored %lld, combined %lld | wave 0 jobs done %lld, wave 1 jobs done %lld | wave 3: S21/S22 done %lld, combined %lld\n", h[8],
            h[6], h[10], h[11], h[7], h[9], h[12], h[14], h[13]);
    fprintf(stderr, "   deferred waves 2..7: S21/S22 jobs done %lld %lld %lld %lld %lld %lld | at the barrier %lld %lld %lld %lld %lld %lld\n", h[26], h[27], h[28],
            h[29], h[30], h[31], h[18], h[19], h[20], h[21], h[22], h[23]);
  }
#endif
  int h = 0;
  VCMI_HIP(hipMemcpyAsync(&h, t->status.p, sizeof(int), hipMemcpyDeviceToHost, st));
  VCMI_HIP(hipStreamSynchronize(st));
  if (h) return fail(VCMI_ERR_NOT_PD, "trajectory normal matrix W'D^-1W is not positive definite");
  return VCMI_OK;
}

// host-pointer batch on the current device: the utterances are gathered straight into the pinned staging slots, run,
// and scattered back the same way
static int traj_host_batch_local(vcmi_traj *t, int64_t n, const double *const *X, const int64_t *T, double *const *Y,
                                 const TrajGV *gv) {
  int64_t nframes = 0;
  for (int64_t u = 0; u < n; ++u) nframes += T[u];
  if (nframes == 0) return VCMI_OK;
  const int D = t->D, D2 = t->D2;
  VCMI_TRY(t->xbuf.reserve((size_t)nframes * D2));
  VCMI_TRY(t->ybuf.reserve((size_t)nframes * D));
  std::vector<TrajUtt> utts(n);
  std::vector<HostPiece> up, down;
  int64_t f0 = 0;
  for (int64_t u = 0; u < n; ++u) {
    if (T[u] > 0) {
      up.push_back(HostPiece{const_cast<double *>(X[u]), sizeof(double) * D2 * T[u]});
      down.push_back(HostPiece{Y[u], sizeof(double) * D * T[u]});
    }
    utts[u] = TrajUtt{t->xbuf.p + (size_t)f0 * D2, t->ybuf.p + (size_t)f0 * D, f0, (int32_t)T[u], (int32_t)u};
    f0 += T[u];
  }
  VCMI_TRY(staged_upload_gather(t->xbuf.p, up, nullptr));
  VCMI_TRY(traj_run(t, utts, nframes, true, t->xbuf.p, nullptr, gv));
  VCMI_TRY(traj_check_status(t, nullptr));
  return staged_download_scatter(down, t->ybuf.p, nullptr);
}

static void traj_sync_replicas(vcmi_traj *t) {
  gmmmap_sync_replicas(t->g);
  const uint64_t ep = group_epoch();
  if (t->replicas_epoch == ep && (int)t->replicas.size() == group_size()) return;
  for (vcmi_traj *r : t->replicas) delete r;
  t->replicas.assign((size_t)group_size(), nullptr);
  t->replicas_epoch = ep;
}

// member i's trajectory converter: t itself when member i got t's own GMMMap, else a replica over the member's GMMMap
static int traj_member(vcmi_traj *t, int member, vcmi_traj **out) {
  vcmi_gmmmap *g = nullptr;
  VCMI_TRY(gmmmap_member(t->g, member, &g));
  if (g == t->g) {
    *out = t;
    return VCMI_OK;
  }
  vcmi_traj *&r = t->replicas[(size_t)member];
  if (!r) VCMI_TRY(vcmi_traj_create(g, t->length, &r));
  *out = r;
  return VCMI_OK;
}

// utterances below which a batch stays on one device even when a device group is set
static constexpr int64_t kGroupMinUtts = 8;

// Utterances (and the chunks of vc) are independent: with a device group they are split by length
// (longest-processing-time) and every member converts its share on its own device; no collective.
static int traj_host_batch(vcmi_traj *t, int64_t n, const double *const *X, const int64_t *T, double *const *Y,
                           const TrajGV *gv = nullptr, vcmi_trajgv *gvh = nullptr) {
  if (!t) return fail(VCMI_ERR_ARG, "trajectory: NULL handle");
  if (n < 0) return fail(VCMI_ERR_ARG, "trajectory: negative batch size");
  if (n == 0) return VCMI_OK;
  if (!X || !T || !Y) return fail(VCMI_ERR_ARG, "trajectory: NULL argument");
  for (int64_t u = 0; u < n; ++u) {
    if (T[u] < 0 || T[u] > INT32_MAX) return fail(VCMI_ERR_DIM, "trajectory: bad utterance length");
    if (T[u] > 0 && (!X[u] || !Y[u])) return fail(VCMI_ERR_ARG, "trajectory: NULL matrix");
  }
  const int m = group_size();
  if (m == 0 || n < kGroupMinUtts) return traj_host_batch_local(t, n, X, T, Y, gv);
  traj_sync_replicas(t);
  if (gvh && (gvh->replicas_epoch != group_epoch() || (int)gvh->replicas.size() != m)) {
    for (vcmi_trajgv *r : gvh->replicas) delete r;
    gvh->replicas.assign((size_t)m, nullptr);
    gvh->replicas_epoch = group_epoch();
  }
  std::vector<int64_t> costs((size_t)n);
  for (int64_t u = 0; u < n; ++u) costs[(size_t)u] = T[u];
  const std::vector<int> part = shard_by_cost(costs, m);
  return group_run(m, [&](int i) -> int {
    std::vector<const double *> x2;
    std::vector<double *> y2;
    std::vector<int64_t> T2;
    for (int64_t u = 0; u < n; ++u) {
      if (part[(size_t)u] != i) continue;
      x2.push_back(X[u]);
      y2.push_back(Y[u]);
      T2.push_back(T[u]);
    }
    if (x2.empty()) return VCMI_OK;
    vcmi_traj *r = nullptr;
    VCMI_TRY(traj_member(t, i, &r));
    TrajGV gv2;
    if (gv) {
      gv2 = *gv;
      if (r != t) {   // the GV statistics on this member's device
        vcmi_trajgv *&gr = gvh->replicas[(size_t)i];
        if (!gr) {
          gr = new (std::nothrow) vcmi_trajgv();
          if (!gr) return fail(VCMI_ERR_OOM, "out of host memory");
          gr->t = r;
          VCMI_TRY(gr->muv.alloc(gvh->h_muv.size()));
          VCMI_TRY(gr->pv.alloc(gvh->h_pv.size()));
          VCMI_TRY(upload_now(gr->muv.p, gvh->h_muv.data(), gvh->h_muv.size() * 8));
          VCMI_TRY(upload_now(gr->pv.p, gvh->h_pv.data(), gvh->h_pv.size() * 8));
        }
        gv2.muv = gr->muv.p;
        gv2.pv = gr->pv.p;
      }
    }
    return traj_host_batch_local(r, (int64_t)x2.size(), x2.data(), T2.data(), y2.data(), gv ? &gv2 : nullptr);
  });
}

}  // namespace vcmi

using namespace vcmi;

extern "C" int vcmi_traj_create(vcmi_gmmmap *g, int64_t T, vcmi_traj **out) {
  if (!g || !out) return fail(VCMI_ERR_ARG, "vcmi_traj_create: NULL argument");
  *out = nullptr;
  if (g->D & 1) return fail(VCMI_ERR_DIM, "TrajectoryGMMMap: dim(g) = %d must be even (static + delta)", g->D);
  if (T < 0) return fail(VCMI_ERR_ARG, "TrajectoryGMMMap: negative length");
  const int D2 = g->D, D = D2 / 2, M = g->M;
  vcmi_traj *t = new (std::nothrow) vcmi_traj();
  if (!t) return fail(VCMI_ERR_OOM, "out of host memory");
  t->big = solve_lds_bytes(D) > 160 * 1024 - 64;      // D >= 47: the window of the solve does not fit LDS -> traj_solve_big_kernel
  t->g = g;
  t->D2 = D2;
  t->D = D;
  t->M = M;
  t->length = T;
  const size_t nn = (size_t)D2 * D2;
  std::vector<double> Q(nn * M), QT(nn * M), AT(nn * M), bv((size_t)D2 * M), tmp(nn), S(nn);
  for (int m = 0; m < M; ++m) {
    // Dy_m = inv(Syy_m - A_m Sxy_m), src/trajectory_gmmmap.jl:24-28
    la::matmul(&g->h_A[nn * m], &g->h_Sxy[nn * m], D2, tmp.data());
    for (size_t k = 0; k < nn; ++k) S[k] = g->h_Syy[nn * m + k] - tmp[k];
    if (!la::inverse(S.data(), D2, &Q[nn * m])) {
      delete t;
      return fail(VCMI_ERR_NOT_PD, "TrajectoryGMMMap: conditional covariance of mixture %d is singular", m + 1);
    }
    for (int r = 0; r < D2; ++r) {
      double ba = 0.0;
      for (int k = 0; k < D2; ++k) {
        QT[nn * m + (size_t)k * D2 + r] = Q[nn * m + (size_t)r * D2 + k];
        AT[nn * m + (size_t)k * D2 + r] = g->h_A[nn * m + (size_t)r * D2 + k];
        ba += g->h_A[nn * m + (size_t)r * D2 + k] * g->h_mux[(size_t)D2 * m + k];
      }
      bv[(size_t)D2 * m + r] = g->h_muy[(size_t)D2 * m + r] - ba;
    }
  }
  // Q in MFMA A-operand order for the GV ascent: lane l of fragment (row tile i, k-step ks) holds Q[16i + (l&15)][4ks + (l>>4)]
  t->NT = (D2 + 15) / 16;
  t->KS = (D2 + 3) / 4;
  std::vector<double> Qf((size_t)M * t->NT * t->KS * 64, 0.0), Af(Qf.size(), 0.0);
  for (int m = 0; m < M; ++m)
    for (int i = 0; i < t->NT; ++i)
      for (int ks = 0; ks < t->KS; ++ks)
        for (int l = 0; l < 64; ++l) {
          const int r = 16 * i + (l & 15), k = 4 * ks + (l >> 4);
          if (r < D2 && k < D2) {
            Qf[(((size_t)m * t->NT + i) * t->KS + ks) * 64 + l] = Q[nn * m + (size_t)r * D2 + k];
            Af[(((size_t)m * t->NT + i) * t->KS + ks) * 64 + l] = g->h_A[nn * m + (size_t)r * D2 + k];
          }
        }
  t->Dpad = traj_blk_padded_dim(D);
  if (t->Dpad) {
    const int Dp = t->Dpad, Dp2 = 2 * Dp;
    std::vector<double> Qp((size_t)M * Dp2 * Dp2, 0.0);
    for (int m = 0; m < M; ++m) {
      double *q = &Qp[(size_t)m * Dp2 * Dp2];
      for (int r = 0; r < D2; ++r)
        for (int c = 0; c < D2; ++c) {
          const int rp = (r / D) * Dp + r % D, cp = (c / D) * Dp + c % D;     // [static ; delta] halves keep their blocks
          q[(size_t)rp * Dp2 + cp] = Q[nn * m + (size_t)r * D2 + c];
        }
      for (int d = D; d < Dp; ++d) q[(size_t)d * Dp2 + d] = 1.0;              // padding: P = I, r = 0 -> y = 0, decoupled
    }
    int rcp = t->Qpad.alloc(Qp.size());
    if (rcp == VCMI_OK && upload_now_hip(t->Qpad.p, Qp.data(), Qp.size() * 8) != hipSuccess) rcp = VCMI_ERR_HIP;
    if (rcp != VCMI_OK) {
      delete t;
      return rcp == VCMI_ERR_HIP ? fail(VCMI_ERR_HIP, "vcmi_traj_create: upload of the padded Q failed") : rcp;
    }
  }
  int rc = VCMI_OK;
  if ((rc = t->Q.alloc(Q.size())) || (rc = t->QT.alloc(QT.size())) || (rc = t->AT.alloc(AT.size())) ||
      (rc = t->bvec.alloc(bv.size())) || (rc = t->Qfrag.alloc(Qf.size())) || (rc = t->Afrag.alloc(Af.size()))) {
    delete t;
    return rc;
  }
  hipError_t e = upload_now_hip(t->Q.p, Q.data(), Q.size() * 8);
  if (e == hipSuccess) e = upload_now_hip(t->QT.p, QT.data(), QT.size() * 8);
  if (e == hipSuccess) e = upload_now_hip(t->AT.p, AT.data(), AT.size() * 8);
  if (e == hipSuccess) e = upload_now_hip(t->bvec.p, bv.data(), bv.size() * 8);
  if (e == hipSuccess) e = upload_now_hip(t->Qfrag.p, Qf.data(), Qf.size() * 8);
  if (e == hipSuccess) e = upload_now_hip(t->Afrag.p, Af.data(), Af.size() * 8);
  if (e != hipSuccess) {
    delete t;
    return fail(VCMI_ERR_HIP, "TrajectoryGMMMap: upload failed: %s", hipGetErrorString(e));
  }
  *out = t;
  return VCMI_OK;
}

extern "C" int vcmi_traj_destroy(vcmi_traj *t) {
  delete t;
  return VCMI_OK;
}
extern "C" int64_t vcmi_traj_length(const vcmi_traj *t) { return t ? t->length : -1; }

extern "C" int vcmi_traj_convert(vcmi_traj *t, const double *X, int64_t T, double *Y) {
  const double *xs[1] = {X};
  double *ys[1] = {Y};
  VCMI_TRY(traj_host_batch(t, 1, xs, &T, ys));
  // The reference rebuilds W when the sequence length differs from length(tgmm) (src/trajectory_gmmmap.jl:70-72), so
  // afterwards length(tgmm) == T -- and with it the chunk length of the next vc call (src/common.jl:41).  Reproduced.
  t->length = T;
  return VCMI_OK;
}

extern "C" int vcmi_traj_convert_batch(vcmi_traj *t, int64_t n, const double *const *X, const int64_t *T, double *const *Y) {
  return traj_host_batch(t, n, X, T, Y);
}

extern "C" int vcmi_traj_convert_batch_dev(vcmi_traj *t, int64_t n, const double *dX, const int64_t *x_off, const int64_t *T,
                                           double *dY, const int64_t *y_off, void *stream) {
  if (!t) return fail(VCMI_ERR_ARG, "vcmi_traj_convert_batch_dev: NULL handle");
  if (n < 0) return fail(VCMI_ERR_ARG, "vcmi_traj_convert_batch_dev: negative batch size");
  if (n == 0) return VCMI_OK;
  if (!dX || !x_off || !T || !dY || !y_off) return fail(VCMI_ERR_ARG, "vcmi_traj_convert_batch_dev: NULL argument");
  std::vector<TrajUtt> utts(n);
  int64_t f0 = 0;
  bool contiguous = true;
  for (int64_t u = 0; u < n; ++u) {
    if (T[u] < 0 || T[u] > INT32_MAX) return fail(VCMI_ERR_DIM, "trajectory: bad utterance length");
    if (x_off[u] != x_off[0] + f0 * t->D2) contiguous = false;
    utts[u] = TrajUtt{dX + x_off[u], dY + y_off[u], f0, (int32_t)T[u], (int32_t)u};
    f0 += T[u];
  }
  if (f0 == 0) return VCMI_OK;   // only empty utterances: nothing was launched, there is no status to read
  VCMI_TRY(traj_run(t, utts, f0, contiguous, dX + x_off[0], as_stream(stream)));
  return traj_check_status(t, as_stream(stream));
}

extern "C" int vcmi_vc_traj(vcmi_traj *t, const double *fm, int64_t T, double *out) {
  if (!t) return fail(VCMI_ERR_ARG, "vcmi_vc_traj: NULL handle");
  if (T < 0 || (T > 0 && (!fm || !out))) return fail(VCMI_ERR_ARG, "vcmi_vc_traj: bad argument");
  if (T == 0) return VCMI_OK;
  if (t->length < 1) return fail(VCMI_ERR_ARG, "vcmi_vc_traj: length(t) must be positive");
  const int D = t->D, D2 = t->D2;
  const int64_t L = t->length, nch = (T + L - 1) / L;   // chunks [kL+1, min((k+1)L, T)], src/common.jl:42-57
  std::vector<double> x((size_t)T * D2), y((size_t)T * D);
  for (int64_t f = 0; f < T; ++f) memcpy(&x[(size_t)f * D2], fm + (size_t)f * (D2 + 1) + 1, sizeof(double) * D2);
  std::vector<const double *> xs(nch);
  std::vector<double *> ys(nch);
  std::vector<int64_t> Ts(nch);
  for (int64_t k = 0; k < nch; ++k) {
    xs[k] = &x[(size_t)k * L * D2];
    ys[k] = &y[(size_t)k * L * D];
    Ts[k] = std::min<int64_t>(L, T - k * L);
  }
  VCMI_TRY(traj_host_batch(t, nch, xs.data(), Ts.data(), ys.data()));
  for (int64_t f = 0; f < T; ++f) {
    out[(size_t)f * (D + 1)] = fm[(size_t)f * (D2 + 1)];   // power row kept, src/common.jl:60
    memcpy(out + (size_t)f * (D + 1) + 1, &y[(size_t)f * D], sizeof(double) * D);
  }
  t->length = Ts[nch - 1];   // the last fvconvert of the loop left W at the last chunk's length (see vcmi_traj_convert)
  return VCMI_OK;
}

// vc(c::TrajectoryConverter, fm) followed by fvpostf!(VarianceScaling(sigma2), converted[2:end, :]) -- src/common.jl:31-63,
// src/gv.jl:10-15 -- with the matrix resident in HBM from the upload of fm to the download of the filtered result: the power
// row and the (2D,T) feature rows are split on the device, the chunks are converted by the device-resident batch path, the
// post-filter runs on the (D,T) result and the output matrix is assembled on the device.
extern "C" int vcmi_vc_traj_postf(vcmi_traj *t, const double *fm, int64_t T, const double *sigma2, double *out) {
  if (!sigma2) return vcmi_vc_traj(t, fm, T, out);
  if (!t) return fail(VCMI_ERR_ARG, "vcmi_vc_traj_postf: NULL handle");
  if (T < 0 || (T > 0 && (!fm || !out))) return fail(VCMI_ERR_ARG, "vcmi_vc_traj_postf: bad argument");
  if (T == 0) return VCMI_OK;
  if (T < 2) return fail(VCMI_ERR_DIM, "vcmi_vc_traj_postf: the variance of a one-frame matrix is undefined");
  if (t->length < 1) return fail(VCMI_ERR_ARG, "vcmi_vc_traj_postf: length(t) must be positive");
  VCMI_TRY(check_device());
  const int D = t->D, D2 = t->D2;
  const int64_t L = t->length, nch = (T + L - 1) / L;   // chunks [kL+1, min((k+1)L, T)], src/common.jl:42-57
  static thread_local DevBuf<double> dfm, dx, dy, dout;
  VCMI_TRY(dfm.reserve((size_t)(D2 + 1) * T));
  VCMI_TRY(dx.reserve((size_t)D2 * T));
  VCMI_TRY(dy.reserve((size_t)D * T));
  VCMI_TRY(dout.reserve((size_t)(D + 1) * T));
  VCMI_TRY(staged_upload(dfm.p, fm, sizeof(double) * (size_t)(D2 + 1) * T, nullptr));
  VCMI_TRY(copy_rows_device(dfm.p, D2 + 1, 1, D2, T, dx.p, D2, 0, nullptr));
  VCMI_TRY(copy_rows_device(dfm.p, D2 + 1, 0, 1, T, dout.p, D + 1, 0, nullptr));         // power row kept, src/common.jl:60
  std::vector<int64_t> xo(nch), yo(nch), Ts(nch);
  for (int64_t k = 0; k < nch; ++k) {
    xo[k] = k * L * D2;
    yo[k] = k * L * D;
    Ts[k] = std::min<int64_t>(L, T - k * L);
  }
  VCMI_TRY(vcmi_traj_convert_batch_dev(t, nch, dx.p, xo.data(), Ts.data(), dy.p, yo.data(), nullptr));
  VCMI_TRY(variance_scaling_device(dy.p, D, D, T, sigma2, dy.p, D, nullptr));
  VCMI_TRY(copy_rows_device(dy.p, D, 0, D, T, dout.p, D + 1, 1, nullptr));
  VCMI_TRY(staged_download(out, dout.p, sizeof(double) * (size_t)(D + 1) * T, nullptr));
  t->length = Ts[nch - 1];   // as vcmi_vc_traj: the last fvconvert of the loop left W at the last chunk's length
  return VCMI_OK;
}

// ---- TrajectoryGVGMMMap, src/trajectory_gmmmap.jl:114-189 ----------------------------------------
extern "C" int vcmi_trajgv_create(vcmi_traj *t, const double *muv, const double *sigmavv, vcmi_trajgv **out) {
  if (!t || !muv || !sigmavv || !out) return fail(VCMI_ERR_ARG, "vcmi_trajgv_create: NULL argument");
  *out = nullptr;
  const int D = t->D;
  if (64 * t->NT > 384) return fail(VCMI_ERR_ARG, "TrajectoryGVGMMMap: feature dimension %d too large", t->D2);
  for (int d = 0; d < D; ++d)
    if (muv[d] < 0.0) return fail(VCMI_ERR_ARG, "TrajectoryGVGMMMap: the GV mean must be non-negative");   // the @assert of :124
  std::vector<double> S((size_t)D * D), P((size_t)D * D);
  for (int r = 0; r < D; ++r)
    for (int c = 0; c < D; ++c) S[(size_t)r * D + c] = sigmavv[r + (size_t)D * c];
  if (!la::inverse(S.data(), D, P.data())) return fail(VCMI_ERR_NOT_PD, "TrajectoryGVGMMMap: the GV covariance is singular");
  std::vector<double> Pj((size_t)D * D);                       // back to the Julia memory image
  for (int r = 0; r < D; ++r)
    for (int c = 0; c < D; ++c) Pj[r + (size_t)D * c] = P[(size_t)r * D + c];
  vcmi_trajgv *h = new (std::nothrow) vcmi_trajgv();
  if (!h) return fail(VCMI_ERR_OOM, "out of host memory");
  h->t = t;
  int rc = VCMI_OK;
  if ((rc = h->muv.alloc(D)) || (rc = h->pv.alloc((size_t)D * D))) {
    delete h;
    return rc;
  }
  hipError_t e = upload_now_hip(h->muv.p, muv, sizeof(double) * D);
  if (e == hipSuccess) e = upload_now_hip(h->pv.p, Pj.data(), sizeof(double) * D * D);
  if (e != hipSuccess) {
    delete h;
    return fail(VCMI_ERR_HIP, "TrajectoryGVGMMMap: upload failed: %s", hipGetErrorString(e));
  }
  h->h_muv.assign(muv, muv + D);
  h->h_pv = Pj;
  *out = h;
  return VCMI_OK;
}

extern "C" int vcmi_trajgv_destroy(vcmi_trajgv *h) {
  delete h;
  return VCMI_OK;
}

static int trajgv_args(const vcmi_trajgv *h, int64_t n, const int64_t *T, int epochs, TrajGV *gv) {
  if (!h) return fail(VCMI_ERR_ARG, "TrajectoryGVGMMMap: NULL handle");
  if (epochs < 0) return fail(VCMI_ERR_ARG, "TrajectoryGVGMMMap: negative epoch count");
  for (int64_t u = 0; T && u < n; ++u)
    if (T[u] == 1) return fail(VCMI_ERR_DIM, "TrajectoryGVGMMMap: the variance of a one-frame trajectory is undefined");
  gv->muv = h->muv.p;
  gv->pv = h->pv.p;
  gv->epochs = epochs;
  return VCMI_OK;
}

extern "C" int vcmi_trajgv_convert_batch(vcmi_trajgv *h, int64_t n, const double *const *X, const int64_t *T, int epochs,
                                         double alpha, double *const *Y) {
  TrajGV gv{};
  gv.alpha = alpha;
  VCMI_TRY(trajgv_args(h, n, T, epochs, &gv));
  return traj_host_batch(h->t, n, X, T, Y, &gv, h);
}

extern "C" int vcmi_trajgv_convert(vcmi_trajgv *h, const double *X, int64_t T, int epochs, double alpha, double *Y) {
  const double *xs[1] = {X};
  double *ys[1] = {Y};
  VCMI_TRY(vcmi_trajgv_convert_batch(h, 1, xs, &T, epochs, alpha, ys));
  h->t->length = T;   // fvconvert(tgv.tgmm, X) rebuilt W for this T, src/trajectory_gmmmap.jl:70-72,146
  return VCMI_OK;
}

extern "C" int vcmi_trajgv_convert_batch_dev(vcmi_trajgv *h, int64_t n, const double *dX, const int64_t *x_off, const int64_t *T,
                                             int epochs, double alpha, double *dY, const int64_t *y_off, void *stream) {
  TrajGV gv{};
  gv.alpha = alpha;
  VCMI_TRY(trajgv_args(h, n, T, epochs, &gv));
  vcmi_traj *t = h->t;
  if (n < 0) return fail(VCMI_ERR_ARG, "vcmi_trajgv_convert_batch_dev: negative batch size");
  if (n == 0) return VCMI_OK;
  if (!dX || !x_off || !T || !dY || !y_off) return fail(VCMI_ERR_ARG, "vcmi_trajgv_convert_batch_dev: NULL argument");
  std::vector<TrajUtt> utts(n);
  int64_t f0 = 0;
  bool contiguous = true;
  for (int64_t u = 0; u < n; ++u) {
    if (T[u] < 0 || T[u] > INT32_MAX) return fail(VCMI_ERR_DIM, "trajectory: bad utterance length");
    if (x_off[u] != x_off[0] + f0 * t->D2) contiguous = false;
    utts[u] = TrajUtt{dX + x_off[u], dY + y_off[u], f0, (int32_t)T[u], (int32_t)u};
    f0 += T[u];
  }
  if (f0 == 0) return VCMI_OK;
  VCMI_TRY(traj_run(t, utts, f0, contiguous, dX + x_off[0], as_stream(stream), &gv));
  return traj_check_status(t, as_stream(stream));
}

// diffgmm(params) -- src/diffgmm.jl:9-25, on the joint parameters mu (2D,M), sigma (2D,2D,M) (host arithmetic: a
// one-time parameter transform).  Feed the result to vcmi_gmmmap_create for the differential converter.
extern "C" int vcmi_diffgmm(const double *mu, const double *sigma, int Dj, int M, double *mu_out, double *sigma_out) {
  if (!mu || !sigma || !mu_out || !sigma_out) return fail(VCMI_ERR_ARG, "vcmi_diffgmm: NULL argument");
  if (Dj < 2 || (Dj & 1) || M < 1) return fail(VCMI_ERR_DIM, "vcmi_diffgmm: joint dimension %d / mixtures %d invalid", Dj, M);
  const int D = Dj / 2;
  for (int m = 0; m < M; ++m) {
    const double *S = sigma + (size_t)Dj * Dj * m;
    double *O = sigma_out + (size_t)Dj * Dj * m;
    for (int d = 0; d < D; ++d) {
      const double mx = mu[d + (size_t)Dj * m], my = mu[D + d + (size_t)Dj * m];
      mu_out[d + (size_t)Dj * m] = mx;
      mu_out[D + d + (size_t)Dj * m] = my - mx;                                  // eq. (6)
    }
    for (int c = 0; c < D; ++c)
      for (int r = 0; r < D; ++r) {
        const double xx = S[r + (size_t)Dj * c], xy = S[r + (size_t)Dj * (D + c)], yx = S[(D + r) + (size_t)Dj * c],
                     yy = S[(D + r) + (size_t)Dj * (D + c)];
        O[r + (size_t)Dj * c] = xx;
        O[r + (size_t)Dj * (D + c)] = xy - xx;                                    // eq. (7)
        O[(D + c) + (size_t)Dj * r] = xy - xx;                                    // its transpose
        O[(D + r) + (size_t)Dj * (D + c)] = xx + yy - xy - yx;                    // eq. (8)
      }
  }
  return VCMI_OK;
}
