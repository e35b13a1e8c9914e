// dtw.hip -- batched dynamic time warping + align post-processing on MI355X (gfx950).
//
// Replaces   fit!(d::DTW, template, sequence) + backward(d)    reference src/dtw.jl:93-145
//            lazy_init!(d, S, T)                               reference src/dtw.jl:44-51
//            transition / observation                          reference src/dtw.jl:23-35
//            align(src, tgt)                                   reference src/align.jl:8-35
//
// This file is compiled with -ffp-contract=off: the parity contract for DTW is BIT-EXACT cost tables and
// back-pointers against the CPU oracle, whose arithmetic is  o = sum_d (v_d - tmpl_d)^2  accumulated
// sequentially in d with separately rounded multiply and add, and  cost = (C[j,t] + o) + transition.
// A GEMM-form distance or a fused multiply-add would change roundings and is therefore not used.
//
// Parallelisation.  Column t+1 of the cost table depends only on column t (src/dtw.jl:110,117), so the S rows
// of a column are independent: one workgroup owns a pair, thread r owns template frame r (its D values live in
// registers), the previous cost column is double-buffered in LDS and one s_barrier separates columns.
// Back-pointers are kept as small step codes (row - predecessor + fstep), 2 bits per cell packed 16 columns
// per LDS word when bstep+fstep <= 3, so the backward pass never touches HBM; the full Float64 / Int64 tables
// of the reference (d.costtable, d.backpointer) are streamed to HBM only when the caller asks for them.
#include "vcmi_common.hpp"

#include <algorithm>
#include <numeric>

namespace vcmi {

struct DtwPair {
  const double *tmpl;   // (D,S)
  const double *seq;    // (D,T)
  int64_t *path;        // (T) 1-based, may be null when only newtgt is wanted
  double *cost;         // (S,T+1) or null
  int64_t *bp;          // (S,T+1) or null
  double *newtgt;       // (D,S) or null: align() output
  unsigned char *codes; // (S,T) bytes in HBM, used only when the step codes do not fit in LDS
  int32_t S, T;
};

__device__ __forceinline__ double dtw_transition(int j, int i) {   // transition(d, j, i), src/dtw.jl:23-31
  return (i == j + 1) ? 0.0 : ((i == j) ? 1.0 : 2.0);
}

// shared epilogue: argmin of the last column, backward pass, path output, align post-processing.
template <bool LDSCODES, int BITS>
__device__ void dtw_finish(const DtwPair &P, int D, int fstep, const double *clast, int32_t *path32, int32_t *owner,
                           int32_t *holes, const uint32_t *codes_lds, int Smax) {
  constexpr int CPW = 32 / BITS;
  const int S = P.S, T = P.T, tid = threadIdx.x, nthr = blockDim.x;
  if (tid == 0) {
    int best = 0;                                   // indmin: first minimum, src/dtw.jl:137
    double bv = clast[0];
    for (int i = 1; i < S; ++i) {
      const double c = clast[i];
      if (c < bv) { bv = c; best = i; }
    }
    path32[T - 1] = best;
    for (int t = T - 1; t >= 1; --t) {              // src/dtw.jl:140-142
      const int row = path32[t];
      int code;
      if (LDSCODES) code = (codes_lds[(t / CPW) * Smax + row] >> (BITS * (t % CPW))) & ((1u << BITS) - 1u);
      else code = P.codes[(size_t)S * t + row];
      path32[t - 1] = row - (code - fstep);
    }
  }
  __syncthreads();
  if (P.path)
    for (int t = tid; t < T; t += nthr) P.path[t] = (int64_t)path32[t] + 1;
  if (!P.newtgt) return;
  // ---- align(), src/align.jl:20-32 ----
  for (int i = tid; i < S; i += nthr) owner[i] = -1;
  __syncthreads();
  for (int t = tid; t < T; t += nthr) atomicMax(&owner[path32[t]], t);   // newtgt[:,path] = tgt: later frames win
  __syncthreads();
  for (int64_t e = tid; e < (int64_t)S * D; e += nthr) {
    const int i = (int)(e / D), d = (int)(e % D);
    const int k = owner[i];
    P.newtgt[e] = (k >= 0) ? P.seq[(size_t)D * k + d] : 0.0;
  }
  __shared__ int nholes;
  if (tid == 0) {                                   // hole = setdiff(path[1]:path[end], path), increasing order
    int n = 0;
    for (int i = path32[0]; i <= path32[T - 1]; ++i)
      if (owner[i] < 0 && i > 0 && i < S - 1) holes[n++] = i;   // 1 < i < S in 1-based terms
    nholes = n;
  }
  __syncthreads();
  // each thread owns feature rows d, d+nthr, ...; a row's holes are filled in increasing i, as the reference does
  for (int d = tid; d < D; d += nthr)
    for (int h = 0; h < nholes; ++h) {
      const int i = holes[h];
      P.newtgt[(size_t)D * i + d] = (P.newtgt[(size_t)D * (i - 1) + d] + P.newtgt[(size_t)D * (i + 1) + d]) / 2.0;
    }
}

// ------------------------------------------------------------------------------------------------
// Fast kernel: S <= blockDim (<= 1024), D <= DMAX; template frame in registers.
// ------------------------------------------------------------------------------------------------
template <int DMAX, int BITS, bool LDSCODES>
__global__ void __launch_bounds__(1024)
dtw_kernel(const DtwPair *__restrict__ pairs, int D, int fstep, int bstep, int Smax, int Tmax) {
  constexpr int CPW = 32 / BITS;
  const DtwPair P = pairs[blockIdx.x];
  const int S = P.S, T = P.T;
  const int r = threadIdx.x;
  const bool active = r < S;
  extern __shared__ unsigned char smem_raw[];
  double *cbuf = reinterpret_cast<double *>(smem_raw);            // [2][Smax]
  int32_t *path32 = reinterpret_cast<int32_t *>(cbuf + 2 * Smax); // [Tmax]
  int32_t *owner = path32 + Tmax;                                 // [Smax]
  int32_t *holes = owner + Smax;                                  // [Smax]
  uint32_t *codes = reinterpret_cast<uint32_t *>(holes + Smax);   // [ceil(Tmax/CPW)][Smax] when LDSCODES
  if (T == 0) return;

  double tm[DMAX];
#pragma unroll
  for (int d = 0; d < DMAX; ++d) tm[d] = (active && d < D) ? P.tmpl[(size_t)D * r + d] : 0.0;

  // lazy_init!: costtable[:,1] = 1:S, backpointer[:,1] = 1:S  (src/dtw.jl:49-50)
  double cprev = (double)(r + 1);
  if (active) {
    cbuf[r] = cprev;
    if (P.cost) P.cost[r] = cprev;
    if (P.bp) P.bp[r] = r + 1;
  }
  __syncthreads();

  uint32_t word = 0;
  for (int t = 0; t < T; ++t) {
    const double *cp = cbuf + (t & 1) * Smax;
    double *cn = cbuf + ((t + 1) & 1) * Smax;
    if (active) {
      const double *__restrict__ v = P.seq + (size_t)D * t;
      double o = 0.0;                                   // observation(d, v, i), src/dtw.jl:33-35
#pragma unroll
      for (int d = 0; d < DMAX; ++d) {
        if (d < D) {
          const double df = v[d] - tm[d];
          const double sq = df * df;
          o = o + sq;
        }
      }
      int arg = r;                                      // minindex = i, src/dtw.jl:106-110
      double best = (cprev + o) + 1.0;
      for (int j = r - bstep; j <= r + fstep; ++j) {    // src/dtw.jl:113-121
        if (j < 0 || j >= S) continue;
        const double c = (cp[j] + o) + dtw_transition(j, r);
        if (c < best) { best = c; arg = j; }
      }
      cn[r] = best;
      cprev = best;
      if (P.cost) P.cost[(size_t)S * (t + 1) + r] = best;
      if (P.bp) P.bp[(size_t)S * (t + 1) + r] = arg + 1;
      const uint32_t code = (uint32_t)(r - arg + fstep);
      if (LDSCODES) {
        word |= code << (BITS * (t % CPW));
        if ((t % CPW) == CPW - 1 || t == T - 1) {
          codes[(t / CPW) * Smax + r] = word;
          word = 0;
        }
      } else {
        P.codes[(size_t)S * t + r] = (unsigned char)code;
      }
    }
    __syncthreads();
  }
  if (!LDSCODES) {
    __threadfence();
    __syncthreads();
  }
  dtw_finish<LDSCODES, BITS>(P, D, fstep, cbuf + (T & 1) * Smax, path32, owner, holes, codes, Smax);
}

// ------------------------------------------------------------------------------------------------
// Generic kernel: any D, S up to the LDS capacity of two cost columns; rows strided over threads,
// template read from HBM/L2, step codes as bytes in HBM.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024)
dtw_generic_kernel(const DtwPair *__restrict__ pairs, int D, int fstep, int bstep, int Smax, int Tmax) {
  const DtwPair P = pairs[blockIdx.x];
  const int S = P.S, T = P.T, tid = threadIdx.x, nthr = blockDim.x;
  extern __shared__ unsigned char smem_raw[];
  double *cbuf = reinterpret_cast<double *>(smem_raw);
  int32_t *path32 = reinterpret_cast<int32_t *>(cbuf + 2 * Smax);
  int32_t *owner = path32 + Tmax;
  int32_t *holes = owner + Smax;
  if (T == 0) return;
  for (int r = tid; r < S; r += nthr) {
    cbuf[r] = (double)(r + 1);
    if (P.cost) P.cost[r] = (double)(r + 1);
    if (P.bp) P.bp[r] = r + 1;
  }
  __syncthreads();
  for (int t = 0; t < T; ++t) {
    const double *cp = cbuf + (t & 1) * Smax;
    double *cn = cbuf + ((t + 1) & 1) * Smax;
    const double *__restrict__ v = P.seq + (size_t)D * t;
    for (int r = tid; r < S; r += nthr) {
      const double *__restrict__ tc = P.tmpl + (size_t)D * r;
      double o = 0.0;
      for (int d = 0; d < D; ++d) {
        const double df = v[d] - tc[d];
        const double sq = df * df;
        o = o + sq;
      }
      int arg = r;
      double best = (cp[r] + o) + 1.0;
      for (int j = r - bstep; j <= r + fstep; ++j) {
        if (j < 0 || j >= S) continue;
        const double c = (cp[j] + o) + dtw_transition(j, r);
        if (c < best) { best = c; arg = j; }
      }
      cn[r] = best;
      if (P.cost) P.cost[(size_t)S * (t + 1) + r] = best;
      if (P.bp) P.bp[(size_t)S * (t + 1) + r] = arg + 1;
      P.codes[(size_t)S * t + r] = (unsigned char)(r - arg + fstep);
    }
    __syncthreads();
  }
  __threadfence();
  __syncthreads();
  dtw_finish<false, 8>(P, D, fstep, cbuf + (T & 1) * Smax, path32, owner, holes, nullptr, Smax);
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
static const size_t kLdsLimit = 160 * 1024 - 256;   // dynamic LDS budget; leaves room for the kernels' static __shared__

struct DtwLaunch {
  std::vector<DtwPair> pairs;
  int D, fstep, bstep;
};

template <int DMAX>
static int launch_fast(const DtwPair *dpairs, int n, int D, int fstep, int bstep, int Smax, int Tmax, int bits,
                       bool ldscodes, size_t shmem, int threads, hipStream_t st) {
#define VCMI_DTW_LAUNCH(B, L)                                                                                  \
  do {                                                                                                         \
    auto kern = dtw_kernel<DMAX, B, L>;                                                                        \
    VCMI_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                 (int)shmem));                                                                 \
    hipLaunchKernelGGL(kern, dim3(n), dim3(threads), shmem, st, dpairs, D, fstep, bstep, Smax, Tmax);          \
  } while (0)
  if (!ldscodes) VCMI_DTW_LAUNCH(8, false);
  else if (bits == 2) VCMI_DTW_LAUNCH(2, true);
  else VCMI_DTW_LAUNCH(8, true);
#undef VCMI_DTW_LAUNCH
  VCMI_HIP(hipGetLastError());
  return VCMI_OK;
}

// Launch one workgroup per pair.  `pairs` holds DEVICE pointers (codes filled in here when needed).
// codes_ws: grow-only device scratch for HBM step codes.
static int dtw_run(std::vector<DtwPair> &pairs, int D, int fstep, int bstep, DevBuf<unsigned char> &codes_ws,
                   DevBuf<DtwPair> &dpairs, hipStream_t st) {
  const int n = (int)pairs.size();
  if (n == 0) return VCMI_OK;
  if (D < 1) return fail(VCMI_ERR_DIM, "DTW: feature dimension %d invalid", D);
  if (fstep < 0 || bstep < 0 || fstep + bstep > 255) return fail(VCMI_ERR_ARG, "DTW: fstep/bstep out of range");
  int Smax = 0, Tmax = 0;
  for (auto &p : pairs) {
    if (p.S < 1) return fail(VCMI_ERR_DIM, "DTW: template must have at least one frame");
    Smax = std::max(Smax, p.S);
    Tmax = std::max(Tmax, p.T);
  }
  if (Tmax == 0) return VCMI_OK;
  // largest pairs first: shortens the tail when n is not a multiple of the resident workgroup count
  std::stable_sort(pairs.begin(), pairs.end(), [](const DtwPair &a, const DtwPair &b) {
    return (int64_t)a.S * a.T > (int64_t)b.S * b.T;
  });
  const int bits = (fstep + bstep + 1 <= 4) ? 2 : 8;
  const size_t base = (size_t)2 * Smax * 8 + (size_t)Tmax * 4 + (size_t)2 * Smax * 4;
  const size_t codes_lds = (size_t)((Tmax + (32 / bits) - 1) / (32 / bits)) * Smax * 4;
  const bool fast = (Smax <= 1024 && D <= 128);
  const bool ldscodes = fast && (base + codes_lds <= kLdsLimit);
  if (base > kLdsLimit) return fail(VCMI_ERR_ARG, "DTW: template of %d frames exceeds the supported length", Smax);
  if (!ldscodes) {
    size_t total = 0;
    for (auto &p : pairs) total += (size_t)p.S * p.T;
    VCMI_TRY(codes_ws.reserve(total));
    size_t off = 0;
    for (auto &p : pairs) {
      p.codes = codes_ws.p + off;
      off += (size_t)p.S * p.T;
    }
  }
  VCMI_TRY(dpairs.reserve(n));
  VCMI_HIP(hipMemcpy(dpairs.p, pairs.data(), sizeof(DtwPair) * n, hipMemcpyHostToDevice));   // tiny, synchronous: `pairs` may die after return
  if (fast) {
    const int threads = std::min(1024, (Smax + 63) / 64 * 64);
    const size_t shmem = base + (ldscodes ? codes_lds : 0);
    if (D <= 8) return launch_fast<8>(dpairs.p, n, D, fstep, bstep, Smax, Tmax, bits, ldscodes, shmem, threads, st);
    if (D <= 16) return launch_fast<16>(dpairs.p, n, D, fstep, bstep, Smax, Tmax, bits, ldscodes, shmem, threads, st);
    if (D <= 32) return launch_fast<32>(dpairs.p, n, D, fstep, bstep, Smax, Tmax, bits, ldscodes, shmem, threads, st);
    if (D <= 40) return launch_fast<40>(dpairs.p, n, D, fstep, bstep, Smax, Tmax, bits, ldscodes, shmem, threads, st);
    if (D <= 64) return launch_fast<64>(dpairs.p, n, D, fstep, bstep, Smax, Tmax, bits, ldscodes, shmem, threads, st);
    return launch_fast<128>(dpairs.p, n, D, fstep, bstep, Smax, Tmax, bits, ldscodes, shmem, threads, st);
  }
  VCMI_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(dtw_generic_kernel),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)base));
  hipLaunchKernelGGL(dtw_generic_kernel, dim3(n), dim3(1024), base, st, dpairs.p, D, fstep, bstep, Smax, Tmax);
  VCMI_HIP(hipGetLastError());
  return VCMI_OK;
}

// per-thread scratch shared by the DTW entry points
struct DtwScratch {
  DevBuf<unsigned char> codes;
  DevBuf<DtwPair> dpairs;
  DevBuf<double> feats, cost, newtgt;
  DevBuf<int64_t> paths, bp;
};
static DtwScratch &scratch() {
  static thread_local DtwScratch s;
  return s;
}

// Host-pointer batch: packs all feature matrices into one upload, runs, downloads paths / tables / newtgt.
static int dtw_host_batch(int64_t n, const double *const *tmpl, const int64_t *S, const double *const *seq,
                          const int64_t *T, int D, int fstep, int bstep, int64_t *const *path, double *cost,
                          int64_t *bp, double *const *newtgt) {
  if (n < 0) return fail(VCMI_ERR_ARG, "DTW: negative batch size");
  if (n == 0) return VCMI_OK;
  if (!tmpl || !S || !seq || !T) return fail(VCMI_ERR_ARG, "DTW: NULL argument");
  if (D < 1) return fail(VCMI_ERR_DIM, "DTW: feature dimension %d invalid", D);
  VCMI_TRY(check_device());
  DtwScratch &sc = scratch();
  size_t nfeat = 0, npath = 0, ntgt = 0;
  for (int64_t p = 0; p < n; ++p) {
    if (S[p] < 1 || T[p] < 0 || S[p] > INT32_MAX || T[p] > INT32_MAX) return fail(VCMI_ERR_DIM, "DTW: bad sequence length");
    if (!tmpl[p] || (T[p] > 0 && !seq[p])) return fail(VCMI_ERR_ARG, "DTW: NULL feature matrix");
    nfeat += (size_t)D * (S[p] + T[p]);
    npath += (size_t)T[p];
    ntgt += (size_t)D * S[p];
  }
  std::vector<double> hfeat(nfeat);
  VCMI_TRY(sc.feats.reserve(nfeat));
  VCMI_TRY(sc.paths.reserve(std::max<size_t>(npath, 1)));
  if (newtgt) VCMI_TRY(sc.newtgt.reserve(ntgt));
  const bool tables = (cost || bp);
  if (tables) {   // single-pair entry point only
    VCMI_TRY(sc.cost.reserve((size_t)S[0] * (T[0] + 1)));
    VCMI_TRY(sc.bp.reserve((size_t)S[0] * (T[0] + 1)));
  }
  std::vector<DtwPair> pairs(n);
  size_t fo = 0, po = 0, to = 0;
  for (int64_t p = 0; p < n; ++p) {
    DtwPair &q = pairs[p];
    memcpy(&hfeat[fo], tmpl[p], sizeof(double) * D * S[p]);
    q.tmpl = sc.feats.p + fo;
    fo += (size_t)D * S[p];
    if (T[p] > 0) memcpy(&hfeat[fo], seq[p], sizeof(double) * D * T[p]);
    q.seq = sc.feats.p + fo;
    fo += (size_t)D * T[p];
    q.path = sc.paths.p + po;
    po += (size_t)T[p];
    q.cost = tables ? sc.cost.p : nullptr;
    q.bp = tables ? sc.bp.p : nullptr;
    q.newtgt = newtgt ? sc.newtgt.p + to : nullptr;
    to += (size_t)D * S[p];
    q.codes = nullptr;
    q.S = (int32_t)S[p];
    q.T = (int32_t)T[p];
  }
  VCMI_HIP(hipMemcpy(sc.feats.p, hfeat.data(), nfeat * 8, hipMemcpyHostToDevice));
  if (tables && T[0] == 0) {   // lazy_init! only: column 1 = 1:S
    for (int64_t i = 0; i < S[0]; ++i) {
      if (cost) cost[i] = (double)(i + 1);
      if (bp) bp[i] = i + 1;
    }
  }
  std::vector<DtwPair> order = pairs;   // dtw_run sorts its argument; keep `pairs` in caller order for the copy-back
  VCMI_TRY(dtw_run(order, D, fstep, bstep, sc.codes, sc.dpairs, nullptr));
  VCMI_HIP(hipDeviceSynchronize());
  std::vector<int64_t> hpath(std::max<size_t>(npath, 1));
  VCMI_HIP(hipMemcpy(hpath.data(), sc.paths.p, npath * 8, hipMemcpyDeviceToHost));
  po = 0;
  to = 0;
  for (int64_t p = 0; p < n; ++p) {
    if (path && path[p] && T[p] > 0) memcpy(path[p], &hpath[po], sizeof(int64_t) * T[p]);
    po += (size_t)T[p];
    if (newtgt && newtgt[p]) {
      if (T[p] > 0) VCMI_HIP(hipMemcpy(newtgt[p], sc.newtgt.p + to, sizeof(double) * D * S[p], hipMemcpyDeviceToHost));
      else memset(newtgt[p], 0, sizeof(double) * D * S[p]);
    }
    to += (size_t)D * S[p];
  }
  if (tables && T[0] > 0) {
    if (cost) VCMI_HIP(hipMemcpy(cost, sc.cost.p, sizeof(double) * S[0] * (T[0] + 1), hipMemcpyDeviceToHost));
    if (bp) VCMI_HIP(hipMemcpy(bp, sc.bp.p, sizeof(int64_t) * S[0] * (T[0] + 1), hipMemcpyDeviceToHost));
  }
  return VCMI_OK;
}

}  // namespace vcmi

using namespace vcmi;

extern "C" int vcmi_dtw_fit(const double *tmpl, int64_t S, const double *seq, int64_t T, int D, int fstep, int bstep,
                            int64_t *path, double *costtable, int64_t *backpointer) {
  int64_t *paths[1] = {path};
  const double *tp[1] = {tmpl}, *sp[1] = {seq};
  return dtw_host_batch(1, tp, &S, sp, &T, D, fstep, bstep, paths, costtable, backpointer, nullptr);
}

extern "C" int vcmi_dtw_fit_batch(int64_t n, const double *const *tmpl, const int64_t *S, const double *const *seq,
                                  const int64_t *T, int D, int fstep, int bstep, int64_t *const *path) {
  if (n > 0 && !path) return fail(VCMI_ERR_ARG, "vcmi_dtw_fit_batch: NULL path array");
  return dtw_host_batch(n, tmpl, S, seq, T, D, fstep, bstep, path, nullptr, nullptr, nullptr);
}

extern "C" int vcmi_dtw_fit_batch_dev(int64_t n, const double *feats, const int64_t *tmpl_off, const int64_t *S,
                                      const int64_t *seq_off, const int64_t *T, int D, int fstep, int bstep,
                                      int64_t *paths, const int64_t *path_off, void *stream) {
  if (n < 0) return fail(VCMI_ERR_ARG, "vcmi_dtw_fit_batch_dev: negative batch size");
  if (n == 0) return VCMI_OK;
  if (!feats || !tmpl_off || !S || !seq_off || !T || !paths || !path_off)
    return fail(VCMI_ERR_ARG, "vcmi_dtw_fit_batch_dev: NULL argument");
  std::vector<DtwPair> pairs(n);
  for (int64_t p = 0; p < n; ++p) {
    if (S[p] < 1 || T[p] < 0 || S[p] > INT32_MAX || T[p] > INT32_MAX) return fail(VCMI_ERR_DIM, "DTW: bad sequence length");
    pairs[p] = DtwPair{feats + tmpl_off[p], feats + seq_off[p], paths + path_off[p], nullptr, nullptr, nullptr, nullptr,
                       (int32_t)S[p], (int32_t)T[p]};
  }
  DtwScratch &sc = scratch();
  return dtw_run(pairs, D, fstep, bstep, sc.codes, sc.dpairs, as_stream(stream));
}

extern "C" int vcmi_align(const double *src, int64_t S, const double *tgt, int64_t T, int D, double *newtgt,
                          int64_t *path) {
  if (!newtgt) return fail(VCMI_ERR_ARG, "vcmi_align: NULL output");
  int64_t *paths[1] = {path};
  const double *tp[1] = {src}, *sp[1] = {tgt};
  double *nt[1] = {newtgt};
  return dtw_host_batch(1, tp, &S, sp, &T, D, /*fstep=*/0, /*bstep=*/2, paths, nullptr, nullptr, nt);   // src/align.jl:16
}

extern "C" int vcmi_align_batch(int64_t n, const double *const *src, const int64_t *S, const double *const *tgt,
                                const int64_t *T, int D, double *const *newtgt) {
  if (n > 0 && !newtgt) return fail(VCMI_ERR_ARG, "vcmi_align_batch: NULL output array");
  return dtw_host_batch(n, src, S, tgt, T, D, 0, 2, nullptr, nullptr, nullptr, newtgt);
}
