// dataset.hip -- the steps either side of DTW and the E-step, MI355X (gfx950).
//
// Replaces   align_mcep(src, tgt, alpha, fftlen; threshold, remove_silence)   reference src/align.jl:38-55
//            mc2e (MelGeneralizedCepstrums, third party; call site src/align.jl:48)
//            ParallelDataset(path; joint=true, diff, ignore0th, add_delta).X   reference src/datasets.jl:52-98
//
// so that alignment -> silence removal -> joint feature matrix -> E-step stays on the device: the aligned pairs never
// go back to the host (the reference writes them to *_parallel.jld files and reads them again in train_gmm.jl).
//
// mc2e(mc, alpha, len) = sum_n h[n]^2 with h = c2ir(freqt(mc, len-1, -alpha), len):
//   * freqt is LINEAR in its input: g = F c with a fixed (len x D) matrix F(alpha), built once on the host by running
//     the published recursion on unit vectors; on the device it is one dense product per frame.
//   * c2ir, h[0] = exp(g[0]), h[n] = (1/n) sum_{k=1..n} k g[k] h[n-k], is sequential in n; one wave per frame splits
//     every sum over its 64 lanes (g and h in LDS) and reduces with cross-lane adds.
#include "vcmi_common.hpp"
#include "hostpipe.hpp"

#include <algorithm>
#include <cmath>
#include <vector>

namespace vcmi {

int dtw_align_on_device(int64_t n, const double *const *src, const int64_t *S, const double *const *tgt, const int64_t *T,
                        int D, const double **d_feats, const double **d_newtgt, std::vector<int64_t> &src_off,
                        std::vector<int64_t> &nt_off);

static constexpr int kMc2eWaves = 4;   // frames per workgroup pass

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// mc: frames of D doubles at mc + frame_off[f]... here a dense (D,nfr) block; Ft: [D][len] (F transposed: row i = freqt of
// unit vector i); e: (nfr)
__global__ void __launch_bounds__(64 * kMc2eWaves)
mc2e_kernel(const double *__restrict__ mc, int D, int64_t nfr, const double *__restrict__ Ft, int len, double *__restrict__ e) {
  extern __shared__ double msm[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double *g = msm + (size_t)wave * 2 * len;   // k * g[k]
  double *h = g + len;
  for (int64_t f = (int64_t)blockIdx.x * kMc2eWaves + wave; f < nfr; f += (int64_t)gridDim.x * kMc2eWaves) {
    const double *c = mc + (size_t)D * f;
    // freqt: g = F c
    double g0 = 0.0;
    for (int j = lane; j < len; j += 64) {
      double s = 0.0;
      for (int i = 0; i < D; ++i) s = fma(Ft[(size_t)i * len + j], c[i], s);
      g[j] = (double)j * s;                    // the recursion only needs k g[k]
      if (j == 0) g0 = s;
    }
    if (lane == 0) h[0] = exp(g0);
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // c2ir
    for (int n = 1; n < len; ++n) {
      double s = 0.0;
      for (int k = 1 + lane; k <= n; k += 64) s = fma(g[k], h[n - k], s);
      s = wave_sum(s);
      if (lane == 0) h[n] = s / (double)n;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
    }
    double en = 0.0;
    for (int j = lane; j < len; j += 64) en = fma(h[j], h[j], en);
    en = wave_sum(en);
    if (lane == 0) e[f] = en;
    __builtin_amdgcn_wave_barrier();
  }
}

// F transposed ([D][len]) on the host: column i of F = freqt(unit vector i, len-1, -alpha) (SPTK recursion)
static void build_freqt_matrix(int D, int len, double alpha, std::vector<double> &Ft) {
  const int m2 = len - 1;
  const double a = -alpha, b = 1.0 - a * a;
  Ft.assign((size_t)D * len, 0.0);
  std::vector<double> g(len), d(len);
  for (int u = 0; u < D; ++u) {
    std::fill(g.begin(), g.end(), 0.0);
    for (int i = D - 1; i >= 0; --i) {
      d = g;
      g[0] = (i == u ? 1.0 : 0.0) + a * d[0];
      if (m2 >= 1) g[1] = b * d[0] + a * d[1];
      for (int j = 2; j <= m2; ++j) g[j] = d[j - 1] + a * (d[j] - g[j - 1]);
    }
    std::copy(g.begin(), g.end(), Ft.begin() + (size_t)u * len);
  }
}

struct DatasetScratch {
  DevBuf<double> Ft, mc, e, out;
  DevBuf<int> idx, cnt;
  DevBuf<int64_t> meta;
  int ft_D = 0, ft_len = 0;
  double ft_alpha = 0.0;
};
static DatasetScratch &dscratch() {
  static thread_local DatasetScratch s;
  return s;
}

static int ensure_freqt(DatasetScratch &sc, int D, int len, double alpha) {
  if (sc.Ft.p && sc.ft_D == D && sc.ft_len == len && sc.ft_alpha == alpha) return VCMI_OK;
  std::vector<double> Ft;
  build_freqt_matrix(D, len, alpha, Ft);
  VCMI_TRY(sc.Ft.reserve(Ft.size()));
  VCMI_TRY(upload_now(sc.Ft.p, Ft.data(), Ft.size() * 8));
  sc.ft_D = D;
  sc.ft_len = len;
  sc.ft_alpha = alpha;
  return VCMI_OK;
}

static int mc2e_device(DatasetScratch &sc, const double *dmc, int D, int64_t nfr, double alpha, int len, double *de,
                       hipStream_t st) {
  if (nfr == 0) return VCMI_OK;
  VCMI_TRY(ensure_freqt(sc, D, len, alpha));
  const size_t shmem = (size_t)kMc2eWaves * 2 * len * sizeof(double);
  if (shmem > 150 * 1024) return fail(VCMI_ERR_ARG, "mc2e: fft length %d too large", len);
  VCMI_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(mc2e_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)shmem));
  const int64_t blocks = std::min<int64_t>((nfr + kMc2eWaves - 1) / kMc2eWaves, 256 * 8);
  hipLaunchKernelGGL(mc2e_kernel, dim3((unsigned)blocks), dim3(64 * kMc2eWaves), shmem, st, dmc, D, nfr, sc.Ft.p, len, de);
  VCMI_HIP(hipGetLastError());
  return VCMI_OK;
}

// per pair: kept source columns in increasing order -> idx[idx_off + k], count -> cnt[p]
// (keep = log(e) > threshold, src/align.jl:49-51; remove_silence == 0 keeps everything)
__global__ void __launch_bounds__(256)
keep_index_kernel(const double *__restrict__ e, const int64_t *__restrict__ col_off, const int64_t *__restrict__ ncols,
                  double threshold, int remove_silence, int *__restrict__ idx, int *__restrict__ cnt) {
  const int p = blockIdx.x;
  const int64_t off = col_off[p];
  const int S = (int)ncols[p];
  __shared__ int base;
  __shared__ int wsum[4];
  if (threadIdx.x == 0) base = 0;
  __syncthreads();
  for (int i0 = 0; i0 < S; i0 += 256) {
    const int i = i0 + threadIdx.x;
    const bool keep = i < S && (!remove_silence || log(e[off + i]) > threshold);
    const unsigned long long m = __ballot(keep);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) wsum[wave] = __popcll(m);
    __syncthreads();
    int pre = base;
    for (int w = 0; w < wave; ++w) pre += wsum[w];
    if (keep) idx[off + pre + __popcll(m & ((1ull << lane) - 1))] = i;
    __syncthreads();
    if (threadIdx.x == 0) base += wsum[0] + wsum[1] + wsum[2] + wsum[3];
    __syncthreads();
  }
  if (threadIdx.x == 0) cnt[p] = base;
}

// one utterance of ParallelDataset(joint=true), src/datasets.jl:60-84, from the kept columns: drop row 1, push_delta on
// the COMPACTED sequence, tgt - src, vcat; frames of pair p go to out columns out_off[p] ...
__global__ void __launch_bounds__(256)
joint_features_kernel(const double *__restrict__ feats, const double *__restrict__ newtgt, const int64_t *__restrict__ src_off,
                      const int64_t *__restrict__ nt_off, const int64_t *__restrict__ col_off, const int *__restrict__ idx,
                      const int *__restrict__ cnt, const int64_t *__restrict__ out_off, int D, int ignore0th, int add_delta,
                      int diff, double *__restrict__ out) {
  const int p = blockIdx.x;
  const int n = cnt[p];
  const int r0 = ignore0th ? 1 : 0, Ds = D - r0, Dh = Ds * (add_delta ? 2 : 1), Dj = 2 * Dh;
  const double *sx = feats + src_off[p], *tx = newtgt + nt_off[p];
  const int *id = idx + col_off[p];
  double *o = out + (size_t)Dj * out_off[p];
  for (int64_t e = (int64_t)blockIdx.y * 256 + threadIdx.x; e < (int64_t)n * Ds; e += (int64_t)gridDim.y * 256) {
    const int k = (int)(e / Ds), d = (int)(e - (int64_t)k * Ds);
    const int c = id[k];
    const double xs = sx[(size_t)D * c + r0 + d], xt = tx[(size_t)D * c + r0 + d];
    double *col = o + (size_t)Dj * k;
    col[d] = xs;
    col[Dh + d] = diff ? xt - xs : xt;
    if (add_delta) {                                          // push_delta, src/datasets.jl:6-13
      double ds = xs, dt = xt;
      if (k >= 1 && k + 1 < n) {
        const int cm = id[k - 1], cp = id[k + 1];
        ds = -0.5 * sx[(size_t)D * cm + r0 + d] + 0.5 * sx[(size_t)D * cp + r0 + d];
        dt = -0.5 * tx[(size_t)D * cm + r0 + d] + 0.5 * tx[(size_t)D * cp + r0 + d];
      }
      col[Ds + d] = ds;
      col[Dh + Ds + d] = diff ? dt - ds : dt;
    }
  }
}

}  // namespace vcmi

using namespace vcmi;

// mc2e(mc, alpha, len) for every column of mc (D,T) -> e (T); call site src/align.jl:48
extern "C" int vcmi_mc2e(const double *mc, int D, int64_t T, double alpha, int fftlen, double *e) {
  if (!mc || !e) return fail(VCMI_ERR_ARG, "vcmi_mc2e: NULL argument");
  if (D < 1 || T < 0 || fftlen < 2) return fail(VCMI_ERR_DIM, "vcmi_mc2e: D=%d T=%lld fftlen=%d invalid", D, (long long)T, fftlen);
  if (T == 0) return VCMI_OK;
  VCMI_TRY(check_device());
  DatasetScratch &sc = dscratch();
  VCMI_TRY(sc.mc.reserve((size_t)D * T));
  VCMI_TRY(sc.e.reserve((size_t)T));
  VCMI_TRY(upload_now(sc.mc.p, mc, sizeof(double) * D * T));
  VCMI_TRY(mc2e_device(sc, sc.mc.p, D, T, alpha, fftlen, sc.e.p, nullptr));
  VCMI_HIP(hipMemcpy(e, sc.e.p, sizeof(double) * T, hipMemcpyDeviceToHost));
  return VCMI_OK;
}

// The pipeline  align_mcep for every pair -> ParallelDataset(joint=true).X , left on the device.
//   src[p] (D,S_p), tgt[p] (D,T_p) host matrices (mel-cepstra with the 0-th coefficient in row 1).
//   do_align = 0: the pairs are already aligned (S_p == T_p), only silence removal / assembly happen.
//   dXY: DEVICE buffer of at least Dj * sum(S_p) doubles, Dj = 2 (D - ignore0th) (1 + add_delta); receives the joint
//   features (Dj, *nframes) column-major, pairs in order; counts[p] (optional, host) = frames kept of pair p.
extern "C" int vcmi_parallel_dataset_dev(int64_t n, const double *const *src, const int64_t *S, const double *const *tgt,
                                         const int64_t *T, int D, int do_align, double alpha, int fftlen, double threshold,
                                         int remove_silence, int ignore0th, int add_delta, int diff, double *dXY,
                                         int64_t capacity_frames, int64_t *nframes, int64_t *counts) {
  if (n < 0 || !nframes) return fail(VCMI_ERR_ARG, "vcmi_parallel_dataset_dev: bad argument");
  *nframes = 0;
  if (n == 0) return VCMI_OK;
  if (!src || !S || !tgt || !T || !dXY) return fail(VCMI_ERR_ARG, "vcmi_parallel_dataset_dev: NULL argument");
  if (D < 1 + (ignore0th ? 1 : 0)) return fail(VCMI_ERR_DIM, "vcmi_parallel_dataset_dev: feature dimension %d too small", D);
  if (remove_silence && fftlen < 2) return fail(VCMI_ERR_ARG, "vcmi_parallel_dataset_dev: fft length %d invalid", fftlen);
  int64_t total = 0;
  for (int64_t p = 0; p < n; ++p) {
    if (S[p] < 1 || T[p] < 1 || S[p] > INT32_MAX) return fail(VCMI_ERR_DIM, "vcmi_parallel_dataset_dev: empty utterance");
    if (!do_align && S[p] != T[p])
      return fail(VCMI_ERR_DIM, "vcmi_parallel_dataset_dev: pair %lld is not aligned (%lld vs %lld frames)", (long long)p + 1,
                  (long long)S[p], (long long)T[p]);
    total += S[p];
  }
  if (capacity_frames < total) return fail(VCMI_ERR_ARG, "vcmi_parallel_dataset_dev: output needs room for %lld frames", (long long)total);
  VCMI_TRY(check_device());
  DatasetScratch &sc = dscratch();
  // (1) align(src, tgt) on the device, src/align.jl:45; results stay in the DTW scratch
  const double *dfeats = nullptr, *dnewtgt = nullptr;
  std::vector<int64_t> src_off, nt_off;
  if (do_align) {
    VCMI_TRY(dtw_align_on_device(n, src, S, tgt, T, D, &dfeats, &dnewtgt, src_off, nt_off));
  } else {
    std::vector<double> hs((size_t)D * total), ht((size_t)D * total);
    src_off.resize(n);
    nt_off.resize(n);
    int64_t o = 0;
    for (int64_t p = 0; p < n; ++p) {
      memcpy(&hs[o], src[p], sizeof(double) * D * S[p]);
      memcpy(&ht[o], tgt[p], sizeof(double) * D * S[p]);
      src_off[p] = nt_off[p] = o;
      o += (int64_t)D * S[p];
    }
    VCMI_TRY(sc.mc.reserve(2 * hs.size()));
    VCMI_TRY(upload_now(sc.mc.p, hs.data(), hs.size() * 8));
    VCMI_TRY(upload_now(sc.mc.p + hs.size(), ht.data(), ht.size() * 8));
    dfeats = sc.mc.p;
    dnewtgt = sc.mc.p + hs.size();
  }
  // (2) frame energies of the source: the sources are not contiguous in the DTW scratch (src_p, tgt_p interleaved), so
  //     one launch per pair would do; instead gather offsets: e is laid out pair after pair (col_off)
  std::vector<int64_t> meta((size_t)5 * n);   // src_off | nt_off | col_off | ncols | out_off
  int64_t co = 0;
  for (int64_t p = 0; p < n; ++p) {
    meta[p] = src_off[p];
    meta[n + p] = nt_off[p];
    meta[2 * n + p] = co;
    meta[3 * n + p] = S[p];
    co += S[p];
  }
  VCMI_TRY(sc.e.reserve((size_t)total));
  VCMI_TRY(sc.idx.reserve((size_t)total));
  VCMI_TRY(sc.cnt.reserve((size_t)n));
  VCMI_TRY(sc.meta.reserve(meta.size()));
  if (remove_silence)
    for (int64_t p = 0; p < n; ++p)
      VCMI_TRY(mc2e_device(sc, dfeats + src_off[p], D, S[p], alpha, fftlen, sc.e.p + meta[2 * n + p], nullptr));
  VCMI_TRY(upload_now(sc.meta.p, meta.data(), sizeof(int64_t) * 4 * n));
  hipLaunchKernelGGL(keep_index_kernel, dim3((unsigned)n), dim3(256), 0, nullptr, sc.e.p, sc.meta.p + 2 * n, sc.meta.p + 3 * n,
                     threshold, remove_silence, sc.idx.p, sc.cnt.p);
  VCMI_HIP(hipGetLastError());
  std::vector<int> hcnt((size_t)n);
  VCMI_HIP(hipMemcpy(hcnt.data(), sc.cnt.p, sizeof(int) * n, hipMemcpyDeviceToHost));
  int64_t oo = 0;
  for (int64_t p = 0; p < n; ++p) {
    meta[4 * n + p] = oo;
    oo += hcnt[p];
    if (counts) counts[p] = hcnt[p];
  }
  VCMI_TRY(upload_now(sc.meta.p + 4 * n, meta.data() + 4 * n, sizeof(int64_t) * n));
  // (3) joint features straight into the caller's device matrix
  hipLaunchKernelGGL(joint_features_kernel, dim3((unsigned)n, 8), dim3(256), 0, nullptr, dfeats, dnewtgt, sc.meta.p,
                     sc.meta.p + n, sc.meta.p + 2 * n, sc.idx.p, sc.cnt.p, sc.meta.p + 4 * n, D, ignore0th, add_delta, diff, dXY);
  VCMI_HIP(hipGetLastError());
  VCMI_HIP(hipDeviceSynchronize());
  *nframes = oo;
  return VCMI_OK;
}

// align_mcep(src, tgt, alpha, fftlen; threshold, remove_silence) -- src/align.jl:38-55.  src (D,S), tgt (D,T) host matrices;
// src_out / newtgt_out (D, up to S) receive the kept columns, *ncols their number.
extern "C" int vcmi_align_mcep(const double *src, int64_t S, const double *tgt, int64_t T, int D, double alpha, int fftlen,
                               double threshold, int remove_silence, double *src_out, double *newtgt_out, int64_t *ncols) {
  if (!src || !tgt || !src_out || !newtgt_out || !ncols) return fail(VCMI_ERR_ARG, "vcmi_align_mcep: NULL argument");
  DatasetScratch &sc = dscratch();
  VCMI_TRY(check_device());
  if (S < 1) return fail(VCMI_ERR_DIM, "vcmi_align_mcep: empty source");
  // the joint matrix with ignore0th = add_delta = diff = 0 is [src ; newtgt] of the kept columns
  VCMI_TRY(sc.out.reserve((size_t)2 * D * S));
  int64_t nfr = 0;
  const double *sp[1] = {src}, *tp[1] = {tgt};
  VCMI_TRY(vcmi_parallel_dataset_dev(1, sp, &S, tp, &T, D, 1, alpha, fftlen, threshold, remove_silence, 0, 0, 0, sc.out.p, S, &nfr,
                                     nullptr));
  std::vector<double> h((size_t)2 * D * nfr);
  if (nfr > 0) VCMI_HIP(hipMemcpy(h.data(), sc.out.p, h.size() * 8, hipMemcpyDeviceToHost));
  for (int64_t k = 0; k < nfr; ++k) {
    memcpy(src_out + (size_t)D * k, &h[(size_t)2 * D * k], sizeof(double) * D);
    memcpy(newtgt_out + (size_t)D * k, &h[(size_t)2 * D * k + D], sizeof(double) * D);
  }
  *ncols = nfr;
  return VCMI_OK;
}

// GVDataset from in-memory feature matrices -- src/datasets.jl:134-183.  Per utterance: drop row 1 (ignore0th), push_delta
// (add_delta; the same rule as vcmi_push_delta: delta_t = (x_{t+1} - x_{t-1}) / 2 for 2 <= t <= T-1, the static value at
// both ends), var(tgt, 2) = sum (x - mean)^2 / (T - 1) per row; utterances with a NaN in it (T = 1) are skipped.
extern "C" int vcmi_gv_dataset(int64_t n, const double *const *fm, const int64_t *T, int D, int ignore0th, int add_delta,
                               double *out, int64_t *nkept) {
  if (n < 0 || (n > 0 && (!fm || !T || !out)) || !nkept) return fail(VCMI_ERR_ARG, "vcmi_gv_dataset: bad argument");
  const int r0 = ignore0th ? 1 : 0, Ds = D - r0, Dout = Ds * (add_delta ? 2 : 1);
  if (Ds < 1) return fail(VCMI_ERR_DIM, "vcmi_gv_dataset: feature dimension %d too small", D);
  for (int64_t i = 0; i < n; ++i)
    if (!fm[i] || T[i] < 0) return fail(VCMI_ERR_ARG, "vcmi_gv_dataset: utterance %lld invalid", (long long)(i + 1));
  std::vector<double> gv((size_t)Dout * n);
  std::vector<char> keep((size_t)n, 0);
  host_parallel_for(n, 1, [&](int64_t lo, int64_t hi) {
    for (int64_t i = lo; i < hi; ++i) {
      const double *x = fm[i];
      const int64_t Ti = T[i];
      double *g = &gv[(size_t)Dout * i];
      // value (row k of [static; delta], frame t) without materialising the matrix
      auto val = [&](int k, int64_t t) -> double {
        if (k < Ds) return x[r0 + k + (size_t)D * t];
        const int d = k - Ds;
        if (t == 0 || t + 1 >= Ti) return x[r0 + d + (size_t)D * t];
        return -0.5 * x[r0 + d + (size_t)D * (t - 1)] + 0.5 * x[r0 + d + (size_t)D * (t + 1)];
      };
      bool ok = Ti >= 2;                      // var of fewer than two frames is NaN in the reference: skipped
      for (int k = 0; k < Dout && ok; ++k) {
        double m = 0.0;
        for (int64_t t = 0; t < Ti; ++t) m += val(k, t);
        m /= (double)Ti;
        double s = 0.0;
        for (int64_t t = 0; t < Ti; ++t) {
          const double dv = val(k, t) - m;
          s += dv * dv;
        }
        g[k] = s / (double)(Ti - 1);
        if (std::isnan(g[k])) ok = false;
      }
      keep[i] = ok ? 1 : 0;
    }
  });
  int64_t c = 0;
  for (int64_t i = 0; i < n; ++i)
    if (keep[i]) {
      memcpy(out + (size_t)Dout * c, &gv[(size_t)Dout * i], sizeof(double) * Dout);
      ++c;
    }
  *nkept = c;
  return VCMI_OK;
}
