// postf.hip -- SURVEY 8(f) rank 4: the steps either side of `vc` as DEVICE-RESIDENT operations, so that
//   src -> push_delta -> vc -> fvpostf!    (bin/vc.jl:75-82, src/datasets.jl:6-13, src/common.jl:7-63, src/gv.jl:10-15)
// is one upload and one download (round 5 had them as separate host-pointer calls: two more PCIe round trips).
//   vcmi_push_delta / _dev           [src; delta]: delta_t = (x_{t+1} - x_{t-1}) / 2 for 2 <= t <= T-1, the static value at t = 1, T
//   vcmi_variance_scaling / _dev     per row sqrt(sigma2 / var) (x - mean) + mean, Julia's corrected variance, in place allowed
//   vcmi_vc_frames_postf             vc(g::GMMMap, fm) with fvpostf! applied to the converted rows before the download
// (vcmi_vc_traj_postf lives in traj.hip beside vcmi_vc_traj.)  Everything is HBM-bound streaming: a frame is D contiguous
// doubles, lanes run along the features of consecutive frames (coalesced), every reduction has a fixed order.
#include "postf.hpp"
#include "gmmmap_handle.hpp"
#include "hostpipe.hpp"

namespace vcmi {

// ---- push_delta, src/datasets.jl:6-13 ------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
push_delta_kernel(const double *__restrict__ src, int64_t lds, int D, int64_t T, double *__restrict__ out, int64_t ldo) {
  const int64_t n = (int64_t)D * T;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
    const int64_t t = e / D;
    const int d = (int)(e - t * D);
    const double x = src[t * lds + d];
    out[t * ldo + d] = x;                                          // repmat(src, 2): the static rows, src/datasets.jl:8
    // t = 2:T-1 (1-based): -0.5 x_{t-1} + 0.5 x_{t+1}, src/datasets.jl:9-11; the first and the last frame keep the copy
    out[t * ldo + D + d] = (t >= 1 && t + 1 < T) ? -0.5 * src[(t - 1) * lds + d] + 0.5 * src[(t + 1) * lds + d] : x;
  }
}

int push_delta_device(const double *dsrc, int64_t lds, int D, int64_t T, double *dout, int64_t ldo, hipStream_t st) {
  if (T == 0) return VCMI_OK;
  const int64_t n = (int64_t)D * T;
  const unsigned grid = (unsigned)std::min<int64_t>((n + 255) / 256, 8192);
  hipLaunchKernelGGL(push_delta_kernel, dim3(grid), dim3(256), 0, st, dsrc, lds, D, T, dout, ldo);
  VCMI_HIP(hipGetLastError());
  return VCMI_OK;
}

// ---- fvpostf!(vs::VarianceScaling, src), src/gv.jl:10-15: three streaming passes (sums, centred squares, scale) over frame
// chunks, every reduction in a fixed order.
//   vs_partial_kernel: part[chunk][d] = sum over the chunk's frames of x (MODE 0) or (x - mean[d])^2 (MODE 1)
//   vs_final_kernel:   stat[d] = sum_chunk part[chunk][d] / denom
static constexpr int kVsChunk = 2048;   // frames per workgroup
template <int MODE>
__global__ void __launch_bounds__(256)
vs_partial_kernel(const double *__restrict__ src, int64_t lds, int D, int64_t T, const double *__restrict__ mean, double *__restrict__ part) {
  extern __shared__ double vred[];       // [NG][D]
  const int tid = threadIdx.x, NG = 256 / D, d = tid % D, g = tid / D;
  const int64_t f0 = (int64_t)blockIdx.x * kVsChunk, f1 = (f0 + kVsChunk < T) ? f0 + kVsChunk : T;
  double s = 0.0;
  if (g < NG) {
    const double m = MODE ? mean[d] : 0.0;
    for (int64_t t = f0 + g; t < f1; t += NG) {
      const double e = src[t * lds + d] - m;
      s = MODE ? fma(e, e, s) : s + e;
    }
    vred[g * D + d] = s;
  }
  __syncthreads();
  if (tid < D) {
    double a = 0.0;
    for (int k = 0; k < NG; ++k) a += vred[k * D + tid];
    part[(size_t)blockIdx.x * D + tid] = a;
  }
}

__global__ void __launch_bounds__(256)
vs_final_kernel(const double *__restrict__ part, int nchunks, int D, double denom, double *__restrict__ stat) {
  const int d = threadIdx.x;
  if (d >= D) return;
  double a = 0.0;
  for (int c = 0; c < nchunks; ++c) a += part[(size_t)c * D + d];
  stat[d] = a / denom;
}

// (src and out may be the same matrix: every element is read and written by the same thread)
__global__ void __launch_bounds__(256)
vs_scale_kernel(const double *src, int64_t lds, int D, int64_t T, const double *__restrict__ sigma2, const double *__restrict__ mean,
                const double *__restrict__ var, double *out, int64_t ldo) {
  const int64_t n = (int64_t)D * T;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
    const int64_t t = e / D;
    const int d = (int)(e - t * D);
    out[t * ldo + d] = sqrt(sigma2[d] / var[d]) * (src[t * lds + d] - mean[d]) + mean[d];
  }
}

struct VsScratch {
  DevBuf<double> part, stat;     // stat: [mean (D) | var (D) | sigma2 (D)]
  StreamOrder order;
};
static VsScratch &vs_scratch() {
  static thread_local VsScratch s;
  return s;
}

int variance_scaling_device(const double *dsrc, int64_t lds, int D, int64_t T, const double *sigma2_host, double *dout,
                            int64_t ldo, hipStream_t st) {
  if (D < 1 || D > 256 || T < 2)
    return fail(VCMI_ERR_DIM, "variance scaling: D=%d T=%lld unsupported (needs 1 <= D <= 256, T >= 2)", D, (long long)T);
  VsScratch &sc = vs_scratch();
  const int nchunks = (int)((T + kVsChunk - 1) / kVsChunk);
  VCMI_TRY(sc.part.reserve((size_t)nchunks * D));
  VCMI_TRY(sc.stat.reserve((size_t)3 * 256));
  VCMI_TRY(sc.order.enter(st));
  double *mean = sc.stat.p, *var = sc.stat.p + 256, *sig = sc.stat.p + 512;
  // (D doubles from pageable memory: the runtime stages them before returning, the caller's vector is free at once)
  VCMI_HIP(hipMemcpyAsync(sig, sigma2_host, sizeof(double) * D, hipMemcpyHostToDevice, st));
  const size_t shm = (size_t)(256 / D) * D * sizeof(double);
  hipLaunchKernelGGL(vs_partial_kernel<0>, dim3(nchunks), dim3(256), shm, st, dsrc, lds, D, T, mean, sc.part.p);
  hipLaunchKernelGGL(vs_final_kernel, dim3(1), dim3(256), 0, st, sc.part.p, nchunks, D, (double)T, mean);
  hipLaunchKernelGGL(vs_partial_kernel<1>, dim3(nchunks), dim3(256), shm, st, dsrc, lds, D, T, mean, sc.part.p);
  hipLaunchKernelGGL(vs_final_kernel, dim3(1), dim3(256), 0, st, sc.part.p, nchunks, D, (double)(T - 1), var);   // Julia's var
  hipLaunchKernelGGL(vs_scale_kernel, dim3(2048), dim3(256), 0, st, dsrc, lds, D, T, sig, mean, var, dout, ldo);
  VCMI_HIP(hipGetLastError());
  return sc.order.leave(st);
}

__global__ void __launch_bounds__(256)
copy_rows_kernel(const double *__restrict__ in, int64_t ldi, int r0, int nrows, int64_t T, double *__restrict__ out, int64_t ldo, int q0) {
  const int64_t n = (int64_t)nrows * T;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
    const int64_t t = e / nrows;
    const int r = (int)(e - t * nrows);
    out[t * ldo + q0 + r] = in[t * ldi + r0 + r];
  }
}

int copy_rows_device(const double *din, int64_t ldi, int r0, int nrows, int64_t T, double *dout, int64_t ldo, int q0, hipStream_t st) {
  if (T == 0 || nrows == 0) return VCMI_OK;
  const int64_t n = (int64_t)nrows * T;
  hipLaunchKernelGGL(copy_rows_kernel, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 8192)), dim3(256), 0, st, din, ldi, r0, nrows, T,
                     dout, ldo, q0);
  VCMI_HIP(hipGetLastError());
  return VCMI_OK;
}

// whole-matrix device buffers of the host-pointer entries below (grow-only, per thread)
struct PostfHostScratch {
  DevBuf<double> in, out;
};
static PostfHostScratch &host_scratch() {
  static thread_local PostfHostScratch s;
  return s;
}

}  // namespace vcmi

using namespace vcmi;

extern "C" int vcmi_push_delta_dev(const double *dsrc, int64_t lds, int D, int64_t T, double *dout, int64_t ldo, void *stream) {
  if (D < 1 || T < 0 || lds < D || ldo < 2 * (int64_t)D) return fail(VCMI_ERR_ARG, "vcmi_push_delta_dev: bad argument");
  if (T > 0 && (!dsrc || !dout)) return fail(VCMI_ERR_ARG, "vcmi_push_delta_dev: NULL argument");
  VCMI_TRY(check_device());
  return push_delta_device(dsrc, lds, D, T, dout, ldo, as_stream(stream));
}

// push_delta(src (D,T)) -> out (2D,T) on HOST matrices: host arithmetic -- three streaming passes over memory the caller already
// holds cost less than moving the matrix over PCIe and back (and the helper works without a device); the device-resident
// form above is the one the vc pipeline uses
extern "C" int vcmi_push_delta(const double *src, int D, int64_t T, double *out) {
  if (!src || !out || D < 1 || T < 0) return fail(VCMI_ERR_ARG, "vcmi_push_delta: bad argument");
  for (int64_t t = 0; t < T; ++t)
    for (int d = 0; d < D; ++d) {   // repmat(src, 2), src/datasets.jl:8
      out[d + (size_t)2 * D * t] = src[d + (size_t)D * t];
      out[D + d + (size_t)2 * D * t] = src[d + (size_t)D * t];
    }
  for (int64_t t = 1; t + 1 < T; ++t)   // t = 2:T-1, src/datasets.jl:9-11
    for (int d = 0; d < D; ++d)
      out[D + d + (size_t)2 * D * t] = -0.5 * src[d + (size_t)D * (t - 1)] + 0.5 * src[d + (size_t)D * (t + 1)];
  return VCMI_OK;
}

extern "C" int vcmi_variance_scaling_dev(const double *dsrc, int64_t lds, int D, int64_t T, const double *sigma2, double *dout,
                                         int64_t ldo, void *stream) {
  if (!dsrc || !sigma2 || !dout) return fail(VCMI_ERR_ARG, "vcmi_variance_scaling_dev: NULL argument");
  if (lds < D || ldo < D) return fail(VCMI_ERR_ARG, "vcmi_variance_scaling_dev: leading dimension below D");
  VCMI_TRY(check_device());
  return variance_scaling_device(dsrc, lds, D, T, sigma2, dout, ldo, as_stream(stream));
}

// fvpostf(vs::VarianceScaling, src) -- src/gv.jl:10-21.  src, out (D,T) host matrices (may alias), sigma2 (D).
extern "C" int vcmi_variance_scaling(const double *src, int D, int64_t T, const double *sigma2, double *out) {
  if (!src || !sigma2 || !out) return fail(VCMI_ERR_ARG, "vcmi_variance_scaling: NULL argument");
  if (D < 1 || D > 256 || T < 2)
    return fail(VCMI_ERR_DIM, "vcmi_variance_scaling: D=%d T=%lld unsupported (needs 1 <= D <= 256, T >= 2)", D, (long long)T);
  VCMI_TRY(check_device());
  PostfHostScratch &hs = host_scratch();
  VCMI_TRY(hs.in.reserve((size_t)D * T));
  VCMI_TRY(staged_upload(hs.in.p, src, sizeof(double) * D * T, nullptr));
  VCMI_TRY(variance_scaling_device(hs.in.p, D, D, T, sigma2, hs.in.p, D, nullptr));
  return staged_download(out, hs.in.p, sizeof(double) * D * T, nullptr);
}

// vc(g::GMMMap, fm) followed by fvpostf!(VarianceScaling(sigma2), converted[2:end, :]) -- src/common.jl:7-26, src/gv.jl:10-15:
// fm, out (D+1,T) host matrices, row 1 (power) passed through.  The post-filter needs the mean and variance of every converted
// row over the WHOLE matrix, so the matrix stays on the device between the two steps: one upload, one download.  (Runs on the
// calling thread's device: a device group does not shard it -- the statistics would need a collective for a 0.1 ms step.)
extern "C" int vcmi_vc_frames_postf(vcmi_gmmmap *g, const double *fm, int64_t T, const double *sigma2, double *out) {
  if (!sigma2) return vcmi_vc_frames(g, fm, T, out);
  if (!g) return fail(VCMI_ERR_ARG, "vcmi_vc_frames_postf: NULL handle");
  if (T < 0 || (T > 0 && (!fm || !out))) return fail(VCMI_ERR_ARG, "vcmi_vc_frames_postf: bad argument");
  if (T == 0) return VCMI_OK;
  if (T < 2) return fail(VCMI_ERR_DIM, "vcmi_vc_frames_postf: the variance of a one-frame matrix is undefined");
  VCMI_TRY(check_device());
  const int64_t ld = g->D + 1;
  const size_t bytes = sizeof(double) * (size_t)ld * T;
  PostfHostScratch &hs = host_scratch();
  VCMI_TRY(hs.in.reserve((size_t)ld * T));
  VCMI_TRY(hs.out.reserve((size_t)ld * T));
  VCMI_TRY(staged_upload(hs.in.p, fm, bytes, nullptr));
  VCMI_TRY(copy_rows_device(hs.in.p, ld, 0, 1, T, hs.out.p, ld, 0, nullptr));                       // power row kept, src/common.jl:23
  VCMI_TRY(gmmmap_convert_device(g, hs.in.p + 1, ld, T, hs.out.p + 1, ld, nullptr));
  VCMI_TRY(variance_scaling_device(hs.out.p + 1, ld, g->D, T, sigma2, hs.out.p + 1, ld, nullptr));
  return staged_download(out, hs.out.p, bytes, nullptr);
}
