// microbench_fetch.hip -- what FETCH_SIZE / WRITE_SIZE (rocprofv3 --pmc) report for KNOWN traffic, by access pattern.
// MI355X_MICROARCH.md calibrates the counters for 16-byte-per-lane streaming only (FETCH_SIZE reads half, WRITE_SIZE exact)
// and calls every other width uncalibrated; the kernels of this library read with other patterns, so each pattern that
// matters is replayed here on a buffer of known size (every byte read / written exactly once, 320 MB like the headline
// workload's x and y), and tools/fetch_calibration.py divides the counter by the bytes:
//   read_x4      16 bytes per lane, consecutive lanes consecutive addresses (the guide's case)
//   read_x2      8 bytes per lane, consecutive
//   read_rows8   the headline kernel's x fragments: a 16-frame tile of (T, 40) doubles, lane (col, grp) reads 8 bytes of
//                frame `col` at dimension 4 ks + grp, ks = 0..9 (16 rows 320 bytes apart per instruction, 32 bytes of each)
//   read_scalar  s_load_dwordx8 of wave-uniform addresses (the DTW kernel's sequence columns)
//   write_x2     8 bytes per lane consecutive;  write_rows8: the headline kernel's y stores (same geometry as read_rows8)
//   write_x4     16 bytes per lane
// build: hipcc -O3 --offload-arch=gfx950 tools/microbench_fetch.hip -o tools/microbench_fetch
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double d2 __attribute__((ext_vector_type(2)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void read_x4(const d2 *__restrict__ p, size_t n2, double *sink) {
  double s = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (size_t)gridDim.x * blockDim.x) {
    const d2 v = __builtin_nontemporal_load(p + i);
    s += v.x + v.y;
  }
  if (s == 12345.678) *sink = s;
}
__global__ void read_x2(const double *__restrict__ p, size_t n, double *sink) {
  double s = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += p[i];
  if (s == 12345.678) *sink = s;
}
// one wave per 16-frame tile; D = 40 doubles per frame
__global__ void read_rows8(const double *__restrict__ X, size_t T, double *sink) {
  const int lane = threadIdx.x & 63, col = lane & 15, grp = lane >> 4;
  const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
  double s = 0;
  for (size_t tile = wave; tile * 16 < T; tile += nwaves) {
    const double *row = X + (tile * 16 + col) * 40 + grp;
#pragma unroll
    for (int ks = 0; ks < 10; ++ks) s += row[4 * ks];
  }
  if (s == 12345.678) *sink = s;
}
__global__ void read_scalar(const double *__restrict__ p, size_t n, double *sink) {
  // every wave reads its own 64-byte pieces through the scalar cache
  const size_t wave = __builtin_amdgcn_readfirstlane((int)((((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6)));
  const size_t nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
  double s = 0;
  for (size_t i = wave * 8; i + 8 <= n; i += nwaves * 8) {
    const double *q = p + i;              // wave-uniform address of read-only memory: the compiler emits s_load_dwordx16
    double a = q[0], b = q[1], c = q[2], d = q[3], e = q[4], f = q[5], g = q[6], h = q[7];
    s += a + b + c + d + e + f + g + h;
  }
  if (s == 12345.678) *sink = s;
}
__global__ void write_x2(double *__restrict__ p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (double)i;
}
__global__ void write_x4(d2 *__restrict__ p, size_t n2) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (size_t)gridDim.x * blockDim.x) {
    d2 v = {(double)i, 1.0};
    p[i] = v;
  }
}
__global__ void write_rows8(double *__restrict__ Y, size_t T) {
  const int lane = threadIdx.x & 63, col = lane & 15, grp = lane >> 4;
  const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
  for (size_t tile = wave; tile * 16 < T; tile += nwaves) {
    double *row = Y + (tile * 16 + col) * 40 + grp;
#pragma unroll
    for (int ks = 0; ks < 10; ++ks) row[4 * ks] = (double)ks;
  }
}

int main() {
  const size_t T = 1000000, n = T * 40;          // 320 MB
  double *buf, *sink;
  CHECK(hipMalloc(&buf, n * 8));
  CHECK(hipMalloc(&sink, 8));
  CHECK(hipMemset(buf, 0, n * 8));
  const dim3 grid(4096), block(256);
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(read_x4, grid, block, 0, 0, reinterpret_cast<const d2 *>(buf), n / 2, sink);
    hipLaunchKernelGGL(read_x2, grid, block, 0, 0, buf, n, sink);
    hipLaunchKernelGGL(read_rows8, grid, block, 0, 0, buf, T, sink);
    hipLaunchKernelGGL(read_scalar, grid, block, 0, 0, buf, n, sink);
    hipLaunchKernelGGL(write_x2, grid, block, 0, 0, buf, n);
    hipLaunchKernelGGL(write_x4, grid, block, 0, 0, reinterpret_cast<d2 *>(buf), n / 2);
    hipLaunchKernelGGL(write_rows8, grid, block, 0, 0, buf, T);
  }
  CHECK(hipDeviceSynchronize());
  printf("bytes per kernel: %zu\n", n * 8);
  return 0;
}
