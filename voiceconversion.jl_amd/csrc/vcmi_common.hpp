// vcmi_common.hpp -- status/error plumbing shared by the libvcmi translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <new>
#include <vector>
#include "../../include/vcmi.h"

namespace vcmi {

// thread-local message behind vcmi_last_error()
char *error_buffer();
int fail(int code, const char *fmt, ...);

#define VCMI_HIP(expr)                                                                                   \
  do {                                                                                                   \
    hipError_t e_ = (expr);                                                                              \
    if (e_ != hipSuccess)                                                                                \
      return vcmi::fail(e_ == hipErrorOutOfMemory ? VCMI_ERR_OOM : VCMI_ERR_HIP, "%s failed: %s (%s:%d)", \
                        #expr, hipGetErrorString(e_), __FILE__, __LINE__);                               \
  } while (0)

#define VCMI_TRY(expr)            \
  do {                            \
    int s_ = (expr);              \
    if (s_ != VCMI_OK) return s_; \
  } while (0)

// Device buffer with RAII; allocation failures surface as status codes through alloc().
template <typename T>
struct DevBuf {
  T *p = nullptr;
  size_t n = 0;
  DevBuf() = default;
  DevBuf(const DevBuf &) = delete;
  DevBuf &operator=(const DevBuf &) = delete;
  ~DevBuf() { release(); }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    n = 0;
  }
  int alloc(size_t count) {
    release();
    if (count == 0) return VCMI_OK;
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&p), count * sizeof(T));
    if (e != hipSuccess) {
      p = nullptr;
      return fail(VCMI_ERR_OOM, "hipMalloc of %zu bytes failed: %s", count * sizeof(T), hipGetErrorString(e));
    }
    n = count;
    return VCMI_OK;
  }
  // grow-only scratch
  int reserve(size_t count) { return count <= n ? VCMI_OK : alloc(count); }
};

int check_device();  // VCMI_ERR_NO_DEVICE when no HIP device is visible

// Orders the calls that share one thread-local device workspace.  The `_dev` entry points run on whatever stream the
// caller passes; two calls of one host thread on DIFFERENT streams would otherwise overwrite each other's operands and
// partial results while the first call's kernels still run.  enter(): the stream waits for the last kernel of the
// previous call; leave(): marks the end of this call's work.  (Same stream twice: the wait is a no-op.)
struct StreamOrder {
  hipEvent_t last_use = nullptr;
  int device = -1;
  StreamOrder() = default;
  StreamOrder(const StreamOrder &) = delete;
  StreamOrder &operator=(const StreamOrder &) = delete;
  ~StreamOrder() {
    if (last_use) (void)hipEventDestroy(last_use);
  }
  int enter(hipStream_t st) {
    int dev = 0;
    VCMI_HIP(hipGetDevice(&dev));
    if (!last_use || dev != device) {
      if (last_use) (void)hipEventDestroy(last_use);
      last_use = nullptr;
      VCMI_HIP(hipEventCreateWithFlags(&last_use, hipEventDisableTiming));
      device = dev;
      return VCMI_OK;   // a new event has nothing recorded
    }
    VCMI_HIP(hipStreamWaitEvent(st, last_use, 0));
    return VCMI_OK;
  }
  int leave(hipStream_t st) {
    if (last_use) VCMI_HIP(hipEventRecord(last_use, st));
    return VCMI_OK;
  }
};

// Test hook (vcmi_debug_force, not part of include/vcmi.h): forces the fallback kernels that a given shape would not
// select by itself, so that the parity tests cover them.  Nothing on the call path reads the environment.
enum : unsigned {
  kDbgTrajGeneric = 1u,      // runtime-D LDS-window trajectory solver instead of the MFMA-blocked one
  kDbgTrajGScalar = 2u,      // one-workgroup-per-frame g_t kernel instead of the MFMA one
  kDbgGvOneTeam = 4u,        // one-team GV kernel (the path of very long utterances)
  kDbgPredictTwoPass = 8u,   // (M,T) log-density matrix + argmax kernel instead of the in-kernel argmax
  kDbgEstepGeneric = 16u,    // generic diagonal E-step kernels instead of the MFMA one
  kDbgDtwTwoKernels = 32u,   // observation + recurrence kernels (the path of tables / D > 48 / wide windows) instead of the fused one
  kDbgDtwNoSegments = 256u,   // fused DTW: whole-length jobs (no column segments)
  kDbgDtwGridOrder = 512u,    // fused DTW: one workgroup per job in grid order instead of persistent workgroups drawing tickets
  kDbgTrajOneWgPerCu = 128u,  // blocked trajectory solver: one workgroup per CU even where two fit
  kDbgConvertNoGrouping = 2048u,  // fvconvert: frames in the caller's order (no grouping by nearest source mean)
  kDbgConvertShapeBroad = 4096u,  // fvconvert: the "broad model" loop (every whitening tile, one branch around the regression) whatever the model
  kDbgConvertShapePeaked = 8192u, // fvconvert: the "peaked model" loop (last whitening tile first, per-tile tests) whatever the model
  kDbgDtwWholeFirst = 32768u,     // fused DTW: whole-length jobs for the full rounds, segments for the rest (round 4 experiment: 3 % slower)
  kDbgEstepFullNoLists = 65536u,  // full-covariance statistics: every workgroup stages every frame of its segment (rounds 1-3) instead of its group's frame list
  kDbgDtwTwoSegments = 16384u,    // fused DTW: at most two column segments per strip (A/B of the traffic / balance trade)
  kDbgConvertShapeScreened = 262144u, // fvconvert: the four-row screening kernel (shape 3) on grouped calls whatever the model
  kDbgScreenRows4 = 2097152u, kDbgScreenRows2 = 524288u, kDbgScreenRows1 = 1048576u,   // shape 3: rows per mixture of the screen, read when a converter is created
  kDbgEstepWaveKernel = 4194304u,    // diagonal E-step, M > 64: the one-barrier-per-block experiment (estep_wave.hpp) instead of estep_mfma_kernel
  kDbgPredictNoScreen = 8388608u,    // predict / trajectory argmax: the early-exit kernel (MODE 3) also for long inputs, instead of grouping + screen
  kDbgPredictScreen = 16777216u,     // predict: grouping + screened arg-max on long inputs whatever the model
  kDbgScreenFp64 = 33554432u,        // fvconvert shape 3: the screen on FP64 MFMAs instead of the certified bf16-split one
  kDbgGroupKeyFp64 = 67108864u,      // grouping keys from FP64 MFMAs (gmmmap_group_key_kernel) instead of the bf16-split ones
  kDbgEstepNoSmall = 1024u,          // diagonal E-step, M <= 32: estep_mfma_kernel with shared tiles instead of estep_small_kernel (estep_small.hpp)
  kDbgEstepNoHard = 134217728u,      // diagonal E-step: every frame through estep_mfma_kernel (no hard-assignment path, estep_hard.hpp)
  kDbgConvertWideTiles = 131072u, // fvconvert: two frame tiles per wave (128-frame workgroups) also for calls of a few thousand frames
  kDbgPredictNoEarlyExit = 64u   // predict / trajectory argmax: every whitening tile of every mixture (MODE 2) instead of the early exit (MODE 3)
};
bool debug_flag(unsigned which);

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

}  // namespace vcmi
