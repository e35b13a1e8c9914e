// estep_path.hpp -- WHICH path a diagonal E-step takes (estep_hard.hpp: the hard-assignment path / estep_mfma_kernel alone) is decided
// from THIS call's data, on the device (round 6; included by estep.hip behind estep_hard.hpp).
//
// Round 5 kept the choice in thread-local state fed by what EARLIER calls had found (their soft counts, polled without
// waiting): the same inputs could take different paths -- different last bits -- depending on the history of the calling thread
// and on timing.  Now: the certified screen runs on a SAMPLE of 16 chunks spread over the call's frames (~20 us),
// estep_path_decide_kernel turns its histograms into control words in device memory, and every kernel of either path starts
// with a look at them.  The launch sequence is fixed (nothing waits for the GPU: consecutive E-steps still queue back to
// back), identical inputs give identical bits, and ranks of one all-reduce may take different paths on their own shards.
// vcmi_estep_set_path(VCMI_ESTEP_HARD / _SOFT) pins one for the calling thread.
// (The one-pass form of the hard-assignment path tried in this round -- per-workgroup register accumulators, no sort -- was
// slower than key -> sort -> sums: profiles/r06_ab/estep_onepass_experiments.txt.)
#pragma once
#include "estep_hard.hpp"

namespace vcmi {

// ctl (int64, device): [0] 1: the hard-assignment path runs / 0: every frame through estep_mfma_kernel; [1] frames of the
// "everything soft" launch (N or 0); [2] soft frames found by the hard-assignment path (estep_hard_gather_kernel)
enum { kCtlHard = 0, kCtlAllSoft = 1, kCtlNSoft = 2, kCtlLen = 4 };

// decision from the sample's histograms (hist[c][k], k = M: no owner): hard iff at most a quarter of the sample has no owner.
// force: -1 decide, 0 / 1 set.
// One workgroup of kDecideThreads (the histograms are nsample x MK ints -- 8 K at M = 128: 64 threads took 19 us over them,
// one dependent load after the other; the sums are integers, so any order gives the same decision).
constexpr int kDecideThreads = 1024;
__global__ void __launch_bounds__(kDecideThreads)
estep_path_decide_kernel(const int *__restrict__ hist, int nsample, int MK, int force, int64_t N, int64_t *__restrict__ ctl) {
  __shared__ int part[2][kDecideThreads / 64];
  int soft = 0, all = 0;
  if (force < 0) {
    for (int e = threadIdx.x; e < nsample * MK; e += kDecideThreads) {
      const int v = hist[e];
      all += v;
      soft += (e % MK == MK - 1) ? v : 0;
    }
#pragma unroll
    for (int sh = 1; sh < 64; sh <<= 1) {
      soft += __shfl_xor(soft, sh);
      all += __shfl_xor(all, sh);
    }
    if ((threadIdx.x & 63) == 0) {
      part[0][threadIdx.x >> 6] = soft;
      part[1][threadIdx.x >> 6] = all;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    if (force < 0) {
      soft = all = 0;
      for (int i = 0; i < kDecideThreads / 64; ++i) {
        soft += part[0][i];
        all += part[1][i];
      }
    }
    const bool hard = force < 0 ? (all > 0 && 4 * (int64_t)soft <= (int64_t)all) : force != 0;
    ctl[kCtlHard] = hard ? 1 : 0;
    ctl[kCtlAllSoft] = hard ? 0 : N;
    ctl[kCtlNSoft] = 0;
  }
}

}  // namespace vcmi
