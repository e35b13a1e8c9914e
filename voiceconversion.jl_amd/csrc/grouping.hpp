// grouping.hpp -- the stable counting sort of frames by an integer key (defined in gmmmap.hip; also used by estep.hip).
#pragma once
#include "vcmi_common.hpp"

namespace vcmi {
constexpr int kGroupChunk = 1024;     // frames per chunk of the grouping sort (64 tiles of 16; 16 wave rows of 64)
// chunkhist[c][k] (counts of key k in chunk c of kGroupChunk consecutive frames) -> its exclusive prefix over the chunks, total[k]
// (gate, optional: a device word; 0 -> the kernel returns at once -- estep_path.hpp)
__global__ void gmmmap_group_scan_kernel(int *__restrict__ chunkhist, int64_t nchunks, int M, int *__restrict__ total, const int64_t *__restrict__ gate);
// perm: frames in key order, inside a key in frame order (needs (17 * M) ints of dynamic LDS, one workgroup of 256 per chunk)
__global__ void gmmmap_group_scatter_kernel(const int *__restrict__ key, int64_t T, int M, const int *__restrict__ chunkhist,
                                            const int *__restrict__ total, int *__restrict__ perm, const int64_t *__restrict__ gate);
}  // namespace vcmi
