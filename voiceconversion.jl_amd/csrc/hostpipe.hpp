// hostpipe.hpp -- how the host-pointer entry points (what a Julia `ccall` passes: pageable arrays) reach the device.
//
// A plain hipMemcpy of pageable memory is synchronous and serial: upload, kernel and download of a 10^6-frame
// conversion add up (68 ms measured in round 1 for a 5.5 ms kernel).  Here every transfer goes through a per-device
// ring of pinned staging slots: worker threads copy user memory <-> pinned slots (first-touch page faults of a fresh
// output array are spread over the workers too), the DMA engines move pinned <-> HBM asynchronously on their own
// streams, and the kernels of chunk c run between the upload of chunk c+1 and the download of chunk c-1
// (PCIe is full duplex).  tools/microbench_pcie.hip measured what the box gives: 57 GB/s per direction pinned,
// 6.9 ms for 2 x 320 MB in both directions at once, 30 GB/s per memcpy thread (100 GB/s with 8).
#pragma once
#include <functional>
#include <vector>
#include "vcmi_common.hpp"

namespace vcmi {

// fn(lo, hi) over [0, n) split into ranges of about `grain` items, on the library's worker threads plus the caller.
void host_parallel_for(int64_t n, int64_t grain, const std::function<void(int64_t, int64_t)> &fn);
// memcpy / strided row copy on the worker threads (inline when small)
void host_copy(void *dst, const void *src, size_t bytes);
void host_copy_rows(void *dst, size_t dst_stride, const void *src, size_t src_stride, size_t row_bytes, int64_t rows);

struct HostPiece {   // one contiguous host range of a gather (upload) or scatter (download) list
  void *host;
  size_t bytes;
};

// hSrc (pageable) -> dDst on the current device; returns once everything is enqueued; `consumer` (may be the null
// stream) is made to wait for the last chunk.  The pieces of the gather variant land back to back at dDst.
int staged_upload(void *dDst, const void *hSrc, size_t bytes, hipStream_t consumer);
// A synchronous upload of model-sized data from ordinary (pageable) host memory THROUGH THE PINNED RING: complete on return, like
// hipMemcpy -- but the runtime never pins the caller's pages in place.  (hipMemcpy from a pageable std::vector of a megabyte or
// more locks the vector's pages for the DMA; in the full test suite -- after arrays had been registered, unregistered and freed
// around the same heap addresses -- that DMA hit an unmapped page about once in three runs: "Memory access fault by GPU" inside
// vcmi_gmmmap_create, round 6.)
int upload_now(void *dDst, const void *hSrc, size_t bytes);
inline hipError_t upload_now_hip(void *dDst, const void *hSrc, size_t bytes) {      // for call sites that chain hipError_t
  return upload_now(dDst, hSrc, bytes) == VCMI_OK ? hipSuccess : hipErrorUnknown;
}
int staged_upload_gather(void *dDst, const std::vector<HostPiece> &pieces, hipStream_t consumer);
// dSrc -> hDst (pageable); waits for `producer`'s work enqueued so far; returns when the data is in host memory.
int staged_download(void *hDst, const void *dSrc, size_t bytes, hipStream_t producer);
int staged_download_scatter(const std::vector<HostPiece> &pieces, const void *dSrc, hipStream_t producer);

// Chunked pipeline over `units` independent records (frames): record u is in_unit bytes at hIn + u*in_stride and
// produces out_unit bytes at hOut + u*out_stride.  launch(dIn, dOut, first, n, stream) enqueues the kernels of one
// chunk: its records are packed densely (in_unit / out_unit apart) in dIn / dOut.  Chunks are sized to fill the
// chip (>= min_chunk_units) and to keep three of them in flight.
using ChunkLaunch = std::function<int(const void *dIn, void *dOut, int64_t first, int64_t n, hipStream_t st)>;
int staged_pipeline(const void *hIn, size_t in_unit, size_t in_stride, void *hOut, size_t out_unit, size_t out_stride,
                    int64_t units, int64_t min_chunk_units, const ChunkLaunch &launch);

// Caller-pinned arrays (include/vcmi.h: vcmi_host_register).  staged_pipeline moves a dense side that lies in pinned memory
// straight between the caller's array and HBM: no staging slot, no host memcpy.
int host_register(void *p, size_t bytes);
int host_unregister(void *p);
int host_is_registered(const void *p, size_t bytes);

}  // namespace vcmi
