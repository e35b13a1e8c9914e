// devgroup.hpp -- one host process driving several GPUs (what a Julia host is: SURVEY 8b/8e).
//
// vcmi_set_devices(devs, n) creates one persistent worker thread per listed device (bound with hipSetDevice; its
// thread-local scratch, staging ring and streams live as long as the group).  The host-pointer entry points then shard
// their frames / pairs / utterances over the members -- no data-path collective -- and only the E-step
// exchanges data: one ncclAllReduce(sum, double) of the packed statistics over RCCL (single-process communicators from
// ncclCommInitAll; librccl is loaded on first use, so hosts that never set a group never load it).
#pragma once
#include <functional>
#include <vector>
#include "vcmi_common.hpp"

namespace vcmi {

int group_size();                               // 0: no group set (single-device behaviour)
int group_device(int member);
uint64_t group_epoch();                         // changes whenever the group is re-made (replica caches key on it)
// fn(member) on every member's worker thread, concurrently; returns the first failing member's status with its message.
// `members` is the group size the caller cut its shards for: if the group was replaced in between (vcmi_set_devices
// must not overlap other calls) the call fails with VCMI_ERR_ARG instead of running on the wrong members.
int group_run(int members, const std::function<int(int)> &fn);
// to be called from inside group_run by EVERY member: in-place sum of `count` doubles at `buf` (device memory of that
// member) over all members, on `st`; returns after the result is complete on this member
// A member that does not see the result within the group's timeout (default 60 s) aborts its communicator and returns
// VCMI_ERR_HIP: a member that died before the collective cannot hang the others.
int group_allreduce_sum(int member, double *buf, size_t count, hipStream_t st);
void group_set_timeout_ms(int64_t ms);

// Platform hooks of the group (devgroup.cpp binds them to HIP + RCCL; the sanitizer driver tests/c/devgroup_stress.cpp
// builds devgroup.cpp with -DVCMI_DEVGROUP_TEST_BACKEND and supplies devgroup_test_backend()).
struct DevGroupBackend {
  int (*device_count)(int *n);                                        // VCMI_ERR_NO_DEVICE without a device
  void (*bind_device)(int device);                                    // once per worker thread
  int (*comm_init_all)(void **comms, int n, const int *devices);      // one communicator per member
  void (*comm_destroy)(void *comm);
  void (*comm_abort)(void *comm);
  int (*all_reduce_start)(void *comm, double *buf, size_t count, hipStream_t st);   // enqueue the in-place sum
  int (*all_reduce_poll)(void *comm, hipStream_t st);                 // 1 done, 0 pending, < 0 failed
};
#ifdef VCMI_DEVGROUP_TEST_BACKEND
const DevGroupBackend &devgroup_test_backend();
#endif

// contiguous balanced shard [lo, hi) of n units for member i of m
inline void shard_range(int64_t n, int i, int m, int64_t *lo, int64_t *hi) {
  const int64_t base = n / m, rem = n % m;
  *lo = i * base + (i < rem ? i : rem);
  *hi = *lo + base + (i < rem ? 1 : 0);
}
// longest-processing-time partition of items by cost: part[k] = member of item k (deterministic)
std::vector<int> shard_by_cost(const std::vector<int64_t> &costs, int m);

}  // namespace vcmi
