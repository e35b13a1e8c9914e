// devgroup.cpp -- persistent per-device worker threads and the RCCL communicators of a single-process device group.
#include "devgroup.hpp"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <memory>
#include <condition_variable>
#include <mutex>
#include <numeric>
#include <string>
#include <thread>

namespace vcmi {

namespace {

// ---------------------------------------------------------------------------------------------------------------
// What the group needs from the platform.  The product binds it to HIP + RCCL (librccl resolved with dlopen on first
// use, so hosts that never set a group never load it).  tests/c/devgroup_stress.cpp is compiled with
// -DVCMI_DEVGROUP_TEST_BACKEND and supplies vcmi::devgroup_test_backend(): host-memory "devices" and an in-process
// collective, so that the worker / retire / timeout logic below runs with four members under ThreadSanitizer on a box
// without a GPU.  The switch is a compile-time one: libvcmi.so contains no test backend.
// ---------------------------------------------------------------------------------------------------------------
struct Rccl {   // entry points resolved from librccl at first use
  void *lib = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  std::mutex mu;
  int load() {
    std::lock_guard<std::mutex> lk(mu);
    if (lib) return VCMI_OK;
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (lib) break;
    }
    if (!lib) return fail(VCMI_ERR_HIP, "cannot load librccl (%s): multi-GPU E-step needs RCCL", dlerror());
    CommInitAll = reinterpret_cast<decltype(CommInitAll)>(dlsym(lib, "ncclCommInitAll"));
    CommDestroy = reinterpret_cast<decltype(CommDestroy)>(dlsym(lib, "ncclCommDestroy"));
    CommAbort = reinterpret_cast<decltype(CommAbort)>(dlsym(lib, "ncclCommAbort"));
    AllReduce = reinterpret_cast<decltype(AllReduce)>(dlsym(lib, "ncclAllReduce"));
    GetErrorString = reinterpret_cast<decltype(GetErrorString)>(dlsym(lib, "ncclGetErrorString"));
    if (!CommInitAll || !CommDestroy || !AllReduce || !GetErrorString) {
      lib = nullptr;
      return fail(VCMI_ERR_HIP, "librccl lacks ncclCommInitAll / ncclAllReduce");
    }
    return VCMI_OK;
  }
};
Rccl g_rccl;

int hip_device_count(int *n) {
  VCMI_TRY(check_device());
  VCMI_HIP(hipGetDeviceCount(n));
  return VCMI_OK;
}
void hip_bind_device(int d) { (void)hipSetDevice(d); }
int rccl_comm_init_all(void **comms, int n, const int *devices) {
  VCMI_TRY(g_rccl.load());
  int cur = 0;
  (void)hipGetDevice(&cur);
  ncclResult_t r = g_rccl.CommInitAll(reinterpret_cast<ncclComm_t *>(comms), n, devices);
  (void)hipSetDevice(cur);   // ncclCommInitAll walks the devices
  if (r != ncclSuccess)
    return fail(VCMI_ERR_HIP, "ncclCommInitAll over %d devices failed: %s", n, g_rccl.GetErrorString(r));
  return VCMI_OK;
}
void rccl_comm_destroy(void *c) {
  if (c && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(static_cast<ncclComm_t>(c));
}
void rccl_comm_abort(void *c) {
  if (!c) return;
  if (g_rccl.CommAbort) (void)g_rccl.CommAbort(static_cast<ncclComm_t>(c));
  else rccl_comm_destroy(c);
}
int rccl_all_reduce_start(void *comm, double *buf, size_t count, hipStream_t st) {
  ncclResult_t r = g_rccl.AllReduce(buf, buf, count, ncclDouble, ncclSum, static_cast<ncclComm_t>(comm), st);
  if (r != ncclSuccess) return fail(VCMI_ERR_HIP, "ncclAllReduce failed: %s", g_rccl.GetErrorString(r));
  return VCMI_OK;
}
int hip_all_reduce_poll(void *, hipStream_t st) {   // 1 done, 0 pending, < 0 failed
  hipError_t e = hipStreamQuery(st);
  if (e == hipSuccess) return 1;
  if (e == hipErrorNotReady) return 0;
  fail(VCMI_ERR_HIP, "all-reduce stream failed: %s", hipGetErrorString(e));
  return -1;
}

const DevGroupBackend kHipBackend = {hip_device_count, hip_bind_device, rccl_comm_init_all, rccl_comm_destroy,
                                     rccl_comm_abort,  rccl_all_reduce_start, hip_all_reduce_poll};

const DevGroupBackend &backend() {
#ifdef VCMI_DEVGROUP_TEST_BACKEND
  return devgroup_test_backend();
#else
  return kHipBackend;
#endif
}

struct Worker {
  int device = 0;
  std::thread th;
  std::mutex m;
  std::condition_variable cv;
  const std::function<int(int)> *job = nullptr;   // pending job (one at a time)
  bool quit = false, busy = false;
  int status = VCMI_OK;
  std::string message;
};

struct Group {
  std::vector<int> devices;
  std::vector<Worker *> workers;
  uint64_t epoch = 0;
  std::mutex run_mu;               // one group_run at a time; vcmi_set_devices retires the group under it
  bool retired = false;            // (run_mu) the workers are gone: a caller that still holds the group must give up
  std::mutex comm_mu;              // communicators: made eagerly by vcmi_set_devices, or by the first all-reduce
  std::vector<void *> comms;
  bool comm_tried = false;
  int comm_status = VCMI_OK;
  std::string comm_message;
  std::atomic<int64_t> timeout_ms{60000};   // a member that never reaches the collective must not hang the others
  ~Group() {
    for (void *c : comms) backend().comm_destroy(c);
  }
};

std::mutex g_mu;
std::shared_ptr<Group> g_group;
std::atomic<uint64_t> g_epoch{1};
std::atomic<int64_t> g_timeout_ms{60000};

std::shared_ptr<Group> current_group() {
  std::lock_guard<std::mutex> lk(g_mu);
  return g_group;
}

thread_local Group *tl_group = nullptr;   // the group a worker thread belongs to (valid while a job runs: group_run holds it)

void worker_loop(Group *g, Worker *w, int member) {
  tl_group = g;
  backend().bind_device(w->device);
  for (;;) {
    const std::function<int(int)> *job;
    {
      std::unique_lock<std::mutex> lk(w->m);
      w->cv.wait(lk, [w] { return w->job != nullptr || w->quit; });
      if (w->quit) return;
      job = w->job;
    }
    error_buffer()[0] = 0;
    const int st = (*job)(member);
    {
      std::lock_guard<std::mutex> lk(w->m);
      w->status = st;
      w->message = error_buffer();
      w->job = nullptr;
      w->busy = false;
    }
    w->cv.notify_all();
  }
}

// Stops the workers of a group that is no longer the current one.  Waits for a group_run in flight (run_mu); callers
// that picked the group up before the swap find `retired` set once they get run_mu and return an error.  The Group
// object itself lives until the last shared_ptr to it is gone.
void retire_group(const std::shared_ptr<Group> &g) {
  if (!g) return;
  std::lock_guard<std::mutex> run(g->run_mu);
  g->retired = true;
  for (Worker *w : g->workers) {
    {
      std::lock_guard<std::mutex> lk(w->m);
      w->quit = true;
    }
    w->cv.notify_all();
    if (w->th.joinable()) w->th.join();
    delete w;
  }
  g->workers.clear();
}

// (comm_mu held) make the communicators of the group once; failure is remembered and reported by the all-reduce
void ensure_comms(Group *g) {
  if (g->comm_tried) return;
  g->comm_tried = true;
  std::vector<int> sorted = g->devices;
  std::sort(sorted.begin(), sorted.end());
  if (std::adjacent_find(sorted.begin(), sorted.end()) != sorted.end()) {
    g->comm_status = fail(VCMI_ERR_ARG, "the E-step all-reduce needs distinct devices in the group (RCCL: one rank per GPU)");
  } else {
    g->comms.assign(g->devices.size(), nullptr);
    g->comm_status = backend().comm_init_all(g->comms.data(), (int)g->devices.size(), g->devices.data());
    if (g->comm_status != VCMI_OK) g->comms.clear();
  }
  if (g->comm_status != VCMI_OK) g->comm_message = error_buffer();
}

}  // namespace

int group_size() {
  std::lock_guard<std::mutex> lk(g_mu);
  return g_group ? (int)g_group->devices.size() : 0;
}

int group_device(int member) {
  std::lock_guard<std::mutex> lk(g_mu);
  return (g_group && member >= 0 && member < (int)g_group->devices.size()) ? g_group->devices[member] : -1;
}

uint64_t group_epoch() { return g_epoch.load(); }

int group_run(int members, const std::function<int(int)> &fn) {
  std::shared_ptr<Group> g = current_group();
  if (!g) return fail(VCMI_ERR_ARG, "no device group set (vcmi_set_devices)");
  std::lock_guard<std::mutex> run(g->run_mu);
  // vcmi_set_devices must not overlap other calls; when it does, the call that lost the race fails cleanly instead of
  // running on workers that are gone or with shards cut for another member count
  if (g->retired || (int)g->workers.size() != members)
    return fail(VCMI_ERR_ARG, "the device group was replaced during the call (vcmi_set_devices overlapped it)");
  for (Worker *w : g->workers) {
    {
      std::lock_guard<std::mutex> lk(w->m);
      w->job = &fn;
      w->busy = true;
    }
    w->cv.notify_all();
  }
  int rc = VCMI_OK;
  for (Worker *w : g->workers) {
    std::unique_lock<std::mutex> lk(w->m);
    w->cv.wait(lk, [w] { return !w->busy; });
    if (rc == VCMI_OK && w->status != VCMI_OK) {
      rc = w->status;
      snprintf(error_buffer(), 512, "%s", w->message.c_str());
    }
  }
  return rc;
}

int group_allreduce_sum(int member, double *buf, size_t count, hipStream_t st) {
  // called on a worker thread from inside group_run, which keeps the worker's group alive and un-retired for the
  // duration.  The worker's OWN group, not the current one: vcmi_set_devices may have published a new group while this
  // run was in flight, and its communicators belong to other members.
  Group *g = tl_group;
  if (!g) return fail(VCMI_ERR_ARG, "all-reduce called outside a device-group worker");
  if (member < 0 || member >= (int)g->devices.size()) return fail(VCMI_ERR_ARG, "all-reduce: member %d out of range", member);
  void *comm = nullptr;
  {
    std::lock_guard<std::mutex> lk(g->comm_mu);
    ensure_comms(g);
    if (g->comm_status != VCMI_OK) {
      snprintf(error_buffer(), 512, "%s", g->comm_message.c_str());
      return g->comm_status;
    }
    comm = g->comms[(size_t)member];
  }
  VCMI_TRY(backend().all_reduce_start(comm, buf, count, st));
  // wait with a deadline instead of hipStreamSynchronize: a member that died before its ncclAllReduce would leave
  // every other member inside the collective for ever
  const auto t0 = std::chrono::steady_clock::now();
  const auto deadline = t0 + std::chrono::milliseconds(g->timeout_ms.load());
  for (unsigned spins = 0;; ++spins) {
    const int done = backend().all_reduce_poll(comm, st);
    if (done > 0) return VCMI_OK;
    if (done < 0) return VCMI_ERR_HIP;
    if (std::chrono::steady_clock::now() > deadline) break;
    if (spins < 4096) std::this_thread::yield();
    else std::this_thread::sleep_for(std::chrono::microseconds(50));
  }
  {
    // give up: abort this member's communicator (releases the stream) and mark the group's communicators unusable
    std::lock_guard<std::mutex> lk(g->comm_mu);
    if (g->comm_status == VCMI_OK) {
      g->comm_status = VCMI_ERR_HIP;
      g->comm_message = "an earlier all-reduce timed out; set the device group again";
    }
    backend().comm_abort(comm);
    g->comms[(size_t)member] = nullptr;
  }
  return fail(VCMI_ERR_HIP, "all-reduce over %zu devices timed out after %lld ms on member %d (a member never joined)",
              g->devices.size(), (long long)g->timeout_ms.load(), member);
}

void group_set_timeout_ms(int64_t ms) {
  g_timeout_ms.store(ms > 0 ? ms : 60000);
  std::shared_ptr<Group> g = current_group();
  if (g) g->timeout_ms.store(g_timeout_ms.load());
}

std::vector<int> shard_by_cost(const std::vector<int64_t> &costs, int m) {
  std::vector<size_t> order(costs.size());
  std::iota(order.begin(), order.end(), (size_t)0);
  std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return costs[a] > costs[b]; });
  std::vector<int64_t> load((size_t)m, 0);
  std::vector<int> part(costs.size(), 0);
  for (size_t k : order) {
    const int r = (int)(std::min_element(load.begin(), load.end()) - load.begin());
    part[k] = r;
    load[(size_t)r] += costs[k];
  }
  return part;
}

}  // namespace vcmi

using namespace vcmi;

extern "C" int vcmi_set_devices(const int *devices, int n) {
  if (n < 0 || (n > 0 && !devices)) return fail(VCMI_ERR_ARG, "vcmi_set_devices: bad argument");
  if (n > 0) {
    int count = 0;
    VCMI_TRY(backend().device_count(&count));
    for (int i = 0; i < n; ++i)
      if (devices[i] < 0 || devices[i] >= count)
        return fail(VCMI_ERR_ARG, "vcmi_set_devices: device %d not visible (%d devices)", devices[i], count);
  }
  std::shared_ptr<Group> g, old;
  if (n > 0) {
    g = std::make_shared<Group>();
    g->devices.assign(devices, devices + n);
    g->timeout_ms.store(g_timeout_ms.load());
    for (int i = 0; i < n; ++i) {
      Worker *w = new Worker();
      w->device = devices[i];
      w->th = std::thread(worker_loop, g.get(), w, i);
      g->workers.push_back(w);
    }
    // several distinct devices: make the RCCL communicators now, from the caller's thread, rather than lazily from
    // inside a worker in the middle of the first E-step.  A failure here does not fail the call (frames / pairs /
    // utterances shard without any collective); the E-step reports it.
    std::vector<int> sorted = g->devices;
    std::sort(sorted.begin(), sorted.end());
    if (n > 1 && std::adjacent_find(sorted.begin(), sorted.end()) == sorted.end()) {
      std::lock_guard<std::mutex> lk(g->comm_mu);
      ensure_comms(g.get());
      error_buffer()[0] = 0;
    }
  }
  {
    std::lock_guard<std::mutex> lk(g_mu);
    old = g_group;
    g_group = g;
    if (g) g->epoch = g_epoch.load() + 1;
    g_epoch.fetch_add(1);
  }
  retire_group(old);
  return VCMI_OK;
}

extern "C" int vcmi_get_devices(int *devices, int capacity, int *n) {
  if (!n) return fail(VCMI_ERR_ARG, "vcmi_get_devices: NULL argument");
  std::lock_guard<std::mutex> lk(g_mu);
  const int m = g_group ? (int)g_group->devices.size() : 0;
  *n = m;
  for (int i = 0; i < m && i < capacity && devices; ++i) devices[i] = g_group->devices[i];
  return VCMI_OK;
}
