// devgroup.cpp -- persistent per-device worker threads and the RCCL communicators of a single-process device group.
#include "devgroup.hpp"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <numeric>
#include <string>
#include <thread>

namespace vcmi {

namespace {

struct Rccl {   // entry points resolved from librccl at first use
  void *lib = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  int load() {
    if (lib) return VCMI_OK;
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (lib) break;
    }
    if (!lib) return fail(VCMI_ERR_HIP, "cannot load librccl (%s): multi-GPU E-step needs RCCL", dlerror());
    CommInitAll = reinterpret_cast<decltype(CommInitAll)>(dlsym(lib, "ncclCommInitAll"));
    CommDestroy = reinterpret_cast<decltype(CommDestroy)>(dlsym(lib, "ncclCommDestroy"));
    AllReduce = reinterpret_cast<decltype(AllReduce)>(dlsym(lib, "ncclAllReduce"));
    GetErrorString = reinterpret_cast<decltype(GetErrorString)>(dlsym(lib, "ncclGetErrorString"));
    if (!CommInitAll || !CommDestroy || !AllReduce || !GetErrorString) {
      lib = nullptr;
      return fail(VCMI_ERR_HIP, "librccl lacks ncclCommInitAll / ncclAllReduce");
    }
    return VCMI_OK;
  }
};

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
  std::vector<ncclComm_t> comms;   // empty until the first all-reduce
  std::mutex run_mu;               // one group_run at a time
  std::mutex comm_mu;
  int comm_status = VCMI_OK;
  std::string comm_message;
};

std::mutex g_mu;
Group *g_group = nullptr;
std::atomic<uint64_t> g_epoch{1};
Rccl g_rccl;

void worker_loop(Worker *w, int member) {
  (void)hipSetDevice(w->device);
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

void destroy_group(Group *g) {
  if (!g) return;
  for (Worker *w : g->workers) {
    {
      std::lock_guard<std::mutex> lk(w->m);
      w->quit = true;
    }
    w->cv.notify_all();
    if (w->th.joinable()) w->th.join();
    delete w;
  }
  for (ncclComm_t c : g->comms)
    if (c && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c);
  delete g;
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

int group_run(const std::function<int(int)> &fn) {
  Group *g;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    g = g_group;
  }
  if (!g) return fail(VCMI_ERR_ARG, "no device group set (vcmi_set_devices)");
  std::lock_guard<std::mutex> run(g->run_mu);
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
  Group *g;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    g = g_group;
  }
  if (!g) return fail(VCMI_ERR_ARG, "no device group set");
  {
    // the communicators are made once, by whichever member gets here first (ncclCommInitAll sets up every rank)
    std::lock_guard<std::mutex> lk(g->comm_mu);
    if (g->comms.empty() && g->comm_status == VCMI_OK) {
      std::vector<int> sorted = g->devices;
      std::sort(sorted.begin(), sorted.end());
      if (std::adjacent_find(sorted.begin(), sorted.end()) != sorted.end())
        g->comm_status = fail(VCMI_ERR_ARG, "the E-step all-reduce needs distinct devices in the group (RCCL: one rank per GPU)");
      else if ((g->comm_status = g_rccl.load()) == VCMI_OK) {
        g->comms.assign(g->devices.size(), nullptr);
        ncclResult_t r = g_rccl.CommInitAll(g->comms.data(), (int)g->devices.size(), g->devices.data());
        if (r != ncclSuccess) {
          g->comms.clear();
          g->comm_status = fail(VCMI_ERR_HIP, "ncclCommInitAll over %zu devices failed: %s", g->devices.size(),
                                g_rccl.GetErrorString(r));
        }
        (void)hipSetDevice(g->devices[member]);   // ncclCommInitAll walks the devices
      }
      if (g->comm_status != VCMI_OK) g->comm_message = error_buffer();
    }
    if (g->comm_status != VCMI_OK) {
      snprintf(error_buffer(), 512, "%s", g->comm_message.c_str());
      return g->comm_status;
    }
  }
  ncclResult_t r = g_rccl.AllReduce(buf, buf, count, ncclDouble, ncclSum, g->comms[member], st);
  if (r != ncclSuccess) return fail(VCMI_ERR_HIP, "ncclAllReduce failed: %s", g_rccl.GetErrorString(r));
  VCMI_HIP(hipStreamSynchronize(st));
  return VCMI_OK;
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
    VCMI_TRY(check_device());
    int count = 0;
    VCMI_HIP(hipGetDeviceCount(&count));
    for (int i = 0; i < n; ++i)
      if (devices[i] < 0 || devices[i] >= count)
        return fail(VCMI_ERR_ARG, "vcmi_set_devices: device %d not visible (%d devices)", devices[i], count);
  }
  Group *old;
  Group *g = nullptr;
  if (n > 0) {
    g = new (std::nothrow) Group();
    if (!g) return fail(VCMI_ERR_OOM, "out of host memory");
    g->devices.assign(devices, devices + n);
    for (int i = 0; i < n; ++i) {
      Worker *w = new Worker();
      w->device = devices[i];
      w->th = std::thread(worker_loop, w, i);
      g->workers.push_back(w);
    }
  }
  {
    std::lock_guard<std::mutex> lk(g_mu);
    old = g_group;
    g_group = g;
    g_epoch.fetch_add(1);
  }
  destroy_group(old);
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
