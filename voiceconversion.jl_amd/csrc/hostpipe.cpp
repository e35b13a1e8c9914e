// hostpipe.cpp -- pinned staging rings, worker-thread copies and the chunked upload / kernel / download pipeline
// behind the host-pointer entry points (see hostpipe.hpp).
#include "hostpipe.hpp"
#include <chrono>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <deque>
#include <memory>
#include <mutex>
#include <thread>

#include <sched.h>
#include <sys/mman.h>
#include <sys/syscall.h>
#include <unistd.h>

namespace vcmi {

// ------------------------------------------------------------------------------------------------
// worker threads
// ------------------------------------------------------------------------------------------------
namespace {

// Completion counter of one host_parallel_for call.  Shared ownership (caller + every task): it cannot disappear under
// a worker that is still inside done(), whatever the caller does; the count is only touched under the mutex.
struct Latch {
  int remaining = 0;
  std::mutex m;
  std::condition_variable c;
  void done() {
    std::lock_guard<std::mutex> lk(m);
    if (--remaining == 0) c.notify_all();
  }
  bool finished() {
    std::lock_guard<std::mutex> lk(m);
    return remaining == 0;
  }
};

struct Task {
  const std::function<void(int64_t, int64_t)> *fn;
  int64_t lo, hi;
  std::shared_ptr<Latch> latch;
  std::shared_ptr<const std::function<void(int64_t, int64_t)>> owned;   // fire-and-forget tasks keep their function alive
};

// ---- NUMA placement (no libnuma in the image: sysfs + raw syscalls; every failure just leaves the default placement) ----
// The boxes have two sockets with four GPUs each; a staging slot or a copy thread on the other socket puts the
// inter-socket link into every transfer (driver-run host path of round 2: 32 ms where the build box gave 15).
int numa_node_of_device(int dev) {
  char bdf[64] = {0};
  if (hipDeviceGetPCIBusId(bdf, (int)sizeof(bdf), dev) != hipSuccess) return -1;
  for (char *c = bdf; *c; ++c) *c = (char)tolower(*c);
  char path[160];
  snprintf(path, sizeof(path), "/sys/bus/pci/devices/%s/numa_node", bdf);
  FILE *f = fopen(path, "r");
  if (!f) return -1;
  int node = -1;
  if (fscanf(f, "%d", &node) != 1) node = -1;
  fclose(f);
  return node;
}
bool cpus_of_node(int node, cpu_set_t *set) {
  char path[96];
  snprintf(path, sizeof(path), "/sys/devices/system/node/node%d/cpulist", node);
  FILE *f = fopen(path, "r");
  if (!f) return false;
  CPU_ZERO(set);
  int a, b, n = 0;
  while (fscanf(f, "%d", &a) == 1) {
    b = a;
    int c = fgetc(f);
    if (c == '-') {
      if (fscanf(f, "%d", &b) != 1) break;
      c = fgetc(f);
    }
    for (int k = a; k <= b && k < CPU_SETSIZE; ++k) {
      CPU_SET(k, set);
      ++n;
    }
    if (c != ',') break;
  }
  fclose(f);
  return n > 0;
}
// memory policy of the calling thread for the duration of a scope: allocations prefer `node` (MPOL_PREFERRED = 1)
// The caller's own policy (an inherited `numactl --membind / --interleave`) is read first and put back afterwards; when it
// cannot be read nothing is changed at all.
struct ScopedPreferNode {
  bool active = false;
  int old_mode = 0;
  unsigned long old_mask[16] = {};            // 1024 nodes
  explicit ScopedPreferNode(int node) {
    if (node < 0 || node >= 64) return;
    if (syscall(SYS_get_mempolicy, &old_mode, old_mask, (unsigned long)(sizeof(old_mask) * 8 + 1), nullptr, 0ul) != 0) return;
    unsigned long mask = 1ul << node;
    active = syscall(SYS_set_mempolicy, 1 /*MPOL_PREFERRED*/, &mask, 65ul) == 0;
  }
  ~ScopedPreferNode() {
    if (!active) return;
    if (syscall(SYS_set_mempolicy, old_mode, old_mask, (unsigned long)(sizeof(old_mask) * 8 + 1)) != 0)
      (void)syscall(SYS_set_mempolicy, 0 /*MPOL_DEFAULT*/, nullptr, 0ul);
  }
};
// Which node the copy workers and the pinned slots of a device go to.  Measured on the two-socket boxes (tools/hostbench.py,
// 2 x 320 MB per call): what matters is that the copies between the user's arrays and the pinned slots stay on ONE
// socket -- the caller's, where its arrays were first touched; the DMA engines reach a slot on the other socket at full
// rate.  Everything on the caller's socket: 11.2 ms; workers and slots on the GPU's socket with the caller on the other
// one: 15.4 ms; unbound workers (rounds 1-2): 13.6-16.5 ms.  VCMI_HOST_NUMA=device|off overrides (read once).
int node_of_calling_thread() {
  unsigned cpu = 0, node = 0;
  if (syscall(SYS_getcpu, &cpu, &node, nullptr) != 0) return -1;
  return (int)node;
}
int placement_node(int dev) {
  static const int mode = [] {
    const char *e = getenv("VCMI_HOST_NUMA");
    return !e ? 0 : !strcmp(e, "device") ? 1 : !strcmp(e, "off") ? 2 : 0;
  }();
  if (mode == 2) return -1;
  return mode == 1 ? numa_node_of_device(dev) : node_of_calling_thread();
}
std::atomic<int> g_pool_node{-2};   // -2: never set; -1: no binding (devices on several nodes / unknown); >= 0: that node

class Pool {
 public:
  static Pool &get() {
    static Pool *p = new Pool();   // never destroyed: the workers may outlive static destruction
    return *p;
  }
  int workers() const { return nworkers_; }
  void push(const Task *tasks, int n) {
    {
      std::lock_guard<std::mutex> lk(mu_);
      for (int i = 0; i < n; ++i) q_.push_back(tasks[i]);
    }
    cv_.notify_all();
  }
  bool try_run_one() {
    Task t;
    {
      std::lock_guard<std::mutex> lk(mu_);
      if (q_.empty()) return false;
      t = q_.front();
      q_.pop_front();
    }
    (*t.fn)(t.lo, t.hi);
    if (t.latch) t.latch->done();
    return true;
  }

 private:
  Pool() {
    // the PROCESS mask (the main thread's: its tid is the pid), not the mask of whichever thread happens to touch the pool
    // first -- a pinned OpenMP / dataloader / Julia thread would confine every copy worker to its one or two CPUs
    if (sched_getaffinity(getpid(), sizeof(inherited_), &inherited_) != 0) {
      CPU_ZERO(&inherited_);
      for (int k = 0; k < CPU_SETSIZE; ++k) CPU_SET(k, &inherited_);
    }
    unsigned hw = std::thread::hardware_concurrency();
    int n = (int)std::min(16u, std::max(2u, hw / 4));
    if (const char *e = getenv("VCMI_HOST_THREADS")) n = std::max(0, std::min(64, atoi(e)));
    nworkers_ = n;
    for (int i = 0; i < n; ++i) std::thread([this] { loop(); }).detach();
  }
  void loop() {
    int bound = -2;
    for (;;) {
      Task t;
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [this] { return !q_.empty(); });
        t = q_.front();
        q_.pop_front();
      }
      // follow the pool's NUMA binding (set when a staging ring is made): the workers run on the socket of the GPU they
      // feed, so their copies and the pages they first touch stay off the inter-socket link
      const int want = g_pool_node.load(std::memory_order_relaxed);
      if (want != bound && want != -2) {
        // never outside the CPU set the process was started with (taskset / numactl / a container's cpuset): the node's
        // CPUs are intersected with the mask the pool inherited, and "no binding" means that mask again, not all CPUs
        cpu_set_t set, both;
        bool narrowed = false;
        if (want >= 0 && cpus_of_node(want, &set)) {
          CPU_AND(&both, &set, &inherited_);
          narrowed = CPU_COUNT(&both) > 0;
        }
        (void)sched_setaffinity(0, sizeof(cpu_set_t), narrowed ? &both : &inherited_);
        bound = want;
      }
      (*t.fn)(t.lo, t.hi);
      if (t.latch) t.latch->done();
    }
  }
  std::mutex mu_;
  std::condition_variable cv_;
  std::deque<Task> q_;
  int nworkers_ = 0;
  cpu_set_t inherited_;      // affinity of the process (its main thread): the workers stay inside it
};

}  // namespace

void host_parallel_for(int64_t n, int64_t grain, const std::function<void(int64_t, int64_t)> &fn) {
  if (n <= 0) return;
  Pool &pool = Pool::get();
  grain = std::max<int64_t>(grain, 1);
  int64_t parts = std::min<int64_t>((n + grain - 1) / grain, pool.workers() + 1);
  if (parts <= 1) {
    fn(0, n);
    return;
  }
  std::vector<Task> tasks;
  std::shared_ptr<Latch> lp = std::make_shared<Latch>();
  Latch &latch = *lp;
  const int64_t per = (n + parts - 1) / parts;
  for (int64_t p = 1; p < parts; ++p) {
    const int64_t lo = p * per, hi = std::min(n, lo + per);
    if (lo < hi) tasks.push_back(Task{&fn, lo, hi, lp, nullptr});
  }
  latch.remaining = (int)tasks.size();
  if (!tasks.empty()) pool.push(tasks.data(), (int)tasks.size());
  fn(0, std::min(n, per));
  // help until our own parts are done (also covers a process that forked away from its worker threads)
  // Help with whatever is queued (ours or another caller's); once the queue is empty every unfinished part of ours is
  // in the hands of a worker, so an untimed wait is safe.  (Tasks are leaf copies: they never wait for anything.)
  while (pool.try_run_one()) {
  }
  std::unique_lock<std::mutex> lk(latch.m);
  latch.c.wait(lk, [&] { return latch.remaining == 0; });
}

// fn(lo, hi) over [0, n) in parts of `grain`, queued for the workers; the caller goes on and MUST call host_async_wait on
// the returned latch before the memory fn touches can go away (used to pre-fault output pages ahead of the pipeline: the
// caller's array must not be freed, unmapped and its range reused while pieces are still queued -- ADVICE r3).
static std::shared_ptr<Latch> host_async_for(int64_t n, int64_t grain, std::function<void(int64_t, int64_t)> fn) {
  if (n <= 0) return nullptr;
  Pool &pool = Pool::get();
  if (pool.workers() == 0) return nullptr;
  auto owned = std::make_shared<const std::function<void(int64_t, int64_t)>>(std::move(fn));
  auto latch = std::make_shared<Latch>();
  std::vector<Task> tasks;
  for (int64_t lo = 0; lo < n; lo += grain) tasks.push_back(Task{owned.get(), lo, std::min(n, lo + grain), latch, owned});
  latch->remaining = (int)tasks.size();
  pool.push(tasks.data(), (int)tasks.size());
  return latch;
}
// Runs queued pieces itself until none is left, then waits for the ones the workers hold.
static void host_async_wait(const std::shared_ptr<Latch> &latch) {
  if (!latch) return;
  Pool &pool = Pool::get();
  while (!latch->finished() && pool.try_run_one()) {
  }
  std::unique_lock<std::mutex> lk(latch->m);
  latch->c.wait(lk, [&] { return latch->remaining == 0; });
}

// memcpy whose stores bypass the caches (AVX2 vmovntdq on 32-byte aligned destinations).  The pipeline's copies are
// parts of a few hundred KB -- under glibc's own non-temporal threshold -- so plain memcpy writes them with ordinary stores:
// every destination line is first READ from memory (read-for-ownership), three transfers per byte copied instead of two,
// on a socket whose memory the two DMA streams are using as well.  VCMI_HOST_NT=0 (read once) goes back to memcpy.
#if defined(__x86_64__)
__attribute__((target("avx2"))) static void copy_stream_avx2(char *d, const char *s, size_t n) {
  while (n && ((uintptr_t)d & 31)) {      // head: up to the destination's alignment
    *d++ = *s++;
    --n;
  }
  size_t v = n / 128;
  for (; v; --v, d += 128, s += 128) {
    const __m256i a = _mm256_loadu_si256((const __m256i *)s), b = _mm256_loadu_si256((const __m256i *)(s + 32)),
                  c = _mm256_loadu_si256((const __m256i *)(s + 64)), e = _mm256_loadu_si256((const __m256i *)(s + 96));
    _mm256_stream_si256((__m256i *)d, a);
    _mm256_stream_si256((__m256i *)(d + 32), b);
    _mm256_stream_si256((__m256i *)(d + 64), c);
    _mm256_stream_si256((__m256i *)(d + 96), e);
  }
  _mm_sfence();
  n &= 127;
  if (n) memcpy(d, s, n);
}
#endif
static void copy_bytes(void *dst, const void *src, size_t n) {
#if defined(__x86_64__)
  static const bool nt = [] {
    const char *e = getenv("VCMI_HOST_NT");
    return !(e && e[0] == '0') && __builtin_cpu_supports("avx2");
  }();
  if (nt && n >= 4096) {
    copy_stream_avx2((char *)dst, (const char *)src, n);
    return;
  }
#endif
  memcpy(dst, src, n);
}

void host_copy(void *dst, const void *src, size_t bytes) {
  if (bytes < ((size_t)1 << 20)) {
    if (bytes) memcpy(dst, src, bytes);
    return;
  }
  const size_t blk = (size_t)256 << 10;   // whole 256 KB blocks per part
  const int64_t nblk = (int64_t)((bytes + blk - 1) / blk);
  host_parallel_for(nblk, 2, [=](int64_t lo, int64_t hi) {
    const size_t a = (size_t)lo * blk, b = std::min(bytes, (size_t)hi * blk);
    copy_bytes((char *)dst + a, (const char *)src + a, b - a);
  });
}

void host_copy_rows(void *dst, size_t dst_stride, const void *src, size_t src_stride, size_t row_bytes, int64_t rows) {
  if (rows <= 0 || row_bytes == 0) return;
  if (dst_stride == row_bytes && src_stride == row_bytes) {
    host_copy(dst, src, row_bytes * (size_t)rows);
    return;
  }
  const int64_t grain = std::max<int64_t>(1, (int64_t)(((size_t)512 << 10) / row_bytes));
  host_parallel_for(rows, grain, [=](int64_t lo, int64_t hi) {
    for (int64_t r = lo; r < hi; ++r)
      memcpy((char *)dst + (size_t)r * dst_stride, (const char *)src + (size_t)r * src_stride, row_bytes);
  });
}

// ------------------------------------------------------------------------------------------------
// per-device staging ring
// ------------------------------------------------------------------------------------------------
namespace {

constexpr int K = 3;                                   // slots in flight
constexpr size_t kXferChunk = (size_t)16 << 20;        // plain uploads / downloads
constexpr size_t kPipeChunk = (size_t)32 << 20;        // pipeline chunks (bytes of the wider side)
constexpr size_t kDirectBytes = (size_t)2 << 20;       // calls up to this many bytes (in + out): kernels on the pinned slots

struct Ring {
  std::mutex mu;
  int device = -1;
  int node = -1;                    // NUMA node of the device (-1: unknown)
  bool ready = false;
  hipStream_t up = nullptr, run = nullptr, down = nullptr;
  hipEvent_t ev_up[K] = {}, ev_run[K] = {}, ev_down[K] = {}, ev_tmp = nullptr;
  char *pin_in[K] = {}, *pin_out[K] = {}, *dev_in[K] = {}, *dev_out[K] = {};
  size_t pin_in_cap = 0, pin_out_cap = 0, dev_in_cap = 0, dev_out_cap = 0;
  // the partial first / last page of a registered array (copy_split): four pinned pages -- two for what goes up, two for what
  // comes down and is copied into the caller's pages once the DMA is complete
  char *edge = nullptr;
  struct EdgeOut {
    char *host;
    const char *bounce;
    size_t n;
  } edge_out[2];
  int n_edge_in = 0, n_edge_out = 0;
  int edge_begin() {
    if (!edge) VCMI_HIP(hipHostMalloc(reinterpret_cast<void **>(&edge), 4 * 4096, hipHostMallocPortable));
    n_edge_in = n_edge_out = 0;
    return VCMI_OK;
  }
  void edge_finish() {            // (behind the synchronisation of the stream that carried the downloads)
    for (int i = 0; i < n_edge_out; ++i) memcpy(edge_out[i].host, edge_out[i].bounce, edge_out[i].n);
    n_edge_out = 0;
  }

  int init() {
    if (ready) return VCMI_OK;
    VCMI_HIP(hipStreamCreateWithFlags(&up, hipStreamNonBlocking));
    VCMI_HIP(hipStreamCreateWithFlags(&run, hipStreamNonBlocking));
    VCMI_HIP(hipStreamCreateWithFlags(&down, hipStreamNonBlocking));
    for (int i = 0; i < K; ++i) {
      VCMI_HIP(hipEventCreateWithFlags(&ev_up[i], hipEventDisableTiming));
      VCMI_HIP(hipEventCreateWithFlags(&ev_run[i], hipEventDisableTiming));
      VCMI_HIP(hipEventCreateWithFlags(&ev_down[i], hipEventDisableTiming));
    }
    VCMI_HIP(hipEventCreateWithFlags(&ev_tmp, hipEventDisableTiming));
    ready = true;
    return VCMI_OK;
  }
  // every user of a set of slots first drains what an earlier call may have left in flight on them
  int quiesce() {
    VCMI_HIP(hipStreamSynchronize(up));
    VCMI_HIP(hipStreamSynchronize(run));
    VCMI_HIP(hipStreamSynchronize(down));
    return VCMI_OK;
  }
  int reserve(char *(&slots)[K], size_t &cap, size_t bytes, bool pinned) {
    if (bytes <= cap) return VCMI_OK;
    VCMI_TRY(quiesce());
    bytes = (bytes + 4095) & ~(size_t)4095;
    for (int i = 0; i < K; ++i) {
      if (slots[i]) (void)(pinned ? hipHostFree(slots[i]) : hipFree(slots[i]));
      slots[i] = nullptr;
    }
    cap = 0;
    ScopedPreferNode prefer(pinned ? node : -1);   // pinned slots on the GPU's socket
    for (int i = 0; i < K; ++i) {
      hipError_t e = pinned ? hipHostMalloc(reinterpret_cast<void **>(&slots[i]), bytes, hipHostMallocPortable)
                            : hipMalloc(reinterpret_cast<void **>(&slots[i]), bytes);
      if (e != hipSuccess) {
        slots[i] = nullptr;
        return fail(VCMI_ERR_OOM, "%s of %zu staging bytes failed: %s", pinned ? "hipHostMalloc" : "hipMalloc", bytes,
                    hipGetErrorString(e));
      }
    }
    cap = bytes;
    return VCMI_OK;
  }
};

constexpr int kMaxDevices = 64;
std::mutex g_rings_mu;
Ring *g_rings[kMaxDevices] = {};

int current_ring(Ring **out) {
  int dev = 0;
  VCMI_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= kMaxDevices) return fail(VCMI_ERR_ARG, "device index %d out of range", dev);
  std::lock_guard<std::mutex> lk(g_rings_mu);
  if (!g_rings[dev]) {
    g_rings[dev] = new (std::nothrow) Ring();
    if (!g_rings[dev]) return fail(VCMI_ERR_OOM, "out of host memory");
    g_rings[dev]->device = dev;
    g_rings[dev]->node = placement_node(dev);
    // the copy workers follow the device while all rings of the process sit on one node; with devices on both sockets
    // (a device group over the whole box) they stay unbound
    int want = g_rings[dev]->node;
    for (int d = 0; d < kMaxDevices; ++d)
      if (g_rings[d] && g_rings[d]->node != want) want = -1;
    g_pool_node.store(want, std::memory_order_relaxed);
  }
  *out = g_rings[dev];
  return VCMI_OK;
}

struct Seg {   // one memcpy of a gather / scatter window
  size_t slot_off;
  char *host;
  size_t bytes;
};

// segments of the window [w0, w1) of the concatenated pieces, split so that no segment exceeds 1 MB
void window_segments(const std::vector<HostPiece> &pieces, const std::vector<size_t> &prefix, size_t w0, size_t w1,
                     std::vector<Seg> &segs) {
  segs.clear();
  size_t p = (size_t)(std::upper_bound(prefix.begin(), prefix.end(), w0) - prefix.begin()) - 1;
  const size_t kMax = (size_t)1 << 20;
  for (; p < pieces.size() && prefix[p] < w1; ++p) {
    const size_t a = std::max(w0, prefix[p]), b = std::min(w1, prefix[p] + pieces[p].bytes);
    for (size_t o = a; o < b; o += kMax)
      segs.push_back(Seg{o - w0, (char *)pieces[p].host + (o - prefix[p]), std::min(kMax, b - o)});
  }
}

}  // namespace
// (below: caller-pinned memory)  lo / hi: the part of the address space around [p, p + bytes) that the DMA engines reach -- the
// whole pages inside an array registered with vcmi_host_register, everything for memory another runtime pinned
bool range_is_pinned(const void *p, size_t bytes, uintptr_t *lo = nullptr, uintptr_t *hi = nullptr);

// host <-> device copy of [h, h + bytes): the part inside [lo, hi) as one DMA on the pinned pages; what is left at either end
// (less than a page each: the partial first and last page of a registered array are not pinned, see host_register) through
// pinned bounce pages of the ring -- up: copied in here, then DMA'd; down: DMA'd into the bounce page and copied into the
// caller's page by Ring::edge_finish once the stream is synchronised.  (Pageable hipMemcpyAsync calls for these few bytes made
// the host wait for the stream in the middle of the pipeline: 13.2 ms per 10^6 frames instead of 7.4.)
// One upload range and one download range per call (a head and a tail each); edge_begin() first.
static int copy_split(Ring *r, char *d, char *h, size_t bytes, bool to_device, hipStream_t st, uintptr_t lo, uintptr_t hi) {
  const uintptr_t a = (uintptr_t)h, b = a + bytes;
  const uintptr_t m0 = std::min(std::max(lo, a), b), m1 = std::max(std::min(hi, b), m0);
  auto edge = [&](uintptr_t x0, uintptr_t x1) -> int {
    if (x1 <= x0) return VCMI_OK;
    char *hp = h + (x0 - a), *dp = d + (x0 - a);
    const size_t n = x1 - x0;
    if (n > 4096 || (to_device ? r->n_edge_in : r->n_edge_out) >= 2) return fail(VCMI_ERR_ARG, "internal: edge copy of %zu bytes", n);
    if (to_device) {
      char *bn = r->edge + 4096 * (r->n_edge_in++);
      memcpy(bn, hp, n);
      VCMI_HIP(hipMemcpyAsync(dp, bn, n, hipMemcpyHostToDevice, st));
    } else {
      char *bn = r->edge + 4096 * (2 + r->n_edge_out);
      VCMI_HIP(hipMemcpyAsync(bn, dp, n, hipMemcpyDeviceToHost, st));
      r->edge_out[r->n_edge_out++] = Ring::EdgeOut{hp, bn, n};
    }
    return VCMI_OK;
  };
  VCMI_TRY(edge(a, m0));
  if (m1 > m0) {
    char *hp = h + (m0 - a), *dp = d + (m0 - a);
    VCMI_HIP(hipMemcpyAsync(to_device ? (void *)dp : (void *)hp, to_device ? (const void *)hp : (const void *)dp, m1 - m0,
                            to_device ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost, st));
  }
  return edge(m1, b);
}
namespace {

int do_upload(Ring *r, char *dDst, const std::vector<HostPiece> *pieces, const char *hSrc, size_t bytes, hipStream_t consumer) {
  if (bytes == 0) return VCMI_OK;
  uintptr_t plo = 0, phi = 0;
  if (!pieces && range_is_pinned(hSrc, bytes, &plo, &phi)) {
    // the caller pinned this array (vcmi_host_register): one DMA straight out of it.  The upload is complete on return -- the
    // staged path has copied the caller's data out of the array by then, and an entry point that only uploads (a resident
    // training matrix) must leave the same freedom to free or overwrite it.
    VCMI_HIP(hipEventRecord(r->ev_tmp, consumer));       // dDst may still be read by work the consumer enqueued earlier
    VCMI_HIP(hipStreamWaitEvent(r->up, r->ev_tmp, 0));
    VCMI_TRY(r->edge_begin());
    VCMI_TRY(copy_split(r, dDst, const_cast<char *>(hSrc), bytes, true, r->up, plo, phi));
    VCMI_HIP(hipStreamSynchronize(r->up));
    return VCMI_OK;
  }
  const size_t chunk = std::min(bytes, kXferChunk);
  VCMI_TRY(r->reserve(r->pin_in, r->pin_in_cap, chunk, true));
  // dDst may still be read by work the consumer enqueued earlier
  VCMI_HIP(hipEventRecord(r->ev_tmp, consumer));
  VCMI_HIP(hipStreamWaitEvent(r->up, r->ev_tmp, 0));
  std::vector<size_t> prefix;
  std::vector<Seg> segs;
  if (pieces) {
    prefix.resize(pieces->size() + 1, 0);
    for (size_t i = 0; i < pieces->size(); ++i) prefix[i + 1] = prefix[i] + (*pieces)[i].bytes;
  }
  int s = 0, last = 0;
  for (size_t off = 0; off < bytes; off += chunk, s = (s + 1) % K) {
    const size_t n = std::min(chunk, bytes - off);
    VCMI_HIP(hipEventSynchronize(r->ev_up[s]));   // the slot's previous upload (this call's or an earlier one's)
    char *slot = r->pin_in[s];
    if (pieces) {
      window_segments(*pieces, prefix, off, off + n, segs);
      const Seg *sg = segs.data();
      host_parallel_for((int64_t)segs.size(), 1, [=](int64_t lo, int64_t hi) {
        for (int64_t i = lo; i < hi; ++i) memcpy(slot + sg[i].slot_off, sg[i].host, sg[i].bytes);
      });
    } else {
      host_copy(slot, hSrc + off, n);
    }
    VCMI_HIP(hipMemcpyAsync(dDst + off, slot, n, hipMemcpyHostToDevice, r->up));
    VCMI_HIP(hipEventRecord(r->ev_up[s], r->up));
    last = s;
  }
  VCMI_HIP(hipStreamWaitEvent(consumer, r->ev_up[last], 0));
  return VCMI_OK;
}

int do_download(Ring *r, const std::vector<HostPiece> *pieces, char *hDst, const char *dSrc, size_t bytes, hipStream_t producer) {
  if (bytes == 0) return VCMI_OK;
  uintptr_t plo = 0, phi = 0;
  if (!pieces && range_is_pinned(hDst, bytes, &plo, &phi)) {        // a pinned destination: one DMA straight into it
    VCMI_HIP(hipEventRecord(r->ev_tmp, producer));
    VCMI_HIP(hipStreamWaitEvent(r->down, r->ev_tmp, 0));
    VCMI_TRY(r->edge_begin());
    VCMI_TRY(copy_split(r, const_cast<char *>(dSrc), hDst, bytes, false, r->down, plo, phi));
    VCMI_HIP(hipStreamSynchronize(r->down));
    r->edge_finish();
    return VCMI_OK;
  }
  const size_t chunk = std::min(bytes, kXferChunk);
  VCMI_TRY(r->reserve(r->pin_out, r->pin_out_cap, chunk, true));
  VCMI_HIP(hipEventRecord(r->ev_tmp, producer));
  VCMI_HIP(hipStreamWaitEvent(r->down, r->ev_tmp, 0));
  // (pin_out slots are always free here: download and pipeline calls return only after the host drained them)
  std::vector<size_t> prefix;
  std::vector<Seg> segs;
  if (pieces) {
    prefix.resize(pieces->size() + 1, 0);
    for (size_t i = 0; i < pieces->size(); ++i) prefix[i + 1] = prefix[i] + (*pieces)[i].bytes;
  }
  const int64_t nch = (int64_t)((bytes + chunk - 1) / chunk);
  for (int64_t c = 0; c <= nch; ++c) {
    if (c < nch) {
      const int s = (int)(c % K);
      const size_t off = (size_t)c * chunk, n = std::min(chunk, bytes - off);
      VCMI_HIP(hipMemcpyAsync(r->pin_out[s], dSrc + off, n, hipMemcpyDeviceToHost, r->down));
      VCMI_HIP(hipEventRecord(r->ev_down[s], r->down));
    }
    const int64_t j = c - 1;
    if (j >= 0) {
      const int s = (int)(j % K);
      const size_t off = (size_t)j * chunk, n = std::min(chunk, bytes - off);
      VCMI_HIP(hipEventSynchronize(r->ev_down[s]));
      const char *slot = r->pin_out[s];
      if (pieces) {
        window_segments(*pieces, prefix, off, off + n, segs);
        const Seg *sg = segs.data();
        host_parallel_for((int64_t)segs.size(), 1, [=](int64_t lo, int64_t hi) {
          for (int64_t i = lo; i < hi; ++i) memcpy(sg[i].host, slot + sg[i].slot_off, sg[i].bytes);
        });
      } else {
        host_copy(hDst + off, slot, n);
      }
    }
  }
  return VCMI_OK;
}

size_t total_bytes(const std::vector<HostPiece> &pieces) {
  size_t n = 0;
  for (auto &p : pieces) n += p.bytes;
  return n;
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// caller-pinned memory (vcmi_host_register): ranges the DMA engines reach without a staging copy
// ------------------------------------------------------------------------------------------------
namespace {
struct PinnedRange {
  uintptr_t base;
  size_t bytes;
  uintptr_t lo, hi;      // the whole pages inside [base, base + bytes) that are registered with the runtime; lo == hi: none
};
std::mutex g_pinned_mu;
std::vector<PinnedRange> g_pinned;      // ranges registered through vcmi_host_register (few: linear scan)
constexpr uintptr_t kPage = 4096;
// Arrays whose whole pages make up less than this are recorded but not pinned: their transfers are latency-bound either way
constexpr size_t kMinPinBytes = (size_t)1 << 20;
}  // namespace

// [p, p + bytes) lies in memory the DMA engines can address: a range registered here, or host memory another party pinned
// (hipHostMalloc / hipHostRegister by the caller's runtime: torch's pin_memory, a Julia AMDGPU.jl pinned array) that the HIP
// runtime knows.  VCMI_HOST_PINNED=0 (read once) switches the detection off (A/B: every call stages).
bool range_is_pinned(const void *p, size_t bytes, uintptr_t *lo, uintptr_t *hi) {
  static const bool enabled = [] {
    const char *e = getenv("VCMI_HOST_PINNED");
    return !(e && e[0] == '0');
  }();
  if (!enabled || !p || bytes == 0) return false;
  const uintptr_t a = (uintptr_t)p;
  {
    std::lock_guard<std::mutex> lk(g_pinned_mu);
    for (const PinnedRange &r : g_pinned)
      if (a >= r.base && a + bytes <= r.base + r.bytes) {
        if (lo) *lo = r.lo;
        if (hi) *hi = r.hi;
        return r.hi > r.lo;              // (a small array: registered with this library, staged all the same)
      }
  }
  // memory pinned by another party: worth two runtime look-ups only for transfers that are not small anyway (they stage)
  if (bytes < ((size_t)64 << 10)) return false;
  hipPointerAttribute_t at;
  if (hipPointerGetAttributes(&at, p) != hipSuccess) {
    (void)hipGetLastError();            // an ordinary pageable pointer: not an error of this call
    return false;
  }
  if (at.type != hipMemoryTypeHost || !at.devicePointer) return false;
  // The attributes name the allocation the FIRST byte lies in, not its extent; two pinned allocations with a pageable gap
  // between them would pass a first-byte / last-byte test.  The extent of the allocation comes from its device
  // mapping: the whole range must lie inside ONE pinned allocation, else the call stages.
  hipDeviceptr_t base = nullptr;
  size_t size = 0;
  if (hipMemGetAddressRange(&base, &size, (hipDeviceptr_t)at.devicePointer) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  const uintptr_t off = (uintptr_t)at.devicePointer - (uintptr_t)base;
  if (!(off <= size && bytes <= size - off)) return false;
  if (lo) *lo = a;
  if (hi) *hi = a + bytes;
  return true;
}

// Only the WHOLE PAGES inside the array are registered with the runtime.  hipHostRegister pins (and hipHostUnregister unmaps)
// every page the range touches; the first and the last page of an array on the heap also hold its neighbours, and after such a
// page had been unmapped, later transfers of the runtime itself from pageable memory in that neighbourhood (its own in-place
// pinning: a model upload of this library, torch's `.cuda()` of a numpy array) died with "Memory access fault by GPU" on a
// page-aligned heap address -- about one run of the test suite in three (round 6; none in seven runs without the two tests that
// registered 320-byte and 960 KB arrays).  The partial pages travel as pageable copies (copy_split).
int host_register(void *p, size_t bytes) {
  if (!p || bytes == 0) return fail(VCMI_ERR_ARG, "vcmi_host_register: NULL pointer or zero length");
  // (one critical section over the overlap test, the registration and the table entry: two threads registering overlapping
  // ranges at once could both pass the test -- ADVICE r5)
  std::lock_guard<std::mutex> lk(g_pinned_mu);
  for (const PinnedRange &r : g_pinned)
    if ((uintptr_t)p < r.base + r.bytes && r.base < (uintptr_t)p + bytes)
      return fail(VCMI_ERR_ARG, "vcmi_host_register: the range overlaps one that is already registered");
  const uintptr_t a = (uintptr_t)p;
  uintptr_t lo = (a + kPage - 1) / kPage * kPage, hi = (a + bytes) / kPage * kPage;
  if (hi <= lo || hi - lo < kMinPinBytes) lo = hi = 0;
  if (hi > lo) {
    const hipError_t e = hipHostRegister((void *)lo, hi - lo, hipHostRegisterPortable);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      return fail(e == hipErrorOutOfMemory ? VCMI_ERR_OOM : VCMI_ERR_HIP, "hipHostRegister of %zu bytes failed: %s", (size_t)(hi - lo),
                  hipGetErrorString(e));
    }
  }
  g_pinned.push_back(PinnedRange{a, bytes, lo, hi});
  return VCMI_OK;
}

int host_unregister(void *p) {
  std::lock_guard<std::mutex> lk(g_pinned_mu);
  auto it = std::find_if(g_pinned.begin(), g_pinned.end(), [p](const PinnedRange &r) { return r.base == (uintptr_t)p; });
  if (it == g_pinned.end()) return fail(VCMI_ERR_ARG, "vcmi_host_unregister: %p was not registered with vcmi_host_register", p);
  // transfers of earlier calls are complete (every host-pointer entry point returns with its data delivered)
  if (it->hi > it->lo) {
    const hipError_t e = hipHostUnregister((void *)it->lo);
    if (e != hipSuccess) {                // the pages are still locked: the entry stays, so that the range is still known as pinned
      (void)hipGetLastError();
      return fail(VCMI_ERR_HIP, "hipHostUnregister failed: %s", hipGetErrorString(e));
    }
  }
  g_pinned.erase(it);
  return VCMI_OK;
}

// registered with vcmi_host_register (pinned or, for a small array, only recorded), or pinned by another runtime
int host_is_registered(const void *p, size_t bytes) {
  if (!p || bytes == 0) return 0;
  {
    std::lock_guard<std::mutex> lk(g_pinned_mu);
    const uintptr_t a = (uintptr_t)p;
    for (const PinnedRange &r : g_pinned)
      if (a >= r.base && a + bytes <= r.base + r.bytes) return 1;
  }
  return range_is_pinned(p, bytes) ? 1 : 0;
}

int staged_upload(void *dDst, const void *hSrc, size_t bytes, hipStream_t consumer) {
  Ring *r = nullptr;
  VCMI_TRY(current_ring(&r));
  std::lock_guard<std::mutex> lk(r->mu);
  VCMI_TRY(r->init());
  return do_upload(r, (char *)dDst, nullptr, (const char *)hSrc, bytes, consumer);
}

int upload_now(void *dDst, const void *hSrc, size_t bytes) {
  if (bytes == 0) return VCMI_OK;
  VCMI_TRY(staged_upload(dDst, hSrc, bytes, nullptr));      // the null stream waits for the last chunk ...
  VCMI_HIP(hipStreamSynchronize(nullptr));                  // ... and the caller for the null stream
  return VCMI_OK;
}

int staged_upload_gather(void *dDst, const std::vector<HostPiece> &pieces, hipStream_t consumer) {
  Ring *r = nullptr;
  VCMI_TRY(current_ring(&r));
  std::lock_guard<std::mutex> lk(r->mu);
  VCMI_TRY(r->init());
  return do_upload(r, (char *)dDst, &pieces, nullptr, total_bytes(pieces), consumer);
}

int staged_download(void *hDst, const void *dSrc, size_t bytes, hipStream_t producer) {
  Ring *r = nullptr;
  VCMI_TRY(current_ring(&r));
  std::lock_guard<std::mutex> lk(r->mu);
  VCMI_TRY(r->init());
  return do_download(r, nullptr, (char *)hDst, (const char *)dSrc, bytes, producer);
}

int staged_download_scatter(const std::vector<HostPiece> &pieces, const void *dSrc, hipStream_t producer) {
  Ring *r = nullptr;
  VCMI_TRY(current_ring(&r));
  std::lock_guard<std::mutex> lk(r->mu);
  VCMI_TRY(r->init());
  return do_download(r, &pieces, nullptr, (const char *)dSrc, total_bytes(pieces), producer);
}

int staged_pipeline(const void *hIn, size_t in_unit, size_t in_stride, void *hOut, size_t out_unit, size_t out_stride,
                    int64_t units, int64_t min_chunk_units, const ChunkLaunch &launch) {
  if (units <= 0) return VCMI_OK;
  Ring *r = nullptr;
  VCMI_TRY(current_ring(&r));
  std::lock_guard<std::mutex> lk(r->mu);
  VCMI_TRY(r->init());
  // Small calls -- one frame, one utterance: what the reference's `vc` and a caller's own frame loop pass -- skip the
  // ring: the three streams, their six event hand-overs and the two DMA transfers cost ~160 us before the first byte
  // of a 1-frame call has moved.  The kernels read the pinned input slot and write the pinned output slot THEMSELVES
  // (hipHostMalloc memory is mapped into the device's address space: loads and stores cross the link, once each, and x
  // and y are touched exactly once by these kernels), on one stream with one synchronisation.  Up to kDirectBytes of
  // input + output (VCMI_HOST_DIRECT_KB, read once; 0 = always the ring).
  static const size_t direct_bytes = [] {
    const char *e = getenv("VCMI_HOST_DIRECT_KB");
    const long kb = e ? atol(e) : -1;
    return kb >= 0 ? (size_t)kb << 10 : kDirectBytes;
  }();
  if ((size_t)units * (in_unit + out_unit) <= direct_bytes) {
    VCMI_TRY(r->reserve(r->pin_in, r->pin_in_cap, (size_t)units * in_unit, true));
    VCMI_TRY(r->reserve(r->pin_out, r->pin_out_cap, (size_t)units * out_unit, true));
    host_copy_rows(r->pin_in[0], in_unit, hIn, in_stride, in_unit, units);
    int rc = launch(r->pin_in[0], r->pin_out[0], 0, units, r->run);
    const hipError_t e = hipStreamSynchronize(r->run);       // also on the error path: nothing stays in flight on the slots
    if (rc != VCMI_OK) return rc;
    if (e != hipSuccess) return fail(VCMI_ERR_HIP, "staged_pipeline (direct): %s", hipGetErrorString(e));
    host_copy_rows(hOut, out_stride, r->pin_out[0], out_unit, out_unit, units);
    return VCMI_OK;
  }
  // A side whose array is PINNED (vcmi_host_register, or pinned by the caller's own runtime) and dense needs no staging copy:
  // the DMA engines read the caller's x / write the caller's y themselves.  The two sides are independent.
  uintptr_t in_lo = 0, in_hi = 0, out_lo = 0, out_hi = 0;
  const bool in_direct = in_stride == in_unit && range_is_pinned(hIn, (size_t)units * in_unit, &in_lo, &in_hi);
  const bool out_direct = out_stride == out_unit && range_is_pinned(hOut, (size_t)units * out_unit, &out_lo, &out_hi);
  const size_t wide = std::max(in_unit, out_unit);
  static const size_t pipe_chunk = [] {          // A/B hook, read once: VCMI_HOST_CHUNK_MB (default: kPipeChunk)
    const char *e = getenv("VCMI_HOST_CHUNK_MB");
    const long mb = e ? atol(e) : 0;
    return (mb >= 1 && mb <= 256) ? (size_t)mb << 20 : kPipeChunk;
  }();
  int64_t chunk = std::max<int64_t>(min_chunk_units, (int64_t)(pipe_chunk / std::max<size_t>(wide, 1)));
  int64_t nch = (units + chunk - 1) / chunk;
  chunk = ((units + nch - 1) / nch + 255) / 256 * 256;
  chunk = std::min(chunk, (units + 255) / 256 * 256);
  nch = (units + chunk - 1) / chunk;   // rounding the chunk up to 256 units can make the last chunk(s) empty: recount
  // Chunk boundaries.  With equal chunks the link idles in one direction for the whole first chunk (gather, upload, kernel)
  // and for the whole last one (download, scatter): ~2 ms of a 8.7 ms call at 32 MB chunks, while uniformly small chunks pay
  // their hand-overs forty times (measured: no better).  So the FIRST and LAST chunks are short -- 1/4, 1/2 of a chunk,
  // then full ones, then 1/2, 1/4 -- and only the ramps shrink.  VCMI_HOST_RAMP=0 (read once) keeps equal chunks.
  std::vector<int64_t> start;           // start[c] = first unit of chunk c; start[nch] = units
  {
    static const bool ramp = [] {
      const char *e = getenv("VCMI_HOST_RAMP");
      return !(e && e[0] == '0');
    }();
    if (ramp && nch >= 6) {
      // 1/4, 1/2 of a chunk at either end.  (Both sides pinned -- nothing but stream hand-overs per chunk -- ramps from 1/16
      // were measured: 9.3-9.6 ms per 10^6 frames against 7.4; so were uniformly smaller chunks, 12-25 ms: transfers of a few
      // MB are not what the copy path is good at.  tools/pinned_probe.py, one box.)
      int64_t steps[2];
      int ns = 0;
      for (int64_t div : {4, 2}) steps[ns++] = std::max<int64_t>(256, chunk / div / 256 * 256);
      int64_t pos = 0, tail = 0;
      for (int i = 0; i < ns; ++i) tail += steps[i];      // units of the short chunks at the end
      for (int i = 0; i < ns; ++i) {
        start.push_back(pos);
        pos += steps[i];
      }
      while (units - pos > tail + chunk) {
        start.push_back(pos);
        pos += chunk;
      }
      // what is left before the tail: one chunk of at most `chunk` units (or two halves of it)
      if (units - pos > tail) {
        const int64_t mid = units - pos - tail;
        if (mid > chunk) {
          start.push_back(pos);
          pos += (mid / 2 + 255) / 256 * 256;
        }
        start.push_back(pos);
        pos = units - tail;
      }
      for (int i = ns - 1; i >= 0; --i) {
        start.push_back(pos);
        pos += steps[i];
      }
      start.push_back(units);
    } else {
      for (int64_t c = 0; c < nch; ++c) start.push_back(c * chunk);
      start.push_back(units);
    }
    nch = (int64_t)start.size() - 1;
  }
  if (!in_direct) VCMI_TRY(r->reserve(r->pin_in, r->pin_in_cap, (size_t)chunk * in_unit, true));
  if (!out_direct) VCMI_TRY(r->reserve(r->pin_out, r->pin_out_cap, (size_t)chunk * out_unit, true));
  VCMI_TRY(r->reserve(r->dev_in, r->dev_in_cap, (size_t)chunk * in_unit, false));
  VCMI_TRY(r->reserve(r->dev_out, r->dev_out_cap, (size_t)chunk * out_unit, false));
  VCMI_TRY(r->quiesce());
  if (in_direct || out_direct) VCMI_TRY(r->edge_begin());
  // A large output array is usually fresh (`similar(X)`, numpy.empty): its first touch is ~80k page faults per 320 MB.
  // Ask for huge pages on the 2 MB-aligned interior before the workers touch it (a hint; ignored where unsupported).
  std::shared_ptr<Latch> prefault;     // joined before this function returns, on the error path too
  {
    const size_t span = (size_t)(units - 1) * out_stride + out_unit, huge = (size_t)2 << 20;
    if (!out_direct && span >= 16 * huge) {      // (a pinned output has its pages already)
      const uintptr_t a = ((uintptr_t)hOut + huge - 1) & ~(uintptr_t)(huge - 1), b = ((uintptr_t)hOut + span) & ~(uintptr_t)(huge - 1);
      if (b > a) (void)madvise(reinterpret_cast<void *>(a), b - a, MADV_HUGEPAGE);
    }
    // ... and fault the pages in AHEAD of the pipeline, on the workers, while the first chunks upload and compute:
    // MADV_POPULATE_WRITE (Linux >= 5.14) takes the write faults of a range in one call without changing its contents
    // (a strided output, vc's (D+1,T) matrix, keeps the rows it does not own).  Otherwise the faults are taken by the
    // scatter copies at the end of the pipeline, on its critical path -- where THP compaction stalls showed up as
    // 25-32 ms calls on some boxes against 14 ms on others.  Older kernels return EINVAL: nothing is lost.
    static const bool populate = [] {
      const char *e = getenv("VCMI_HOST_POPULATE");      // read once; "0" switches the pre-faulting off
      return !(e && e[0] == '0');
    }();
    if (populate && !out_direct && span >= 4 * huge) {
      char *base = reinterpret_cast<char *>(hOut);
      constexpr size_t piece = (size_t)8 << 20;
      prefault = host_async_for((int64_t)((span + piece - 1) / piece), 1, [base, span](int64_t lo, int64_t hi) {
        constexpr size_t piece = (size_t)8 << 20;
        const long pg = 4096;
        for (int64_t k = lo; k < hi; ++k) {
          uintptr_t a = (uintptr_t)(base + (size_t)k * piece), b = (uintptr_t)(base + std::min(span, (size_t)(k + 1) * piece));
          a = (a + pg - 1) & ~(uintptr_t)(pg - 1);       // whole pages inside the range only
          b &= ~(uintptr_t)(pg - 1);
          if (b > a) (void)madvise(reinterpret_cast<void *>(a), b - a, 23 /*MADV_POPULATE_WRITE*/);
        }
      });
    }
  }
  constexpr int LAG = K - 1;
  int rc = VCMI_OK;
  // (Round 4 tried to run the gather of chunk c+1 on the workers BESIDE the caller's scatter of chunk c-LAG instead of one
  // after the other: 15-19 ms per 10^6 frames against 9.2-9.8 ms on the same box -- two host copies and two DMA streams at
  // once saturate the socket's memory, and the DMAs are what gets slower.  One host copy at a time it stays.)
  auto body = [&]() -> int {
    for (int64_t c = 0; c < nch + LAG; ++c) {
      if (c < nch) {
        const int s = (int)(c % K);
        const int64_t first = start[(size_t)c], n = start[(size_t)c + 1] - first;
        const char *src = (const char *)hIn + (size_t)first * in_stride;
        if (!in_direct) {
          if (c >= K) VCMI_HIP(hipEventSynchronize(r->ev_up[s]));               // pinned slot: upload of chunk c-K done
          host_copy_rows(r->pin_in[s], in_unit, src, in_stride, in_unit, n);
          src = r->pin_in[s];
        }
        if (c >= K) VCMI_HIP(hipStreamWaitEvent(r->up, r->ev_run[s], 0));        // device slot: kernels of chunk c-K done
        if (in_direct) VCMI_TRY(copy_split(r, r->dev_in[s], const_cast<char *>(src), (size_t)n * in_unit, true, r->up, in_lo, in_hi));
        else VCMI_HIP(hipMemcpyAsync(r->dev_in[s], src, (size_t)n * in_unit, hipMemcpyHostToDevice, r->up));
        VCMI_HIP(hipEventRecord(r->ev_up[s], r->up));
        VCMI_HIP(hipStreamWaitEvent(r->run, r->ev_up[s], 0));
        if (c >= K) VCMI_HIP(hipStreamWaitEvent(r->run, r->ev_down[s], 0));      // output slot: download of chunk c-K done
        VCMI_TRY(launch(r->dev_in[s], r->dev_out[s], first, n, r->run));
        VCMI_HIP(hipEventRecord(r->ev_run[s], r->run));
        VCMI_HIP(hipStreamWaitEvent(r->down, r->ev_run[s], 0));
        // pin_out[s] was drained by the host LAG < K iterations ago
        if (out_direct) VCMI_TRY(copy_split(r, r->dev_out[s], (char *)hOut + (size_t)first * out_stride, (size_t)n * out_unit, false, r->down, out_lo, out_hi));
        else VCMI_HIP(hipMemcpyAsync(r->pin_out[s], r->dev_out[s], (size_t)n * out_unit, hipMemcpyDeviceToHost, r->down));
        VCMI_HIP(hipEventRecord(r->ev_down[s], r->down));
      }
      const int64_t j = c - LAG;
      if (out_direct) {
        if (j == nch - 1) {
          VCMI_HIP(hipStreamSynchronize(r->down));               // the caller's y is complete on return
          r->edge_finish();
        }
        continue;
      }
      if (j >= 0 && j < nch) {
        const int s = (int)(j % K);
        const int64_t first = start[(size_t)j], n = start[(size_t)j + 1] - first;
        VCMI_HIP(hipEventSynchronize(r->ev_down[s]));
        host_copy_rows((char *)hOut + (size_t)first * out_stride, out_stride, r->pin_out[s], out_unit, out_unit, n);
      }
    }
    return VCMI_OK;
  };
  rc = body();
  if (rc != VCMI_OK) (void)r->quiesce();   // leave nothing in flight on the slots
  host_async_wait(prefault);               // no pre-fault piece outlives the call (the caller may free hOut right away)
  return rc;
}

}  // namespace vcmi

// include/vcmi.h: caller-pinned arrays
extern "C" int vcmi_host_register(void *ptr, size_t bytes) {
  VCMI_TRY(vcmi::check_device());
  return vcmi::host_register(ptr, bytes);
}
extern "C" int vcmi_host_unregister(void *ptr) { return vcmi::host_unregister(ptr); }
extern "C" int vcmi_host_is_registered(const void *ptr, size_t bytes, int *flag) {
  if (!flag) return vcmi::fail(VCMI_ERR_ARG, "vcmi_host_is_registered: NULL argument");
  *flag = vcmi::host_is_registered(ptr, bytes);
  return VCMI_OK;
}

// Measurement hook (not part of include/vcmi.h; bench.py's `host_inclusive.pcie`): what the link gives THIS library's own
// staging path -- the ring's pinned slots and its upload / download streams, chunk by chunk as staged_pipeline moves them,
// without any host memcpy or kernel: out[0] = H2D alone, out[1] = D2H alone, out[2] = both directions at once (GB/s per
// direction), for `bytes` per direction.
extern "C" int vcmi_debug_pcie_probe(size_t bytes, double *out) {
  using namespace vcmi;
  if (!out || bytes == 0) return fail(VCMI_ERR_ARG, "vcmi_debug_pcie_probe: bad argument");
  Ring *r = nullptr;
  VCMI_TRY(current_ring(&r));
  std::lock_guard<std::mutex> lk(r->mu);
  VCMI_TRY(r->init());
  const size_t chunk = kPipeChunk;
  VCMI_TRY(r->reserve(r->pin_in, r->pin_in_cap, chunk, true));
  VCMI_TRY(r->reserve(r->pin_out, r->pin_out_cap, chunk, true));
  VCMI_TRY(r->reserve(r->dev_in, r->dev_in_cap, chunk, false));
  VCMI_TRY(r->reserve(r->dev_out, r->dev_out_cap, chunk, false));
  VCMI_TRY(r->quiesce());
  const int64_t nch = (int64_t)((bytes + chunk - 1) / chunk);
  for (int mode = 0; mode < 3; ++mode) {
    double best = 1e30;
    for (int rep = 0; rep < 4; ++rep) {
      VCMI_TRY(r->quiesce());
      const auto t0 = std::chrono::steady_clock::now();
      for (int64_t c = 0; c < nch; ++c) {
        const int s = (int)(c % K);
        const size_t n = std::min(chunk, bytes - (size_t)c * chunk);
        if (mode != 1) VCMI_HIP(hipMemcpyAsync(r->dev_in[s], r->pin_in[s], n, hipMemcpyHostToDevice, r->up));
        if (mode != 0) VCMI_HIP(hipMemcpyAsync(r->pin_out[s], r->dev_out[s], n, hipMemcpyDeviceToHost, r->down));
      }
      VCMI_TRY(r->quiesce());
      best = std::min(best, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    }
    out[mode] = (double)bytes / best / 1e9;
  }
  return VCMI_OK;
}

// test hooks (not part of include/vcmi.h): the worker-thread copies without a device, for the CPU stress / sanitizer tests
extern "C" int vcmi_debug_host_copy(void *dst, const void *src, size_t bytes) {
  vcmi::host_copy(dst, src, bytes);
  return VCMI_OK;
}
extern "C" int vcmi_debug_host_copy_rows(void *dst, size_t dst_stride, const void *src, size_t src_stride, size_t row_bytes,
                                         int64_t rows) {
  vcmi::host_copy_rows(dst, dst_stride, src, src_stride, row_bytes, rows);
  return VCMI_OK;
}
