// devgroup_stress.cpp -- sanitizer driver for csrc/devgroup.cpp with FOUR members on a box without a GPU.
// devgroup.cpp is compiled with -DVCMI_DEVGROUP_TEST_BACKEND, which swaps its HIP + RCCL hooks for the ones below:
// eight host-memory "devices" and an in-process all-reduce (members register their buffers; the last arrival sums in
// rank order).  What runs is the product's own worker / group_run / retire / eager-communicator / timeout logic:
//   1. sharded jobs + the two-phase E-step protocol (local statistics, then one all-reduce) -- exact sums;
//   2. vcmi_set_devices racing with group_run from other threads: every call either completes with the right sum or
//      fails with the "group was replaced" status; nothing touches freed workers (ASan) or races (TSan);
//   3. a member that never joins the collective: the others time out, the call fails, the group reports the broken
//      communicator until it is set again;
//   4. a slow member: the others wait for it and the result is still exact.
// Built by tests/test_sanitizers.py with -fsanitize=address,undefined and with -fsanitize=thread.
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#include "../../voiceconversion.jl_amd/csrc/devgroup.hpp"

namespace {

struct StubShared {
  int n = 0;
  std::mutex mu;
  std::vector<double *> bufs;
  std::vector<size_t> counts;
  int arrived = 0;
  uint64_t generation = 0;
};
struct StubComm {
  StubShared *sh;
  int rank;
  uint64_t want = 0;   // generation at which this member's pending all-reduce is complete
  bool aborted = false;
};
std::atomic<int> g_comm_inits{0}, g_comm_aborts{0};

int stub_device_count(int *n) {
  *n = 8;
  return VCMI_OK;
}
void stub_bind(int) {}
int stub_comm_init_all(void **comms, int n, const int *) {
  StubShared *sh = new StubShared();   // leaked deliberately (process-lifetime, like a real communicator's shared state)
  sh->n = n;
  sh->bufs.assign((size_t)n, nullptr);
  sh->counts.assign((size_t)n, 0);
  for (int i = 0; i < n; ++i) comms[i] = new StubComm{sh, i};
  g_comm_inits++;
  return VCMI_OK;
}
void stub_comm_destroy(void *c) { delete static_cast<StubComm *>(c); }
void stub_comm_abort(void *c) {
  g_comm_aborts++;
  delete static_cast<StubComm *>(c);
}
int stub_start(void *c, double *buf, size_t count, hipStream_t) {
  StubComm *m = static_cast<StubComm *>(c);
  StubShared *sh = m->sh;
  std::lock_guard<std::mutex> lk(sh->mu);
  sh->bufs[(size_t)m->rank] = buf;
  sh->counts[(size_t)m->rank] = count;
  m->want = sh->generation + 1;
  if (++sh->arrived == sh->n) {
    std::vector<double> sum(count, 0.0);
    for (int r = 0; r < sh->n; ++r)
      for (size_t k = 0; k < count; ++k) sum[k] += sh->bufs[(size_t)r][k];
    for (int r = 0; r < sh->n; ++r) memcpy(sh->bufs[(size_t)r], sum.data(), count * sizeof(double));
    sh->arrived = 0;
    sh->generation++;
  }
  return VCMI_OK;
}
int stub_poll(void *c, hipStream_t) {
  StubComm *m = static_cast<StubComm *>(c);
  std::lock_guard<std::mutex> lk(m->sh->mu);
  return m->sh->generation >= m->want ? 1 : 0;
}
const vcmi::DevGroupBackend kStub = {stub_device_count, stub_bind, stub_comm_init_all, stub_comm_destroy,
                                     stub_comm_abort,   stub_start, stub_poll};

}  // namespace

namespace vcmi {
const DevGroupBackend &devgroup_test_backend() { return kStub; }
}  // namespace vcmi

using namespace vcmi;

static int bad = 0;
#define CHECK(cond)                                              \
  do {                                                           \
    if (!(cond)) {                                               \
      fprintf(stderr, "CHECK failed line %d: %s\n", __LINE__, #cond); \
      bad++;                                                     \
    }                                                            \
  } while (0)

// the two-phase E-step protocol of csrc/estep.hip on host buffers: returns VCMI_OK and the exact sum, or a status
static int two_phase(int m, int len, std::vector<std::vector<double>> &stats, int skip_member = -1, int slow_member = -1) {
  stats.assign((size_t)m, std::vector<double>((size_t)len, 0.0));
  int rc = group_run(m, [&](int i) -> int {
    for (int k = 0; k < len; ++k) stats[(size_t)i][(size_t)k] = (double)(i + 1) * (k + 1);
    return VCMI_OK;
  });
  if (rc != VCMI_OK) return rc;
  return group_run(m, [&](int i) -> int {
    if (i == skip_member) return VCMI_OK;   // fault injection: this member never reaches the collective
    if (i == slow_member) std::this_thread::sleep_for(std::chrono::milliseconds(150));
    return group_allreduce_sum(i, stats[(size_t)i].data(), (size_t)len, nullptr);
  });
}

int main() {
  int devs4[4] = {0, 1, 2, 3}, devs4b[4] = {4, 5, 6, 7}, devs2[2] = {1, 2}, got[8], n = -1;
  // 1. four members, eager communicators, sharding and the exact all-reduce
  CHECK(vcmi_set_devices(devs4, 4) == VCMI_OK);
  CHECK(g_comm_inits.load() == 1);                       // made by vcmi_set_devices, not by the first all-reduce
  CHECK(vcmi_get_devices(got, 8, &n) == VCMI_OK && n == 4 && got[3] == 3);
  CHECK(group_size() == 4 && group_device(2) == 2 && group_device(4) == -1);
  {
    std::vector<int64_t> seen(4, 0);
    CHECK(group_run(4, [&](int i) -> int {
            int64_t lo, hi;
            shard_range(1003, i, 4, &lo, &hi);
            seen[(size_t)i] = hi - lo;
            return VCMI_OK;
          }) == VCMI_OK);
    CHECK(seen[0] + seen[1] + seen[2] + seen[3] == 1003);
    std::vector<std::vector<double>> st;
    for (int rep = 0; rep < 50; ++rep) {
      CHECK(two_phase(4, 257, st) == VCMI_OK);
      for (int i = 0; i < 4; ++i) CHECK(st[(size_t)i][256] == 10.0 * 257);
    }
    CHECK(g_comm_inits.load() == 1);
    // a failing member's status and message come back; the others complete
    int rc = group_run(4, [&](int i) -> int { return i == 2 ? fail(VCMI_ERR_DIM, "member %d failed on purpose", i) : VCMI_OK; });
    CHECK(rc == VCMI_ERR_DIM && strstr(vcmi_last_error(), "member 2 failed on purpose"));
    // shards cut for another member count are refused
    CHECK(group_run(3, [&](int) -> int { return VCMI_OK; }) == VCMI_ERR_ARG);
  }
  // 4. a slow member only delays the result
  {
    std::vector<std::vector<double>> st;
    CHECK(two_phase(4, 33, st, -1, /*slow=*/1) == VCMI_OK);
    CHECK(st[0][32] == 10.0 * 33 && st[3][0] == 10.0);
  }
  // 2. vcmi_set_devices racing with runs from two threads
  {
    std::atomic<bool> stop{false};
    std::atomic<int> ok{0}, replaced{0}, other{0};
    auto runner = [&] {
      std::vector<std::vector<double>> st;
      while (!stop.load()) {
        const int m = group_size();
        if (m == 0) continue;
        const int rc = two_phase(m, 64, st);
        if (rc == VCMI_OK) {
          const double want = 0.5 * m * (m + 1) * 64;
          bool good = true;
          for (int i = 0; i < m; ++i) good = good && st[(size_t)i][63] == want;
          good ? ok++ : other++;
        } else if (rc == VCMI_ERR_ARG && strstr(vcmi_last_error(), "replaced")) {
          replaced++;
        } else if (rc == VCMI_ERR_ARG && strstr(vcmi_last_error(), "no device group")) {
          replaced++;
        } else {
          fprintf(stderr, "unexpected status %d: %s\n", rc, vcmi_last_error());
          other++;
        }
      }
    };
    std::thread a(runner), b(runner);
    for (int it = 0; it < 120; ++it) {
      const int *d = (it % 3 == 0) ? devs4 : (it % 3 == 1) ? devs4b : devs2;
      CHECK(vcmi_set_devices(d, it % 3 == 2 ? 2 : 4) == VCMI_OK);
      std::this_thread::sleep_for(std::chrono::microseconds(300));
    }
    stop.store(true);
    a.join();
    b.join();
    CHECK(other.load() == 0);
    CHECK(ok.load() > 0);
    printf("devgroup_stress: %d runs completed, %d refused after a group swap\n", ok.load(), replaced.load());
  }
  // 3. a member that never joins: timeout, clean failure, sticky until the group is set again
  {
    CHECK(vcmi_set_devices(devs4, 4) == VCMI_OK);
    group_set_timeout_ms(200);
    std::vector<std::vector<double>> st;
    const auto t0 = std::chrono::steady_clock::now();
    int rc = two_phase(4, 16, st, /*skip=*/3);
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    CHECK(rc == VCMI_ERR_HIP && strstr(vcmi_last_error(), "timed out"));
    CHECK(dt > 0.15 && dt < 5.0);
    CHECK(g_comm_aborts.load() == 3);
    rc = two_phase(4, 16, st);
    CHECK(rc == VCMI_ERR_HIP && strstr(vcmi_last_error(), "set the device group again"));
    CHECK(vcmi_set_devices(devs4, 4) == VCMI_OK);      // new communicators
    group_set_timeout_ms(60000);
    CHECK(two_phase(4, 16, st) == VCMI_OK && st[2][15] == 10.0 * 16);
  }
  // duplicate devices: sharding works, the collective is refused (RCCL: one rank per GPU)
  {
    int dup[2] = {0, 0};
    CHECK(vcmi_set_devices(dup, 2) == VCMI_OK);
    std::vector<std::vector<double>> st;
    CHECK(two_phase(2, 8, st) == VCMI_ERR_ARG && strstr(vcmi_last_error(), "distinct devices"));
  }
  CHECK(vcmi_set_devices(nullptr, 0) == VCMI_OK);
  CHECK(group_size() == 0);
  CHECK(group_run(1, [&](int) -> int { return VCMI_OK; }) == VCMI_ERR_ARG);
  int nine[1] = {9};
  CHECK(vcmi_set_devices(nine, 1) == VCMI_ERR_ARG);
  printf("devgroup_stress: %s\n", bad ? "FAILED" : "ok");
  return bad ? 1 : 0;
}
