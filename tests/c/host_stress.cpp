// host_stress.cpp -- sanitizer driver for the host-only part of libvcmi (core.cpp, hostpipe.cpp, devgroup.cpp):
// several threads run parallel copies at once (the completion latch of each copy lives on its caller's stack), the
// device-group entry points are exercised on their no-device error paths, and oracle-independent invariants are
// checked.  Built twice by tests/test_sanitizers.py: -fsanitize=address,undefined and -fsanitize=thread.
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <thread>
#include <vector>

#include "../../include/vcmi.h"

extern "C" int vcmi_debug_host_copy(void *dst, const void *src, size_t bytes);
extern "C" int vcmi_debug_host_copy_rows(void *dst, size_t dst_stride, const void *src, size_t src_stride, size_t row_bytes,
                                         int64_t rows);

int main() {
  std::atomic<int> bad{0};
  std::vector<std::thread> th;
  for (int t = 0; t < 6; ++t)
    th.emplace_back([t, &bad] {
      std::mt19937_64 rng(1234 + t);
      for (int it = 0; it < 40; ++it) {
        const size_t n = 1 + rng() % 5000000;
        std::vector<unsigned char> src(n), dst(n + 16, 0);
        for (size_t i = 0; i < n; i += 97) src[i] = (unsigned char)(rng() & 0xff);
        vcmi_debug_host_copy(dst.data() + 8, src.data(), n);
        if (memcmp(dst.data() + 8, src.data(), n) != 0 || dst[7] != 0 || dst[8 + n] != 0) bad++;
        const int64_t rows = 1 + (int64_t)(rng() % 3000);
        const size_t rb = 8 + 8 * (rng() % 60), ss = rb + 8 * (rng() % 3), ds = rb + 8 * (rng() % 3);
        std::vector<unsigned char> a(rows * ss, 3), b(rows * ds, 9);
        vcmi_debug_host_copy_rows(b.data(), ds, a.data(), ss, rb, rows);
        for (int64_t r = 0; r < rows; r += 17)
          if (b[r * ds] != 3 || (ds > rb && b[r * ds + rb] != 9)) bad++;
      }
    });
  for (auto &x : th) x.join();
  // device-group entry points without a device: clean errors, no leaks, no races
  int devs[2] = {0, 1}, n = -1, got[4];
  int rc = vcmi_set_devices(devs, 2);
  if (rc == VCMI_OK) {   // a GPU is present: make and drop a group
    rc = vcmi_get_devices(got, 4, &n);
    if (rc != VCMI_OK || n != 2) bad++;
  } else if (rc != VCMI_ERR_NO_DEVICE && rc != VCMI_ERR_ARG) {
    bad++;
  }
  if (vcmi_set_devices(nullptr, 0) != VCMI_OK) bad++;
  if (vcmi_get_devices(got, 4, &n) != VCMI_OK || n != 0) bad++;
  if (vcmi_set_devices(nullptr, 3) != VCMI_ERR_ARG) bad++;
  if (!strstr(vcmi_last_error(), "vcmi_set_devices")) bad++;
  printf("host_stress: %s\n", bad.load() ? "FAILED" : "ok");
  return bad.load() ? 1 : 0;
}
