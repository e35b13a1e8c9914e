// core.cpp -- error plumbing, device selection, version string of libvcmi.
#include <cstdlib>
#include "vcmi_common.hpp"

#include <atomic>

namespace vcmi {

char *error_buffer() {
  static thread_local char buf[512] = "";
  return buf;
}

int fail(int code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(error_buffer(), 512, fmt, ap);
  va_end(ap);
  return code;
}

int check_device() {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n < 1) {
    (void)hipGetLastError();
    return fail(VCMI_ERR_NO_DEVICE, "no HIP device visible (%s)", e == hipSuccess ? "count is 0" : hipGetErrorString(e));
  }
  return VCMI_OK;
}

static std::atomic<unsigned> g_debug_flags{0};
bool debug_flag(unsigned which) { return (g_debug_flags.load(std::memory_order_relaxed) & which) != 0; }

}  // namespace vcmi

// Test hook, deliberately absent from include/vcmi.h: forces kernels a shape would not select by itself (A/B partners in
// the parity tests, `bench.py --debug-force`).  It is a process-global switch, so it is inert unless the PROCESS was started
// with VCMI_TEST_HOOKS=1 in its environment (tests/conftest.py sets it): a product process cannot be switched by accident.
extern "C" int vcmi_debug_force(unsigned flags) {
  static const bool enabled = [] {
    const char *e = getenv("VCMI_TEST_HOOKS");
    return e && e[0] == '1';
  }();
  if (!enabled) return vcmi::fail(VCMI_ERR_ARG, "vcmi_debug_force: test hooks are disabled (start the process with VCMI_TEST_HOOKS=1)");
  vcmi::g_debug_flags.store(flags);
  return VCMI_OK;
}

extern "C" const char *vcmi_last_error(void) { return vcmi::error_buffer(); }
extern "C" const char *vcmi_version(void) { return "vcmi 0.1 (gfx950, FP64)"; }

extern "C" int vcmi_device_count(int *count) {
  if (!count) return vcmi::fail(VCMI_ERR_ARG, "vcmi_device_count: NULL argument");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    n = 0;
  }
  *count = n;
  return VCMI_OK;
}

extern "C" int vcmi_set_device(int device) {
  VCMI_TRY(vcmi::check_device());
  VCMI_HIP(hipSetDevice(device));
  return VCMI_OK;
}
