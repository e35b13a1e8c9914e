// core.cpp -- error plumbing, device selection, version string of libvcmi.
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

// test hook, deliberately absent from include/vcmi.h
extern "C" int vcmi_debug_force(unsigned flags) {
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
