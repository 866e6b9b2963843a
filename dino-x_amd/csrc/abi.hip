// abi.hip -- error plumbing and library-level entry points of libdinox_hip.so.
#include <cstdarg>
#include <cstring>
#include <mutex>
#include <unordered_map>

#include "common.h"

namespace dinox {

static thread_local char g_err[512] = {0};

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

// Raise a kernel's dynamic-LDS limit, ONCE per (kernel, size).  hipFuncSetAttribute is not a stream operation and must not be
// issued while a stream is being captured into a hipGraph; after the first eager launch of a shape it is never called again, so
// a captured step (dinox.engine.TrainEngine(use_graph=True)) holds kernel nodes only.
int reserve_lds(const void* kern, size_t bytes, const char* what) {
  if (bytes <= 64 * 1024) return 0;
  static std::mutex mu;
  static std::unordered_map<const void*, size_t> granted;
  std::lock_guard<std::mutex> lk(mu);
  size_t& have = granted[kern];
  if (bytes <= have) return 0;
  const hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return fail((int)e, "%s: cannot reserve %zu B of LDS (%s)", what, bytes, hipGetErrorString(e));
  }
  have = bytes;
  return 0;
}

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return (int)e;
  }
  return 0;
}

}  // namespace dinox

extern "C" int dinox_version(void) { return DINOX_ABI_VERSION; }

extern "C" const char* dinox_last_error(void) { return dinox::g_err; }

extern "C" int dinox_device_ok(void) {
  int n = 0;
  const hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) {
    (void)hipGetLastError();
    dinox::set_error("no HIP device visible (hipGetDeviceCount: %s, %d devices)", hipGetErrorString(e), n);
    return 0;
  }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, 0) != hipSuccess) {
    (void)hipGetLastError();
    dinox::set_error("hipGetDeviceProperties failed");
    return 0;
  }
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    dinox::set_error("device 0 is %s, this library is built for gfx950 only", prop.gcnArchName);
    return 0;
  }
  return 1;
}
