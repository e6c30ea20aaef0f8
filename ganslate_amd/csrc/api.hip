// Library lifecycle + error plumbing for libganslate_hip.so (see include/ganslate_hip.h).
#include "common.hpp"
#include <cstdarg>
#include <cstdio>
#include <cstring>

namespace {
char g_err[512] = "";
void* g_zero = nullptr;      // 256-byte zero page: source of masked LDS-DMA lanes
float* g_reduce_ws = nullptr;  // 1024 partials + 1 arrival counter for the loss reductions
}  // namespace

void gs_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
const void* gs_zero_page() { return g_zero; }
float* gs_reduce_workspace() { return g_reduce_ws; }

extern "C" const char* gs_last_error(void) { return g_err; }

extern "C" int gs_init(int device) {
  GS_CHECK_HIP(hipSetDevice(device));
  if (!g_zero) {
    GS_CHECK_HIP(hipMalloc(&g_zero, 256));
    GS_CHECK_HIP(hipMemset(g_zero, 0, 256));
  }
  if (!g_reduce_ws) {
    GS_CHECK_HIP(hipMalloc(reinterpret_cast<void**>(&g_reduce_ws), 1040 * sizeof(float)));
    GS_CHECK_HIP(hipMemset(g_reduce_ws, 0, 1040 * sizeof(float)));
  }
  hipDeviceProp_t prop;
  GS_CHECK_HIP(hipGetDeviceProperties(&prop, device));
  GS_REQUIRE(strncmp(prop.gcnArchName, "gfx950", 6) == 0,
             "gs_init: device %d is %s; this library is built for gfx950 (MI355X) only", device, prop.gcnArchName);
  return 0;
}

extern "C" void gs_shutdown(void) {
  if (g_zero) { (void)hipFree(g_zero); g_zero = nullptr; }
  if (g_reduce_ws) { (void)hipFree(g_reduce_ws); g_reduce_ws = nullptr; }
}
