// Library-level entry points of libfyc_hip.so: init, capabilities, error text.
#include "fyc_common.h"

thread_local char g_fyc_err[512] = {0};
const void* g_fyc_zero_page = nullptr;
int g_fyc_tuning[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

extern "C" int fyc_version(void) { return FYC_VERSION; }

extern "C" const char* fyc_last_error(void) { return g_fyc_err; }

extern "C" int fyc_init(const void* zero_page) {
  FYC_REQUIRE(zero_page != nullptr, "fyc_init: zero_page is null");
  FYC_REQUIRE(((uintptr_t)zero_page % 256) == 0, "fyc_init: zero_page must be 256-byte aligned");
  g_fyc_zero_page = zero_page;
  return 0;
}

#ifdef FYC_TRACE
unsigned long long* g_fyc_trace = nullptr;
// timing builds only (not part of include/fyc.h): [blocks][2][128] u64, zeroed by the caller before each launch
extern "C" int fyc_set_trace(void* buf) { g_fyc_trace = (unsigned long long*)buf; return 0; }
#endif

extern "C" int fyc_set_tuning(int key, int value) {
  FYC_REQUIRE(key >= 0 && key < 16, "fyc_set_tuning: key %d", key);
  g_fyc_tuning[key] = value;
  return 0;
}

extern "C" int fyc_device_caps(int64_t* caps) {
  FYC_REQUIRE(caps != nullptr, "fyc_device_caps: null");
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) FYC_FAIL(-3, "fyc_device_caps: hipGetDevice: %s", hipGetErrorString(e));
  hipDeviceProp_t pr;
  e = hipGetDeviceProperties(&pr, dev);
  if (e != hipSuccess) FYC_FAIL(-3, "fyc_device_caps: hipGetDeviceProperties: %s", hipGetErrorString(e));
  caps[0] = pr.multiProcessorCount;
  caps[1] = (int64_t)pr.maxSharedMemoryPerMultiProcessor;
  caps[2] = pr.warpSize;
  int arch = 0;
  const char* s = strstr(pr.gcnArchName, "gfx");
  if (s) arch = (int)strtol(s + 3, nullptr, 10);
  caps[3] = arch;
  caps[4] = pr.clockRate;
  caps[5] = pr.l2CacheSize;
  caps[6] = (int64_t)(pr.totalGlobalMem >> 20);
  caps[7] = 0;
  return 0;
}
