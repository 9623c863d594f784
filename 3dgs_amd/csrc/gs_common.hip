// gs_common.hip -- status/error text, device-pointer validation, scratch arena.
#include "gs_common.h"

#include <mutex>

namespace gs {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int check_device_ptr(const void *p, const char *name, const char *fn) {
  if (p == nullptr) {
    set_error("%s: pointer '%s' is null", fn, name);
    return GSPLAT_ERR_NULL_POINTER;
  }
  hipPointerAttribute_t attr;
  hipError_t e = hipPointerGetAttributes(&attr, p);
  if (e != hipSuccess) {
    (void)hipGetLastError();  // clear the sticky error
    set_error("%s: pointer '%s' is not known to the HIP runtime (%s)", fn, name, hipGetErrorString(e));
    return GSPLAT_ERR_NOT_DEVICE;
  }
  if (attr.type != hipMemoryTypeDevice) {
    set_error("%s: pointer '%s' is not a device pointer", fn, name);
    return GSPLAT_ERR_NOT_DEVICE;
  }
  return GSPLAT_OK;
}

int DeviceBuffer::reserve(size_t want) {
  if (want <= bytes) return GSPLAT_OK;
  size_t grow = want + want / 4 + 256;
  if (ptr) {
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) { set_error("scratch: hipDeviceSynchronize: %s", hipGetErrorString(e)); return GSPLAT_ERR_HIP; }
    (void)hipFree(ptr);
    ptr = nullptr;
    bytes = 0;
  }
  ++generation;
  hipError_t e = hipMalloc(&ptr, grow);
  if (e != hipSuccess) {
    ptr = nullptr;
    set_error("scratch: hipMalloc(%zu) failed: %s", grow, hipGetErrorString(e));
    return GSPLAT_ERR_HIP;
  }
  bytes = grow;
  return GSPLAT_OK;
}

void DeviceBuffer::release() {
  if (ptr) (void)hipFree(ptr);
  ptr = nullptr;
  bytes = 0;
  ++generation;
}

static DeviceBuffer g_scratch[SCR_NUM];
DeviceBuffer &scratch(ScratchSlot slot) { return g_scratch[slot]; }
static std::recursive_mutex g_scratch_mutex;
ScratchLock::ScratchLock() { g_scratch_mutex.lock(); }
ScratchLock::~ScratchLock() { g_scratch_mutex.unlock(); }

int SortFork::create() {
  if (ready) return GSPLAT_OK;
  if (hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming) != hipSuccess) { set_error("hipEventCreate failed"); return GSPLAT_ERR_HIP; }
  for (int k = 0; k < 3; ++k)
    if (hipStreamCreateWithFlags(&side[k], hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&ev_join[k], hipEventDisableTiming) != hipSuccess) {
      set_error("could not create the sort side streams");
      destroy();
      return GSPLAT_ERR_HIP;
    }
  ready = true;
  return GSPLAT_OK;
}

void SortFork::destroy() {
  for (int k = 0; k < 3; ++k) {
    if (ev_join[k]) (void)hipEventDestroy(ev_join[k]);
    if (side[k]) (void)hipStreamDestroy(side[k]);
    ev_join[k] = nullptr; side[k] = nullptr;
  }
  if (ev_fork) (void)hipEventDestroy(ev_fork);
  ev_fork = nullptr;
  ready = false;
}

static HostWords g_words;
int HostWords::ensure() {
  if (p) return GSPLAT_OK;
  hipError_t e = hipHostMalloc((void **)&p, 64 * sizeof(int), hipHostMallocDefault);
  if (e != hipSuccess) { p = nullptr; set_error("hipHostMalloc failed: %s", hipGetErrorString(e)); return GSPLAT_ERR_HIP; }
  return GSPLAT_OK;
}
HostWords &host_words() { return g_words; }

}  // namespace gs

extern "C" {
const char *gsplat_last_error(void) { return gs::g_err; }
int gsplat_abi_version(void) { return GSPLAT_ABI_VERSION; }
int gsplat_release_scratch(void) {
  gs::ScratchLock lock;
  (void)hipDeviceSynchronize();
  for (int i = 0; i < gs::SCR_NUM; ++i) gs::g_scratch[i].release();
  return GSPLAT_OK;
}
}
