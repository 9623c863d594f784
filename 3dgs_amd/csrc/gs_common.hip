// gs_common.hip -- status/error text, device-pointer validation, scratch arena.
#include "gs_common.h"

#include <map>
#include <mutex>
#include <unordered_map>
#include <vector>

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

int DeviceBuffer::reserve(size_t want, hipStream_t user) {
  if (want > wanted) wanted = want;
  if (want <= bytes) return GSPLAT_OK;
  size_t grow = want + want / 4 + 256;
  if (pooled) {
    // The request is the plain size when the buffer comes back after a detach (the block it gave away is of exactly
    // that class: no allocator call in the steady state of a loop that hands its outputs over every iteration), with
    // headroom when it grows.  GROWING returns the old block to the pool, whose next taker may sit on another stream
    // (two contexts on two torch streams): the device is synchronised first, as the hipMalloc path does before its
    // hipFree -- growth only happens while sizes are still settling.
    if (ptr) {
      hipError_t e = hipDeviceSynchronize();
      if (e != hipSuccess) { set_error("scratch: hipDeviceSynchronize: %s", hipGetErrorString(e)); return GSPLAT_ERR_HIP; }
      (void)pool_free_quiet(ptr);  // (the device has just been synchronised)
      ptr = nullptr;
      bytes = 0;
    } else {
      grow = want;
    }
    ++generation;
    void *fresh = nullptr;
    const int rc = gsplat_pool_alloc_on(&fresh, grow, user);  // r06: ordered behind whoever returned the block
    if (rc != GSPLAT_OK) return rc;
    ptr = fresh;
    bytes = grow;  // (the block may be larger -- its size class -- but only this much is promised)
    return GSPLAT_OK;
  }
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

void *DeviceBuffer::detach() {
  detached_bytes = bytes;  // what reserve_again() asks for: the same headroom, hence the same pool class
  void *p = ptr;
  ptr = nullptr;
  bytes = 0;
  ++generation;
  return p;
}

void DeviceBuffer::release() {
  if (ptr) {
    if (pooled) (void)pool_free_quiet(ptr);  // (release() runs behind a device synchronisation: context destruction)
    else (void)hipFree(ptr);
  }
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
  hipError_t e = hipHostMalloc((void **)&p, 512 * sizeof(int), hipHostMallocDefault);
  if (e != hipSuccess) { p = nullptr; set_error("hipHostMalloc failed: %s", hipGetErrorString(e)); return GSPLAT_ERR_HIP; }
  return GSPLAT_OK;
}
HostWords &host_words() { return g_words; }

}  // namespace gs

// ---- device block pool (gsplat_pool_alloc / gsplat_pool_free): see include/gsplat_hip.h
namespace gs {
namespace {
// `freed_on`: the stream whose queued work may still touch the block when it was returned (gsplat_pool_free_on), or
// kQuiet when nothing can (the returner had synchronised the device).  r06: a request from ANOTHER stream first orders
// itself behind that stream (pool_order_behind) -- until r05 a freed block was reusable at once by anybody, correct only
// because every caller sat on the NULL stream (the reference host owns a second stream: cuda/trainer.cu:1257-1259).
struct PoolBlock { size_t cls; int device; hipStream_t freed_on; };
const hipStream_t kQuiet = reinterpret_cast<hipStream_t>(~(uintptr_t)0);
hipEvent_t g_pool_event[64] = {};  // scratch event of pool_order_behind, one per device (used under the pool mutex)
unsigned long long g_pool_cross_stream = 0;  // reuses that needed the ordering (gsplat_pool_cross_stream_reuses)
std::mutex g_pool_mutex;
std::unordered_map<void *, PoolBlock> g_pool_live, g_pool_idle_info;
std::map<std::pair<int, size_t>, std::vector<void *>> g_pool_idle;  // (device, class size) -> cached blocks
size_t g_pool_idle_bytes = 0, g_pool_live_bytes = 0;
std::unordered_map<const void *, const unsigned char **> g_pool_watch;      // block -> slot to clear when it is freed
std::unordered_map<const unsigned char **, const void *> g_pool_watch_of;   // slot -> the block it watches

// size classes with three mantissa bits: at most 12.5 % of a block is slack, and the per-view sizes of a training run
// (they follow the visible count and the instance count, which change a little from view to view) fall into the same
// few classes iteration after iteration
size_t pool_class(size_t bytes) {
  if (bytes <= 4096) return 4096;
  int e = 63 - __builtin_clzll((unsigned long long)(bytes - 1));
  const size_t step = (size_t)1 << (e > 3 ? e - 3 : 0);
  return (bytes + step - 1) / step * step;
}

int pool_drop_idle_locked() {  // hipFree of every cached block (synchronises each device that owns one)
  if (g_pool_idle_bytes == 0) return GSPLAT_OK;
  int here = 0;
  (void)hipGetDevice(&here);
  int synced = -1;
  for (auto &kv : g_pool_idle) {  // the map is ordered by (device, class): one synchronisation per device
    if (kv.second.empty()) continue;
    const int dev = kv.first.first;
    if (dev != synced) {  // work that still reads a block runs on the block's OWN device
      (void)hipSetDevice(dev);
      (void)hipDeviceSynchronize();
      synced = dev;
    }
    for (void *p : kv.second) (void)hipFree(p);
  }
  (void)hipSetDevice(here);
  g_pool_idle.clear();
  g_pool_idle_info.clear();
  g_pool_idle_bytes = 0;
  return GSPLAT_OK;
}
}  // namespace
}  // namespace gs

namespace gs {
void pool_unwatch(const unsigned char **slot) {
  std::lock_guard<std::mutex> lock(g_pool_mutex);
  auto it = g_pool_watch_of.find(slot);
  if (it == g_pool_watch_of.end()) return;
  g_pool_watch.erase(it->second);
  g_pool_watch_of.erase(it);
}
void pool_watch(const void *block, const unsigned char **slot) {
  std::lock_guard<std::mutex> lock(g_pool_mutex);
  auto it = g_pool_watch_of.find(slot);
  if (it != g_pool_watch_of.end()) { g_pool_watch.erase(it->second); g_pool_watch_of.erase(it); }
  if (!block) return;
  auto old = g_pool_watch.find(block);  // (another slot watched this block: the newest watcher wins, the old one is cleared)
  if (old != g_pool_watch.end()) { *old->second = nullptr; g_pool_watch_of.erase(old->second); g_pool_watch.erase(old); }
  g_pool_watch[block] = slot;
  g_pool_watch_of[slot] = block;
}
}  // namespace gs

namespace gs {
namespace {
// Work queued on `taker` from now on runs behind everything queued on `giver` so far (which includes whatever was queued
// when the block was returned: the record happens later).  Same stream: stream order already says so.  The legacy NULL
// stream orders itself with every BLOCKING stream by definition, but not with hipStreamNonBlocking ones (torch's side
// streams, the context's sort streams), so it gets the event like any other pair.  A giver that has been destroyed
// meanwhile cannot be recorded on: the device is synchronised instead (its queued work may still be running).
int pool_order_behind(hipStream_t giver, hipStream_t taker) {
  if (giver == kQuiet || giver == taker) return GSPLAT_OK;
  ++g_pool_cross_stream;
  int dev = -1;  // (an event belongs to the device it was created on: blocks are only ever reused on their own device)
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = -1;
  if (dev >= 0 && !g_pool_event[dev] && hipEventCreateWithFlags(&g_pool_event[dev], hipEventDisableTiming) != hipSuccess) {
    (void)hipGetLastError();
    g_pool_event[dev] = nullptr;
  }
  if (dev >= 0 && g_pool_event[dev] && hipEventRecord(g_pool_event[dev], giver) == hipSuccess &&
      hipStreamWaitEvent(taker, g_pool_event[dev], 0) == hipSuccess)
    return GSPLAT_OK;
  (void)hipGetLastError();
  if (hipDeviceSynchronize() != hipSuccess) {
    set_error("gsplat_pool_alloc: could not order the new owner of a block behind the stream that returned it");
    return GSPLAT_ERR_HIP;
  }
  return GSPLAT_OK;
}

int pool_free_impl(void *ptr, hipStream_t freed_on) {
  if (!ptr) return GSPLAT_OK;
  std::lock_guard<std::mutex> lock(g_pool_mutex);
  auto it = g_pool_live.find(ptr);
  if (it == g_pool_live.end()) {
    set_error("gsplat_pool_free: %p was not allocated by gsplat_pool_alloc (or was freed twice)", ptr);
    return GSPLAT_ERR_INVALID_ARG;
  }
  PoolBlock b = it->second;
  b.freed_on = freed_on;
  g_pool_live.erase(it);
  {
    auto w = g_pool_watch.find(ptr);
    if (w != g_pool_watch.end()) {  // somebody remembered this block by pointer: it stops meaning what it meant
      *w->second = nullptr;
      g_pool_watch_of.erase(w->second);
      g_pool_watch.erase(w);
    }
  }
  g_pool_live_bytes -= b.cls;
  g_pool_idle[{b.device, b.cls}].push_back(ptr);
  g_pool_idle_info[ptr] = b;
  g_pool_idle_bytes += b.cls;
  return GSPLAT_OK;
}
}  // namespace
// a block nothing on the device can still be touching (the caller has synchronised): reusable by any stream at once
int pool_free_quiet(void *ptr) { return pool_free_impl(ptr, kQuiet); }
}  // namespace gs

extern "C" {
int gsplat_pool_alloc_on(void **ptr, size_t bytes, void *stream) {
  GS_REQUIRE(ptr != nullptr, "ptr is null");
  *ptr = nullptr;
  if (bytes == 0) return GSPLAT_OK;
  int dev = 0;
  GS_HIP(hipGetDevice(&dev));
  const size_t cls = gs::pool_class(bytes);
  const hipStream_t taker = (hipStream_t)stream;
  std::lock_guard<std::mutex> lock(gs::g_pool_mutex);
  auto it = gs::g_pool_idle.find({dev, cls});
  if (it != gs::g_pool_idle.end() && !it->second.empty()) {
    // a block this stream returned itself (or a quiet one) needs no ordering: look for one among the most recent returns
    std::vector<void *> &v = it->second;
    size_t pick = v.size() - 1;
    for (size_t k = v.size(), seen = 0; k-- > 0 && seen < 8; ++seen) {
      const hipStream_t f = gs::g_pool_idle_info[v[k]].freed_on;
      if (f == taker || f == gs::kQuiet) { pick = k; break; }
    }
    void *p = v[pick];
    const gs::PoolBlock b = gs::g_pool_idle_info[p];
    const int rc = gs::pool_order_behind(b.freed_on, taker);
    if (rc != GSPLAT_OK) return rc;
    v.erase(v.begin() + (long)pick);
    gs::g_pool_idle_info.erase(p);
    gs::g_pool_idle_bytes -= cls;
    gs::g_pool_live[p] = {cls, dev, gs::kQuiet};
    gs::g_pool_live_bytes += cls;
    *ptr = p;
    return GSPLAT_OK;
  }
  void *p = nullptr;
  hipError_t e = hipMalloc(&p, cls);
  if (e != hipSuccess) {  // out of memory with blocks cached: give them back and try once more
    (void)hipGetLastError();
    gs::pool_drop_idle_locked();
    e = hipMalloc(&p, cls);
  }
  if (e != hipSuccess) {
    (void)hipGetLastError();
    gs::set_error("gsplat_pool_alloc: hipMalloc(%zu) failed: %s", cls, hipGetErrorString(e));
    return GSPLAT_ERR_HIP;
  }
  gs::g_pool_live[p] = {cls, dev, gs::kQuiet};
  gs::g_pool_live_bytes += cls;
  *ptr = p;
  return GSPLAT_OK;
}

int gsplat_pool_alloc(void **ptr, size_t bytes) { return gsplat_pool_alloc_on(ptr, bytes, nullptr); }

int gsplat_pool_free_on(void *ptr, void *stream) { return gs::pool_free_impl(ptr, (hipStream_t)stream); }
int gsplat_pool_free(void *ptr) { return gs::pool_free_impl(ptr, (hipStream_t)nullptr); }

unsigned long long gsplat_pool_cross_stream_reuses(void) {
  std::lock_guard<std::mutex> lock(gs::g_pool_mutex);
  return gs::g_pool_cross_stream;
}

// r06: what a context's destruction calls instead of gsplat_pool_release -- that one synchronises EVERY
// device owning an idle block and frees the blocks cached for other live contexts, the shim's vectors and other rank
// threads' devices.  This one looks at the CURRENT device only, which the caller has just synchronised (so its idle
// blocks are quiet), and frees idle blocks -- largest classes first -- only while more than `keep_bytes` of them are
// cached: a process that creates and destroys contexts of different sizes hoards at most that much.
int gsplat_pool_trim(size_t keep_bytes) {
  int dev = 0;
  GS_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lock(gs::g_pool_mutex);
  size_t idle_here = 0;
  for (auto &kv : gs::g_pool_idle)
    if (kv.first.first == dev) idle_here += kv.first.second * kv.second.size();
  if (idle_here <= keep_bytes) return GSPLAT_OK;
  bool synced = false;
  for (auto it = gs::g_pool_idle.rbegin(); it != gs::g_pool_idle.rend() && idle_here > keep_bytes; ++it) {
    if (it->first.first != dev) continue;
    std::vector<void *> &v = it->second;
    while (!v.empty() && idle_here > keep_bytes) {
      void *p = v.back();
      if (gs::g_pool_idle_info[p].freed_on != gs::kQuiet && !synced) {  // returned behind the caller's synchronisation
        (void)hipDeviceSynchronize();
        synced = true;
      }
      v.pop_back();
      gs::g_pool_idle_info.erase(p);
      (void)hipFree(p);
      gs::g_pool_idle_bytes -= it->first.second;
      idle_here -= it->first.second;
    }
  }
  return GSPLAT_OK;
}

int gsplat_pool_release(void) {
  std::lock_guard<std::mutex> lock(gs::g_pool_mutex);
  return gs::pool_drop_idle_locked();
}

size_t gsplat_pool_bytes(int idle_only) {
  std::lock_guard<std::mutex> lock(gs::g_pool_mutex);
  return idle_only ? gs::g_pool_idle_bytes : gs::g_pool_idle_bytes + gs::g_pool_live_bytes;
}

const char *gsplat_last_error(void) { return gs::g_err; }
int gsplat_abi_version(void) { return GSPLAT_ABI_VERSION; }
int gsplat_release_scratch(void) {
  gs::ScratchLock lock;
  (void)hipDeviceSynchronize();
  for (int i = 0; i < gs::SCR_NUM; ++i) gs::g_scratch[i].release();
  return GSPLAT_OK;
}
}
