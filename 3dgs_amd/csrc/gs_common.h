// gs_common.h -- host-side plumbing shared by all translation units of libgsplat_hip.so:
// status codes, argument validation with the reference's device-pointer rule
// (cuda/checks.cuh:17-38, but returning a status instead of exiting), launch checks and
// the library-owned scratch arena used by the stand-alone operators.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstring>

#include "../../include/gsplat_hip.h"

namespace gs {

void set_error(const char *fmt, ...);

// NULL -> GSPLAT_ERR_NULL_POINTER, host/unknown memory -> GSPLAT_ERR_NOT_DEVICE
int check_device_ptr(const void *p, const char *name, const char *fn);

#define GS_REQUIRE_DEV(p)                                                  \
  do {                                                                     \
    int _st = ::gs::check_device_ptr((p), #p, __func__);                   \
    if (_st != GSPLAT_OK) return _st;                                      \
  } while (0)

#define GS_REQUIRE(cond, msg)                                              \
  do {                                                                     \
    if (!(cond)) {                                                         \
      ::gs::set_error("%s: invalid argument: %s", __func__, msg);          \
      return GSPLAT_ERR_INVALID_ARG;                                       \
    }                                                                      \
  } while (0)

#define GS_HIP(call)                                                                          \
  do {                                                                                        \
    hipError_t _e = (call);                                                                   \
    if (_e != hipSuccess) {                                                                   \
      ::gs::set_error("%s: %s failed: %s", __func__, #call, hipGetErrorString(_e));           \
      return GSPLAT_ERR_HIP;                                                                  \
    }                                                                                         \
  } while (0)

#define GS_LAUNCH_CHECK()                                                                     \
  do {                                                                                        \
    hipError_t _e = hipGetLastError();                                                        \
    if (_e != hipSuccess) {                                                                   \
      ::gs::set_error("%s: kernel launch failed: %s", __func__, hipGetErrorString(_e));       \
      return GSPLAT_ERR_HIP;                                                                  \
    }                                                                                         \
  } while (0)

// Counting-sort binning (gs_binning.hip) and the per-gaussian forward (gs_fused.hip) split the gaussians into the same
// kBinBlocks contiguous slices: workgroup b of either kernel owns global indices [N*b/kBinBlocks, N*(b+1)/kBinBlocks).
constexpr int kBinBlocks = 256, kBinThreads = 1024;
// r03: the gaussians are dealt to those workgroups in CHUNKS of 64 consecutive entries (one wave trip), round robin:
// chunk c belongs to workgroup c % kBinBlocks, and inside it to wave (c / kBinBlocks) % 16.  A training run keeps its
// gaussians in Morton order, where contiguous index ranges differ several times in tile instances per gaussian (near /
// far, inside / outside the view): with one contiguous slice per workgroup the slowest slice decided both kernels
// (garden-shaped workload: 5.6x the mean; config 3 in Morton order: 3.1x; dealt in chunks: 1.1x).
// Only the cull's compaction ranks need contiguous slices (a local scan); those are kBinBlocks runs of whole chunks.
constexpr int kBinChunk = 64;
__host__ __device__ inline int bin_chunks(int n) { return (n + kBinChunk - 1) / kBinChunk; }
// first chunk of the cull's slice s when the index space has C chunks
__host__ __device__ inline int bin_slice_first_chunk(int C, int s) { return (int)((long long)C * s / kBinBlocks); }
// the slice that holds chunk c: the largest s with bin_slice_first_chunk(C, s) <= c
__host__ __device__ inline int bin_slice_of_chunk(int C, int c) {
  return (int)((((long long)c + 1) * kBinBlocks - 1) / C);
}
// The forward's host record: pinned, mapped host memory the GPU writes and the host polls.  Five 64-bit words, each
// {value << 32 | low half of the forward's ticket}: M, S, candidate pairs (low, high), longest tile list.  A word is
// one aligned 8-byte store, so it cannot tear, and the host takes the record when all five carry its ticket -- no
// ordering between the stores is needed, hence no __threadfence_system() (a system-scope release writes back the
// XCD's whole L2: several microseconds on the GPU's critical path, behind kernels that left megabytes dirty).
constexpr int kRecordWords = 5;
__host__ __device__ inline unsigned long long record_word(unsigned int value, unsigned long long ticket) {
  return ((unsigned long long)value << 32) | (ticket & 0xFFFFFFFFull);
}
#ifdef __HIPCC__
__device__ inline void publish_record(volatile unsigned long long *pub, unsigned long long ticket, unsigned int M,
                                      unsigned int S, unsigned long long pairs, unsigned int longest) {
  pub[0] = record_word(M, ticket);
  pub[1] = record_word(S, ticket);
  pub[2] = record_word((unsigned int)(pairs & 0xFFFFFFFFull), ticket);
  pub[3] = record_word((unsigned int)(pairs >> 32), ticket);
  pub[4] = record_word(longest, ticket);
}
#endif

constexpr int kBinMaxTiles = 16384;  // 64 KB of LDS counters; larger tile grids take the radix-sort route

static inline unsigned int div_up(long long a, long long b) { return (unsigned int)((a + b - 1) / b); }

// A growable device buffer (never shrinks).  Growing synchronises the device: it only
// happens while sizes are still settling, never in the steady state of a training loop.
struct DeviceBuffer {
  void *ptr = nullptr;
  size_t bytes = 0;
  unsigned long long generation = 0;  // bumped by every allocation and release: state cached about the CONTENTS keys on it
  // r05: a POOLED buffer takes its storage from the library's block pool (gsplat_pool_alloc) instead of hipMalloc, so
  // that its block can be handed to the caller (detach: the caller then owns it and returns it with gsplat_pool_free) and
  // the next reserve() finds a block of the same size class in the pool without touching the allocator -- how the
  // rasterize_image shim gives ForwardPassData the forward's own output arrays instead of copies (raster.cuh).
  bool pooled = false;
  size_t wanted = 0;  // the largest size ever asked for
  size_t detached_bytes = 0;  // size of the block detach() gave away: what reserve_again() restores (headroom included)
  // `user`: the stream whose work will touch the storage next -- a pooled block that another stream returned is ordered
  // behind that stream first (gsplat_pool_alloc_on); the hipMalloc path synchronises the device when it grows anyway
  int reserve(size_t want, hipStream_t user = nullptr);
  int reserve_again(hipStream_t user = nullptr) {
    const size_t w = wanted;
    const int rc = reserve(detached_bytes > wanted ? detached_bytes : wanted, user);
    wanted = w;
    return rc;
  }
  void *detach();
  void release();
  template <typename T> T *as() const { return reinterpret_cast<T *>(ptr); }
};

// "Tell me when this pool block is returned": *slot is set to nullptr when `block` goes back to the pool
// (gsplat_pool_free) -- a context remembers the mask array of its last forward by POINTER (gsplat_context_last_compaction),
// and once the caller has freed that block the pointer may name somebody else's data.  One watch per slot: watching
// again replaces the previous block; pool_unwatch(slot) before the slot's owner dies.
void pool_watch(const void *block, const unsigned char **slot);
int pool_free_quiet(void *ptr);  // gsplat_pool_free for a block nothing queued on the device can still touch
void pool_unwatch(const unsigned char **slot);

// Scratch slots of the stand-alone operators (the reference allocates thrust::device_vector
// temporaries inside the same operators: cuda/culling.cu:400-463, cuda/spherical_harmonics.cu:76-79).
enum ScratchSlot {
  SCR_COUNTS = 0, SCR_OFFSETS, SCR_KEYS_A, SCR_KEYS_B, SCR_VALS_B, SCR_TEMP, SCR_SPLATS, SCR_MISC, SCR_GRADROWS,
  SCR_LOSS_MU, SCR_LOSS_S1, SCR_LOSS_S12, SCR_LOSS_ACC,
  SCR_NUM
};
DeviceBuffer &scratch(ScratchSlot slot);
// The scratch slots are process-wide: entry points that a multi-threaded host may call concurrently (the loss of a
// training loop with several in-process ranks) hold this lock from their first scratch access to their last launch /
// read-back.  Recursive, so that helpers may take it again.
struct ScratchLock {
  ScratchLock();
  ~ScratchLock();
  ScratchLock(const ScratchLock &) = delete;
  ScratchLock &operator=(const ScratchLock &) = delete;
};

// Side streams of a context for the per-tile sorts of long lists (gs_binning.hip sort_tiles_by_depth): the kernels of the
// 2049..4096 / ..8192 / ..16384-entry classes run beside the one-wave-per-tile kernel instead of behind it.
struct SortFork {
  bool ready = false;
  hipStream_t side[3] = {nullptr, nullptr, nullptr};
  hipEvent_t ev_fork = nullptr, ev_join[3] = {nullptr, nullptr, nullptr};
  int create();
  void destroy();
};

// pinned host words for count read-backs
struct HostWords {
  int *p = nullptr;
  int ensure();
};
HostWords &host_words();

}  // namespace gs
