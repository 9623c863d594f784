// gs_fused.hip -- the device-resident per-view pass: gsplat_rasterize_image (the work of
// rasterize_image, cuda/raster.cu:12-136) and gsplat_backward_pass (the operator chain of
// TrainerImpl::backward_pass, cuda/trainer.cu:941-1012) on a persistent workspace.
//
// Forward:   project+cull (N)  ->  scan(mask)  ->  preprocess (fused SH / Sigma / J / conic /
//            radius / splat record / exact tile count, written in compacted order)
//            ->  scan(counts)  ->  ONE host read-back {M, S}  ->  emit keys  ->  radix sort
//            ->  tile ranges  ->  compositing.
// Backward:  zero gradient rows -> compositing backward (64-byte row atomics) -> one fused
//            per-gaussian kernel for the whole SH / conic / Jacobian / Sigma / projection chain.
// The reference runs ~20 thrust compactions, ~15 kernels and 5 blocking read-backs for the
// same work; results are laid out exactly like its ForwardPassData / GaussianGradients.
#include <cstring>
#include <cmath>
#include <new>
#include <rocprim/rocprim.hpp>

#include <cstdlib>

#include "gs_common.h"
#include "gs_math.h"
#include "gs_render.h"
#include "gs_rows.h"

namespace gs {
// from gs_binning.hip / gs_render.hip
size_t binning_temp_bytes(size_t N, size_t S, int num_tiles);
int scan_counts(int N, const int *counts, int *offsets, void *temp, size_t temp_bytes, hipStream_t st);
bool binning_supports_counting_sort(int num_tiles);
bool binning_prefers_radix(size_t S, int num_tiles);
bool binning_next_route_is_radix(bool was_counting_sort, size_t S, int num_tiles, long long longest);
size_t binning_table_bytes(int num_tiles);
int binning_offsets(int ntx, int nty, int *table, int *long_tiles, hipStream_t st);
int binning_scatter_and_sort(const float *uv, const float *xyz_c, const float *radius,
                             const unsigned long long *hitmask, const int *rank, int N, int ntx, int nty,
                             const int *table, int *ranges, size_t S, unsigned long long *payload,
                             int *long_tiles, int *sorted_out, long long longest, const int *m_total,
                             const unsigned long long *pair_counters, unsigned long long *pub,
                             unsigned long long ticket, hipStream_t st, const SortFork *fork, bool compact_walk,
                             bool keys_ok);
int emit_sort_ranges(const float *uv, const float *xyz_c, const float *radius, int ntx, int nty, int N,
                     const unsigned char *mask, const int *rank, const int *offsets, size_t S, unsigned int *tkeys_a,
                     unsigned int *tkeys_b, unsigned long long *pay_a, unsigned long long *pay_b, int *sorted_out,
                     int *ranges, void *temp, size_t temp_bytes, hipStream_t st, bool already_emitted,
                     const unsigned long long *hitmask);
int launch_tile_emit(const float *uv, const float *xyz_c, const float *radius, int ntx, int nty, int N,
                     const unsigned char *mask, const int *rank, const int *offsets,
                     const unsigned long long *hitmask, long long capacity, unsigned int *tkeys,
                     unsigned long long *payload, hipStream_t st);
struct RawSplats;
int launch_render_fwd(const float4 *recs, const RawSplats *raw, const int *sorted, const int *ranges, int width,
                      int height, float bg, int *n_out, float *T_out, float *image, hipStream_t st, float4 *zero, long long zero_vec,
                      unsigned short *masks_out, const int *order, int *tops_out, const TileSegments *segments,
                      const FwdSegments *fwd_segments);
int launch_fwd_segments_table(const int *ranges, int num_tiles, const FwdSegments &fs, hipStream_t st);
int launch_render_bwd(const float4 *recs, const RawSplats *raw, const int *sorted, const int *ranges, const int *n_px,
                      const float *T_px, const float *grad_image, int width, int height, float bg, float *rows,
                      float *g_rgb, float *g_opacity, float *g_uv, float *g_conic, hipStream_t st,
                      const unsigned short *masks_in, hipEvent_t ev_start, hipEvent_t ev_stop, const int *order,
                      const TileSegments *segments);
int launch_tile_segments(const int *ranges, const int *tops, int num_tiles, const TileSegments &seg, hipStream_t st);
int launch_tile_order(const int *work, const int *ranges, int num_tiles, int *order, hipStream_t st);
bool tile_order_supported(int num_tiles);
}  // namespace gs

// GSPLAT_PRE_SPLIT=0|1|2: the per-gaussian forward as one kernel (the default), as sh_colour_kernel + preprocess_geom_kernel
// one behind the other, or side by side on two streams (r06: built, bit-identical, measured slower -- see sh_colour_kernel);
// per context: gsplat_context_set_preprocess_split
static int gs_pre_split_default() {
  static const int v = [] { const char *e = getenv("GSPLAT_PRE_SPLIT"); return e && e[0] >= '0' && e[0] <= '2' ? e[0] - '0' : 0; }();
  return v;
}
// GSPLAT_PRE_JOIN=late: the caller's stream waits for the colour kernel only in front of render_fwd (the first reader of
// the records' colour), not in front of the binning kernels
static bool gs_pre_join_late() {
  static const bool v = [] { const char *e = getenv("GSPLAT_PRE_JOIN"); return e && e[0] == 'l'; }();
  return v;
}

struct gsplat_context {
  int max_gaussians = 0, max_width = 0, max_height = 0;
  // per-gaussian, global order
  gs::DeviceBuffer mask, counters, rank, xyz_c_all, uv_all;
  // per-gaussian, compacted order
  gs::DeviceBuffer c2g, xyz_c, uv, sigma, conic, J, rgb, radius, recs, counts, offsets, grad_rows, hitmask;
  // instances
  gs::DeviceBuffer keys_a, keys_b, pay_a, pay_b, sorted, temp, bin_table;
  gs::DeviceBuffer blockmasks;  // per instance: the 16 block bits of the compositing kernels (forward -> backward)
  // per tile / pixel
  gs::DeviceBuffer ranges, image, T_px, n_px;
  // r04: the backward's tiles heaviest first (tile_order_kernel): per tile the largest stop index of its pixels, written
  // by render_fwd, and the order table made from it right behind the forward (off the backward's critical path)
  gs::DeviceBuffer tile_tops, tile_order;
  bool order_ready = false;  // tile_order belongs to the recorded forward
  // r05: long lists split into segments for the backward (gs_render.h: TileSegments); allocated by the first forward that
  // follows one with a list beyond kSegSplitMin
  gs::DeviceBuffer seg_first, seg_extra, seg_chk;
  // ... and for the forward (gs_render.h: FwdSegments)
  gs::DeviceBuffer fseg_first, fseg_blocks, fseg_gran, fseg_part, fseg_stop;
  void *fseg_gran_zeroed = nullptr;  // the granule block whose tags have been cleared (a fresh block holds anything)
  size_t fseg_gran_zeroed_bytes = 0;
  unsigned int fseg_epoch = 0;
  int fseg_poll_budget = gs::kFwdPollBudget, fseg_thin_layer = gs::kFwdThinLayerDefault;
  double fseg_gate = -1.0;  // < 0: GSPLAT_FWD_SEGMENTS_GATE / its default
  unsigned long long n_segmented_forwards = 0;
  // r06: the figures the segment kernels publish, as the host last took them from a slot whose ticket it could trust
  // (queue_tail): the largest stop index of any tile, the sum over the tiles, the segments / segment blocks asked for
  long long fig_max = 0, fig_sum = 0, fig_asked_bwd = 0, fig_asked_fwd = 0;
  int seg_cap = 0;          // extra segments the recorded forward had room for
  bool seg_ready = false;   // the recorded forward wrote the table and the checkpoints
  unsigned long long n_segmented_backwards = 0;
  const unsigned char *last_mask = nullptr;  // the mask array the last COMPLETED forward wrote (gsplat_context_last_compaction)
  int *h_words = nullptr;  // pinned
  bool dense_route = false;  // binning route of the next forward (follows the last one's density)
  int forced_route = 0;      // gsplat_context_set_binning_route: 0 auto, 1 counting sort, 2 radix sorts
  bool rows_ready = false;  // gsplat_backward_render has filled grad_rows for the recorded forward
  bool backward_seen = false, rows_zeroed = false;  // training use: the forward clears grad_rows for the backward
  bool render_only = false;  // gsplat_context_set_render_only: forwards skip what only a backward would read
  bool lean = false;         // gsplat_context_set_lean_forward: Sigma / J / conic / colour are not materialised
  // r06: the per-gaussian forward: 0 one fused kernel (r01-r05), 1 sh_colour_kernel then preprocess_geom_kernel on the
  // caller's stream, 2 sh_colour_kernel BESIDE preprocess_geom_kernel on a stream of the context (pre_side) -- a stream
  // of memory requests next to a kernel bound by its arithmetic (gsplat_context_set_preprocess_split)
  int pre_split = gs_pre_split_default();
  hipStream_t pre_side = nullptr;
  hipEvent_t ev_pre_fork = nullptr, ev_pre_join = nullptr;
  gs::DeviceBuffer chunk_first;  // [chunks of the index space]: slice-local rank of each chunk's first index (mode 2)
  gs::DeviceBuffer kept;     // slice-local lists of the kept gaussians (project_cull -> preprocess' compacted walk)
  gs::DeviceBuffer dir_grad;  // [M,3]: sh_adam_dir_kernel -> preprocess_bwd_kernel<L, 3> (gsplat_adam_fused.mode 2)
  gs::SortFork fork;         // side streams for the per-tile sorts of long lists (created when a forward first needs them)
  // the forward's record (gs_common.h: publish_record): pinned host memory the GPU writes and the host polls
  volatile unsigned long long *h_pub = nullptr;
  unsigned long long *d_pub = nullptr, ticket = 0;
  // small device counters: 64 spread counters of candidate pairs, then the kBinBlocks slice counts of the cull
  unsigned long long *pair_counters() const { return counters.as<unsigned long long>(); }
  int *slice_counts() const { return counters.as<int>() + 128; }
  int *fseg_fallbacks() const { return counters.as<int>() + 128 + gs::kBinBlocks; }  // (zeroed at creation, then only added to)
  // optional per-stage HIP-event timing (gsplat_context_set_timing)
  static constexpr int kStages = 8, kSlots = 32;
  unsigned int timing = 0;  // bit k: stage k is timed
  hipEvent_t ev[kSlots][2 * kStages] = {};
  unsigned char pending[kSlots][kStages] = {};
  double stage_ms[kStages] = {};
  long long stage_n[kStages] = {};
  long long fwd_calls = 0;
  // what the speculative forward did (gsplat_context_get_counters): forwards, forwards whose queued tail had to be redone
  // (instances outgrew the room, or the longest list needed a sort kernel that was not queued), forwards that walked the
  // compacted slots, growths of the instance buffers
  long long n_forwards = 0, n_tail_redone = 0, n_compact_walks = 0, n_instance_growths = 0, n_ordered_backwards = 0;
  int slot = 0;
  void mark(int stage, bool stop, hipStream_t st) {
    if (!((timing >> stage) & 1u)) return;
    (void)hipEventRecord(ev[slot][2 * stage + (stop ? 1 : 0)], st);
    if (stop) pending[slot][stage] = 1;
  }
  void harvest(int sl) {
    for (int k = 0; k < kStages; ++k)
      if (pending[sl][k]) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, ev[sl][2 * k], ev[sl][2 * k + 1]) == hipSuccess) { stage_ms[k] += ms; stage_n[k] += 1; }
        pending[sl][k] = 0;
      }
  }
  // state of the last forward
  int N = 0, M = 0, l_max = 0, width = 0, height = 0;
  float tan_fovx = 0.f, tan_fovy = 0.f, mh_dist = 0.f;  // what the per-gaussian backward needs to redo Sigma, J, conic
  size_t S = 0;
  long long last_longest = -1;  // longest tile list of the last counting-sort forward (-1: unknown)
  bool have_forward = false;
  // r05: the thirteen arrays of the reference's ForwardPassData (cuda_data.cuh:70-86).  They are pooled buffers, so that
  // gsplat_context_detach_forward_outputs can hand their blocks to the caller instead of the caller copying them.
  void forward_outputs(gs::DeviceBuffer *out[13]) {
    gs::DeviceBuffer *all[13] = {&mask, &uv_all, &xyz_c_all, &sigma, &conic, &J, &rgb, &radius, &sorted, &ranges, &image, &T_px, &n_px};
    for (int k = 0; k < 13; ++k) out[k] = all[k];
  }
  size_t bytes() const {
    const gs::DeviceBuffer *all[] = {&mask, &counters, &rank, &xyz_c_all, &uv_all, &c2g, &xyz_c, &uv, &sigma, &conic, &J,
                                     &rgb, &radius, &recs, &counts, &offsets, &grad_rows, &hitmask, &keys_a, &keys_b, &pay_a, &pay_b,
                                     &sorted, &temp, &ranges, &image, &T_px, &n_px, &blockmasks, &kept, &tile_tops, &tile_order,
                                     &seg_first, &seg_extra, &seg_chk, &fseg_first, &fseg_blocks, &fseg_gran, &fseg_part, &fseg_stop, &dir_grad};
    size_t b = 0;
    for (auto *p : all) b += p->bytes;
    return b;
  }
  void release() {
    gs::pool_unwatch(&last_mask);
    last_mask = nullptr;
    gs::DeviceBuffer *all[] = {&mask, &counters, &rank, &xyz_c_all, &uv_all, &c2g, &xyz_c, &uv, &sigma, &conic, &J,
                               &rgb, &radius, &recs, &counts, &offsets, &grad_rows, &hitmask, &keys_a, &keys_b, &pay_a, &pay_b,
                               &sorted, &temp, &bin_table, &ranges, &image, &T_px, &n_px, &blockmasks, &kept, &tile_tops, &tile_order,
                               &seg_first, &seg_extra, &seg_chk, &fseg_first, &fseg_blocks, &fseg_gran, &fseg_part, &fseg_stop, &dir_grad};
    for (auto *p : all) p->release();
    fseg_gran_zeroed = nullptr;
    fork.destroy();
    chunk_first.release();
    if (pre_side) { (void)hipStreamDestroy(pre_side); pre_side = nullptr; }
    if (ev_pre_fork) { (void)hipEventDestroy(ev_pre_fork); ev_pre_fork = nullptr; }
    if (ev_pre_join) { (void)hipEventDestroy(ev_pre_join); ev_pre_join = nullptr; }
    if (h_words) (void)hipHostFree(h_words);
    h_words = nullptr;
    if (h_pub) { (void)hipHostFree((void *)h_pub); h_pub = nullptr; d_pub = nullptr; }
    for (int a = 0; a < kSlots; ++a)
      for (int b = 0; b < 2 * kStages; ++b)
        if (ev[a][b]) { (void)hipEventDestroy(ev[a][b]); ev[a][b] = nullptr; }
  }
};

namespace {

constexpr int kBlock = 256;

// GSPLAT_NO_TILE_ORDER=1: the backward takes its tiles in the plain XCD-run order (A/B of r04's heaviest-first order)
bool gs_no_tile_order() {
  static const bool v = [] { const char *e = getenv("GSPLAT_NO_TILE_ORDER"); return e && e[0] == '1'; }();
  return v;
}

// GSPLAT_NO_SEGMENTS=1: long lists stay whole in the backward (A/B of r05's segment split)
bool gs_no_segments() {
  static const bool v = [] { const char *e = getenv("GSPLAT_NO_SEGMENTS"); return e && e[0] == '1'; }();
  return v;
}

// GSPLAT_FWD_SEGMENTS_GATE=<x>: the forward splits its long lists when the longest chain exceeds x times the work per
// resident workgroup (default 3; 0: always)
double gs_fwd_segments_gate() {
  static const double v = [] { const char *e = getenv("GSPLAT_FWD_SEGMENTS_GATE"); return e && e[0] ? atof(e) : 3.0; }();
  return v;
}
// GSPLAT_NO_FWD_SEGMENTS=1: one block per tile in the forward, whatever the list lengths (A/B)
bool gs_no_fwd_segments() {
  static const bool v = [] { const char *e = getenv("GSPLAT_NO_FWD_SEGMENTS"); return e && e[0] == '1'; }();
  return v;
}

// ---- A: world -> camera -> pixel -> keep-mask, for all N, and the compaction ranks
// kBinBlocks workgroups of kBinThreads; workgroup b owns the global indices of the chunks [C*b/kBinBlocks,
// C*(b+1)/kBinBlocks) (gs_common.h: bin_slice_first_chunk).  rank[i] leaves this kernel as the exclusive count of kept
// gaussians INSIDE the slice and slice_counts[b] as the slice's total; preprocess_kernel adds the counts of the slices
// before the gaussian's own and stores the global rank.  That replaces a two-kernel rocPRIM scan over N + 1 flags (12 us of
// launch + drain on the forward's critical path and 12 MB of traffic) by two barriers in a kernel that is here anyway.
// Pass 1 has no barrier, so the loads of all trips overlap; it leaves one ballot per (trip, wave) in LDS, wave 0 scans
// their counts, pass 2 only stores.
__global__ __launch_bounds__(gs::kBinThreads) void project_cull_kernel(const float *__restrict__ xyz,
                                                                       const float *__restrict__ view,
                                                                       const float *__restrict__ proj, int N, int width,
                                                                       int height, float near_thresh, int padding,
                                                                       float *__restrict__ xyz_c, float *__restrict__ uv,
                                                                       unsigned char *__restrict__ mask,
                                                                       int *__restrict__ rank,
                                                                       int *__restrict__ slice_counts,
                                                                       unsigned long long *__restrict__ pair_counters,
                                                                       int *__restrict__ kept,
                                                                       int *__restrict__ chunk_first) {
  extern __shared__ unsigned long long s_ballot[];  // [trips * 16] ballots, then [trips * 16] exclusive counts (int)
  constexpr int kWaves = gs::kBinThreads / 64;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int C = gs::bin_chunks(N);
  const int lo = gs::kBinChunk * gs::bin_slice_first_chunk(C, blockIdx.x);
  const int hi = min(N, gs::kBinChunk * gs::bin_slice_first_chunk(C, blockIdx.x + 1));
  const int trips = (hi - lo + gs::kBinThreads - 1) / gs::kBinThreads;
  int *s_before = reinterpret_cast<int *>(s_ballot + trips * kWaves);
  if (blockIdx.x == 0 && threadIdx.x < 64) pair_counters[threadIdx.x] = 0ull;  // consumed by preprocess_kernel
  const gs::Mat34 v = gs::load_view(view);
  const gs::Mat44 p = gs::load_proj(proj);
  for (int t = 0; t < trips; ++t) {
    const int i = lo + t * gs::kBinThreads + (int)threadIdx.x;
    bool k = false;
    if (i < hi) {
      float x, y, z, u, q;
      gs::camera_space(v, xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], x, y, z);
      gs::to_screen(p, x, y, z, width, height, u, q);
      k = gs::keep(u, q, z, near_thresh, padding, width, height);
      if (xyz_c) {  // ForwardPassData's uncompacted d_xyz_c / d_uv; a lean context does not materialise them
        xyz_c[3 * i] = x; xyz_c[3 * i + 1] = y; xyz_c[3 * i + 2] = z;
        uv[2 * i] = u; uv[2 * i + 1] = q;
      }
      mask[i] = k ? 1 : 0;
    }
    const unsigned long long bal = __ballot(k);
    if (lane == 0) s_ballot[t * kWaves + w] = bal;
  }
  __syncthreads();
  if (w == 0) {  // exclusive scan of the trips * 16 ballot counts, slice order = (trip, wave)
    const int n = trips * kWaves, per = (n + 63) / 64;
    int sum = 0;
    for (int e = lane * per; e < min(lane * per + per, n); ++e) sum += __popcll(s_ballot[e]);
    int incl = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int u = __shfl_up(incl, off, 64);
      if (lane >= off) incl += u;
    }
    int run = incl - sum;
    for (int e = lane * per; e < min(lane * per + per, n); ++e) { s_before[e] = run; run += __popcll(s_ballot[e]); }
    if (lane == 63) slice_counts[blockIdx.x] = incl;
  }
  __syncthreads();
  for (int t = 0; t < trips; ++t) {
    const int i = lo + t * gs::kBinThreads + (int)threadIdx.x;
    if (i < hi) {
      const unsigned long long bal = s_ballot[t * kWaves + w];
      const int r = s_before[t * kWaves + w] + __popcll(bal & ((1ull << lane) - 1ull));
      rank[i] = r;
      // r06: the slice-local rank of every 64-entry chunk's first index (a slice starts on a chunk boundary and a trip is
      // sixteen chunks), for sh_colour_kernel when it runs BESIDE preprocess_geom_kernel, which rewrites rank[] in place
      if (chunk_first && lane == 0) chunk_first[(lo >> 6) + t * kWaves + w] = s_before[t * kWaves + w];
      // the slice's kept gaussians as a list (preprocess_kernel's compacted walk; null when the next kernel walks all)
      if (kept && ((bal >> lane) & 1ull)) kept[lo + r] = i;
    }
  }
}

// Rows of K floats of a wave's active lanes -- compacted positions p = 0 .. nact-1, consecutive global rows j0 + p -- staged in
// the wave's LDS area and written as ONE linear span of 16-byte stores, instead of K strided 4-byte stores per lane (r04).
// Measured on the forward that stores every ForwardPassData array (Sigma, J, conic, colour: 24 strided dword stores per
// gaussian more than the lean one): preprocess 0.132 -> 0.104 ms.  The lean kernel's own strided stores (the 48-byte record,
// the camera-space position) gain nothing from the same treatment (0.0885 -> 0.0883 .. 0.0909) and keep their direct form.
template <int K>
__device__ __forceinline__ void wave_rows_store(float *__restrict__ dst, int j0, int p, int nact, const float *v, float *wsh) {
#pragma unroll
  for (int k = 0; k < K; ++k) wsh[p * K + k] = v[k];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  const int total = nact * K;
  float *out = dst + (size_t)j0 * K;
  for (int e = p * 4; e + 3 < total; e += nact * 4)
    *reinterpret_cast<gs::f4u *>(out + e) = __builtin_bit_cast(gs::f4u, *reinterpret_cast<const float4 *>(wsh + e));
  for (int e = (total & ~3) + p; e < total; e += nact) out[e] = wsh[e];
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");  // (the area is reused by the next array)
  __builtin_amdgcn_wave_barrier();
}

constexpr int kCoopTiles = 64;  // candidate tiles above which a splat's tile tests are shared by its wave

struct PreOut {
  int *c2g;
  float *xyz_c, *uv, *sigma, *conic, *J, *rgb, *radius;
  float4 *recs;
  int *counts;
  unsigned long long *hitmask;  // per gaussian: one bit per tile of its coarse rectangle (rectangles of <= 64 tiles)
  unsigned long long *pairs;  // 64 spread counters of coarse candidate pairs (reporting only)
};

// ---- B: everything per kept gaussian, written at its compacted slot
// Launched as kBinBlocks workgroups of kBinThreads; workgroup b owns every kBinBlocks-th chunk of 64 entries
// (gs_common.h: kBinChunk).  When `table` is given it also histograms the tiles its gaussians hit in LDS -- the count
// phase of the counting-sort binning, for free next to the separating-axis tests -- and writes the row table[b][0..T).
//   kStoreMid: Sigma, J, conic and the SH colour are stored (ForwardPassData, cuda_data.cuh:70-86).  The fused backward
//     recomputes the first three and never reads the colour, so a training context that does not hand them to its
//     caller (gsplat_context_set_lean_forward) skips 72 of the 176 bytes this kernel writes per gaussian.
//   kCompact: the chunks are chunks of the M compacted slots, resolved through the slice-local lists project_cull_kernel
//     left in `kept` (kept[first index of slice s + k] = global index of the slice's k-th kept gaussian), instead of
//     chunks of all indices with the culled lanes idle: a view that culls every other gaussian runs half the trips
//     (r02: 23 % of HBM on such a view).
// Everything a gaussian stores leaves BEFORE its tile loop, which then holds six numbers per lane: kept across the loop,
// Sigma, J, conic, colour and the 48-byte record pushed the SH-3 instance past its 128 registers (6 spilled in r02).
template <int L, bool kStoreMid, bool kCompact>
__global__ __launch_bounds__(gs::kBinThreads) void preprocess_kernel(gsplat_gaussians g, const float *__restrict__ view,
                                                            const unsigned char *__restrict__ mask,
                                                            int *__restrict__ rank,
                                                            const int *__restrict__ slice_counts,
                                                            const int *__restrict__ kept,
                                                            const float *__restrict__ proj, int width, int height,
                                                            float fx, float fy,
                                                            float tan_fovx, float tan_fovy, float mh_dist, float cx,
                                                            float cy, float cz, int ntx, int nty, PreOut o,
                                                            int *__restrict__ table) {
  extern __shared__ int s_hist[];
  __shared__ int s_base[gs::kBinBlocks + 1];  // kept gaussians before each of the cull's slices; [kBinBlocks]: all of them
  __shared__ int s_wsum[4];
  const int N = g.num_gaussians, T = ntx * nty;
  static_assert(gs::kBinBlocks == 256, "the scan below is written for four waves of slice counts");
  {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int cnt = 0, incl = 0;
    if (threadIdx.x < gs::kBinBlocks) {
      cnt = slice_counts[threadIdx.x];
      incl = cnt;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const int u = __shfl_up(incl, off, 64);
        if (lane >= off) incl += u;
      }
      if (lane == 63) s_wsum[w] = incl;
    }
    if (table)
      for (int t = threadIdx.x; t < T; t += gs::kBinThreads) s_hist[t] = 0;
    __syncthreads();
    if (threadIdx.x < gs::kBinBlocks) {
      int before = 0;
      for (int q = 0; q < w; ++q) before += s_wsum[q];
      s_base[threadIdx.x] = before + incl - cnt;
      if (threadIdx.x == gs::kBinBlocks - 1) s_base[gs::kBinBlocks] = before + incl;
    }
    __syncthreads();
  }
  const int M = s_base[gs::kBinBlocks];
  const int C = gs::bin_chunks(N);  // chunks of the index space: the cull's slices are runs of them
  unsigned long long coarse = 0;
  const int lane = threadIdx.x & 63;
  // what a walk over all indices in order did on the side: counts[M..N] must read 0 in the scan that may follow, and
  // rank[N] holds the total
  for (long long i = (long long)M + (long long)blockIdx.x * gs::kBinThreads + threadIdx.x; i <= N;
       i += (long long)gs::kBinBlocks * gs::kBinThreads)
    o.counts[i] = 0;
  if (blockIdx.x == 0 && threadIdx.x == 0) rank[N] = M;
  // This workgroup's chunks, one wave trip each (gs_common.h): chunk c of the walked space -- all N indices, or the M
  // compacted slots (kCompact) -- belongs to workgroup c % kBinBlocks and to its wave (c / kBinBlocks) % 16.  The trip
  // count is wave-uniform: the tiles of LARGE splats are tested by the whole wave together (below).
  const int walk_chunks = gs::bin_chunks(kCompact ? M : N);
  for (int c = (int)blockIdx.x + gs::kBinBlocks * (int)(threadIdx.x >> 6); c < walk_chunks;
       c += gs::kBinBlocks * (gs::kBinThreads / 64)) {
  const int e = c * gs::kBinChunk + lane;  // kCompact: the compacted slot; else the global index
  int i = e, j = e;
  bool act;
  if constexpr (kCompact) {
    act = e < M;
    // the slice of the cull that holds compacted slot e: the last s with s_base[s] <= e (binary search in LDS)
    int sl = 0;
#pragma unroll
    for (int step = gs::kBinBlocks / 2; step > 0; step >>= 1) sl += (s_base[sl + step] <= e) ? step : 0;
    i = act ? kept[gs::kBinChunk * gs::bin_slice_first_chunk(C, sl) + (e - s_base[sl])] : 0;
    if (act) rank[i] = j;
  } else {
    // local -> global rank, for every index (the pack kernels read it through the mask, bin_scatter_kernel at the
    // chunk starts): a chunk lies inside one slice of the cull
    const int sl = gs::bin_slice_of_chunk(C, c);
    if (i < N) {
      j = s_base[sl] + rank[i];
      rank[i] = j;
    }
    act = i < N && mask[i];
  }
  int hits = 0, span_n = 0;
  unsigned long long hm = 0ull;
  float bu = 0.0f, bv = 0.0f, br0 = 0.0f, br1 = 0.0f, br2 = 0.0f, br3 = 0.0f;  // what the tile tests need of a lane
  if (act) {
    constexpr int n = (L + 1) * (L + 1);
    const gs::Mat34 vw = gs::load_view(view);
    // colour
    float dx, dy, dz, len, rgb[3];
    gs::view_dir(g.xyz[3 * i], g.xyz[3 * i + 1], g.xyz[3 * i + 2], cx, cy, cz, dx, dy, dz, len);
    gs::sh_to_rgb<L>(g.sh + (size_t)i * (n - 1) * 3, g.rgb + 3 * i, dx, dy, dz, rgb);
    // covariance -> conic
    const float4 q = reinterpret_cast<const float4 *>(g.quaternion)[i];
    const gs::RotScale rs = gs::rot_scale(q.x, q.y, q.z, q.w, g.scale[3 * i], g.scale[3 * i + 1], g.scale[3 * i + 2]);
    float sg[6], J[6], con[3], rad[4];
    gs::sigma_from(rs, sg);
    // kStoreMid: the stores go through the wave's staging area (wave_rows_store).  All active lanes of the wave are here
    // together: their compacted positions, and the row of the first one (the rows of a chunk's active lanes are consecutive)
    __shared__ __attribute__((aligned(16))) float s_stage[kStoreMid ? gs::kBinThreads / 64 : 1][kStoreMid ? 64 * 12 : 4];
    int pp = 0, nact = 0, j0 = 0;
    float *wsh = s_stage[0];
    if constexpr (kStoreMid) {
      const unsigned long long actm = __ballot(true);
      nact = __popcll(actm);
      pp = __popcll(actm & ((1ull << lane) - 1ull));
      j0 = __builtin_amdgcn_readfirstlane(j - pp);
      wsh = s_stage[threadIdx.x >> 6];
      // each of the four as soon as it exists: all of them pending at once cost spilled registers
      wave_rows_store<6>(o.sigma, j0, pp, nact, sg, wsh);
      wave_rows_store<3>(o.rgb, j0, pp, nact, rgb, wsh);
    }
    // camera-space position and pixel coordinates: recomputed with project_cull_kernel's functions on the same inputs
    // (bit for bit its values) from a second, cache-resident read of the position, instead of a 20-byte round trip
    // through the uncompacted arrays
    float x, y, z, u, v;
    {
      int i2 = i;
      asm volatile("" : "+v"(i2));  // an opaque copy: the first read's registers are not kept alive across the SH sum
      gs::camera_space(vw, g.xyz[3 * i2], g.xyz[3 * i2 + 1], g.xyz[3 * i2 + 2], x, y, z);
      gs::to_screen(gs::load_proj(proj), x, y, z, width, height, u, v);
    }
    gs::jacobian(x, y, z, fx, fy, tan_fovx, tan_fovy, J);
    gs::conic_radius(J, sg, vw, mh_dist, con, rad);
    // stores (compacted order); counts and hitmask follow the tile tests
    o.c2g[j] = i;
    o.uv[2 * j] = u; o.uv[2 * j + 1] = v;
    reinterpret_cast<float4 *>(o.radius)[j] = make_float4(rad[0], rad[1], rad[2], rad[3]);
    const gs::SplatRec rec = gs::make_record(u, v, con[0], con[1], con[2], g.opacity[i], rgb[0], rgb[1], rgb[2]);
    if constexpr (kStoreMid) {
      const float xyzc[3] = {x, y, z};
      wave_rows_store<3>(o.xyz_c, j0, pp, nact, xyzc, wsh);
      wave_rows_store<6>(o.J, j0, pp, nact, J, wsh);
      wave_rows_store<3>(o.conic, j0, pp, nact, con, wsh);
      const float rr[12] = {rec.r0.x, rec.r0.y, rec.r0.z, rec.r0.w, rec.r1.x, rec.r1.y, rec.r1.z, rec.r1.w,
                            rec.r2.x, rec.r2.y, rec.r2.z, rec.r2.w};
      wave_rows_store<12>(reinterpret_cast<float *>(o.recs), j0, pp, nact, rr, wsh);
    } else {
      o.xyz_c[3 * j] = x; o.xyz_c[3 * j + 1] = y; o.xyz_c[3 * j + 2] = z;
      o.recs[3 * j] = rec.r0; o.recs[3 * j + 1] = rec.r1; o.recs[3 * j + 2] = rec.r2;
    }
    bu = u; bv = v; br0 = rad[0]; br1 = rad[1]; br2 = rad[2]; br3 = rad[3];
  }
  if (act) {
    // exact tile count
    const gs::TileRect r = gs::coarse_rect(bu, bv, br0, ntx, nty);
    if (r.x1 > r.x0 && r.y1 > r.y0) {
      coarse += (unsigned long long)(r.x1 - r.x0) * (unsigned long long)(r.y1 - r.y0);
      const gs::Obb ob = gs::make_obb(bu, bv, br0, br1, br2, br3);
      const gs::TileRect sp = gs::obb_span(ob, r);
      const int rh = r.y1 - r.y0;
      span_n = max(0, sp.x1 - sp.x0) * max(0, sp.y1 - sp.y0);
      if (span_n <= kCoopTiles) {
        for (int tx = sp.x0; tx < sp.x1; ++tx)
          for (int ty = sp.y0; ty < sp.y1; ++ty) {
            const bool h = gs::obb_hits_tile(ob, tx, ty);
            const int bit = (tx - r.x0) * rh + (ty - r.y0);  // position in the full coarse rectangle
            hits += h ? 1 : 0;
            hm |= (h && bit < 64) ? (1ull << bit) : 0ull;  // read by the binning kernels when the rectangle has <= 64 tiles
            if (h && table) atomicAdd(&s_hist[ty * ntx + tx], 1);
          }
      }
    }
  }
  // Large splats (more than kCoopTiles candidate tiles after the axis-aligned clipping: a capture early in training has
  // gaussians over hundreds or thousands of tiles): one lane walking them alone decided the kernel's duration.  The
  // wave takes them one at a time: the owner's six numbers are broadcast, every lane rebuilds the same OBB (the same
  // functions on the same inputs: bit-identical decisions) and tests every 64th tile of the span.  Such splats have
  // rectangles of more than 64 tiles, so they carry no hit mask.
  for (unsigned long long big = __ballot(span_n > kCoopTiles); big != 0ull; big &= big - 1ull) {
    const int owner = __builtin_ctzll(big);
    const float ou = __shfl(bu, owner, 64), ov = __shfl(bv, owner, 64);
    const float o0 = __shfl(br0, owner, 64), o1 = __shfl(br1, owner, 64), o2 = __shfl(br2, owner, 64), o3 = __shfl(br3, owner, 64);
    const gs::TileRect r = gs::coarse_rect(ou, ov, o0, ntx, nty);
    const gs::Obb ob = gs::make_obb(ou, ov, o0, o1, o2, o3);
    const gs::TileRect sp = gs::obb_span(ob, r);
    const int sh = sp.y1 - sp.y0, total = (sp.x1 - sp.x0) * sh;
    int found = 0;
    for (int p = lane; p < total + lane; p += 64) {  // uniform trip count: every lane reaches the ballot
      bool h = false;
      if (p < total) {
        const int tx = sp.x0 + p / sh, ty = sp.y0 + p % sh;
        h = gs::obb_hits_tile(ob, tx, ty);
        if (h && table) atomicAdd(&s_hist[ty * ntx + tx], 1);
      }
      found += __popcll(__ballot(h));
    }
    if (lane == owner) hits = found;
  }
  if (act) {
    o.counts[j] = hits;
    o.hitmask[j] = hm;
  }
  }
  // candidate-pair count (what call 1 of get_sorted_gaussian_list reports): one atomic per wave, spread over 64
  // counters so that no single address serialises the chip
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) coarse += __shfl_down(coarse, off, 64);
  if ((threadIdx.x & 63) == 0 && coarse) atomicAdd(&o.pairs[(blockIdx.x * 16 + (threadIdx.x >> 6)) & 63], coarse);
  if (table) {
    __syncthreads();
    for (int t = threadIdx.x; t < T; t += gs::kBinThreads) table[(size_t)blockIdx.x * T + t] = s_hist[t];
  }
}

// ---- r06: the per-gaussian forward as TWO kernels (gsplat_context_set_preprocess_split / GSPLAT_PRE_SPLIT) ------------
// VERDICT r05 asked for the split that r04 had only priced: preprocess_kernel<3> needs 99-128 VGPRs (four waves per SIMD in
// the 1024-thread shape of the LDS histogram) and reaches 36 % of HBM on its algorithmic bytes.  Built here, bit-identical
// (tests/test_fused_gpu.py: test_preprocess_split_is_bit_identical), switchable -- and NOT the default, because it loses
// (profiles/r06_ab_preprocess_split.txt, same box, alternating):
//   sh_colour_kernel        SH -> colour as a plain stream: every wave moves its 64 rows (180 B each at degree 3) through
//                           LDS as one linear span of 16-byte loads (gs_rows.h), writes 12 bytes per gaussian.
//                           47 us for 228 MB = 4.8 TB/s.
//   preprocess_geom_kernel  everything else in the persistent 256 x 1024 shape; a gaussian's inputs are 14 dwords, so the
//                           chunk loop requests the next chunk's inputs a whole trip early, in front of the tile loop,
//                           and defers the tile count / hit mask stores to the next trip (one wait per trip, on loads
//                           and stores that are a tile loop old).  64 us lean / 69 us full.
//   mode 1 (one behind the other) 112 / 121 us against the fused kernel's 86 / 103; mode 2 (side by side on two streams,
//   the colour written straight into the records) 124 / 140: the geometry kernel slows down beside a stream of loads.
// Where the geometry kernel's 64 us go (diagnostic builds, profiles/r06_ab_preprocess_split.txt): without its tile loop
// 42 us, without tile loop and conic / radius arithmetic 40 us -- the floor is 150 MB, two thirds of it WRITTEN, at
// 3.7 TB/s, and the order of loads and stores inside a trip does not move it (three orders measured).  The fused kernel
// moves 350 MB (lean) in 86 us = 4.1 TB/s and 420 MB (every array stored) in 103 us: it already runs at the rate this
// mix of strided 4..16-byte stores reaches, with the SH loads riding in the shadow of the arithmetic; the figure the
// verdict compares it with, preprocess_bwd's 72 %, is a kernel whose traffic is linear spans read and written in equal
// parts.  What is left in the fused kernel is its tile loop (9 us) and bytes (the tile count array, 4 B, is dead weight
// on the sparse route).  cuda/raster.cu:78-100 runs the colour and the covariance chain as separate kernels too.
// `rank`: the cull's slice-local ranks (sequential form, in front of preprocess_geom_kernel) or, when the kernel runs
// beside that one -- which rewrites rank[] in place --, null: the rank then comes from `chunk_first` and the chunk's mask
// ballot.  `recs`: the colour goes into the 48-byte records (r2.xyz; the geometry kernel writes the rest of the record,
// r2.w included: disjoint bytes), `rgb_out`: and / or into ForwardPassData's colour array.
template <int L, bool kCompact>
__global__ __launch_bounds__(kBlock) void sh_colour_kernel(gsplat_gaussians g, const unsigned char *__restrict__ mask,
                                                           const int *__restrict__ rank,
                                                           const int *__restrict__ chunk_first,
                                                           const int *__restrict__ slice_counts,
                                                           const int *__restrict__ kept, float cx, float cy, float cz,
                                                           float *__restrict__ rgb_out, float4 *__restrict__ recs) {
  constexpr int n = (L + 1) * (L + 1), kRest = (n - 1) * 3;
  static_assert(kRest > 0, "degree 0 has no rest coefficients: preprocess_geom_kernel forms the colour itself");
  static_assert(gs::kBinBlocks == kBlock, "one thread per slice of the cull in the scan below");
  __shared__ __attribute__((aligned(16))) float s_sh[kBlock * kRest];
  __shared__ int s_base[gs::kBinBlocks + 1];
  __shared__ int s_wsum[4];
  const int N = g.num_gaussians;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  {  // kept gaussians before each of the cull's slices (as preprocess_kernel)
    const int cnt = slice_counts[threadIdx.x];
    int incl = cnt;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int u = __shfl_up(incl, off, 64);
      if (lane >= off) incl += u;
    }
    if (lane == 63) s_wsum[w] = incl;
    __syncthreads();
    int before = 0;
    for (int q = 0; q < w; ++q) before += s_wsum[q];
    s_base[threadIdx.x] = before + incl - cnt;
    if (threadIdx.x == gs::kBinBlocks - 1) s_base[gs::kBinBlocks] = before + incl;
    __syncthreads();
  }
  const int M = s_base[gs::kBinBlocks];
  const int C = gs::bin_chunks(N);
  const int c = (int)blockIdx.x * (kBlock / 64) + w;  // this wave's chunk of the walked space
  if (c >= gs::bin_chunks(kCompact ? M : N)) return;
  const int e = c * gs::kBinChunk + lane;
  int i = e, j = e;
  bool act;
  if constexpr (kCompact) {
    act = e < M;
    int sl = 0;
#pragma unroll
    for (int step = gs::kBinBlocks / 2; step > 0; step >>= 1) sl += (s_base[sl + step] <= e) ? step : 0;
    i = act ? kept[gs::kBinChunk * gs::bin_slice_first_chunk(C, sl) + (e - s_base[sl])] : 0;
  } else {
    act = i < N && mask[i];
  }
  const unsigned long long actm = __ballot(act);
  if (actm == 0ull) return;
  if constexpr (!kCompact) {
    if (rank) {
      if (i < N) j = s_base[gs::bin_slice_of_chunk(C, c)] + rank[i];  // rank[]: still the cull's slice-local counts
    } else {
      j = s_base[gs::bin_slice_of_chunk(C, c)] + chunk_first[c] + __popcll(actm & ((1ull << lane) - 1ull));
    }
  }
  float *wsh = s_sh + (threadIdx.x - lane) * kRest;
  // position and band 0 of the lane's own gaussian: requested before the rows, used behind them
  float px = 0.0f, py = 0.0f, pz = 0.0f, b0[3] = {0.0f, 0.0f, 0.0f};
  if (act) {
    px = g.xyz[3 * i]; py = g.xyz[3 * i + 1]; pz = g.xyz[3 * i + 2];
    b0[0] = g.rgb[3 * i]; b0[1] = g.rgb[3 * i + 1]; b0[2] = g.rgb[3 * i + 2];
  }
  if (!kCompact && __popcll(actm) * 2 >= min(64, N - c * gs::kBinChunk)) {
    // a chunk of consecutive indices of which most are kept: one linear span, culled rows included
    gs::rows_to_lds<kRest>(g.sh + (size_t)c * gs::kBinChunk * kRest, wsh, min(64, N - c * gs::kBinChunk), lane);
  } else {
    // scattered rows (the compacted walk, or a chunk most of which is culled): twelve lanes fetch one row in 16-byte
    // pieces, all of a wave's loads issued before the first LDS store (as preprocess_bwd_kernel)
    constexpr int kPieces = (kRest + 3) / 4, kPer = 64 / kPieces, kIter = (64 + kPer - 1) / kPer;
    const int grp = lane / kPieces, piece = lane - grp * kPieces;
    const bool in_grp = grp < kPer;
    float4 v[kIter];
#pragma unroll
    for (int it = 0; it < kIter; ++it) {
      const int r = it * kPer + grp;
      const int ir = __shfl(i, r < 64 ? r : 0, 64);
      const bool want = in_grp && r < 64 && ((actm >> r) & 1ull);
      v[it] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      if (want) {
        const float *src = g.sh + (size_t)ir * kRest + 4 * piece;
        if (4 * piece + 3 < kRest) {
          v[it] = __builtin_bit_cast(float4, *reinterpret_cast<const gs::f4u *>(src));
        } else {
          v[it].x = src[0];
          if (4 * piece + 1 < kRest) v[it].y = src[1];
          if (4 * piece + 2 < kRest) v[it].z = src[2];
        }
      }
    }
#pragma unroll
    for (int it = 0; it < kIter; ++it) {
      const int r = it * kPer + grp;
      if (in_grp && r < 64 && ((actm >> r) & 1ull)) {
        float *dst = wsh + r * kRest + 4 * piece;
        dst[0] = v[it].x;
        if (4 * piece + 1 < kRest) dst[1] = v[it].y;
        if (4 * piece + 2 < kRest) dst[2] = v[it].z;
        if (4 * piece + 3 < kRest) dst[3] = v[it].w;
      }
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  if (act) {
    float dx, dy, dz, len, rgb[3];
    gs::view_dir(px, py, pz, cx, cy, cz, dx, dy, dz, len);
    gs::sh_to_rgb<L>(wsh + lane * kRest, b0, dx, dy, dz, rgb);
    if (rgb_out) { rgb_out[3 * j] = rgb[0]; rgb_out[3 * j + 1] = rgb[1]; rgb_out[3 * j + 2] = rgb[2]; }
    if (recs) {
      float *r2 = reinterpret_cast<float *>(recs + 3 * (size_t)j + 2);
      r2[0] = rgb[0]; r2[1] = rgb[1]; r2[2] = rgb[2];
    }
  }
}

// what a lane of preprocess_geom_kernel holds for a chunk it has not started yet
struct GeomData { float x, y, z, sx, sy, sz, op; float4 q; };

// kColour 0: degree 0 -- the colour is band 0 alone, formed here (no sh_colour_kernel); 1: read from o.rgb, where
// sh_colour_kernel has left it (the two kernels one behind the other); 2: none -- sh_colour_kernel runs BESIDE this kernel
// on a stream of its own and writes the colour into the records itself; this kernel stores r0, r1 and r2.w only
template <int kColour, bool kStoreMid, bool kCompact>
__global__ __launch_bounds__(gs::kBinThreads) void preprocess_geom_kernel(gsplat_gaussians g, const float *__restrict__ view,
                                                                  const unsigned char *__restrict__ mask,
                                                                  int *__restrict__ rank,
                                                                  const int *__restrict__ slice_counts,
                                                                  const int *__restrict__ kept,
                                                                  const float *__restrict__ proj, int width, int height,
                                                                  float fx, float fy, float tan_fovx, float tan_fovy,
                                                                  float mh_dist, float cx, float cy, float cz, int ntx,
                                                                  int nty, PreOut o, int *__restrict__ table) {
  extern __shared__ int s_hist[];
  __shared__ int s_base[gs::kBinBlocks + 1];
  __shared__ int s_wsum[4];
  __shared__ __attribute__((aligned(16))) float s_stage[kStoreMid ? gs::kBinThreads / 64 : 1][kStoreMid ? 64 * 12 : 4];
  const int N = g.num_gaussians, T = ntx * nty;
  static_assert(gs::kBinBlocks == 256, "the scan below is written for four waves of slice counts");
  {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int cnt = 0, incl = 0;
    if (threadIdx.x < gs::kBinBlocks) {
      cnt = slice_counts[threadIdx.x];
      incl = cnt;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const int u = __shfl_up(incl, off, 64);
        if (lane >= off) incl += u;
      }
      if (lane == 63) s_wsum[w] = incl;
    }
    if (table)
      for (int t = threadIdx.x; t < T; t += gs::kBinThreads) s_hist[t] = 0;
    __syncthreads();
    if (threadIdx.x < gs::kBinBlocks) {
      int before = 0;
      for (int q = 0; q < w; ++q) before += s_wsum[q];
      s_base[threadIdx.x] = before + incl - cnt;
      if (threadIdx.x == gs::kBinBlocks - 1) s_base[gs::kBinBlocks] = before + incl;
    }
    __syncthreads();
  }
  const int M = s_base[gs::kBinBlocks];
  const int C = gs::bin_chunks(N);
  unsigned long long coarse = 0;
  const int lane = threadIdx.x & 63;
  for (long long i = (long long)M + (long long)blockIdx.x * gs::kBinThreads + threadIdx.x; i <= N;
       i += (long long)gs::kBinBlocks * gs::kBinThreads)
    o.counts[i] = 0;
  if (blockIdx.x == 0 && threadIdx.x == 0) rank[N] = M;
  const int walk_chunks = gs::bin_chunks(kCompact ? M : N);
  constexpr int kStep = gs::kBinBlocks * (gs::kBinThreads / 64);
  const int c_first = (int)blockIdx.x + gs::kBinBlocks * (int)(threadIdx.x >> 6);
  // The pipeline.  s_waitcnt vmcnt counts a wave's loads AND stores in issue order on gfx950: waiting for a load that was
  // issued behind stores waits for those stores' acknowledgements too, and a value's first use is where the compiler puts
  // the wait.  So everything a chunk needs from memory is REQUESTED one trip early, at the top of the trip, in front of
  // that trip's thirteen stores, and left untouched until the chunk's own trip begins: by then the loads are a whole
  // trip old and the wait lets the younger stores stay in flight (r05's kernel waited three times per trip with all of
  // its earlier stores still queued in front of the loads).
  auto load_data = [&](int i) {
    GeomData d;
    d.x = g.xyz[3 * i]; d.y = g.xyz[3 * i + 1]; d.z = g.xyz[3 * i + 2];
    d.q = reinterpret_cast<const float4 *>(g.quaternion)[i];
    d.sx = g.scale[3 * i]; d.sy = g.scale[3 * i + 1]; d.sz = g.scale[3 * i + 2];
    d.op = g.opacity[i];
    return d;
  };
  const GeomData kNoData = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, make_float4(0.0f, 0.0f, 0.0f, 0.0f)};
  // all indices in order: a chunk's data are at its own indices (requested whether kept or not: this walk runs when at
  // most a fifth is culled), rank[] and mask[] are requested with them.  compacted walk: the chunk's global indices come
  // from `kept`, requested TWO trips early (the data loads of the trip in between take them as addresses).
  int rk_n = 0, mk_n = 0, ix_n = 0, ix_nn = 0;  // raw: rank / mask of the next chunk | kept index of the next, next-next
  GeomData d_n = kNoData;
  auto kept_index = [&](int c) {  // global index of this lane's slot in chunk c of the compacted slots (0 past the end)
    const int e = c * gs::kBinChunk + lane;
    if (c >= walk_chunks || e >= M) return 0;
    int sl = 0;
#pragma unroll
    for (int step = gs::kBinBlocks / 2; step > 0; step >>= 1) sl += (s_base[sl + step] <= e) ? step : 0;
    return kept[gs::kBinChunk * gs::bin_slice_first_chunk(C, sl) + (e - s_base[sl])];
  };
  auto request = [&](int c) {  // the loads of chunk c (the chunk after the one being worked on)
    rk_n = 0; mk_n = 0; d_n = kNoData;
    const int e = c * gs::kBinChunk + lane;
    if constexpr (kCompact) {
      if (c < walk_chunks && e < M) d_n = load_data(ix_n);
    } else {
      if (c < walk_chunks && e < N) { rk_n = rank[e]; mk_n = mask[e]; d_n = load_data(e); }
    }
  };
  if constexpr (kCompact) { ix_n = kept_index(c_first); }
  request(c_first);
  if constexpr (kCompact) { ix_nn = kept_index(c_first + kStep); }
  bool late_act = false;  // the previous chunk's tile count and hit mask, stored with this trip's stores
  int late_j = 0, late_hits = 0;
  unsigned long long late_hm = 0ull;
  for (int c = c_first; c < walk_chunks; c += kStep) {
  // this chunk's values (requested a trip ago, in front of the previous tile loop)
  const int e0 = c * gs::kBinChunk + lane;
  const int rk = rk_n, mk = mk_n, i = kCompact ? ix_n : e0;
  const GeomData cur = d_n;
  bool act;
  int j = e0;
  if constexpr (kCompact) {
    act = e0 < M;
  } else {
    act = e0 < N && mk != 0;
    if (e0 < N) j = s_base[gs::bin_slice_of_chunk(C, c)] + rk;  // local -> global rank, for every index (see preprocess_kernel)
  }
  int hits = 0, span_n = 0;
  unsigned long long hm = 0ull;
  float bu = 0.0f, bv = 0.0f, br0 = 0.0f, br1 = 0.0f, br2 = 0.0f, br3 = 0.0f;
  if (act) {
    const gs::Mat34 vw = gs::load_view(view);
    // the colour: requested here, used when the record is made
    float rgb[3] = {0.0f, 0.0f, 0.0f};
    if constexpr (kColour == 0) {
      float dx, dy, dz, len;
      gs::view_dir(cur.x, cur.y, cur.z, cx, cy, cz, dx, dy, dz, len);
      gs::sh_to_rgb<0>(nullptr, g.rgb + 3 * i, dx, dy, dz, rgb);
    } else if constexpr (kColour == 1) {
      rgb[0] = o.rgb[3 * j]; rgb[1] = o.rgb[3 * j + 1]; rgb[2] = o.rgb[3 * j + 2];
    }
    const gs::RotScale rs = gs::rot_scale(cur.q.x, cur.q.y, cur.q.z, cur.q.w, cur.sx, cur.sy, cur.sz);
    float sg[6], J[6], con[3], rad[4];
    gs::sigma_from(rs, sg);
    int pp = 0, nact = 0, j0 = 0;
    float *wsh = s_stage[0];
    if constexpr (kStoreMid) {
      const unsigned long long actm = __ballot(true);
      nact = __popcll(actm);
      pp = __popcll(actm & ((1ull << lane) - 1ull));
      j0 = __builtin_amdgcn_readfirstlane(j - pp);
      wsh = s_stage[threadIdx.x >> 6];
      wave_rows_store<6>(o.sigma, j0, pp, nact, sg, wsh);
    }
    float x, y, z, u, v;
    gs::camera_space(vw, cur.x, cur.y, cur.z, x, y, z);
    gs::to_screen(gs::load_proj(proj), x, y, z, width, height, u, v);
    gs::jacobian(x, y, z, fx, fy, tan_fovx, tan_fovy, J);
    gs::conic_radius(J, sg, vw, mh_dist, con, rad);
    o.c2g[j] = i;
    o.uv[2 * j] = u; o.uv[2 * j + 1] = v;
    reinterpret_cast<float4 *>(o.radius)[j] = make_float4(rad[0], rad[1], rad[2], rad[3]);
    const gs::SplatRec rec = gs::make_record(u, v, con[0], con[1], con[2], cur.op, rgb[0], rgb[1], rgb[2]);
    if constexpr (kStoreMid) {
      const float xyzc[3] = {x, y, z};
      wave_rows_store<3>(o.xyz_c, j0, pp, nact, xyzc, wsh);
      wave_rows_store<6>(o.J, j0, pp, nact, J, wsh);
      wave_rows_store<3>(o.conic, j0, pp, nact, con, wsh);
      if constexpr (kColour == 0) wave_rows_store<3>(o.rgb, j0, pp, nact, rgb, wsh);
      if constexpr (kColour == 2) {  // (the colour's twelve bytes of the record belong to sh_colour_kernel)
        o.recs[3 * j] = rec.r0; o.recs[3 * j + 1] = rec.r1;
        reinterpret_cast<float *>(o.recs + 3 * (size_t)j + 2)[3] = rec.r2.w;
      } else {
        const float rr[12] = {rec.r0.x, rec.r0.y, rec.r0.z, rec.r0.w, rec.r1.x, rec.r1.y, rec.r1.z, rec.r1.w,
                              rec.r2.x, rec.r2.y, rec.r2.z, rec.r2.w};
        wave_rows_store<12>(reinterpret_cast<float *>(o.recs), j0, pp, nact, rr, wsh);
      }
    } else {
      o.xyz_c[3 * j] = x; o.xyz_c[3 * j + 1] = y; o.xyz_c[3 * j + 2] = z;
      o.recs[3 * j] = rec.r0; o.recs[3 * j + 1] = rec.r1;
      if constexpr (kColour == 2) reinterpret_cast<float *>(o.recs + 3 * (size_t)j + 2)[3] = rec.r2.w;
      else o.recs[3 * j + 2] = rec.r2;
    }
    bu = u; bv = v; br0 = rad[0]; br1 = rad[1]; br2 = rad[2]; br3 = rad[3];
  }
  // the rest of the trip's stores -- the global rank (every index: the pack kernels read it through the mask) and the
  // PREVIOUS chunk's tile count and hit mask, which its tile loop produced behind that trip's stores --, then the next
  // chunk's requests, then the tile loop: no memory instruction in it, so when the loop top waits for the requests,
  // everything this wave has in flight is a tile loop old
  if constexpr (kCompact) {
    if (act) rank[i] = j;
  } else {
    if (e0 < N) rank[e0] = j;
  }
  if (late_act) {
    o.counts[late_j] = late_hits;
    o.hitmask[late_j] = late_hm;
  }
  asm volatile("" ::: "memory");
  if constexpr (kCompact) { ix_n = ix_nn; }
  request(c + kStep);
  if constexpr (kCompact) { ix_nn = kept_index(c + 2 * kStep); }
  asm volatile("" ::: "memory");
  if (act) {
    const gs::TileRect r = gs::coarse_rect(bu, bv, br0, ntx, nty);
    if (r.x1 > r.x0 && r.y1 > r.y0) {
      coarse += (unsigned long long)(r.x1 - r.x0) * (unsigned long long)(r.y1 - r.y0);
      const gs::Obb ob = gs::make_obb(bu, bv, br0, br1, br2, br3);
      const gs::TileRect sp = gs::obb_span(ob, r);
      const int rh = r.y1 - r.y0;
      span_n = max(0, sp.x1 - sp.x0) * max(0, sp.y1 - sp.y0);
      if (span_n <= kCoopTiles) {
        for (int tx = sp.x0; tx < sp.x1; ++tx)
          for (int ty = sp.y0; ty < sp.y1; ++ty) {
            const bool h = gs::obb_hits_tile(ob, tx, ty);
            const int bit = (tx - r.x0) * rh + (ty - r.y0);
            hits += h ? 1 : 0;
            hm |= (h && bit < 64) ? (1ull << bit) : 0ull;
            if (h && table) atomicAdd(&s_hist[ty * ntx + tx], 1);
          }
      }
    }
  }
  for (unsigned long long big = __ballot(span_n > kCoopTiles); big != 0ull; big &= big - 1ull) {  // (as preprocess_kernel)
    const int owner = __builtin_ctzll(big);
    const float ou = __shfl(bu, owner, 64), ov = __shfl(bv, owner, 64);
    const float o0 = __shfl(br0, owner, 64), o1 = __shfl(br1, owner, 64), o2 = __shfl(br2, owner, 64), o3 = __shfl(br3, owner, 64);
    const gs::TileRect r = gs::coarse_rect(ou, ov, o0, ntx, nty);
    const gs::Obb ob = gs::make_obb(ou, ov, o0, o1, o2, o3);
    const gs::TileRect sp = gs::obb_span(ob, r);
    const int sh = sp.y1 - sp.y0, total = (sp.x1 - sp.x0) * sh;
    int found = 0;
    for (int p = lane; p < total + lane; p += 64) {
      bool h = false;
      if (p < total) {
        const int tx = sp.x0 + p / sh, ty = sp.y0 + p % sh;
        h = gs::obb_hits_tile(ob, tx, ty);
        if (h && table) atomicAdd(&s_hist[ty * ntx + tx], 1);
      }
      found += __popcll(__ballot(h));
    }
    if (lane == owner) hits = found;
  }
  late_act = act; late_j = j; late_hits = hits; late_hm = hm;
  }
  if (late_act) {
    o.counts[late_j] = late_hits;
    o.hitmask[late_j] = late_hm;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) coarse += __shfl_down(coarse, off, 64);
  if ((threadIdx.x & 63) == 0 && coarse) atomicAdd(&o.pairs[(blockIdx.x * 16 + (threadIdx.x >> 6)) & 63], coarse);
  if (table) {
    __syncthreads();
    for (int t = threadIdx.x; t < T; t += gs::kBinThreads) table[(size_t)blockIdx.x * T + t] = s_hist[t];
  }
}

// {M, S, candidate pairs} gathered into one 16-byte record so that the forward's only read-back is one copy
__global__ __launch_bounds__(64) void publish_counts_kernel(const int *__restrict__ rank_total,
                                                            const int *__restrict__ offsets_total,
                                                            const unsigned long long *__restrict__ pair_counters,
                                                            volatile unsigned long long *out,
                                                            unsigned long long ticket) {
  // `out` is pinned host memory mapped into the device: the host polls it for its ticket instead of paying a
  // copy + stream synchronisation, and the kernels queued behind this one keep the GPU busy meanwhile
  unsigned long long v = pair_counters[threadIdx.x];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  if (threadIdx.x == 0)
    gs::publish_record(out, ticket, (unsigned int)*rank_total, (unsigned int)*offsets_total, v, 0u);
}

struct BwdOut {
  float *xyz, *rgb, *sh, *opacity, *scale, *quaternion;
  float *conic, *uv, *J, *sigma, *xyz_c, *pre_rgb;
  // r05, view-sharded step: instead of the six compacted leaf arrays the kernel writes the row the exchange sums --
  // common[i] = {xyz 3, opacity, scale 3, quaternion 4, 1.0 (this view saw the gaussian)} at the gaussian's GLOBAL index i
  // (48 bytes, three 16-byte stores) -- and |grad_uv| at uv_norm[i]; the rows of culled gaussians were zeroed by
  // gsplat_backward_render_split.  No compacted gradient array, no pack pass.
  float *common, *uv_norm;
};

// Number of entries of the increasing array c2g[0..M) that are below `key` = the first compacted slot whose global index
// is >= key.  A 64-ary search by the whole wave: every step the lanes probe 64 evenly spaced entries and a ballot keeps
// the one sub-interval that contains the answer (four steps of one load each at 1e6 gaussians; every wave of the
// grid does it for itself, no barrier).
__device__ __forceinline__ int first_slot_not_below(const int *__restrict__ c2g, int M, int key) {
  const int lane = threadIdx.x & 63;
  int lo = 0, hi = M;  // c2g[j] < key for j < lo, c2g[j] >= key for j >= hi
  while (hi > lo) {
    const int step = (hi - lo + 63) >> 6;
    const long long p = (long long)lo + (long long)lane * step;
    const bool below = p < hi && c2g[p] < key;
    const int t = __popcll(__ballot(below));  // the array is increasing: the first t probes are below the key
    if (t == 0) {
      hi = lo;
    } else {
      const long long next = (long long)lo + (long long)t * step;  // the probe after the last one below (if it exists)
      hi = next < hi ? (int)next : hi;
      lo = lo + (t - 1) * step + 1;
    }
  }
  return lo;
}

// r06: what the per-gaussian backward needs to apply the optimizer step itself (kAdam; gsplat_backward_gaussians_adam):
// the moments of the six parameter groups in GLOBAL order next to the parameters (`g`, updated in place), the groups'
// learning rates, Adam's constants and the densification statistics.
struct AdamFused {
  float *m_xyz, *v_xyz, *m_rgb, *v_rgb, *m_sh, *v_sh, *m_op, *v_op, *m_sc, *v_sc, *m_q, *v_q;
  float lr_xyz, lr_rgb, lr_sh, lr_op, lr_sc, lr_q;
  float b1, b2, eps, bias1, bias2;
  float *uv_accum;
  int *accum_dur;
  float *dir;  // kAdam 3: [M,3] d loss / d position through the view direction of the colour (sh_adam_dir_kernel -> here)
};

// ---- backward of everything per gaussian, one thread per compacted slot
// kAdam (r06, single-GPU training): the kernel applies the masked Adam step of TrainerImpl::optimizer_step
// (cuda/trainer.cu:1027-1158) to the gaussian it has just differentiated instead of storing six gradient arrays for a
// second and third kernel to read back next to parameters this one has just read.  A gaussian's gradients depend on its
// own parameters only, so updating in place is safe once the thread -- for the SH rows: the wave -- has read what it
// needs.  Gradient values and update are the separate kernels' (gs::sh_bwd's product, gs::adam_values): parameters and
// moments come out bit-identical to gsplat_backward_gaussians + gsplat_optimizer_step_sh_factored + gsplat_optimizer_step
// (tests/test_optimizer_gpu.py).  The SH group: the coefficient rows stay in the wave's LDS rows (gs::sh_bwd<L, false>),
// every lane parks its direction and colour gradient next to them, and the wave walks its 64 rows element by element --
// thread e owns element e of the span, rebuilds gradient = g_rgb[channel] * Y_k(direction) as
// optimizer_sh_factored_kernel does and streams sh / exp_avg / exp_avg_sq through the update: consecutive lanes,
// consecutive addresses.
// kAdam 2: all six groups here.  kAdam 1: band 0, opacity, scale, rotation and the statistics here; the SH group
// (gsplat_optimizer_step_sh_factored, which needs the positions the backward saw) and the position group
// (gsplat_optimizer_step on grad_xyz) stay with the optimizer kernels behind this one.  kAdam 3: the five small groups
// here and NOTHING of the SH rows -- sh_adam_dir_kernel (below) has run in front of this kernel: it read the coefficient
// rows once, for their Adam step AND for sh_bwd's sums over them, and left the position gradient through the view
// direction in ad.dir; no LDS, and without sh_bwd's basis tables far fewer registers.
#ifndef GS_BWD3_WAVES
#define GS_BWD3_WAVES 3  // waves per SIMD the kAdam 3 form is compiled for (r06 A/B)
#endif
template <int L, int kAdam = 0>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(kAdam == 3 ? GS_BWD3_WAVES : 3, 8))) void preprocess_bwd_kernel(gsplat_gaussians g, const float *__restrict__ view,
                                                                const float *__restrict__ proj, int M,
                                                                const int *__restrict__ c2g,
                                                                const float *__restrict__ xyz_c_sel,
                                                                const float4 *__restrict__ rows_in, float fx, float fy,
                                                                float tan_fovx, float tan_fovy, float fwd_tan_fovx,
                                                                float fwd_tan_fovy, float mh_dist, float cx, float cy,
                                                                float cz, int width, int height, BwdOut o,
                                                                int ranged, int i_lo, int i_hi, AdamFused ad) {
  // ranged: only the gaussians with global index in [i_lo, i_hi), i.e. the compacted slots [first slot whose global index
  // is >= i_lo, first slot whose global index is >= i_hi) (chunked backward of a view-sharded step: the exchange of one
  // chunk runs while the next is computed); the grid covers the largest possible chunk, blocks past its end leave at once.
  // The two slots come from compact_to_global, which is increasing, NOT from rank[]: after a forward that walked the
  // compacted slots (preprocess_kernel<.., kCompact>) rank[i] of a CULLED index still holds the cull's slice-local
  // count, and a range bound may well be a culled index.
  int j_first = 0;
  if (ranged) {
    j_first = first_slot_not_below(c2g, M, i_lo);
    M = first_slot_not_below(c2g, M, i_hi);
  }
  const int j = j_first + blockIdx.x * kBlock + threadIdx.x;
  constexpr int n = (L + 1) * (L + 1), kRest = (n - 1) * 3;
  // The SH rows (kRest floats per gaussian, 180 B at degree 3) go through LDS: a lane reading ITS row touches 64
  // different cache lines per wave instruction; the wave's 64 rows as one linear span touch 8.  Same for the
  // gradient rows on the way out.  Each wave stages only its own rows (no workgroup barrier).
  __shared__ __attribute__((aligned(16))) float s_sh[kRest > 0 && kAdam != 3 ? kBlock * kRest : 4];
  // kAdam: per row the unit direction and the colour gradient (what the SH gradients are made of) and the global row
  __shared__ float s_dir[kAdam == 2 && kRest > 0 ? kBlock * 7 : 1];
  const int lane = threadIdx.x & 63, wave_first = threadIdx.x - lane;
  const int jw = j_first + blockIdx.x * kBlock + wave_first;  // first compacted slot of this wave
  if (jw >= M) return;
  const int rows = min(64, M - jw);
  const bool live = j < M;
  const int i = c2g[live ? j : jw];
  float *wsh = s_sh + (kAdam != 3 ? wave_first * kRest : 0);
  if constexpr (kRest > 0 && kAdam != 3) {
    const int i0 = __builtin_amdgcn_readfirstlane(i);
    if (__all(!live || i == i0 + lane)) {  // consecutive gaussians (no culling in between): one linear span
      gs::rows_to_lds<kRest>(g.sh + (size_t)i0 * kRest, wsh, rows, lane);
    } else {
      // Culled gaussians in between (every real training view): the wave's rows are scattered.  Twelve lanes fetch one
      // row -- eleven 16-byte pieces and the last float(s) -- so a wave instruction brings in five whole rows (880 B)
      // instead of one (180 B with one float per lane: 64 load instructions per wave, r01), and all of a wave's loads
      // are issued before the first LDS store.  Rows sit at a pitch of kRest words in LDS (conflict-free for the
      // per-lane walk of sh_bwd), which is not 16-byte aligned: the pieces are stored as four words.
      constexpr int kPieces = (kRest + 3) / 4, kPer = 64 / kPieces, kIter = (64 + kPer - 1) / kPer;
      static_assert(kPieces <= 64 && kPer >= 1, "row too long for one lane group");
      const int grp = lane / kPieces, piece = lane - grp * kPieces;
      const bool in_grp = grp < kPer;
      float4 v[kIter];
#pragma unroll
      for (int it = 0; it < kIter; ++it) {
        const int r = it * kPer + grp;
        const int ir = __shfl(i, r < rows ? r : 0, 64);
        v[it] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (in_grp && r < rows) {
          const float *src = g.sh + (size_t)ir * kRest + 4 * piece;
          if (4 * piece + 3 < kRest) {
            v[it] = __builtin_bit_cast(float4, *reinterpret_cast<const gs::f4u *>(src));
          } else {  // the row's last, partial piece
            v[it].x = src[0];
            if (4 * piece + 1 < kRest) v[it].y = src[1];
            if (4 * piece + 2 < kRest) v[it].z = src[2];
          }
        }
      }
#pragma unroll
      for (int it = 0; it < kIter; ++it) {
        const int r = it * kPer + grp;
        if (in_grp && r < rows) {
          float *dst = wsh + r * kRest + 4 * piece;
          dst[0] = v[it].x;
          if (4 * piece + 1 < kRest) dst[1] = v[it].y;
          if (4 * piece + 2 < kRest) dst[2] = v[it].z;
          if (4 * piece + 3 < kRest) dst[3] = v[it].w;
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  // (kAdam: the SH group first, while nothing else of the thread's state is live -- the kernel sits at the 168-register
  // step of three workgroups per CU; sh_bwd below still finds the coefficients in LDS, the update went to global memory)
  if constexpr (kRest > 0 && kAdam == 2) {
    {  // the direction exactly as gs::sh_bwd / optimizer_sh_factored_kernel form it (and the colour gradient: rows_in[.].xyz)
      const float4 ga = rows_in[4 * (live ? j : jw)];
      float ux, uy, uz, len;
      gs::view_dir(g.xyz[3 * i], g.xyz[3 * i + 1], g.xyz[3 * i + 2], cx, cy, cz, ux, uy, uz, len);
      float *d = s_dir + (wave_first + lane) * 7;
      d[0] = ux; d[1] = uy; d[2] = uz; d[3] = ga.x; d[4] = ga.y; d[5] = ga.z; d[6] = __int_as_float(i);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // Twelve lanes own one row (the layout of the scattered row loads above): lane `piece` of a group its floats
    // 4 piece .. 4 piece + 3, i.e. 16-byte pieces of parameter and moment rows, five rows per wave instruction, whatever
    // gaps culling has left between the rows; a lane evaluates the basis once per row.  The moment loads of kAhead
    // trips are requested before the first of them is used.
    float *sh_p = const_cast<float *>(g.sh);
    const float *dirs = s_dir + wave_first * 7;
    constexpr int kPieces = (kRest + 3) / 4, kPer = 64 / kPieces, kIter = (64 + kPer - 1) / kPer, kAhead = 4;
    const int grp = lane / kPieces, piece = lane - grp * kPieces;
    const bool in_grp = grp < kPer;
    constexpr int kLast = kRest - 4 * (kPieces - 1);  // floats of a row's last piece (1..4)
    const int nval = piece == kPieces - 1 ? kLast : 4;
#pragma unroll 1
    for (int it0 = 0; it0 < kIter; it0 += kAhead) {
      float4 mq[kAhead], vq[kAhead];
      size_t at[kAhead];
#pragma unroll
      for (int u = 0; u < kAhead; ++u) {
        const int r = (it0 + u) * kPer + grp;
        mq[u] = vq[u] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        at[u] = 0;
        if (it0 + u < kIter && in_grp && r < rows) {
          at[u] = (size_t)__float_as_int(dirs[r * 7 + 6]) * kRest + 4 * piece;
          if (nval == 4) {
            mq[u] = __builtin_bit_cast(float4, *reinterpret_cast<const gs::f4u *>(ad.m_sh + at[u]));
            vq[u] = __builtin_bit_cast(float4, *reinterpret_cast<const gs::f4u *>(ad.v_sh + at[u]));
          } else {
            mq[u].x = ad.m_sh[at[u]]; vq[u].x = ad.v_sh[at[u]];
            if (nval > 1) { mq[u].y = ad.m_sh[at[u] + 1]; vq[u].y = ad.v_sh[at[u] + 1]; }
            if (nval > 2) { mq[u].z = ad.m_sh[at[u] + 2]; vq[u].z = ad.v_sh[at[u] + 2]; }
          }
        }
      }
#pragma unroll
      for (int u = 0; u < kAhead; ++u) {
        const int r = (it0 + u) * kPer + grp;
        if (it0 + u < kIter && in_grp && r < rows) {
          const float *d = dirs + r * 7;
          float Y[n];
          gs::sh_basis<L>(d[0], d[1], d[2], Y);
          const float *prow = wsh + r * kRest + 4 * piece;
          float pv[4] = {prow[0], nval > 1 ? prow[1] : 0.0f, nval > 2 ? prow[2] : 0.0f, nval > 3 ? prow[3] : 0.0f};
          float mv[4] = {mq[u].x, mq[u].y, mq[u].z, mq[u].w}, vv[4] = {vq[u].x, vq[u].y, vq[u].z, vq[u].w};
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            if (t < nval) {
              const int c = 4 * piece + t, k = c / 3, ch = c - 3 * k;
              float yv = 0.0f;
#pragma unroll
              for (int q = 0; q < n - 1; ++q) yv = k == q ? Y[q + 1] : yv;
              const float grad = d[3 + ch] * yv;
              gs::adam_values(pv[t], mv[t], vv[t], grad, ad.lr_sh, ad.b1, ad.b2, ad.eps, ad.bias1, ad.bias2);
            }
          }
          if (nval == 4) {
            *reinterpret_cast<gs::f4u *>(sh_p + at[u]) = gs::f4u{pv[0], pv[1], pv[2], pv[3]};
            *reinterpret_cast<gs::f4u *>(ad.m_sh + at[u]) = gs::f4u{mv[0], mv[1], mv[2], mv[3]};
            *reinterpret_cast<gs::f4u *>(ad.v_sh + at[u]) = gs::f4u{vv[0], vv[1], vv[2], vv[3]};
          } else {
            for (int t = 0; t < nval; ++t) { sh_p[at[u] + t] = pv[t]; ad.m_sh[at[u] + t] = mv[t]; ad.v_sh[at[u] + t] = vv[t]; }
          }
        }
      }
    }
  }
  float gx = 0.0f, gy = 0.0f, gz = 0.0f, b0g[3] = {0.0f, 0.0f, 0.0f};
  const gs::Mat34 vw = gs::load_view(view);
  const gs::Mat44 pr = gs::load_proj(proj);
  const int jr = live ? j : jw;  // dead lanes of the last wave recompute row jw and store nothing
  const float4 a = rows_in[4 * jr], b = rows_in[4 * jr + 1], c = rows_in[4 * jr + 2];
  const float g_rgb[3] = {a.x, a.y, a.z};
  const float g_op = a.w;
  const float g_con[3] = {b.x, b.y, b.z};
  const float g_u = b.w, g_v = c.x;
  if constexpr (kAdam == 3 && kRest > 0) {
    // gs::sh_bwd's two results from elsewhere: band 0's gradient is its own expression (Y_0 is the constant), the position
    // gradient through the view direction is what sh_adam_dir_kernel left (the same sums in the same order)
    b0g[0] = g_rgb[0] * GS_SH_C0; b0g[1] = g_rgb[1] * GS_SH_C0; b0g[2] = g_rgb[2] * GS_SH_C0;
    gx = ad.dir[3 * (size_t)jr]; gy = ad.dir[3 * (size_t)jr + 1]; gz = ad.dir[3 * (size_t)jr + 2];
  } else {
    float *row = wsh + lane * kRest;  // read as coefficients, overwritten with their gradients (kAdam: left as they are)
    gs::sh_bwd<L, kAdam == 0>(row, g.rgb + 3 * i, g.xyz[3 * i], g.xyz[3 * i + 1], g.xyz[3 * i + 2], cx, cy, cz, g_rgb, row, b0g,
                          gx, gy, gz);
  }
  if constexpr (kRest > 0 && kAdam == 0) {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (o.sh) gs::rows_from_lds<kRest>(o.sh + (size_t)jw * kRest, wsh, rows, lane);
  }
  if (!live) return;
  gx = 0.0f + gx; gy = 0.0f + gy; gz = 0.0f + gz;
  // kAdam: one three-vector group of this lane's gaussian: parameter and moments at the GLOBAL row i
  auto step3 = [&](const float *pc, float *m, float *v, const float *gr, float lr) {
    float *p = const_cast<float *>(pc);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      float pv = p[3 * i + k], mv = m[3 * i + k], vv = v[3 * i + k];
      gs::adam_values(pv, mv, vv, gr[k], lr, ad.b1, ad.b2, ad.eps, ad.bias1, ad.bias2);
      p[3 * i + k] = pv; m[3 * i + k] = mv; v[3 * i + k] = vv;
    }
  };
  if constexpr (kAdam != 0) {
    // what is final already -- band 0 (nothing below reads it again), the opacity, the densification statistics
    // (cuda/trainer.cu:1136-1157, optimizer_step_kernel's expressions) -- leaves now, not across the covariance chain
    step3(g.rgb, ad.m_rgb, ad.v_rgb, b0g, ad.lr_rgb);
    {
      float pv = g.opacity[i], mv = ad.m_op[i], vv = ad.v_op[i];
      gs::adam_values(pv, mv, vv, g_op, ad.lr_op, ad.b1, ad.b2, ad.eps, ad.bias1, ad.bias2);
      const_cast<float *>(g.opacity)[i] = pv; ad.m_op[i] = mv; ad.v_op[i] = vv;
    }
    if (ad.uv_accum) ad.uv_accum[i] += sqrtf(g_u * g_u + g_v * g_v);
    if (ad.accum_dur) ad.accum_dur[i] += 1;
  }
  // conic -> (J, Sigma).  Sigma, J and the conic are RECOMPUTED from what this kernel reads anyway (quaternion, scale,
  // camera-space position) with the forward's functions and the forward's tan(fov): bit for bit the values
  // preprocess_kernel stored, without reading 60 bytes per gaussian back (the kernel is HBM bound).
  const float x = xyz_c_sel[3 * j], y = xyz_c_sel[3 * j + 1], z = xyz_c_sel[3 * j + 2];
  float Jv[6], sg[6], con[3], rad_unused[4], dJ[6], dS[6];
  {
    const float4 q = reinterpret_cast<const float4 *>(g.quaternion)[i];
    const gs::RotScale rs = gs::rot_scale(q.x, q.y, q.z, q.w, g.scale[3 * i], g.scale[3 * i + 1], g.scale[3 * i + 2]);
    gs::sigma_from(rs, sg);
  }
  gs::jacobian(x, y, z, fx, fy, fwd_tan_fovx, fwd_tan_fovy, Jv);
  gs::conic_radius(Jv, sg, vw, mh_dist, con, rad_unused);
  gs::conic_bwd(Jv, sg, vw, con, g_con, dJ, dS);
#pragma unroll
  for (int k = 0; k < 6; ++k) { dJ[k] = 0.0f + dJ[k]; dS[k] = 0.0f + dS[k]; }
  // J -> xyz_c
  float cxg, cyg, czg;
  gs::jacobian_bwd(x, y, z, fx, fy, tan_fovx, tan_fovy, dJ, cxg, cyg, czg);
  cxg = 0.0f + cxg; cyg = 0.0f + cyg; czg = 0.0f + czg;
  // Sigma -> quaternion, scale.  The rotation/scale terms are built a second time from a second (L2-resident) read
  // through an opaque copy of the index: kept alive across conic_bwd they cost 58 VGPRs and a third of the occupancy.
  int i2 = i;
  asm volatile("" : "+v"(i2));
  const float4 q = reinterpret_cast<const float4 *>(g.quaternion)[i2];
  const gs::RotScale rs = gs::rot_scale(q.x, q.y, q.z, q.w, g.scale[3 * i2], g.scale[3 * i2 + 1], g.scale[3 * i2 + 2]);
  float dQ[4], dSc[3];
  gs::sigma_bwd(rs, dS, dQ, dSc);
  // uv -> xyz_c
  float px_, py_, pz_;
  gs::to_screen_bwd(pr, x, y, z, g_u, g_v, width, height, px_, py_, pz_);
  cxg += px_; cyg += py_; czg += pz_;
  // xyz_c -> xyz
  float wx, wy, wz;
  gs::camera_space_bwd(vw, cxg, cyg, czg, wx, wy, wz);
  gx += wx; gy += wy; gz += wz;
  if constexpr (kAdam != 0) {
    // the groups whose gradients the covariance chain produced: position (kAdam 2), scale, rotation
    if constexpr (kAdam == 2 || kAdam == 3) {
      const float gxyz[3] = {gx, gy, gz};
      step3(g.xyz, ad.m_xyz, ad.v_xyz, gxyz, ad.lr_xyz);
    }
    step3(g.scale, ad.m_sc, ad.v_sc, dSc, ad.lr_sc);
    {
      float4 pq = reinterpret_cast<const float4 *>(g.quaternion)[i];
      float4 mq = reinterpret_cast<const float4 *>(ad.m_q)[i], vq = reinterpret_cast<const float4 *>(ad.v_q)[i];
      gs::adam_values(pq.x, mq.x, vq.x, dQ[0], ad.lr_q, ad.b1, ad.b2, ad.eps, ad.bias1, ad.bias2);
      gs::adam_values(pq.y, mq.y, vq.y, dQ[1], ad.lr_q, ad.b1, ad.b2, ad.eps, ad.bias1, ad.bias2);
      gs::adam_values(pq.z, mq.z, vq.z, dQ[2], ad.lr_q, ad.b1, ad.b2, ad.eps, ad.bias1, ad.bias2);
      gs::adam_values(pq.w, mq.w, vq.w, dQ[3], ad.lr_q, ad.b1, ad.b2, ad.eps, ad.bias1, ad.bias2);
      reinterpret_cast<float4 *>(const_cast<float *>(g.quaternion))[i] = pq;
      reinterpret_cast<float4 *>(ad.m_q)[i] = mq; reinterpret_cast<float4 *>(ad.v_q)[i] = vq;
    }
    if (!o.xyz) return;  // (gradient arrays: only when the caller asked for them)
  }
  // stores
  if constexpr (kAdam == 1) {
    // what the optimizer kernels behind this one need: the position gradient and the colour gradient the SH gradients
    // are made of; the other arrays only where the caller gave them
    o.xyz[3 * j] = gx; o.xyz[3 * j + 1] = gy; o.xyz[3 * j + 2] = gz;
    if (o.opacity) o.opacity[j] = g_op;
    if (o.scale) { o.scale[3 * j] = dSc[0]; o.scale[3 * j + 1] = dSc[1]; o.scale[3 * j + 2] = dSc[2]; }
    if (o.quaternion) reinterpret_cast<float4 *>(o.quaternion)[j] = make_float4(dQ[0], dQ[1], dQ[2], dQ[3]);
  } else if (o.common) {  // the exchange's row, in global order (the same twelve values pack_split_kernel gathers)
    gs::f4u *row = reinterpret_cast<gs::f4u *>(o.common + (size_t)i * 12);
    row[0] = gs::f4u{gx, gy, gz, g_op};
    row[1] = gs::f4u{dSc[0], dSc[1], dSc[2], dQ[0]};
    row[2] = gs::f4u{dQ[1], dQ[2], dQ[3], 1.0f};
    if (o.uv_norm) o.uv_norm[i] = sqrtf(g_u * g_u + g_v * g_v);  // pack_uv_norm_kernel's expression
  } else {
    o.xyz[3 * j] = gx; o.xyz[3 * j + 1] = gy; o.xyz[3 * j + 2] = gz;
    o.opacity[j] = g_op;
    o.scale[3 * j] = dSc[0]; o.scale[3 * j + 1] = dSc[1]; o.scale[3 * j + 2] = dSc[2];
    reinterpret_cast<float4 *>(o.quaternion)[j] = make_float4(dQ[0], dQ[1], dQ[2], dQ[3]);
  }
  if (o.rgb) { o.rgb[3 * j] = b0g[0]; o.rgb[3 * j + 1] = b0g[1]; o.rgb[3 * j + 2] = b0g[2]; }
  if (o.conic) { o.conic[3 * j] = g_con[0]; o.conic[3 * j + 1] = g_con[1]; o.conic[3 * j + 2] = g_con[2]; }
  if (o.uv) { o.uv[2 * j] = g_u; o.uv[2 * j + 1] = g_v; }
  if (o.pre_rgb) { o.pre_rgb[3 * j] = g_rgb[0]; o.pre_rgb[3 * j + 1] = g_rgb[1]; o.pre_rgb[3 * j + 2] = g_rgb[2]; }
  if (o.J) {
#pragma unroll
    for (int k = 0; k < 6; ++k) o.J[6 * j + k] = dJ[k];
  }
  if (o.sigma) {
#pragma unroll
    for (int k = 0; k < 6; ++k) o.sigma[6 * j + k] = dS[k];
  }
  if (o.xyz_c) { o.xyz_c[3 * j] = cxg; o.xyz_c[3 * j + 1] = cyg; o.xyz_c[3 * j + 2] = czg; }
}

// r06 (gsplat_adam_fused.mode 2): the SH group's Adam step and gs::sh_bwd's sums over the coefficient rows in ONE read of
// the rows.  The unfused iteration reads them twice -- preprocess_bwd_kernel for d colour / d direction, then
// optimizer_sh_factored_kernel for the update -- and the one fat kernel (kAdam 2) reads them once but at three waves per
// SIMD.  Here sixteen lanes own one gaussian: lane k its coefficient k + 1 (three channels: 12 bytes of the parameter row
// and of both moment rows, neighbouring lanes neighbouring pieces, as optimizer_sh_factored_kernel).  Every lane evaluates
// the basis and its gradient at the row's direction and keeps its own coefficient's entries; the update is
// optimizer_sh_factored_kernel's (gradient = g_rgb[channel] * Y_k), from the OLD coefficients the lane also forms its
// nine products dY_k/d(axis) * coefficient[channel], and the nine sums over k run through the row as a chain of DPP adds
// (row_shr:1, one instruction per step and sum): lane k ends with exactly gs::sh_bwd's left-to-right partial sum
// ((0 + band-0 term) + k = 1) + ... + k, so the
// last coefficient's lane holds sh_bwd's sums BIT FOR BIT and finishes its three outputs.  No LDS, ~60 registers.
template <int L>
__global__ __launch_bounds__(kBlock) void sh_adam_dir_kernel(int M, const int *__restrict__ c2g, float *__restrict__ sh,
                                                             float *__restrict__ m, float *__restrict__ v, float lr,
                                                             float b1, float b2, float eps, float bias1, float bias2,
                                                             const float *__restrict__ xyz,
                                                             const float *__restrict__ band0, float cx, float cy, float cz,
                                                             const float4 *__restrict__ rows_in, float *__restrict__ dir_out) {
  constexpr int n = (L + 1) * (L + 1), kCoef = n - 1;
  static_assert(L >= 1 && kCoef <= 15, "one 16-lane row per gaussian");
  const unsigned int t = blockIdx.x * (unsigned int)kBlock + threadIdx.x;
  const unsigned int r = t >> 4;
  const int k = (int)(t & 15u);
  if (r >= (unsigned int)M) return;  // (a whole DPP row leaves together)
  const long long row = c2g[r];
  const float4 ga = rows_in[4 * (size_t)r];  // d loss / d colour of this view: rows[.][0..2]
  const float gr[3] = {ga.x, ga.y, ga.z};
  float ux, uy, uz, len;
  gs::view_dir(xyz[3 * row], xyz[3 * row + 1], xyz[3 * row + 2], cx, cy, cz, ux, uy, uz, len);
  float Y[n], dY[n][3];
  gs::sh_basis<L>(ux, uy, uz, Y);
  gs::sh_basis_grad<L>(ux, uy, uz, dY);
  float yv = 0.0f, dd[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int q = 0; q < kCoef; ++q) {
    yv = k == q ? Y[q + 1] : yv;
    dd[0] = k == q ? dY[q + 1][0] : dd[0]; dd[1] = k == q ? dY[q + 1][1] : dd[1]; dd[2] = k == q ? dY[q + 1][2] : dd[2];
  }
  const bool act = k < kCoef;
  const long long o = (row * kCoef + k) * 3;
  float p[3] = {0.0f, 0.0f, 0.0f};
  if (act) {
    float mm[3], vv[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) { p[c] = sh[o + c]; mm[c] = m[o + c]; vv[c] = v[o + c]; }
    float pn[3] = {p[0], p[1], p[2]};
#pragma unroll
    for (int c = 0; c < 3; ++c) gs::adam_values(pn[c], mm[c], vv[c], gr[c] * yv, lr, b1, b2, eps, bias1, bias2);
#pragma unroll
    for (int c = 0; c < 3; ++c) { sh[o + c] = pn[c]; m[o + c] = mm[c]; v[o + c] = vv[c]; }
  }
  // acc[axis][channel]: gs::sh_bwd's dRx dGx dBx | dRy .. | dRz ..  x = what a lane adds: its product, in lane 0 sh_bwd's
  // start (0 + band-0 term) + product.  One DPP add per step and sum: lane l takes lane l - 1's partial sum + x; lane 0 gets a
  // zero shifted in and re-forms 0 + x = its start (never -0: a sum that began with + 0 is not), so there is no select.
  float x[3][3], acc[3][3];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float leaf = dd[a] * p[c];
      float first = 0.0f;
      first += dY[0][a] * band0[3 * row + c];
      first += leaf;
      x[a][c] = acc[a][c] = k == 0 ? first : leaf;
    }
#pragma unroll
  for (int step = 1; step < kCoef; ++step)
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int c = 0; c < 3; ++c)
        acc[a][c] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(acc[a][c]), 0x111 /* row_shr:1 */, 0xF, 0xF, true)) + x[a][c];
  if (k == kCoef - 1) {  // gs::sh_bwd's last lines
    const float tx = gr[0] * acc[0][0] + gr[1] * acc[0][1] + gr[2] * acc[0][2];
    const float ty = gr[0] * acc[1][0] + gr[1] * acc[1][1] + gr[2] * acc[1][2];
    const float tz = gr[0] * acc[2][0] + gr[1] * acc[2][1] + gr[2] * acc[2][2];
    const float dot = tx * ux + ty * uy + tz * uz;
    dir_out[3 * (size_t)r] = (tx - dot * ux) / len;
    dir_out[3 * (size_t)r + 1] = (ty - dot * uy) / len;
    dir_out[3 * (size_t)r + 2] = (tz - dot * uz) / len;
  }
}

// ---- global-order gradient rows for the view-sharded all-reduce
__global__ __launch_bounds__(kBlock) void pack_global_kernel(const unsigned char *__restrict__ mask,
                                                             const int *__restrict__ rank, int N, int n_coeffs,
                                                             gsplat_gradients gr, float *__restrict__ packed) {
  const int width = 12 + 3 * n_coeffs;
  const long long e = (long long)blockIdx.x * kBlock + threadIdx.x;
  if (e >= (long long)N * width) return;
  const int i = (int)(e / width), k = (int)(e % width);
  float val = 0.0f;
  if (mask[i]) {
    const size_t j = (size_t)rank[i];
    const int n_rest3 = 3 * (n_coeffs - 1);
    if (k < 3) val = gr.grad_xyz[3 * j + k];
    else if (k < 6) val = gr.grad_rgb[3 * j + (k - 3)];
    else if (k < 6 + n_rest3) val = gr.grad_sh[j * n_rest3 + (k - 6)];
    else if (k == 6 + n_rest3) val = gr.grad_opacity[j];
    else if (k < 10 + n_rest3) val = gr.grad_scale[3 * j + (k - 7 - n_rest3)];
    else if (k < 14 + n_rest3) val = gr.grad_quaternion[4 * j + (k - 10 - n_rest3)];
    else val = 1.0f;
  }
  packed[e] = val;
}

// ---- factored exchange rows (view-sharded training).  Every SH-coefficient gradient of a view is the outer product
// g_rgb (3) x Y_k(view direction): instead of shipping 3*n_coeffs floats per gaussian through the all-reduce, a
// rank ships only its g_rgb in a slot of its own (3 floats; the other ranks' slots stay zero, so the SUM all-reduce
// acts as an all-gather for those columns) and every rank rebuilds sum_r g_rgb^r * Y_k(dir^r) locally from the
// camera positions.  Row layout: [xyz3 | opacity1 | scale3 | quat4 | visible1 | g_rgb slot 0 .. slot W-1].
__global__ __launch_bounds__(kBlock) void pack_factored_kernel(const unsigned char *__restrict__ mask,
                                                               const int *__restrict__ rank_of, int N, int my_rank,
                                                               int world, gsplat_gradients gr,
                                                               float *__restrict__ packed) {
  const int width = 12 + 3 * world;
  const long long e = (long long)blockIdx.x * kBlock + threadIdx.x;
  if (e >= (long long)N * width) return;
  const int i = (int)(e / width), k = (int)(e % width);
  float val = 0.0f;
  if (mask[i]) {
    const size_t j = (size_t)rank_of[i];
    if (k < 3) val = gr.grad_xyz[3 * j + k];
    else if (k == 3) val = gr.grad_opacity[j];
    else if (k < 7) val = gr.grad_scale[3 * j + (k - 4)];
    else if (k < 11) val = gr.grad_quaternion[4 * j + (k - 7)];
    else if (k == 11) val = 1.0f;
    else if ((k - 12) / 3 == my_rank) val = gr.grad_precompute_rgb[3 * j + (k - 12) % 3];
  }
  packed[e] = val;
}

template <int L>
__global__ __launch_bounds__(kBlock) void unpack_factored_kernel(const float *__restrict__ xyz,
                                                                 const float *__restrict__ campos_all, int N,
                                                                 int world, const float *__restrict__ packed,
                                                                 float *__restrict__ full) {
  constexpr int n = (L + 1) * (L + 1);
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= N) return;
  const int wf = 12 + 3 * world, wo = 12 + 3 * n;
  const float *row = packed + (size_t)i * wf;
  float *out = full + (size_t)i * wo;
  float acc[n][3];
#pragma unroll
  for (int k = 0; k < n; ++k) acc[k][0] = acc[k][1] = acc[k][2] = 0.0f;
  const float px = xyz[3 * i], py = xyz[3 * i + 1], pz = xyz[3 * i + 2];
  for (int r = 0; r < world; ++r) {
    const float g0 = row[12 + 3 * r], g1 = row[13 + 3 * r], g2 = row[14 + 3 * r];
    if (g0 == 0.0f && g1 == 0.0f && g2 == 0.0f) continue;
    float dx, dy, dz, len, Y[n];
    gs::view_dir(px, py, pz, campos_all[3 * r], campos_all[3 * r + 1], campos_all[3 * r + 2], dx, dy, dz, len);
    gs::sh_basis<L>(dx, dy, dz, Y);
#pragma unroll
    for (int k = 0; k < n; ++k) { acc[k][0] += g0 * Y[k]; acc[k][1] += g1 * Y[k]; acc[k][2] += g2 * Y[k]; }
  }
  out[0] = row[0]; out[1] = row[1]; out[2] = row[2];  // xyz
#pragma unroll
  for (int k = 0; k < n; ++k) { out[3 + 3 * k] = acc[k][0]; out[4 + 3 * k] = acc[k][1]; out[5 + 3 * k] = acc[k][2]; }
  out[3 + 3 * n] = row[3];                                             // opacity
  out[4 + 3 * n] = row[4]; out[5 + 3 * n] = row[5]; out[6 + 3 * n] = row[6];  // scale
  out[7 + 3 * n] = row[7]; out[8 + 3 * n] = row[8]; out[9 + 3 * n] = row[9]; out[10 + 3 * n] = row[10];  // quaternion
  out[11 + 3 * n] = row[11];                                           // visibility count
}

// g_rgb of this view in global gaussian order, straight from the compositing backward's rows (rows[j][0..2]):
// available before the per-gaussian backward has run, so its all-gather can overlap that kernel.
__global__ __launch_bounds__(kBlock) void scatter_rgb_rows_kernel(const unsigned char *__restrict__ mask,
                                                                  const int *__restrict__ rank_of, int N,
                                                                  const float *__restrict__ rows,
                                                                  float *__restrict__ rgb, float *__restrict__ common,
                                                                  float *__restrict__ uv_norm) {
  const long long e = (long long)blockIdx.x * kBlock + threadIdx.x;
  if (e >= (long long)N * 3) return;
  const int i = (int)(e / 3), k = (int)(e % 3);
  const bool kept = mask[i] != 0;
  rgb[e] = kept ? rows[(size_t)rank_of[i] * 16 + k] : 0.0f;
  // r05: the exchange's twelve common columns are written in place by preprocess_bwd_kernel for the gaussians this view
  // saw; the rows of the culled ones are cleared here (thread k of a row: its k-th 16 bytes), where the mask is read anyway
  if (common && !kept) {
    reinterpret_cast<gs::f4u *>(common + (size_t)i * 12)[k] = gs::f4u{0.0f, 0.0f, 0.0f, 0.0f};
    if (uv_norm && k == 0) uv_norm[i] = 0.0f;
  }
}

// |grad_uv| of this view in global gaussian order (0 where culled): the densification statistic of a view-sharded step
__global__ __launch_bounds__(kBlock) void pack_uv_norm_kernel(const unsigned char *__restrict__ mask,
                                                              const int *__restrict__ rank_of, int N,
                                                              const float *__restrict__ grad_uv,
                                                              float *__restrict__ out) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= N) return;
  float val = 0.0f;
  if (mask[i]) {
    const float2 g = reinterpret_cast<const float2 *>(grad_uv)[rank_of[i]];
    val = sqrtf(g.x * g.x + g.y * g.y);
  }
  out[i] = val;
}

// Split exchange: the 12 direction-independent columns (SUM all-reduce) and this view's g_rgb (all-gather).
__global__ __launch_bounds__(kBlock) void pack_split_kernel(const unsigned char *__restrict__ mask,
                                                            const int *__restrict__ rank_of, int i_first, int N,
                                                            gsplat_gradients gr, float *__restrict__ common,
                                                            float *__restrict__ rgb) {
  const int i = i_first + blockIdx.x * kBlock + threadIdx.x;  // one gaussian per thread: three 16-byte stores per row
  if (i >= N) return;
  gs::f4u r0 = {0.0f, 0.0f, 0.0f, 0.0f}, r1 = r0, r2 = r0;
  float c0 = 0.0f, c1 = 0.0f, c2 = 0.0f;
  if (mask[i]) {
    const size_t j = (size_t)rank_of[i];
    const float *gq = gr.grad_quaternion + 4 * j;
    r0 = gs::f4u{gr.grad_xyz[3 * j], gr.grad_xyz[3 * j + 1], gr.grad_xyz[3 * j + 2], gr.grad_opacity[j]};
    r1 = gs::f4u{gr.grad_scale[3 * j], gr.grad_scale[3 * j + 1], gr.grad_scale[3 * j + 2], gq[0]};
    r2 = gs::f4u{gq[1], gq[2], gq[3], 1.0f};
    if (rgb) {
      c0 = gr.grad_precompute_rgb[3 * j]; c1 = gr.grad_precompute_rgb[3 * j + 1]; c2 = gr.grad_precompute_rgb[3 * j + 2];
    }
  }
  gs::f4u *row = reinterpret_cast<gs::f4u *>(common + (size_t)i * 12);
  row[0] = r0; row[1] = r1; row[2] = r2;
  if (rgb) { rgb[(size_t)i * 3] = c0; rgb[(size_t)i * 3 + 1] = c1; rgb[(size_t)i * 3 + 2] = c2; }
}

// rgb_all: world blocks of `stride` floats; block r = [N,3] g_rgb of rank r followed by that rank's campos[3].
// kSh: rebuild the SH-coefficient columns from rgb_all; kCommon: place the twelve all-reduced columns.  The two
// halves touch disjoint columns of `full`, so the SH half can run while the all-reduce of `common` is in flight.
template <int L, bool kSh, bool kCommon>
__global__ __launch_bounds__(kBlock) void unpack_split_kernel(const float *__restrict__ xyz, int N, int world,
                                                              const float *__restrict__ common,
                                                              const float *__restrict__ rgb_all, size_t stride,
                                                              float *__restrict__ full) {
  constexpr int n = (L + 1) * (L + 1);
  constexpr int wo = 12 + 3 * n;
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if constexpr (!kSh) {  // the twelve all-reduced columns only: two short pieces per row
    if (i >= N) return;
    float *out = full + (size_t)i * wo;
    const float *row = common + (size_t)i * 12;
    out[0] = row[0]; out[1] = row[1]; out[2] = row[2];                          // xyz
    out[3 + 3 * n] = row[3];                                                    // opacity
    out[4 + 3 * n] = row[4]; out[5 + 3 * n] = row[5]; out[6 + 3 * n] = row[6];  // scale
    out[7 + 3 * n] = row[7]; out[8 + 3 * n] = row[8]; out[9 + 3 * n] = row[9]; out[10 + 3 * n] = row[10];  // quaternion
    out[11 + 3 * n] = row[11];                                                  // visibility count
  } else {
    // The rebuilt rows leave through LDS, one row per wave instruction (a lane storing ITS 240-byte row touches 64
    // cache lines per instruction).  kCols floats of every row are written, starting at column kFirst.
    constexpr int kCols = kCommon ? wo : 3 * n, kFirst = kCommon ? 0 : 3, kPitch = kCols | 1;  // odd pitch: no bank conflicts
    __shared__ float s_rows[kBlock * kPitch];
    const int lane = threadIdx.x & 63, wave_first = threadIdx.x - lane;
    const int iw = blockIdx.x * kBlock + wave_first;
    if (iw >= N) return;
    float *mine = s_rows + (wave_first + lane) * kPitch;
    if (i < N) {
      float acc[n][3];
#pragma unroll
      for (int k = 0; k < n; ++k) acc[k][0] = acc[k][1] = acc[k][2] = 0.0f;
      const float px = xyz[3 * i], py = xyz[3 * i + 1], pz = xyz[3 * i + 2];
      for (int r = 0; r < world; ++r) {
        const float *blk = rgb_all + (size_t)r * stride;
        const float g0 = blk[3 * (size_t)i], g1 = blk[3 * (size_t)i + 1], g2 = blk[3 * (size_t)i + 2];
        if (g0 == 0.0f && g1 == 0.0f && g2 == 0.0f) continue;
        const float *cp = blk + 3 * (size_t)N;
        float dx, dy, dz, len, Y[n];
        gs::view_dir(px, py, pz, cp[0], cp[1], cp[2], dx, dy, dz, len);
        gs::sh_basis<L>(dx, dy, dz, Y);
#pragma unroll
        for (int k = 0; k < n; ++k) { acc[k][0] += g0 * Y[k]; acc[k][1] += g1 * Y[k]; acc[k][2] += g2 * Y[k]; }
      }
      constexpr int o = 3 - kFirst;  // where the SH block starts inside the staged row
#pragma unroll
      for (int k = 0; k < n; ++k) { mine[o + 3 * k] = acc[k][0]; mine[o + 3 * k + 1] = acc[k][1]; mine[o + 3 * k + 2] = acc[k][2]; }
      if constexpr (kCommon) {
        const float *row = common + (size_t)i * 12;
        mine[0] = row[0]; mine[1] = row[1]; mine[2] = row[2];
#pragma unroll
        for (int k = 3; k < 12; ++k) mine[3 * n + k] = row[k];
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int rows = min(64, N - iw);
    const float *wrows = s_rows + wave_first * kPitch;
    float *dst = full + (size_t)iw * wo + kFirst;
    if constexpr (kCols <= 64) {
      for (int r = 0; r < rows; ++r)
        if (lane < kCols) dst[(size_t)r * wo + lane] = wrows[r * kPitch + lane];
    } else {
      static_assert(kCols <= 128, "row wider than two wave instructions");
      for (int r = 0; r < rows; ++r) {
        dst[(size_t)r * wo + lane] = wrows[r * kPitch + lane];
        if (lane + 64 < kCols) dst[(size_t)r * wo + lane + 64] = wrows[r * kPitch + lane + 64];
      }
    }
  }
}

// Room of the instance buffers, in INSTANCES: the smallest element count of the per-instance arrays, minus the one spare
// slot every user keeps.  Each array is a DeviceBuffer grown by `want + want / 4 + 256` BYTES, so arrays of different
// element size end up with different element counts for the same S (8-byte payloads: 32 entries of slack, 4-byte keys:
// 64, 2-byte masks: 128) -- a bound taken from one array alone is not a bound for the others.  Until r05 the radix
// route bounded its speculative tile_emit_payload_kernel by keys_a's count only, 31 entries more than pay_a holds: a
// forward whose instances outgrew the room wrote 248 bytes past the end of pay_a (test_instance_buffers_grow's radix
// case: a 12 266-byte allocation that ends 22 bytes before its page does; whether the next page is mapped depends on
// what the allocator placed there -- the unexplained abort of r04's suite).
size_t instance_room(const gsplat_context *c) {
  size_t room = c->keys_a.bytes / sizeof(unsigned int);
  room = std::min(room, c->keys_b.bytes / sizeof(unsigned int));
  room = std::min(room, c->pay_a.bytes / sizeof(unsigned long long));
  room = std::min(room, c->pay_b.bytes / sizeof(unsigned long long));
  room = std::min(room, c->sorted.bytes / sizeof(int));
  room = std::min(room, c->blockmasks.bytes / sizeof(unsigned short));
  return room;
}

int reserve_instances(gsplat_context *c, size_t S, int num_tiles, hipStream_t st) {
  int rc;
  const void *pay_before = c->pay_a.ptr, *sorted_before = c->sorted.ptr;
  if ((rc = c->keys_a.reserve((S + 1) * sizeof(unsigned int)))) return rc;
  if ((rc = c->keys_b.reserve((S + 1) * sizeof(unsigned int)))) return rc;
  if ((rc = c->pay_a.reserve((S + 1) * sizeof(unsigned long long)))) return rc;
  if ((rc = c->pay_b.reserve((S + 1) * sizeof(unsigned long long)))) return rc;
  if ((rc = c->sorted.reserve((S + 1) * sizeof(int)))) return rc;
  // Fresh instance buffers start as zeros.  The sparse forward queues its sorts and render_fwd before the host has
  // seen S; when S outgrows the room those kernels run on truncated lists whose slots may not have been written by
  // this forward -- whatever they hold must still be a valid gaussian id (0, or one of an earlier forward).
  // The fill goes on the CALLER's stream: on the NULL stream nothing would order it against the placement and the sorts
  // that a non-blocking stream (a torch side stream) runs right behind it, and the zeros could land on top of them.
  if (c->pay_a.ptr != pay_before) GS_HIP(hipMemsetAsync(c->pay_a.ptr, 0, c->pay_a.bytes, st));
  if (c->sorted.ptr != sorted_before) GS_HIP(hipMemsetAsync(c->sorted.ptr, 0, c->sorted.bytes, st));
  if ((rc = c->blockmasks.reserve((S + 1) * sizeof(unsigned short)))) return rc;
  if ((rc = c->temp.reserve(gs::binning_temp_bytes((size_t)c->max_gaussians, S ? S : 1, num_tiles)))) return rc;
  return GSPLAT_OK;
}

}  // namespace

extern "C" {

int gsplat_packed_gradient_width(int l_max) { return 12 + 3 * (l_max + 1) * (l_max + 1); }
int gsplat_factored_gradient_width(int world_size) { return 12 + 3 * world_size; }

int gsplat_pack_gradients_factored(gsplat_context *c, const gsplat_gradients *grads, int num_gaussians, int rank,
                                   int world_size, float *packed, void *stream) {
  GS_REQUIRE(c && grads, "null argument struct");
  GS_REQUIRE(c->have_forward && num_gaussians == c->N, "does not match the recorded forward");
  GS_REQUIRE(world_size >= 1 && rank >= 0 && rank < world_size, "bad rank / world_size");
  GS_REQUIRE_DEV(packed);
  GS_REQUIRE_DEV(grads->grad_precompute_rgb);  // backward must have been asked for this intermediate
  const long long total = (long long)num_gaussians * (12 + 3 * world_size);
  pack_factored_kernel<<<gs::div_up(total, kBlock), kBlock, 0, (hipStream_t)stream>>>(
      c->mask.as<unsigned char>(), c->rank.as<int>(), num_gaussians, rank, world_size, *grads, packed);
  GS_LAUNCH_CHECK();
  return GSPLAT_OK;
}

int gsplat_unpack_gradients_factored(const float *xyz, const float *campos_all, const float *packed, int l_max,
                                     int num_gaussians, int world_size, float *full, void *stream) {
  GS_REQUIRE_DEV(xyz); GS_REQUIRE_DEV(campos_all); GS_REQUIRE_DEV(packed); GS_REQUIRE_DEV(full);
  GS_REQUIRE(l_max >= 0 && l_max <= 3 && num_gaussians >= 0 && world_size >= 1, "bad sizes");
  if (num_gaussians == 0) return GSPLAT_OK;
  const dim3 g(gs::div_up(num_gaussians, kBlock)), b(kBlock);
  hipStream_t st = (hipStream_t)stream;
  switch (l_max) {
    case 0: unpack_factored_kernel<0><<<g, b, 0, st>>>(xyz, campos_all, num_gaussians, world_size, packed, full); break;
    case 1: unpack_factored_kernel<1><<<g, b, 0, st>>>(xyz, campos_all, num_gaussians, world_size, packed, full); break;
    case 2: unpack_factored_kernel<2><<<g, b, 0, st>>>(xyz, campos_all, num_gaussians, world_size, packed, full); break;
    default: unpack_factored_kernel<3><<<g, b, 0, st>>>(xyz, campos_all, num_gaussians, world_size, packed, full); break;
  }
  GS_LAUNCH_CHECK();
  return GSPLAT_OK;
}

int gsplat_pack_gradients_split_range(gsplat_context *c, const gsplat_gradients *grads, int num_gaussians,
                                      int first_gaussian, int end_gaussian, float *common, float *rgb, void *stream) {
  GS_REQUIRE(c && grads, "null argument struct");
  GS_REQUIRE(c->have_forward && num_gaussians == c->N, "does not match the recorded forward");
  GS_REQUIRE(0 <= first_gaussian && first_gaussian <= end_gaussian && end_gaussian <= num_gaussians, "bad gaussian range");
  GS_REQUIRE_DEV(common);
  if (rgb) {
    GS_REQUIRE_DEV(rgb);
    GS_REQUIRE_DEV(grads->grad_precompute_rgb);  // backward must have been asked for this intermediate
  }
  if (end_gaussian == first_gaussian) return GSPLAT_OK;
  pack_split_kernel<<<gs::div_up((long long)(end_gaussian - first_gaussian), kBlock), kBlock, 0, (hipStream_t)stream>>>(
      c->mask.as<unsigned char>(), c->rank.as<int>(), first_gaussian, end_gaussian, *grads, common, rgb);
  GS_LAUNCH_CHECK();
  return GSPLAT_OK;
}

int gsplat_pack_gradients_split(gsplat_context *c, const gsplat_gradients *grads, int num_gaussians, float *common,
                                float *rgb, void *stream) {
  return gsplat_pack_gradients_split_range(c, grads, num_gaussians, 0, num_gaussians, common, rgb, stream);
}

int gsplat_pack_uv_grad_norm(gsplat_context *c, const gsplat_gradients *grads, int num_gaussians, float *uv_norm,
                             void *stream) {
  GS_REQUIRE(c && grads, "null argument struct");
  GS_REQUIRE(c->have_forward && num_gaussians == c->N, "does not match the recorded forward");
  GS_REQUIRE_DEV(uv_norm);
  GS_REQUIRE_DEV(grads->grad_uv);  // backward must have been asked for this intermediate
  pack_uv_norm_kernel<<<gs::div_up((long long)num_gaussians, kBlock), kBlock, 0, (hipStream_t)stream>>>(
      c->mask.as<unsigned char>(), c->rank.as<int>(), num_gaussians, grads->grad_uv, uv_norm);
  GS_LAUNCH_CHECK();
  return GSPLAT_OK;
}

int gsplat_unpack_gradients_split(const float *xyz, const float *common, const float *rgb_all, size_t rank_stride,
                                  int l_max, int num_gaussians, int world_size, float *full, void *stream) {
  GS_REQUIRE_DEV(full);
  GS_REQUIRE(common != nullptr || rgb_all != nullptr, "nothing to unpack");
  GS_REQUIRE(l_max >= 0 && l_max <= 3 && num_gaussians >= 0 && world_size >= 1, "bad sizes");
  if (common) GS_REQUIRE_DEV(common);
  if (rgb_all) {
    GS_REQUIRE_DEV(rgb_all); GS_REQUIRE_DEV(xyz);
    GS_REQUIRE(rank_stride >= 3 * (size_t)num_gaussians + 3, "rank_stride must cover [N,3] g_rgb + campos[3]");
  }
  if (num_gaussians == 0) return GSPLAT_OK;
  const dim3 g(gs::div_up(num_gaussians, kBlock)), b(kBlock);
  hipStream_t st = (hipStream_t)stream;
#define GS_UNPACK(LL)                                                                                                  \
  do {                                                                                                                 \
    if (common && rgb_all)                                                                                             \
      unpack_split_kernel<LL, true, true><<<g, b, 0, st>>>(xyz, num_gaussians, world_size, common, rgb_all, rank_stride, full);  \
    else if (rgb_all)                                                                                                  \
      unpack_split_kernel<LL, true, false><<<g, b, 0, st>>>(xyz, num_gaussians, world_size, nullptr, rgb_all, rank_stride, full); \
    else                                                                                                               \
      unpack_split_kernel<LL, false, true><<<g, b, 0, st>>>(nullptr, num_gaussians, world_size, common, nullptr, 0, full);        \
  } while (0)
  switch (l_max) {
    case 0: GS_UNPACK(0); break;
    case 1: GS_UNPACK(1); break;
    case 2: GS_UNPACK(2); break;
    default: GS_UNPACK(3); break;
  }
#undef GS_UNPACK
  GS_LAUNCH_CHECK();
  return GSPLAT_OK;
}

int gsplat_context_create(gsplat_context **out, int max_gaussians, int max_width, int max_height) {
  GS_REQUIRE(out != nullptr, "out is null");
  GS_REQUIRE(max_gaussians > 0 && max_width > 0 && max_height > 0, "capacities must be positive");
  gsplat_context *c = new (std::nothrow) gsplat_context();
  GS_REQUIRE(c != nullptr, "out of host memory");
  c->max_gaussians = max_gaussians; c->max_width = max_width; c->max_height = max_height;
  const size_t N = (size_t)max_gaussians, P = (size_t)max_width * max_height;
  const size_t T = (size_t)((max_width + 15) / 16) * ((max_height + 15) / 16);
  int rc = GSPLAT_OK;
  {
    gs::DeviceBuffer *outs[13];
    c->forward_outputs(outs);
    for (gs::DeviceBuffer *b : outs) b->pooled = true;
  }
  auto R = [&](gs::DeviceBuffer &b, size_t bytes) { if (!rc) rc = b.reserve(bytes); };
  R(c->mask, N + 16); R(c->counters, 512 + gs::kBinBlocks * 4 + 64); R(c->rank, (N + 1) * 4); R(c->xyz_c_all, N * 12); R(c->uv_all, N * 8);
  R(c->c2g, N * 4); R(c->xyz_c, N * 12); R(c->uv, N * 8); R(c->sigma, N * 24); R(c->conic, N * 12); R(c->J, N * 24);
  R(c->rgb, N * 12); R(c->radius, N * 16); R(c->recs, N * 48); R(c->counts, (N + 1) * 4); R(c->offsets, (N + 1) * 4);
  R(c->grad_rows, N * 64); R(c->hitmask, N * 8);
  R(c->ranges, (T + 1) * 4); R(c->image, P * 12); R(c->T_px, P * 4); R(c->n_px, P * 4);
  R(c->tile_tops, (T + 8) * 4); R(c->tile_order, (T + 8) * 4);
  if (!rc) {
    size_t sb1 = 0;
    (void)rocprim::exclusive_scan(nullptr, sb1, (int *)nullptr, (int *)nullptr, 0, N + 1, rocprim::plus<int>(), (hipStream_t)0);
    const size_t sb2 = gs::binning_temp_bytes(N, 4 * N, (int)T);
    rc = c->temp.reserve(sb1 > sb2 ? sb1 : sb2);
  }
  if (!rc && hipMemsetAsync(c->counters.ptr, 0, c->counters.bytes, (hipStream_t)0) != hipSuccess) rc = GSPLAT_ERR_HIP;
  if (!rc) rc = reserve_instances(c, 4 * N, (int)T, (hipStream_t)0);
  if (!rc && hipStreamSynchronize((hipStream_t)0) != hipSuccess) {  // the fills above: done before any stream uses the context
    gs::set_error("gsplat_context_create: hipStreamSynchronize failed");
    rc = GSPLAT_ERR_HIP;
  }
  if (!rc) {
    void *h = nullptr, *d = nullptr;
    if (hipHostMalloc(&h, 256, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess ||
        hipHostGetDevicePointer(&d, h, 0) != hipSuccess) {
      gs::set_error("gsplat_context_create: could not map the count record");
      rc = GSPLAT_ERR_HIP;
    } else {
      memset(h, 0, 256);  // words 0..4: the forward's record; 8..15: two slots of four tagged figures (queue_tail)
      c->h_pub = (volatile unsigned long long *)h;
      c->d_pub = (unsigned long long *)d;
    }
  }
  if (!rc && hipHostMalloc((void **)&c->h_words, 1024, hipHostMallocDefault) != hipSuccess) {
    gs::set_error("gsplat_context_create: hipHostMalloc failed");
    rc = GSPLAT_ERR_HIP;
  }
  if (rc) { c->release(); delete c; return rc; }
  *out = c;
  return GSPLAT_OK;
}

int gsplat_context_destroy(gsplat_context *ctx) {
  if (!ctx) return GSPLAT_OK;
  (void)hipDeviceSynchronize();
  ctx->release();
  delete ctx;
  // the context's pooled output arrays went back to the pool's idle lists.  r06: they stay there for the next context /
  // the shim's vectors of this device unless more than a gigabyte is idle on it (gsplat_pool_trim) -- r05 called the
  // process-wide gsplat_pool_release here, which synchronised every device and emptied the caches of live peers
  (void)gsplat_pool_trim((size_t)1 << 30);
  return GSPLAT_OK;
}

size_t gsplat_context_bytes(const gsplat_context *ctx) { return ctx ? ctx->bytes() : 0; }

int gsplat_rasterize_image(gsplat_context *c, const gsplat_gaussians *g, const gsplat_camera *cam,
                           const gsplat_raster_config *cfg, float bg_color, int l_max, gsplat_forward_view *out,
                           void *stream) {
  GS_REQUIRE(c && g && cam && cfg, "null argument struct");
  GS_REQUIRE(l_max >= 0 && l_max <= 3, "l_max must be 0..3");  // cuda/raster.cu:58-60
  const int N = g->num_gaussians, W = cam->width, H = cam->height;
  GS_REQUIRE(N > 0, "num_gaussians must be positive");
  if (N > c->max_gaussians || W > c->max_width || H > c->max_height || W <= 0 || H <= 0) {
    gs::set_error("gsplat_rasterize_image: %d gaussians / %dx%d exceed the context capacity %d / %dx%d", N, W, H,
                  c->max_gaussians, c->max_width, c->max_height);
    return GSPLAT_ERR_CAPACITY;
  }
  GS_REQUIRE_DEV(g->xyz); GS_REQUIRE_DEV(g->rgb); GS_REQUIRE_DEV(g->opacity); GS_REQUIRE_DEV(g->scale);
  GS_REQUIRE_DEV(g->quaternion); GS_REQUIRE_DEV(cam->view); GS_REQUIRE_DEV(cam->proj);
  if (l_max > 0) GS_REQUIRE_DEV(g->sh);
  GS_REQUIRE(((uintptr_t)g->quaternion & 15) == 0, "quaternion must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  c->have_forward = false;
  c->rows_ready = false;
  c->order_ready = false;
  gs::pool_unwatch(&c->last_mask);
  c->last_mask = nullptr;  // rank[] and compact_to_global are about to be overwritten
  {  // outputs the caller took over (gsplat_context_detach_forward_outputs) come back from the pool, at their old sizes
    gs::DeviceBuffer *outs[13];
    c->forward_outputs(outs);
    for (gs::DeviceBuffer *b : outs)
      if (b->ptr == nullptr) {
        const int r = b->reserve_again(st);  // (ordered behind the stream that returned the block, if another)
        if (r) return r;
        // a fresh `sorted` must hold valid gaussian ids wherever the speculative tail may read it (reserve_instances)
        if (b == &c->sorted) GS_HIP(hipMemsetAsync(b->ptr, 0, b->bytes, st));
      }
  }
  const int ntx = (W + 15) / 16, nty = (H + 15) / 16, num_tiles = ntx * nty;
  const float fx = cam->focal_x, fy = cam->focal_y;
  const float tan_fovx = (float)W / (2.0f * fx), tan_fovy = (float)H / (2.0f * fy);  // cuda/raster.cu:92-93

  if (c->timing) {  // next timing slot; its events are >= kSlots forwards old, hence complete
    c->slot = (int)(c->fwd_calls % gsplat_context::kSlots);
    c->harvest(c->slot);
  }
  c->fwd_calls++;
  GS_REQUIRE(N <= (64 << 20), "more than 64 Mi gaussians: the cull's per-slice ballots would not fit its LDS");
  c->mark(0, false, st);
  // A view that culled a fifth of the scene or more last time gets the compacted walk in preprocess_kernel (the previous
  // forward of this context decides: views of a training run look alike; the first call walks all indices).
  const bool compact = c->N == N && c->M > 0 && (long long)c->M * 5 < (long long)N * 4;
  int rc = GSPLAT_OK;
  if (compact && (rc = c->kept.reserve((size_t)c->max_gaussians * sizeof(int)))) return rc;
  // the colour kernel beside the geometry kernel (pre_split 2): its stream and the two events, made on first use
  const bool beside = c->pre_split == 2 && l_max > 0;
  if (beside) {
    if ((rc = c->chunk_first.reserve(((size_t)gs::bin_chunks(c->max_gaussians) + gs::kBinThreads / 64 + 1) * sizeof(int)))) return rc;
    if (!c->pre_side) {
      int lo_prio = 0, hi_prio = 0;
      GS_HIP(hipDeviceGetStreamPriorityRange(&lo_prio, &hi_prio));  // (numerically largest = lowest priority)
      GS_HIP(hipStreamCreateWithPriority(&c->pre_side, hipStreamNonBlocking, lo_prio));
      GS_HIP(hipEventCreateWithFlags(&c->ev_pre_fork, hipEventDisableTiming));
      GS_HIP(hipEventCreateWithFlags(&c->ev_pre_join, hipEventDisableTiming));
    }
  }
  {
    // LDS of the cull: one ballot + one count per (trip, wave) of the largest slice (a run of whole 64-entry chunks)
    const size_t slice_max = ((size_t)gs::bin_chunks(N) / gs::kBinBlocks + 2) * gs::kBinChunk;
    const size_t trips = (slice_max + gs::kBinThreads - 1) / gs::kBinThreads + 1;
    project_cull_kernel<<<gs::kBinBlocks, gs::kBinThreads, trips * (gs::kBinThreads / 64) * 12, st>>>(
        g->xyz, cam->view, cam->proj, N, W, H, cfg->near_thresh, cfg->cull_mask_padding,
        c->lean ? nullptr : c->xyz_c_all.as<float>(), c->lean ? nullptr : c->uv_all.as<float>(),
        c->mask.as<unsigned char>(), c->rank.as<int>(), c->slice_counts(), c->pair_counters(),
        compact ? c->kept.as<int>() : nullptr, beside && !compact ? c->chunk_first.as<int>() : nullptr);
    GS_LAUNCH_CHECK();
  }
  c->mark(0, true, st);
  c->mark(1, false, st);
  const bool ro = c->render_only;
  const bool mid = !ro && !c->lean;  // Sigma, J, conic, colour: stored for the caller / the stand-alone backward operators
  PreOut po = {c->c2g.as<int>(), c->xyz_c.as<float>(), c->uv.as<float>(), mid ? c->sigma.as<float>() : nullptr,
               mid ? c->conic.as<float>() : nullptr, mid ? c->J.as<float>() : nullptr, mid ? c->rgb.as<float>() : nullptr,
               c->radius.as<float>(), c->recs.as<float4>(), c->counts.as<int>(),
               c->hitmask.as<unsigned long long>(), c->pair_counters()};
  // Two binning routes, both exact for any scene; the choice only affects speed, so it follows the LAST forward's
  // density (the first call starts sparse): sparse = LDS counting sort + per-tile depth sort, dense (more than
  // ~768 list entries per tile) or very large tile grids = stable radix sorts (gs_binning.hip).
  const bool want_dense = c->forced_route == 2 || (c->forced_route == 0 && c->dense_route);
  const bool sparse = !want_dense && gs::binning_supports_counting_sort(num_tiles);
  int *bin_table = nullptr;
  if (sparse) {
    if ((rc = c->bin_table.reserve(gs::binning_table_bytes(num_tiles)))) return rc;
    bin_table = c->bin_table.as<int>();
  }
  const size_t hist_bytes = sparse ? (size_t)num_tiles * sizeof(int) : 0;
#define GS_PRE3(LL, MID, CMP)                                                                                          \
  preprocess_kernel<LL, MID, CMP><<<gs::kBinBlocks, gs::kBinThreads, hist_bytes, st>>>(                                \
      *g, cam->view, c->mask.as<unsigned char>(), c->rank.as<int>(), c->slice_counts(), c->kept.as<int>(),              \
      cam->proj, W, H, fx, fy, tan_fovx, tan_fovy, cfg->mh_dist, cam->campos[0], cam->campos[1], cam->campos[2], ntx, nty, po, bin_table)
#define GS_PRE(LL)                                                                                                     \
  do {                                                                                                                 \
    if (mid) { if (compact) GS_PRE3(LL, true, true); else GS_PRE3(LL, true, false); }                                  \
    else { if (compact) GS_PRE3(LL, false, true); else GS_PRE3(LL, false, false); }                                    \
  } while (0)
  bool join_pending = false;
  if (c->pre_split) {
    // r06: colour and geometry as two kernels (see sh_colour_kernel)
    const bool seq = l_max > 0 && c->pre_split == 1;  // one behind the other: the colour travels through c->rgb
    if (seq) po.rgb = c->rgb.as<float>();
    hipStream_t cst = st;
    if (beside) {  // the colour kernel's stream starts where the cull has finished
      GS_HIP(hipEventRecord(c->ev_pre_fork, st));
      GS_HIP(hipStreamWaitEvent(c->pre_side, c->ev_pre_fork, 0));
      cst = c->pre_side;
    }
    auto launch_colour = [&]() -> int {
      if (l_max == 0) return GSPLAT_OK;
      const dim3 cg(gs::div_up(gs::bin_chunks(N), kBlock / 64)), cb(kBlock);
      const int *rk = beside ? nullptr : c->rank.as<int>();
      float *rgb_to = beside ? (mid ? c->rgb.as<float>() : nullptr) : c->rgb.as<float>();
      float4 *rec_to = beside ? c->recs.as<float4>() : nullptr;
#define GS_SHC(LL)                                                                                                     \
  do {                                                                                                                 \
    if (compact)                                                                                                       \
      sh_colour_kernel<LL, true><<<cg, cb, 0, cst>>>(*g, c->mask.as<unsigned char>(), rk, nullptr, c->slice_counts(),   \
                                                     c->kept.as<int>(), cam->campos[0], cam->campos[1], cam->campos[2], rgb_to, rec_to); \
    else                                                                                                               \
      sh_colour_kernel<LL, false><<<cg, cb, 0, cst>>>(*g, c->mask.as<unsigned char>(), rk, c->chunk_first.as<int>(),    \
                                                      c->slice_counts(), nullptr, cam->campos[0], cam->campos[1], cam->campos[2], rgb_to, rec_to); \
  } while (0)
      switch (l_max) {
        case 1: GS_SHC(1); break;
        case 2: GS_SHC(2); break;
        default: GS_SHC(3); break;
      }
#undef GS_SHC
      GS_LAUNCH_CHECK();
      return GSPLAT_OK;
    };
    if (seq && (rc = launch_colour())) return rc;
#define GS_GEOM3(COL, MID, CMP)                                                                                        \
  preprocess_geom_kernel<COL, MID, CMP><<<gs::kBinBlocks, gs::kBinThreads, hist_bytes, st>>>(                          \
      *g, cam->view, c->mask.as<unsigned char>(), c->rank.as<int>(), c->slice_counts(), c->kept.as<int>(),              \
      cam->proj, W, H, fx, fy, tan_fovx, tan_fovy, cfg->mh_dist, cam->campos[0], cam->campos[1], cam->campos[2], ntx, nty, po, bin_table)
#define GS_GEOM(COL)                                                                                                   \
  do {                                                                                                                 \
    if (mid) { if (compact) GS_GEOM3(COL, true, true); else GS_GEOM3(COL, true, false); }                              \
    else { if (compact) GS_GEOM3(COL, false, true); else GS_GEOM3(COL, false, false); }                                \
  } while (0)
    // (the geometry kernel is queued FIRST: its 256 workgroups want a CU each, the colour kernel's take what is left)
    if (l_max == 0) GS_GEOM(0); else if (beside) GS_GEOM(2); else GS_GEOM(1);
#undef GS_GEOM
#undef GS_GEOM3
    GS_LAUNCH_CHECK();
    if (beside) {
      if ((rc = launch_colour())) return rc;
      GS_HIP(hipEventRecord(c->ev_pre_join, c->pre_side));
      if (gs_pre_join_late()) join_pending = true;
      else GS_HIP(hipStreamWaitEvent(st, c->ev_pre_join, 0));
    }
  } else {
  switch (l_max) {
    case 0: GS_PRE(0); break;
    case 1: GS_PRE(1); break;
    case 2: GS_PRE(2); break;
    default: GS_PRE(3); break;
  }
  }
#undef GS_PRE
#undef GS_PRE3
  GS_LAUNCH_CHECK();
  size_t inst_cap = 0;
  // the one host read-back of the forward: M, S (and the candidate count)
  const unsigned long long ticket = ++c->ticket;
  size_t spec_cap = 0;  // sparse route: room of the instance buffers, what the kernels queued before the wait may use
  if (sparse) {
    if (instance_room(c) < 2 && (rc = reserve_instances(c, 4 * (size_t)N, num_tiles, st))) return rc;
    spec_cap = instance_room(c) - 1;
    c->mark(1, true, st);
    c->mark(2, false, st);
    rc = gs::binning_offsets(ntx, nty, bin_table, c->keys_a.as<int>(), st);
    if (rc) return rc;
  } else {
    rc = gs::scan_counts(N, c->counts.as<int>(), c->offsets.as<int>(), c->temp.ptr, c->temp.bytes, st);
    if (rc) return rc;
    c->mark(1, true, st);
    publish_counts_kernel<<<1, 64, 0, st>>>(c->rank.as<int>() + N, c->offsets.as<int>() + N, c->pair_counters(), c->d_pub,
                                           ticket);
    GS_LAUNCH_CHECK();
  }
  if (!sparse) {
    // Emit does not need the totals on the host, only room for its writes: launch it bounded by the buffers'
    // capacity and sleep on the read-back while it runs (the reference blocks five times per forward, GPU idle).
    // (the bound is the room of ALL instance arrays: see instance_room)
    if (instance_room(c) < 2 && (rc = reserve_instances(c, 4 * (size_t)N, num_tiles, st))) return rc;
    inst_cap = instance_room(c) - 1;
    c->mark(2, false, st);
    rc = gs::launch_tile_emit(c->uv.as<float>(), c->xyz_c.as<float>(), c->radius.as<float>(), ntx, nty, N,
                              c->mask.as<unsigned char>(), c->rank.as<int>(), c->offsets.as<int>(),
                              c->hitmask.as<unsigned long long>(), (long long)inst_cap,
                              c->keys_a.as<unsigned int>(), c->pay_a.as<unsigned long long>(), st);
    if (rc) return rc;
  }
  // Sparse route: nothing behind the counts needs them on the HOST -- the placement only needs room for its writes,
  // the per-tile sorts read `ranges` on the device, the compositing nothing at all.  So scatter, sorts and render_fwd
  // are queued right here, bounded by the buffers' capacity, and the host sleeps on the read-back while they run (r01
  // launched them after the wake-up: the GPU idled for the round trip, ~13 us per forward).  Which long-list kernels
  // to queue follows the previous forward's longest list; afterwards the record is checked, and a forward whose
  // instances outgrew the buffers or whose longest list needed a kernel that was not queued is redone from the
  // placement on (results are unaffected: every launch overwrites).
  // (r06: class 0 -- no list beyond the wave kernel's 1024 entries and every depth key an ordinary positive float: the
  // hand-over kernel is not queued at all, gs_binning.hip sort_tiles_by_depth)
  const bool keys_ok = cfg->near_thresh >= 1e-30f;
  auto list_class = [keys_ok](long long longest) {  // which of the workgroup sort kernels a list of that length needs
    return longest > 8 * 1024 ? 4 : longest > 4 * 1024 ? 3 : longest > 2 * 1024 ? 2 : (longest > 1024 || !keys_ok) ? 1 : 0;
  };
  bool segmented_this_forward = false;
  auto queue_tail = [&](size_t cap, long long longest_hint, bool publish) -> int {  // the placement publishes the record
    if (cap + 1 > instance_room(c)) {  // every kernel below indexes the instance arrays up to `cap` (inclusive: the spare slot)
      gs::set_error("gsplat_rasterize_image: internal: %zu instances queued into room for %zu", cap, instance_room(c));
      return GSPLAT_ERR_CAPACITY;
    }
    if ((longest_hint < 0 || longest_hint > 2048) && !c->fork.ready) {  // lists beyond two register-sorted runs: see SortFork
      const int fr = c->fork.create();
      if (fr) return fr;
    }
    int r = gs::binning_scatter_and_sort(c->uv.as<float>(), c->xyz_c.as<float>(), c->radius.as<float>(),
                                         c->hitmask.as<unsigned long long>(), c->rank.as<int>(), N, ntx, nty,
                                         c->bin_table.as<int>(), c->ranges.as<int>(), cap, c->pay_a.as<unsigned long long>(),
                                         c->keys_a.as<int>(), c->sorted.as<int>(), longest_hint, c->rank.as<int>() + N,
                                         c->pair_counters(), publish ? c->d_pub : nullptr, ticket, st, &c->fork, compact,
                                         keys_ok);
    if (r) return r;
    c->mark(2, true, st);
    c->mark(4, false, st);
    // Heaviest-first for the backward only where the tiles differ enough in work to pay for it: the order breaks up the
    // XCD runs' spatial adjacency (neighbouring tiles share records in one L2), which on the uniform benchmark scene cost
    // +39 % HBM traffic in render_bwd (594 instead of 428 MB, profiles/r04_pmc_summary_all_tiles_ordered.json) for 2 % of
    // its time; on a skewed scene (garden-shaped workload: longest list 7x the average) it is worth 6.5 %.  Decided like
    // the rest of the queued tail by the previous forward's figures.
    const bool ordered = !ro && gs::tile_order_supported(num_tiles) && !gs_no_tile_order() && c->last_longest > 0 &&
                         c->S > 0 && c->last_longest * (long long)num_tiles > 3ll * (long long)c->S;
    // (The forward itself keeps the plain XCD-run order: dealt heaviest first by list length it was 10-14 us SLOWER on
    // the garden-shaped workload, profiles/r04_tile_order_ab.txt -- the list length says little about a dense tile's
    // forward, whose pixels saturate early, and neighbouring tiles no longer run side by side on one XCD's L2.)
    // Lists beyond kSegSplitMin are split for the backward (gs_render.h: TileSegments) -- decided, like the order, by the
    // previous forward: its longest list says whether there is anything to split; the room for extra blocks follows what
    // the tiles of the previous forward asked for (a list that does not fit stays whole).
    // r06: the figures the segment kernels publish -- how uneven the tiles' work is, how many segments the
    // lists asked for -- decide whether THIS forward splits and how much room it reserves, and a split forward sums
    // per-segment partials where an unsplit one runs one fma chain: the decision must not follow host / GPU timing.
    // The kernels write each figure as {ticket << 32 | value} into the slot of their forward's ticket parity.  This
    // forward (ticket t) takes the slot of ticket t - 2: the host has seen the record of forward t - 1, which
    // bin_scatter_kernel published BEHIND everything forward t - 2 queued on the stream, so that slot is complete -- and
    // nobody writes it again before this forward's own kernels, queued below, run.  (The figures of forward t - 1 may or
    // may not have landed yet: they are never looked at.)  A slot that does not carry ticket t - 2 -- the forward two back
    // published nothing -- leaves the figures as they were.
    {
      const unsigned int want_tag = (unsigned int)((ticket - 2) & 0xFFFFFFFFull);
      const volatile unsigned long long *slot = c->h_pub + 8 + 4 * (ticket & 1ull);
      const unsigned long long w0 = __atomic_load_n(&slot[0], __ATOMIC_RELAXED), w1 = __atomic_load_n(&slot[1], __ATOMIC_RELAXED);
      const unsigned long long w2 = __atomic_load_n(&slot[2], __ATOMIC_RELAXED), w3 = __atomic_load_n(&slot[3], __ATOMIC_RELAXED);
      if (ticket >= 3 && (unsigned int)(w0 >> 32) == want_tag && (unsigned int)(w1 >> 32) == want_tag) {
        c->fig_max = (long long)(int)(unsigned int)w0;
        c->fig_sum = (long long)(int)(unsigned int)w1;
      }
      if (ticket >= 3 && (unsigned int)(w2 >> 32) == want_tag) c->fig_asked_bwd = (long long)(int)(unsigned int)w2;
      if (ticket >= 3 && (unsigned int)(w3 >> 32) == want_tag) c->fig_asked_fwd = (long long)(int)(unsigned int)w3;
    }
    unsigned long long *d_slot = c->d_pub + 8 + 4 * (ticket & 1ull);  // where THIS forward's kernels publish
    const unsigned int my_tag = (unsigned int)(ticket & 0xFFFFFFFFull);
    gs::TileSegments seg = {nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr, nullptr, my_tag};
    const bool split = !ro && !gs_no_segments() && c->last_longest > gs::kSegSplitMin && num_tiles <= 16384;  // (the table kernels' reach)
    if (split) {
      const size_t slots = cap / gs::kSegEntries + 2;  // (gs_render.h: segment_slot)
      const size_t asked = (size_t)c->fig_asked_bwd;
      const size_t want = std::min(slots, (asked + asked / 2 + 256 + 7) & ~(size_t)7);
      if ((r = c->seg_first.reserve(((size_t)num_tiles + 8) * 4))) return r;
      if ((r = c->seg_extra.reserve((want + 2) * sizeof(int2)))) return r;  // [want]: the count
      if ((r = c->seg_chk.reserve(slots * 256 * sizeof(float4)))) return r;
      seg = {c->seg_first.as<int>(), c->seg_extra.as<int2>(), reinterpret_cast<int *>(c->seg_extra.as<int2>() + want),
             c->seg_chk.as<float4>(), c->image.as<float>(), (int)want, d_slot + 2, nullptr, my_tag};
    }
    // the tiles' largest stop indices of this forward, for the next one's decision below
    const bool figures = !gs_no_fwd_segments() && c->last_longest > gs::kSegSplitMin && num_tiles <= 16384;
    if (figures) seg.stats = d_slot;
    // ... and for the forward itself (gs_render.h: FwdSegments): every segment of a long list a block of its own
    gs::FwdSegments fs = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, nullptr, my_tag, nullptr, 0, 0};
    // Only where the tiles' work is uneven enough for ONE list to set the launch's duration: the previous forward's longest
    // chain (the largest stop index of any tile) against the work per resident workgroup (the sum over the tiles / 2048).
    // A throughput-bound scene gains nothing from the split and pays for its table, its combine pass and the product passes
    // (garden-shaped synthetic scene 0.197 -> 0.271 ms, dense4m 0.183 -> 0.26 when split regardless).
    const long long top_max = c->fig_max, top_sum = c->fig_sum;
    const double gate = c->fseg_gate >= 0.0 ? c->fseg_gate : gs_fwd_segments_gate();
    const bool fsplit = figures && top_sum > 0 && top_max * 2048ll > (long long)(gate * (double)top_sum);
    if (fsplit) {
      const size_t most = cap / gs::kSegEntries + (size_t)num_tiles + 8;  // sum of ceil(len / kSegEntries) over any lists
      const size_t asked = (size_t)c->fig_asked_fwd;
      const size_t want = std::min(most, asked + asked / 4 + 512) + 7 & ~(size_t)7;
      if ((r = c->fseg_first.reserve(((size_t)num_tiles + 8 + 136) * 4))) return r;  // ranks | layer bases
      if ((r = c->fseg_blocks.reserve((want + 2) * sizeof(int2)))) return r;  // [want]: the count
      if ((r = c->fseg_gran.reserve(2 * want * 256 * sizeof(unsigned long long)))) return r;  // t products | final Ts
      if ((r = c->fseg_part.reserve(want * 256 * sizeof(float4)))) return r;
      if ((r = c->fseg_stop.reserve(want * 256 * sizeof(int)))) return r;
      if (c->fseg_gran.ptr != c->fseg_gran_zeroed || c->fseg_gran.bytes != c->fseg_gran_zeroed_bytes) {
        GS_HIP(hipMemsetAsync(c->fseg_gran.ptr, 0, c->fseg_gran.bytes, st));  // tags of no epoch
        c->fseg_gran_zeroed = c->fseg_gran.ptr;
        c->fseg_gran_zeroed_bytes = c->fseg_gran.bytes;
      }
      if (++c->fseg_epoch == 0) c->fseg_epoch = 1;
      fs = {c->fseg_first.as<int>(), c->fseg_first.as<int>() + num_tiles + 8, c->fseg_blocks.as<int2>(),
            reinterpret_cast<int *>(c->fseg_blocks.as<int2>() + want),
            c->fseg_gran.as<unsigned long long>(), c->fseg_part.as<float4>(), c->fseg_stop.as<int>(), (int)want,
            c->fseg_epoch, d_slot + 3, my_tag, c->fseg_fallbacks(), c->fseg_poll_budget, c->fseg_thin_layer};
      if ((r = gs::launch_fwd_segments_table(c->ranges.as<int>(), num_tiles, fs, st))) return r;
      segmented_this_forward = true;  // (counted once per forward below: a redone tail comes through here twice)
    }
    if (join_pending) GS_HIP(hipStreamWaitEvent(st, c->ev_pre_join, 0));  // (late join: the records' colour is first read here)
    r = gs::launch_render_fwd(c->recs.as<float4>(), nullptr, c->sorted.as<int>(), c->ranges.as<int>(), W, H, bg_color,
                              c->n_px.as<int>(), c->T_px.as<float>(), c->image.as<float>(), st,
                              c->rows_zeroed ? c->grad_rows.as<float4>() : nullptr, (long long)N * 4,  // M <= N is not known here yet
                              ro ? nullptr : c->blockmasks.as<unsigned short>(), nullptr,
                              (ordered || split || figures) ? c->tile_tops.as<int>() : nullptr, split ? &seg : nullptr,
                              fsplit ? &fs : nullptr);
    if (r) return r;
    c->seg_ready = split;
    c->seg_cap = seg.extra_cap;
    c->mark(4, true, st);
    // the backward's tile order, behind the forward: nothing waits for it until the loss has produced dL/dimage
    if (ordered && (r = gs::launch_tile_order(c->tile_tops.as<int>(), nullptr, num_tiles, c->tile_order.as<int>(), st))) return r;
    c->order_ready = ordered;
    // ... and which segments of the long lists get a block of their own
    if ((split || figures) && (r = gs::launch_tile_segments(c->ranges.as<int>(), c->tile_tops.as<int>(), num_tiles, seg, st))) return r;
    return GSPLAT_OK;
  };
  // once a backward has been seen, the forward clears the gradient rows on the side (see render_fwd_kernel)
  if (c->rows_zeroed) c->backward_seen = false;  // the last forward's cleared rows were never used: rendering only
  c->rows_zeroed = c->backward_seen && !ro;
  long long spec_hint = -1;
  if (sparse) {
    // `ranges` on the device are clamped to spec_cap (bin_scatter_kernel): whatever S turns out to be, the queued kernels
    // stay inside the buffers
    spec_hint = c->last_longest >= 0 ? c->last_longest + c->last_longest / 2 + 64 : -1;  // unknown: every kernel
    if ((rc = queue_tail(spec_cap, spec_hint, true))) return rc;
  }
  {
    // Poll the mapped record; every few hundred polls ask the runtime about the stream, which both keeps its
    // submission path moving and tells us when everything queued so far has drained.
    volatile unsigned long long *pub = c->h_pub;
    auto arrived = [&]() {
      for (int k = 0; k < gs::kRecordWords; ++k)
        if ((__atomic_load_n(&pub[k], __ATOMIC_RELAXED) & 0xFFFFFFFFull) != (ticket & 0xFFFFFFFFull)) return false;
      __atomic_thread_fence(__ATOMIC_ACQUIRE);
      return true;
    };
    long long polls = 0;
    while (!arrived()) {
      if ((++polls & 255) == 0) {
        const hipError_t q = hipStreamQuery(st);
        if (q == hipSuccess) {  // stream drained: the record must be there now
          if (arrived()) break;
          gs::set_error("gsplat_rasterize_image: the count record never arrived");
          return GSPLAT_ERR_HIP;
        }
        if (q != hipErrorNotReady) {
          gs::set_error("gsplat_rasterize_image: %s while waiting for the counts", hipGetErrorString(q));
          return GSPLAT_ERR_HIP;
        }
      }
      __builtin_ia32_pause();
    }
  }
  const int M = (int)(unsigned int)(c->h_pub[0] >> 32);
  const size_t S = (size_t)(c->h_pub[1] >> 32);
  const unsigned long long pairs = (c->h_pub[2] >> 32) | (c->h_pub[3] & 0xFFFFFFFF00000000ull);
  if (M == 0) {
    gs::set_error("gsplat_rasterize_image: no gaussians in view");  // cuda/raster.cu:38-41
    return GSPLAT_ERR_NO_VISIBLE;
  }
  const long long longest = sparse ? (long long)(c->h_pub[4] >> 32) : -1;
  c->dense_route = gs::binning_next_route_is_radix(sparse, S, num_tiles, longest);
  c->last_longest = longest;
  const bool emitted = S <= inst_cap;  // dense route: else grow the instance buffers (synchronises) and emit again
  c->n_forwards++;
  if (compact) c->n_compact_walks++;
  if (sparse) {
    const bool fits = S <= spec_cap;
    if (!fits || !(spec_hint < 0 || list_class(longest) <= list_class(spec_hint))) {
      c->n_tail_redone++;
      if (!fits) c->n_instance_growths++;
      // grow (synchronises); the placement queued below writes the true ranges
      if (!fits && (rc = reserve_instances(c, S + S / 4, num_tiles, st))) return rc;
      // the long-tile counter lives at the head of keys_a: zeroed by bin_offsets, then used by the queued sorts
      GS_HIP(hipMemsetAsync(c->keys_a.ptr, 0, sizeof(int), st));
      if ((rc = queue_tail(S, longest, false))) return rc;
    }
  } else {
    if ((rc = reserve_instances(c, S, num_tiles, st))) return rc;
    if (S + 1 > instance_room(c)) {
      gs::set_error("gsplat_rasterize_image: internal: %zu instances sorted in room for %zu", S, instance_room(c));
      return GSPLAT_ERR_CAPACITY;
    }
    rc = gs::emit_sort_ranges(c->uv.as<float>(), c->xyz_c.as<float>(), c->radius.as<float>(), ntx, nty, N,
                              c->mask.as<unsigned char>(), c->rank.as<int>(), c->offsets.as<int>(), S,
                              c->keys_a.as<unsigned int>(), c->keys_b.as<unsigned int>(),
                              c->pay_a.as<unsigned long long>(), c->pay_b.as<unsigned long long>(),
                              c->sorted.as<int>(), c->ranges.as<int>(), c->temp.ptr, c->temp.bytes, st, emitted,
                              c->hitmask.as<unsigned long long>());
    if (rc) return rc;
    c->mark(2, true, st);
    c->mark(4, false, st);
    const bool ordered = false;  // (radix route: the longest list is not known; see queue_tail)
    if (join_pending) GS_HIP(hipStreamWaitEvent(st, c->ev_pre_join, 0));
    rc = gs::launch_render_fwd(c->recs.as<float4>(), nullptr, c->sorted.as<int>(), c->ranges.as<int>(), W, H, bg_color,
                               c->n_px.as<int>(), c->T_px.as<float>(), c->image.as<float>(), st,
                               c->rows_zeroed ? c->grad_rows.as<float4>() : nullptr, (long long)M * 4,
                               ro ? nullptr : c->blockmasks.as<unsigned short>(), nullptr,
                               ordered ? c->tile_tops.as<int>() : nullptr, nullptr, nullptr);
    if (rc) return rc;
    c->seg_ready = false;
    c->mark(4, true, st);
    if (ordered && (rc = gs::launch_tile_order(c->tile_tops.as<int>(), nullptr, num_tiles, c->tile_order.as<int>(), st))) return rc;
    c->order_ready = ordered;
  }
  if (segmented_this_forward) c->n_segmented_forwards++;
  c->N = N; c->M = M; c->S = S; c->l_max = l_max; c->width = W; c->height = H;
  c->last_mask = c->mask.as<unsigned char>();
  gs::pool_watch(c->last_mask, &c->last_mask);  // cleared when whoever ends up owning the block returns it to the pool
  c->tan_fovx = tan_fovx; c->tan_fovy = tan_fovy; c->mh_dist = cfg->mh_dist;
  c->have_forward = !ro;  // a render-only forward leaves nothing for a backward
  if (out) {
    out->num_culled = (size_t)M; out->num_pairs = (size_t)pairs; out->num_splats = S;
    out->mask = c->mask.as<unsigned char>();
    out->uv = c->lean ? nullptr : c->uv_all.as<float>(); out->xyz_c = c->lean ? nullptr : c->xyz_c_all.as<float>();
    out->compact_to_global = c->c2g.as<int>();
    out->sigma = mid ? c->sigma.as<float>() : nullptr; out->conic = mid ? c->conic.as<float>() : nullptr;
    out->J = mid ? c->J.as<float>() : nullptr; out->precomputed_rgb = mid ? c->rgb.as<float>() : nullptr;
    out->radius = c->radius.as<float>();
    out->uv_selected = c->uv.as<float>(); out->xyz_c_selected = c->xyz_c.as<float>();
    out->sorted_gaussians = c->sorted.as<int>();
    out->splat_start_end_idx_by_tile_idx = c->ranges.as<int>();
    out->image = c->image.as<float>(); out->weight_per_pixel = c->T_px.as<float>();
    out->splats_per_pixel = c->n_px.as<int>();
  }
  return GSPLAT_OK;
}

int gsplat_context_last_compaction(gsplat_context *c, const unsigned char **mask, const int **slots,
                                   const int **compact_to_global, int *num_gaussians, int *num_culled) {
  GS_REQUIRE(c != nullptr, "null context");
  const bool have = c->n_forwards > 0 && c->last_mask != nullptr;
  if (mask) *mask = have ? c->last_mask : nullptr;
  if (slots) *slots = have ? c->rank.as<int>() : nullptr;
  if (compact_to_global) *compact_to_global = have ? c->c2g.as<int>() : nullptr;
  if (num_gaussians) *num_gaussians = have ? c->N : 0;
  if (num_culled) *num_culled = have ? c->M : 0;
  return GSPLAT_OK;
}

int gsplat_context_detach_forward_outputs(gsplat_context *c) {
  GS_REQUIRE(c != nullptr, "null context");
  GS_REQUIRE(c->n_forwards > 0 && c->mask.ptr != nullptr, "no forward whose outputs could be handed over");
  GS_REQUIRE(!c->lean && !c->render_only, "a lean / render-only forward does not fill all ForwardPassData arrays");
  gs::DeviceBuffer *outs[13];
  c->forward_outputs(outs);
  for (gs::DeviceBuffer *b : outs) (void)b->detach();  // the caller owns the blocks now (gsplat_pool_free)
  c->have_forward = false;  // the fused backward would read arrays this context no longer has
  c->rows_ready = false;
  return GSPLAT_OK;
}

int gsplat_backward_render(gsplat_context *c, const float *grad_image, float bg_color, float *rgb_global,
                           void *stream) {
  return gsplat_backward_render_split(c, grad_image, bg_color, rgb_global, nullptr, nullptr, stream);
}

int gsplat_backward_render_split(gsplat_context *c, const float *grad_image, float bg_color, float *rgb_global,
                                 float *common, float *uv_norm, void *stream) {
  GS_REQUIRE(c != nullptr, "null context");
  GS_REQUIRE(c->have_forward, "no forward pass recorded in this context");
  GS_REQUIRE_DEV(grad_image);
  if (rgb_global) GS_REQUIRE_DEV(rgb_global);
  if (common) {
    GS_REQUIRE(rgb_global != nullptr, "the common rows are cleared by the pass that scatters g_rgb: rgb_global is needed");
    GS_REQUIRE_DEV(common);
    GS_REQUIRE(((uintptr_t)common & 15) == 0, "common must be 16-byte aligned");
  }
  if (uv_norm) { GS_REQUIRE(common != nullptr, "uv_norm goes with common"); GS_REQUIRE_DEV(uv_norm); }
  hipStream_t st = (hipStream_t)stream;
  const int M = c->M, W = c->width, H = c->height;
  c->rows_ready = false;
  c->backward_seen = true;
  if (!c->rows_zeroed) {  // first backward of the context, or a second backward of the same forward
    c->mark(5, false, st);
    GS_HIP(hipMemsetAsync(c->grad_rows.ptr, 0, (size_t)M * 64, st));
    c->mark(5, true, st);
  }
  c->rows_zeroed = false;
  // stage 6 is this one launch: when it is timed, the launch itself stamps the two events (see launch_render_bwd)
  const bool timed = (c->timing >> 6) & 1u;
  gs::TileSegments seg = {nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr, nullptr, 0u};
  if (c->seg_ready)
    seg = {c->seg_first.as<int>(), c->seg_extra.as<int2>(), reinterpret_cast<int *>(c->seg_extra.as<int2>() + c->seg_cap),
           c->seg_chk.as<float4>(), c->image.as<float>(), c->seg_cap, nullptr, nullptr, 0u};
  int rc = gs::launch_render_bwd(c->recs.as<float4>(), nullptr, c->sorted.as<int>(), c->ranges.as<int>(),
                                 c->n_px.as<int>(), c->T_px.as<float>(), grad_image, W, H, bg_color,
                                 c->grad_rows.as<float>(), nullptr, nullptr, nullptr, nullptr, st,
                                 c->blockmasks.as<unsigned short>(), timed ? c->ev[c->slot][12] : nullptr,
                                 timed ? c->ev[c->slot][13] : nullptr,
                                 (c->order_ready && !gs_no_tile_order()) ? c->tile_order.as<int>() : nullptr,
                                 c->seg_ready ? &seg : nullptr);
  if (rc) return rc;
  if (c->seg_ready) c->n_segmented_backwards++;
  if (c->order_ready && !gs_no_tile_order()) c->n_ordered_backwards++;
  if (timed) c->pending[c->slot][6] = 1;
  if (rgb_global) {
    scatter_rgb_rows_kernel<<<gs::div_up((long long)c->N * 3, kBlock), kBlock, 0, st>>>(
        c->mask.as<unsigned char>(), c->rank.as<int>(), c->N, c->grad_rows.as<float>(), rgb_global, common, uv_norm);
    GS_LAUNCH_CHECK();
  }
  c->rows_ready = true;
  return GSPLAT_OK;
}

int gsplat_backward_gaussians(gsplat_context *c, const gsplat_gaussians *g, const gsplat_camera *cam, int l_max,
                              const gsplat_gradients *out, void *stream) {
  return gsplat_backward_gaussians_range(c, g, cam, l_max, out, 0, g ? g->num_gaussians : 0, stream);
}

static int backward_gaussians_impl(gsplat_context *c, const gsplat_gaussians *g, const gsplat_camera *cam, int l_max,
                                   const gsplat_gradients *out, float *common, float *uv_norm, int first_gaussian,
                                   int end_gaussian, void *stream, const AdamFused *adam = nullptr, int adam_mode = 2);

int gsplat_backward_gaussians_range(gsplat_context *c, const gsplat_gaussians *g, const gsplat_camera *cam, int l_max,
                                    const gsplat_gradients *out, int first_gaussian, int end_gaussian, void *stream) {
  GS_REQUIRE(out != nullptr, "null argument struct");
  return backward_gaussians_impl(c, g, cam, l_max, out, nullptr, nullptr, first_gaussian, end_gaussian, stream);
}

int gsplat_backward_gaussians_split(gsplat_context *c, const gsplat_gaussians *g, const gsplat_camera *cam, int l_max,
                                    float *common, float *uv_norm, int first_gaussian, int end_gaussian, void *stream) {
  GS_REQUIRE_DEV(common);
  GS_REQUIRE(((uintptr_t)common & 15) == 0, "common must be 16-byte aligned");
  if (uv_norm) GS_REQUIRE_DEV(uv_norm);
  return backward_gaussians_impl(c, g, cam, l_max, nullptr, common, uv_norm, first_gaussian, end_gaussian, stream);
}

int gsplat_backward_gaussians_adam(gsplat_context *c, const gsplat_gaussians *g, const gsplat_camera *cam, int l_max,
                                   const gsplat_adam_fused *opt, const gsplat_gradients *out, void *stream) {
  GS_REQUIRE(opt != nullptr && g != nullptr, "null argument struct");
  const int n_groups = l_max > 0 ? 6 : 5;
  for (int k = 0; k < 6; ++k) {
    if (k == 2 && l_max == 0) continue;  // no coefficients beyond band 0
    GS_REQUIRE_DEV(opt->exp_avg[k]); GS_REQUIRE_DEV(opt->exp_avg_sq[k]);
  }
  (void)n_groups;
  GS_REQUIRE(((uintptr_t)opt->exp_avg[5] & 15) == 0 && ((uintptr_t)opt->exp_avg_sq[5] & 15) == 0,
             "the quaternion moments must be 16-byte aligned");
  if (opt->uv_grad_accum) GS_REQUIRE_DEV(opt->uv_grad_accum);
  if (opt->grad_accum_dur) GS_REQUIRE_DEV(opt->grad_accum_dur);
  AdamFused ad = {opt->exp_avg[0], opt->exp_avg_sq[0], opt->exp_avg[1], opt->exp_avg_sq[1], opt->exp_avg[2], opt->exp_avg_sq[2],
                  opt->exp_avg[3], opt->exp_avg_sq[3], opt->exp_avg[4], opt->exp_avg_sq[4], opt->exp_avg[5], opt->exp_avg_sq[5],
                  opt->lr[0], opt->lr[1], opt->lr[2], opt->lr[3], opt->lr[4], opt->lr[5],
                  opt->b1, opt->b2, opt->eps, opt->bias1, opt->bias2, opt->uv_grad_accum, opt->grad_accum_dur, nullptr};
  GS_REQUIRE(opt->mode >= 0 && opt->mode <= 2,
             "mode: 0 all six groups in one kernel, 1 SH and position left to the optimizer kernels, 2 the SH group in a kernel of its own in front");
  if (opt->mode == 1) {
    GS_REQUIRE(out != nullptr, "mode 1 hands grad_xyz and grad_precompute_rgb to the optimizer kernels: `out` is needed");
    GS_REQUIRE_DEV(out->grad_xyz);
    if (l_max > 0) GS_REQUIRE_DEV(out->grad_precompute_rgb);
  }
  if (opt->mode == 2 && l_max > 0 && c && c->have_forward && c->M > 0) {
    // the SH rows are read ONCE, by sh_adam_dir_kernel: their Adam step and sh_bwd's sums over them; the per-gaussian
    // backward behind it (kAdam 3) takes the position gradient through the view direction from c->dir_grad
    GS_REQUIRE(c->rows_ready, "gsplat_backward_render has not run for this forward pass");
    GS_REQUIRE(l_max == c->l_max && g->num_gaussians == c->N, "backward arguments do not match the recorded forward pass");
    GS_REQUIRE(cam != nullptr, "null argument struct");
    hipStream_t st = (hipStream_t)stream;
    int rc = c->dir_grad.reserve((size_t)c->M * 3 * sizeof(float), st);
    if (rc) return rc;
    ad.dir = c->dir_grad.as<float>();
    const unsigned int blocks = (unsigned int)(((long long)c->M * 16 + kBlock - 1) / kBlock);
    float *sh_p = const_cast<float *>(g->sh);
#define GS_SHA(LL)                                                                                                     \
  sh_adam_dir_kernel<LL><<<blocks, kBlock, 0, st>>>(c->M, c->c2g.as<int>(), sh_p, ad.m_sh, ad.v_sh, ad.lr_sh, ad.b1, ad.b2,  \
                                                   ad.eps, ad.bias1, ad.bias2, g->xyz, g->rgb, cam->campos[0], cam->campos[1], \
                                                   cam->campos[2], c->grad_rows.as<float4>(), ad.dir)
    switch (l_max) {
      case 1: GS_SHA(1); break;
      case 2: GS_SHA(2); break;
      default: GS_SHA(3); break;
    }
#undef GS_SHA
    GS_LAUNCH_CHECK();
  }
  return backward_gaussians_impl(c, g, cam, l_max, out, nullptr, nullptr, 0, g->num_gaussians, stream, &ad,
                                 opt->mode == 1 ? 1 : opt->mode == 2 ? 3 : 2);
}

static int backward_gaussians_impl(gsplat_context *c, const gsplat_gaussians *g, const gsplat_camera *cam, int l_max,
                                   const gsplat_gradients *out, float *common, float *uv_norm, int first_gaussian,
                                   int end_gaussian, void *stream, const AdamFused *adam, int adam_mode) {
  GS_REQUIRE(c && g && cam, "null argument struct");
  GS_REQUIRE(0 <= first_gaussian && first_gaussian <= end_gaussian && end_gaussian <= g->num_gaussians, "bad gaussian range");
  GS_REQUIRE(c->have_forward && c->rows_ready, "gsplat_backward_render has not run for this forward pass");
  GS_REQUIRE(l_max == c->l_max && g->num_gaussians == c->N && cam->width == c->width && cam->height == c->height,
             "backward arguments do not match the recorded forward pass");
  static const gsplat_gradients kNoArrays = {};
  if (!out) out = &kNoArrays;  // split form: the twelve common columns go to `common`, nothing else is stored
  if (!common && !(adam && (out == &kNoArrays || adam_mode == 1))) {  // (the Adam forms store only the arrays they are given)
    GS_REQUIRE_DEV(out->grad_xyz); GS_REQUIRE_DEV(out->grad_rgb); GS_REQUIRE_DEV(out->grad_opacity);
    GS_REQUIRE_DEV(out->grad_scale); GS_REQUIRE_DEV(out->grad_quaternion);
    GS_REQUIRE(((uintptr_t)out->grad_quaternion & 15) == 0, "grad_quaternion must be 16-byte aligned");
  }
  if (l_max > 0 && out->grad_sh) GS_REQUIRE_DEV(out->grad_sh);  // NULL: the caller rebuilds them (gsplat_optimizer_step_sh_factored)
  hipStream_t st = (hipStream_t)stream;
  const int M = c->M, W = c->width, H = c->height;
  // cuda/trainer.cu:992-995
  const float fx = cam->focal_x, fy = cam->focal_y;
  const float fov_x = (float)(2.0 * atan((double)W / (2.0 * (double)fx)));
  const float fov_y = (float)(2.0 * atan((double)H / (2.0 * (double)fy)));
  const float tan_fovx = tanf(fov_x * 0.5f), tan_fovy = tanf(fov_y * 0.5f);
  const float fwd_tan_fovx = c->tan_fovx, fwd_tan_fovy = c->tan_fovy;  // the recorded forward's (cuda/raster.cu:92-93)
  BwdOut bo = {out->grad_xyz, out->grad_rgb, out->grad_sh, out->grad_opacity, out->grad_scale, out->grad_quaternion,
               out->grad_conic, out->grad_uv, out->grad_J, out->grad_sigma, out->grad_xyz_c, out->grad_precompute_rgb,
               common, uv_norm};
  // a range of global indices holds at most that many visible gaussians (and never more than M); the kernel finds the
  // range's compacted slots in compact_to_global on the device (first_slot_not_below)
  const bool whole = first_gaussian == 0 && end_gaussian == g->num_gaussians;
  const int span = whole ? M : std::min(M, end_gaussian - first_gaussian);
  if (span == 0) return GSPLAT_OK;
  const int ranged = whole ? 0 : 1;
  const dim3 grid(gs::div_up(span, kBlock)), block(kBlock);
  c->mark(7, false, st);
  static const AdamFused kNoAdam = {};  // (value-initialised: every pointer null)
#define GS_BWD(LL)                                                                                                     \
  do {                                                                                                                 \
    if (adam && adam_mode == 1)                                                                                        \
      preprocess_bwd_kernel<LL, 1><<<grid, block, 0, st>>>(*g, cam->view, cam->proj, M, c->c2g.as<int>(),              \
                                                    c->xyz_c.as<float>(), c->grad_rows.as<float4>(), fx, fy, tan_fovx, \
                                                    tan_fovy, fwd_tan_fovx, fwd_tan_fovy, c->mh_dist,                  \
                                                    cam->campos[0], cam->campos[1], cam->campos[2], W, H, bo,          \
                                                    ranged, first_gaussian, end_gaussian, *adam);                      \
    else if (adam && adam_mode == 3)                                                                                   \
      preprocess_bwd_kernel<LL, 3><<<grid, block, 0, st>>>(*g, cam->view, cam->proj, M, c->c2g.as<int>(),              \
                                                    c->xyz_c.as<float>(), c->grad_rows.as<float4>(), fx, fy, tan_fovx, \
                                                    tan_fovy, fwd_tan_fovx, fwd_tan_fovy, c->mh_dist,                  \
                                                    cam->campos[0], cam->campos[1], cam->campos[2], W, H, bo,          \
                                                    ranged, first_gaussian, end_gaussian, *adam);                      \
    else if (adam)                                                                                                     \
      preprocess_bwd_kernel<LL, 2><<<grid, block, 0, st>>>(*g, cam->view, cam->proj, M, c->c2g.as<int>(),           \
                                                    c->xyz_c.as<float>(), c->grad_rows.as<float4>(), fx, fy, tan_fovx, \
                                                    tan_fovy, fwd_tan_fovx, fwd_tan_fovy, c->mh_dist,                  \
                                                    cam->campos[0], cam->campos[1], cam->campos[2], W, H, bo,          \
                                                    ranged, first_gaussian, end_gaussian, *adam);                      \
    else                                                                                                               \
      preprocess_bwd_kernel<LL, 0><<<grid, block, 0, st>>>(*g, cam->view, cam->proj, M, c->c2g.as<int>(),          \
                                                    c->xyz_c.as<float>(), c->grad_rows.as<float4>(), fx, fy, tan_fovx, \
                                                    tan_fovy, fwd_tan_fovx, fwd_tan_fovy, c->mh_dist,                  \
                                                    cam->campos[0], cam->campos[1], cam->campos[2], W, H, bo,          \
                                                    ranged, first_gaussian, end_gaussian, kNoAdam);                    \
  } while (0)
  switch (l_max) {
    case 0: GS_BWD(0); break;
    case 1: GS_BWD(1); break;
    case 2: GS_BWD(2); break;
    default: GS_BWD(3); break;
  }
#undef GS_BWD
  GS_LAUNCH_CHECK();
  c->mark(7, true, st);
  return GSPLAT_OK;
}

int gsplat_backward_pass(gsplat_context *c, const gsplat_gaussians *g, const gsplat_camera *cam,
                         const float *grad_image, float bg_color, int l_max, const gsplat_gradients *out,
                         void *stream) {
  GS_REQUIRE(c && g && cam && out, "null argument struct");
  GS_REQUIRE(c->have_forward, "no forward pass recorded in this context");
  GS_REQUIRE(l_max == c->l_max && g->num_gaussians == c->N && cam->width == c->width && cam->height == c->height,
             "backward arguments do not match the recorded forward pass");
  GS_REQUIRE_DEV(out->grad_xyz); GS_REQUIRE_DEV(out->grad_rgb); GS_REQUIRE_DEV(out->grad_opacity);
  GS_REQUIRE_DEV(out->grad_scale); GS_REQUIRE_DEV(out->grad_quaternion);
  if (l_max > 0 && out->grad_sh) GS_REQUIRE_DEV(out->grad_sh);  // NULL: the caller rebuilds them (gsplat_optimizer_step_sh_factored)
  int rc = gsplat_backward_render(c, grad_image, bg_color, nullptr, stream);
  if (rc) return rc;
  return gsplat_backward_gaussians(c, g, cam, l_max, out, stream);
}

int gsplat_context_set_binning_route(gsplat_context *c, int route) {
  GS_REQUIRE(c != nullptr, "null context");
  GS_REQUIRE(route >= 0 && route <= 2, "route: 0 auto, 1 counting sort, 2 radix sorts");
  c->forced_route = route;
  return GSPLAT_OK;
}

int gsplat_context_set_segment_options(gsplat_context *c, int poll_budget, int thin_layer_blocks, float gate) {
  GS_REQUIRE(c != nullptr, "null context");
  if (poll_budget >= 0) c->fseg_poll_budget = poll_budget > 0 ? poll_budget : 1;
  if (thin_layer_blocks >= 0) c->fseg_thin_layer = thin_layer_blocks;
  if (gate >= 0.0f) c->fseg_gate = gate;
  return GSPLAT_OK;
}

int gsplat_context_set_timing_stages(gsplat_context *c, unsigned int stage_mask) {
  GS_REQUIRE(c != nullptr, "null context");
  if (stage_mask && !c->ev[0][0]) {
    for (int a = 0; a < gsplat_context::kSlots; ++a)
      for (int b = 0; b < 2 * gsplat_context::kStages; ++b) GS_HIP(hipEventCreate(&c->ev[a][b]));
  }
  GS_HIP(hipDeviceSynchronize());
  for (int a = 0; a < gsplat_context::kSlots; ++a) c->harvest(a);
  for (int k = 0; k < gsplat_context::kStages; ++k) { c->stage_ms[k] = 0; c->stage_n[k] = 0; }
  c->timing = stage_mask & ((1u << gsplat_context::kStages) - 1u);
  return GSPLAT_OK;
}

int gsplat_context_get_counters(gsplat_context *c, long long *out, int n) {
  GS_REQUIRE(c && out && n >= 0, "null argument");
  // (reporting only: the freshest pair of figures either slot holds -- the forward's decisions use queue_tail's rule)
  long long fmax = c->fig_max, fsum = c->fig_sum;
  unsigned int best = 0;
  for (int sl = 0; sl < 2; ++sl) {
    const unsigned long long w0 = __atomic_load_n(&c->h_pub[8 + 4 * sl], __ATOMIC_RELAXED);
    const unsigned long long w1 = __atomic_load_n(&c->h_pub[9 + 4 * sl], __ATOMIC_RELAXED);
    const unsigned int tag = (unsigned int)(w0 >> 32);
    if (tag != 0 && tag == (unsigned int)(w1 >> 32) && tag >= best && tag <= (unsigned int)(c->ticket & 0xFFFFFFFFull)) {
      best = tag; fmax = (long long)(int)(unsigned int)w0; fsum = (long long)(int)(unsigned int)w1;
    }
  }
  int fallbacks = 0;  // (a blocking 4-byte copy: this is a diagnostic call)
  if (n > 9 && c->counters.ptr) GS_HIP(hipMemcpy(&fallbacks, c->fseg_fallbacks(), sizeof(int), hipMemcpyDeviceToHost));
  const long long v[10] = {c->n_forwards, c->n_tail_redone, c->n_compact_walks, c->n_instance_growths, c->n_ordered_backwards,
                           (long long)c->n_segmented_backwards, (long long)c->n_segmented_forwards, fmax, fsum, fallbacks};
  for (int k = 0; k < n && k < 10; ++k) out[k] = v[k];
  return 10;
}

int gsplat_context_set_render_only(gsplat_context *c, int enabled) {
  GS_REQUIRE(c != nullptr, "null context");
  c->render_only = enabled != 0;
  if (c->render_only) c->have_forward = false;
  return GSPLAT_OK;
}

int gsplat_context_set_preprocess_split(gsplat_context *c, int mode) {
  GS_REQUIRE(c != nullptr, "null context");
  GS_REQUIRE(mode >= 0 && mode <= 2, "mode: 0 one kernel, 1 two kernels in sequence, 2 two kernels side by side");
  c->pre_split = mode;
  return GSPLAT_OK;
}

int gsplat_context_set_lean_forward(gsplat_context *c, int enabled) {
  GS_REQUIRE(c != nullptr, "null context");
  c->lean = enabled != 0;
  return GSPLAT_OK;
}

int gsplat_context_set_timing(gsplat_context *c, int enabled) {
  return gsplat_context_set_timing_stages(c, enabled ? ~0u : 0u);
}

int gsplat_context_get_timing(gsplat_context *c, double *stage_ms_sum, long long *stage_count, int max_stages) {
  GS_REQUIRE(c && stage_ms_sum && stage_count, "null argument");
  GS_HIP(hipDeviceSynchronize());
  for (int a = 0; a < gsplat_context::kSlots; ++a) c->harvest(a);
  for (int k = 0; k < max_stages && k < gsplat_context::kStages; ++k) {
    stage_ms_sum[k] = c->stage_ms[k];
    stage_count[k] = c->stage_n[k];
  }
  return gsplat_context::kStages;
}

int gsplat_pack_gradients_global(gsplat_context *c, const gsplat_gradients *grads, int l_max, int num_gaussians,
                                 float *packed, void *stream) {
  GS_REQUIRE(c && grads, "null argument struct");
  GS_REQUIRE(c->have_forward && num_gaussians == c->N && l_max == c->l_max, "does not match the recorded forward");
  GS_REQUIRE_DEV(packed);
  const int n_coeffs = (l_max + 1) * (l_max + 1);
  const long long total = (long long)num_gaussians * (12 + 3 * n_coeffs);
  pack_global_kernel<<<gs::div_up(total, kBlock), kBlock, 0, (hipStream_t)stream>>>(
      c->mask.as<unsigned char>(), c->rank.as<int>(), num_gaussians, n_coeffs, *grads, packed);
  GS_LAUNCH_CHECK();
  return GSPLAT_OK;
}

}  // extern "C"
