// gs_render.hip -- per-tile alpha compositing, forward and backward, for gfx950.
//
// Semantics: render_image / render_image_backward of the reference (cuda/render.cu:6-135,
// cuda/render_backward.cu:11-258) -- front-to-back blending of each 16x16 tile's depth-
// sorted list with the 1/255 alpha floor, the 0.99 cap and the T < 1e-4 stop; the backward
// walks the list back to front rebuilding T and the colour behind each splat.
//
// Design (not the reference's one-warp-per-tile, 8-pixels-per-lane layout):
//   * one 256-thread workgroup per tile = four wave64s; every 16-lane ROW of a wave owns one 4x4 pixel block,
//     one pixel per lane;
//   * the tile's list is consumed in batches (256 entries forward, 128 backward): every thread gathers ONE gaussian
//     (three 16-byte loads of a 48-byte record, or the reference's four arrays), works out which of the tile's 16
//     blocks the ellipse alpha >= 1/255 can reach (gs::block_hits; the forward hands these masks to the backward),
//     and parks the record in LDS in the form the loops evaluate (gs::stage_record);
//   * each wave compacts, per row, the batch's slots that hit the row's block into a byte list in LDS, and every
//     row walks ITS OWN list: in one loop trip the four rows of a wave work on four different (gaussian, block)
//     pairs and a wave's trip count is the longest of its four lists -- on the benchmark scene 1.65x fewer trips
//     than visiting (gaussian, 8x8 quadrant) pairs (tests/analysis/model_trips.py).  Skipping is exact: a skipped
//     gaussian has alpha < 1/255 on every pixel of the block;
//   * forward: two list entries per trip; a row whose 16 pixels are saturated gets no list, a wave stops when its 64
//     pixels are saturated, the workgroup when all four waves have; on the side it clears the gradient rows the
//     backward accumulates into (the kernel leaves most of the HBM bandwidth unused).  33 VALU per two-entry trip (r04:
//     the 0.99 cap lives in the exponent's clamp, a pixel's T is zeroed only in the trip that saturates it);
//   * backward: 39 VALU per trip (r01: 51).  The colour behind a splat is carried as ONE number, its dot product with
//     the pixel's gradient (the recurrence is linear and nothing else ever reads the colour; the background is the
//     layer behind everything, which also absorbs the reference's T_final (bg . grad) / (1 - alpha) term); the nine
//     partial sums of a (gaussian, block) pair are reduced across the row's 16 lanes only (gs::row_moments9r: 14 VALU --
//     the products are folded into the quad stage, and since the four lanes of a quad share their cy, three of the six
//     moments are cy or cy^2 times a quad-partial of another; the nine totals land in nine lanes of one register),
//     merged across the tile's blocks with ONE ds_add_f64 per trip, converted once per gaussian and flushed per batch
//     to HBM as whole 64-byte gradient rows (or into the reference's four gradient arrays).
#include <hip/hip_ext.h>
#include "gs_common.h"
#include "gs_render.h"

#include <cstdlib>
#include <type_traits>

#ifndef GS_ABLATE
#define GS_ABLATE 0
#endif
// Diagnostic build (-DGS_STAMP=1, tools/bwd_timeline.py): every wave of render_bwd records where its cycles went
// (s_memtime around the barriers, staging, list building, the trip loop and the flush) into a device array that
// gsplat_debug_read_stamps copies out.  No stamp code exists in the normal build.
#ifndef GS_STAMP
#define GS_STAMP 0
#endif
#ifndef GS_EXTRA_FMA
#define GS_EXTRA_FMA 0
#endif
// r04: the nine-value row reduction with two butterfly stages folded into the products (gs_render.h: 1 = row_moments9q,
// 17 VALU + 1 wait state; 2 = row_moments9r, 14 VALU with the quad's constant cy factored out of three of the moments
// and the atomic's address in the wait state) instead of one (0 = row_moments9, 22 + 3: the r01-r03 form; A/B: tools/ab).
#ifndef GS_ROWSUM_QUAD
#define GS_ROWSUM_QUAD 2
#endif
// r04 A/B switches of the backward's loop (profiles/r04_ab_*.txt): the exponent chain as one asm block; the issue priority
// of the trip loop (0: none)
#ifndef GS_BWD_ALPHA_ASM
#define GS_BWD_ALPHA_ASM 1
#endif
#ifndef GS_BWD_PRIO
#define GS_BWD_PRIO 1
#endif
#ifndef GS_BWD_CAP_AS_FORWARD
#define GS_BWD_CAP_AS_FORWARD 1
#endif
// r05: waves 2 and 3 -- idle while waves 0 and 1 staged a batch and while they run the flush's first step (124 slots
// occupy two waves: 10 % of the kernel's wave cycles were waves 2 and 3 waiting at those two barriers) -- take the staging
// over and REQUEST the next batch's list entries and records while the flush of the current batch runs; the values wait
// in registers (live only between the trip loops: 12 bytes of scratch, touched twice per batch) for the LDS arrays to
// become free.  The two dependent global round trips of a batch leave the tile's critical path: launch 0.2764 -> 0.2719 ms
// on the benchmark scene, garden-shaped workload 0.319 -> 0.308 ms (profiles/r05_ab_split_staging.txt).  0: staging by the
// slots' own threads (waves 0, 1) at the top of the batch (r04).
#ifndef GS_BWD_SPLIT_STAGING
#define GS_BWD_SPLIT_STAGING 1
#endif
#ifndef GS_BWD_ACC_CLEAR
#define GS_BWD_ACC_CLEAR 0  // who clears the batch's sums under split staging: 0 waves 2-3 (measured best), 1 all four, 2 waves 0-1
#endif
#if GS_STAMP
#define GS_STAMP_WORDS 16
__device__ unsigned long long gs_stamp_buf[(1 << 16) * GS_STAMP_WORDS];
__device__ unsigned long long gs_stamp_fwd[(1 << 16) * GS_STAMP_WORDS];  // render_fwd: same layout, phases mapped below
#define GS_NOW() ((unsigned long long)__builtin_readcyclecounter())
#define GS_LAP(acc) do { const unsigned long long _n = GS_NOW(); (acc) += _n - st_last; st_last = _n; } while (0)
#else
#define GS_LAP(acc) do { } while (0)
#endif
// Backward batch size: 124 slots keep the block at 20 392 B of LDS (records 4 KB, f64 accumulators 9.7 KB, lists + third
// record array 5.8 KB), i.e. EIGHT workgroups per CU (160 KB / 8 = 20 480 B), matching the 64 VGPRs the kernel is held
// to below: 8 waves per SIMD and 2048 resident workgroups, which 1080p's 8160 tiles fill in 3.98 rounds (r02: 0.350 ->
// 0.342 ms against 128 slots / 72 VGPRs / 7 workgroups per CU).  256 slots (41 KB, 3 blocks/CU) measured 0.76 ms vs
// 0.57 ms when tried in r01, 64 slots 0.63 ms (twice the barriers and per-batch work).
#ifndef GS_BWD_BATCH
#define GS_BWD_BATCH 124
#endif

namespace gs {

// Forward batch size: 248 slots keep the block at 20 KB of LDS = EIGHT workgroups per CU at the kernel's 62 VGPRs (256
// slots: 20 800 B, seven).  r02 measured +-0 for this; with r04's shorter loop it is -6 us (profiles/r04_ab_priority...).
#ifndef GS_FWD_BATCH
#define GS_FWD_BATCH 248
#endif
constexpr int kBatch = GS_FWD_BATCH;

struct RawSplats {  // the reference operator's input arrays
  const float *uv, *opacity, *conic, *rgb;
};

template <bool kPacked>
__device__ __forceinline__ SplatRec load_record(int g, const float4 *__restrict__ recs, const RawSplats &raw) {
  if constexpr (kPacked) {
    SplatRec s;
    s.r0 = recs[3 * g]; s.r1 = recs[3 * g + 1]; s.r2 = recs[3 * g + 2];
    return s;
  } else {
    return make_record(raw.uv[2 * g], raw.uv[2 * g + 1], raw.conic[3 * g], raw.conic[3 * g + 1], raw.conic[3 * g + 2],
                       raw.opacity[g], raw.rgb[3 * g], raw.rgb[3 * g + 1], raw.rgb[3 * g + 2]);
  }
}

struct GradOut {         // either whole rows ...
  float *rows;           // [M,16]: rgb3 opacity1 conic3 uv2, 7 pad
  // ... or the reference operator's four arrays
  float *rgb, *opacity, *uv, *conic;
};

constexpr int kListStride = kBatch;  // entries per row list (a batch can put all of its 256 slots on one list)

// bit of (wave, row) in the 16-bit block mask: block (bx, by) = ((wave & 1) * 2 + (row & 1), (wave >> 1) * 2 + (row >> 1))
__device__ __forceinline__ int wave_first_bit(int wave) { return (wave >> 1) * 8 + (wave & 1) * 2; }

struct RowCounts { int c0, c1, c2, c3; };

// Compacts, for each of the wave's four rows, the slots of the staged batch that hit the row's block
// (and lie below the row's stop index `lim*` in the backward).  `s_r2`: the record array whose .w holds the 16 block bits
// (the backward's third array, the forward's second).  A list entry is the slot's byte offset into the
// record arrays (slot * 16), so the loop needs no shift; counts are wave-uniform.
template <int kStride>
__device__ __forceinline__ RowCounts build_row_lists(const float4 *s_r2, unsigned short *lists, int count, int wave,
                                                     int lane, int lim0, int lim1, int lim2, int lim3, int round,
                                                     int slot0 = 0) {  // slot0: the record arrays' slot of `s_r2[0]` (r06: two sets)
  RowCounts rc = {0, 0, 0, 0};
  const int b0 = wave_first_bit(wave);
  for (int sb = 0; sb < count; sb += 64) {
    const int slot = sb + lane;
    const unsigned int nib = slot < count ? (__float_as_uint(s_r2[slot].w) >> b0) : 0u;
    const bool h0 = (nib & 1u) && slot < lim0, h1 = (nib & 2u) && slot < lim1;
    const bool h2 = (nib & 16u) && slot < lim2, h3 = (nib & 32u) && slot < lim3;
    const unsigned long long m0 = __ballot(h0), m1 = __ballot(h1), m2 = __ballot(h2), m3 = __ballot(h3);
    if ((m0 | m1 | m2 | m3) == 0ull) continue;
#define GS_MBCNT(m) __builtin_amdgcn_mbcnt_hi((unsigned int)((m) >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)(m), 0u))
    if (h0) lists[0 * kStride + rc.c0 + GS_MBCNT(m0)] = (unsigned short)((slot0 + slot) << 4);
    if (h1) lists[1 * kStride + rc.c1 + GS_MBCNT(m1)] = (unsigned short)((slot0 + slot) << 4);
    if (h2) lists[2 * kStride + rc.c2 + GS_MBCNT(m2)] = (unsigned short)((slot0 + slot) << 4);
    if (h3) lists[3 * kStride + rc.c3 + GS_MBCNT(m3)] = (unsigned short)((slot0 + slot) << 4);
#undef GS_MBCNT
    rc.c0 += __popcll(m0); rc.c1 += __popcll(m1); rc.c2 += __popcll(m2); rc.c3 += __popcll(m3);
  }
  // Pad the shorter lists up to the wave's trip count (rounded up to `round`) with the sentinel slot kStride, whose
  // record is all zeros (opacity 0 -> alpha 0): a row past the end of its list then needs no "am I active" test.
  const int longest = max(max(rc.c0, rc.c1), max(rc.c2, rc.c3));
  const int padded = (longest + round - 1) / round * round;
  const unsigned short sentinel = (unsigned short)((slot0 + kStride) << 4);
  for (int k = lane; k < padded; k += 64) {
    if (k >= rc.c0) lists[0 * kStride + k] = sentinel;
    if (k >= rc.c1) lists[1 * kStride + k] = sentinel;
    if (k >= rc.c2) lists[2 * kStride + k] = sentinel;
    if (k >= rc.c3) lists[3 * kStride + k] = sentinel;
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  return rc;
}

// ---- r05: long lists in segments, forward (gs_render.h: FwdSegments) ------------------------------------------------
typedef __attribute__((address_space(1))) unsigned long long gs_gu64;
__device__ __forceinline__ void granule_store(unsigned long long *g, unsigned int epoch, float v) {
  __hip_atomic_store((gs_gu64 *)g, ((unsigned long long)epoch << 32) | __float_as_uint(v), __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool granule_load(const unsigned long long *g, unsigned int epoch, float &v) {
  const unsigned long long x = __hip_atomic_load((gs_gu64 *)g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  v = __uint_as_float((unsigned int)x);
  return (unsigned int)(x >> 32) == epoch;
}
// T in front of segment k's first entry from what the blocks in front have published: the nearest block q < k whose
// final T is out (F_q = P_(q+1)) and the products t_(q+1) .. t_(k-1) of the blocks behind it, multiplied up from the left
// -- whichever q a thread finds, the value is the same left fold.  false: the poll budget ran out.
// (segment j of the tile lives in storage slot base[j] + rank: t at gran[slot], F at gran[cap + slot])
__device__ __forceinline__ bool resolve_prefix(const FwdSegments &fs, int rank, int tid, int k, int budget, float &prefix) {
  const unsigned long long *gt = fs.granules + (size_t)rank * 256 + tid, *gf = gt + (size_t)fs.cap * 256;
  for (int polls = 0;;) {
    int q = -1;
    float v = 0.0f;
    for (int j = k - 1; j >= 0; --j) {
      const size_t at = (size_t)fs.base[j] * 256;
      if (granule_load(gf + at, fs.epoch, v)) { q = j; break; }
      float t;
      if (!granule_load(gt + at, fs.epoch, t)) break;
    }
    if (q >= 0) {
      for (int j = q + 1; j < k; ++j) {
        float t;
        (void)granule_load(gt + (size_t)fs.base[j] * 256, fs.epoch, t);  // (seen on the way down: published values stay)
        v = v * t;
      }
      prefix = v;
      return true;
    }
    if (++polls >= budget) return false;
    __builtin_amdgcn_s_sleep(32);
  }
}

// Phase A: the product of (1 - alpha) over the entries [begin, end) of the tile's list, per pixel -- the forward's loop
// without colour, stop test and saturation bookkeeping, the same alpha values in the same order, so that the running
// product is bit for bit the one phase C multiplies up.  `fold`: the product segment by segment from the left,
// ((t_0 t_1) t_2) ..., the order in which collect_prefix takes the published values.
template <bool kPacked>
__device__ __forceinline__ float fwd_t_product(const float4 *__restrict__ recs, const RawSplats &raw,
                                               const int *__restrict__ sorted, int start, int begin, int end, bool fold,
                                               float tx0, float ty0, float fpx, float fpy, int tid, float4 *s_r0,
                                               float4 *s_r1, float4 *s_r2, unsigned short *s_list) {
  const char *r0b = reinterpret_cast<const char *>(s_r0), *r1b = reinterpret_cast<const char *>(s_r1);
  const char *r2b = reinterpret_cast<const char *>(s_r2);
  float tl = 1.0f, pref = 1.0f;
  for (int base = begin; base < end; base += kBatch) {
    const int count = min(kBatch, end - base);
    int t = tid;
    asm volatile("" : "+v"(t));
    if (fold && base > begin && base % kSegEntries == 0) { pref = pref * tl; tl = 1.0f; }
    __syncthreads();
    if (t < count) {
      const int g = sorted[start + base + t];
      SplatRec s = load_record<kPacked>(g, recs, raw);
      const unsigned int hits = block_hits(s, tx0, ty0);
      stage_record(s);
      s.r1.w = __uint_as_float(hits);
      s.r2.w = s.r1.y > kLog2AlphaMax ? kLog2AlphaMax : s.r1.y;
      s_r0[t] = s.r0; s_r1[t] = s.r1; s_r2[t] = s.r2;
    }
    __syncthreads();
    unsigned short *lists = s_list + (t >> 6) * 4 * kListStride;
    const unsigned int list_lds =
        (unsigned int)(size_t)(__attribute__((address_space(3))) const unsigned short *)(lists + ((t >> 4) & 3) * kListStride);
    const RowCounts rc = build_row_lists<kListStride>(s_r1, lists, count, t >> 6, t & 63, kBatch, kBatch, kBatch, kBatch, 2);
    const int trips = max(max(rc.c0, rc.c1), max(rc.c2, rc.c3));
    for (int i = 0; i < trips; i += 2) {
      int off0, off1;
      asm volatile("ds_read_u16 %0, %2\n\tds_read_u16 %1, %2 offset:2\n\ts_waitcnt lgkmcnt(0)"
                   : "=&v"(off0), "=&v"(off1) : "v"(list_lds + 2 * i) : "memory");
      const float4 a0 = *reinterpret_cast<const float4 *>(r0b + off0), a1 = *reinterpret_cast<const float4 *>(r0b + off1);
      const float2 b0 = *reinterpret_cast<const float2 *>(r1b + off0), b1 = *reinterpret_cast<const float2 *>(r1b + off1);
      const float l0 = *reinterpret_cast<const float *>(r2b + off0 + 12), l1 = *reinterpret_cast<const float *>(r2b + off1 + 12);
      float al0 = staged_alpha_capped(a0.z, a0.w, b0.x, b0.y, l0, a0.x - fpx, a0.y - fpy);
      float al1 = staged_alpha_capped(a1.z, a1.w, b1.x, b1.y, l1, a1.x - fpx, a1.y - fpy);
      al0 = al0 > kAlphaMin ? al0 : 0.0f;
      al1 = al1 > kAlphaMin ? al1 : 0.0f;
      asm volatile("" : "+v"(al0), "+v"(al1));
      tl = __builtin_fmaf(-al0, tl, tl);
      tl = __builtin_fmaf(-al1, tl, tl);
    }
  }
  return fold ? pref * tl : tl;
}

// One segment of a long list (gs_render.h: FwdSegments).
template <bool kPacked>
__device__ __forceinline__ void fwd_segment_block(const float4 *__restrict__ recs, const RawSplats &raw,
                                                  const int *__restrict__ sorted, const int *__restrict__ ranges,
                                                  int width, int height, int ntx, unsigned short *__restrict__ masks_out,
                                                  const FwdSegments &fs, int blk, float4 *s_r0, float4 *s_r1,
                                                  float4 *s_r2, unsigned short *s_list) {
  const int2 bk = fs.blocks[blk];
  const int tile = bk.x, k = bk.y & 0xFFFF;
  const bool side_by_side = (bk.y >> 30) != 0;  // a thin layer (fwd_segments_table_kernel)
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, row = lane >> 4, j = lane & 15;
  if (tid == 0) {
    s_r0[kBatch] = s_r2[kBatch] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    s_r1[kBatch] = sentinel_r1();
  }
  const int tile_x = tile % ntx, tile_y = tile / ntx;
  const int px = tile_x * 16 + (wave & 1) * 8 + (row & 1) * 4 + (j & 3);
  const int py = tile_y * 16 + (wave >> 1) * 8 + (row >> 1) * 4 + (j >> 2);
  const bool inside = px < width && py < height;
  const float fpx = (float)px, fpy = (float)py;
  const float tx0 = (float)(tile_x * 16), ty0 = (float)(tile_y * 16);
  const int start = ranges[tile], total = ranges[tile + 1] - start;
  const int m = (total + kSegEntries - 1) / kSegEntries;
  const int a = k * kSegEntries, b = min(total, a + kSegEntries);
  // this block's storage slot is its index; the slot of the segment in front: base[k - 1] + the tile's rank
  const int rank = fs.rank[tile];
  unsigned long long *gran_t = fs.granules + (size_t)blk * 256 + tid, *gran_f = gran_t + (size_t)fs.cap * 256;
  float4 *part = fs.part + (size_t)blk * 256 + tid;
  int *stop = fs.stop + (size_t)blk * 256 + tid;

  float prefix = 1.0f;
  if (k > 0) {
    // Is the block in front done already?  (The blocks are dispatched layer by layer -- all segments 0, then all segments
    // 1, ...: where a layer fills the chip, the layer behind it starts when it is through.)  Then T is simply what it left
    // behind, nobody behind this block can be waiting for its product yet either, so phase A is skipped: its final T will
    // do.  And if no pixel of the tile is left, neither has this segment anything to add.
    float f = 0.0f;
    const bool have = granule_load(fs.granules + ((size_t)fs.cap + fs.base[k - 1] + rank) * 256 + tid, fs.epoch, f);
    const bool all_have = __syncthreads_and(have ? 1 : 0);
    if (all_have) {
      prefix = f;
      if (__syncthreads_and(!inside || f < kTMin ? 1 : 0)) {
        if (k < m - 1) granule_store(gran_f, fs.epoch, 0.0f);
        *part = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        *stop = -2;
        return;
      }
    } else {
      // In a thin layer the segments of a list run side by side: the blocks behind need this segment's product before
      // its compositing.  In a thick one the block in front is about to finish (it was dispatched a layer earlier) and the
      // other blocks of this CU keep its SIMDs busy meanwhile: waiting costs less than the product pass.
      if (side_by_side && k < m - 1) {
        const float tk = fwd_t_product<kPacked>(recs, raw, sorted, start, a, b, false, tx0, ty0, fpx, fpy, tid, s_r0, s_r1, s_r2, s_list);
        granule_store(gran_t, fs.epoch, tk);
      }
      const bool got = resolve_prefix(fs, rank, tid, k, fs.poll_budget, prefix);
      if (__syncthreads_or(got ? 0 : 1)) {  // (rare: the blocks in front have not run yet and may be waiting for this CU)
        if (tid == 0 && fs.fallbacks) atomicAdd(fs.fallbacks, 1);
        prefix = fwd_t_product<kPacked>(recs, raw, sorted, start, 0, a, true, tx0, ty0, fpx, fpy, tid, s_r0, s_r1, s_r2, s_list);
      }
    }
  }

  // Phase C: the forward's loop with T = prefix * (running product of the segment)
  const char *r0b = reinterpret_cast<const char *>(s_r0), *r1b = reinterpret_cast<const char *>(s_r1);
  const char *r2b = reinterpret_cast<const char *>(s_r2);
  const bool alive = inside && prefix >= kTMin;
  float tl = alive ? 1.0f : 0.0f, T = alive ? prefix : 0.0f, T_fin = -1.0f, ar = 0.0f, ag = 0.0f, ab = 0.0f;
  int n = -1;
  unsigned long long satmask = __ballot(!alive);
  int live = satmask != ~0ull ? 1 : 0;
  for (int base = a; base < b; base += kBatch) {
    const int count = min(kBatch, b - base);
    int t = tid;
    asm volatile("" : "+v"(t));
    __syncthreads();
    if (t < count) {
      const int g = sorted[start + base + t];
      SplatRec s = load_record<kPacked>(g, recs, raw);
      const unsigned int hits = block_hits(s, tx0, ty0);
      if (masks_out) masks_out[start + base + t] = (unsigned short)hits;
      stage_record(s);
      s.r1.w = __uint_as_float(hits);
      s.r2.w = s.r1.y > kLog2AlphaMax ? kLog2AlphaMax : s.r1.y;
      s_r0[t] = s.r0; s_r1[t] = s.r1; s_r2[t] = s.r2;
    }
    __syncthreads();
    if (live > 0) {
      const int big = kBatch;
      unsigned short *lists = s_list + (t >> 6) * 4 * kListStride;
      const unsigned int list_lds =
          (unsigned int)(size_t)(__attribute__((address_space(3))) const unsigned short *)(lists + ((t >> 4) & 3) * kListStride);
      const RowCounts rc = build_row_lists<kListStride>(s_r1, lists, count, t >> 6, t & 63,
                                           (unsigned int)(satmask & 0xFFFFull) == 0xFFFFu ? 0 : big,
                                           (unsigned int)((satmask >> 16) & 0xFFFFull) == 0xFFFFu ? 0 : big,
                                           (unsigned int)((satmask >> 32) & 0xFFFFull) == 0xFFFFu ? 0 : big,
                                           (unsigned int)((satmask >> 48) & 0xFFFFull) == 0xFFFFu ? 0 : big, 2);
      const int trips = max(max(rc.c0, rc.c1), max(rc.c2, rc.c3));
      for (int i = 0; i < trips; i += 2) {
        int off0, off1;
        asm volatile("ds_read_u16 %0, %2\n\tds_read_u16 %1, %2 offset:2\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(off0), "=&v"(off1) : "v"(list_lds + 2 * i) : "memory");
        const float4 a0 = *reinterpret_cast<const float4 *>(r0b + off0), c0 = *reinterpret_cast<const float4 *>(r2b + off0);
        const float2 b0 = *reinterpret_cast<const float2 *>(r1b + off0);
        const float4 a1 = *reinterpret_cast<const float4 *>(r0b + off1), c1 = *reinterpret_cast<const float4 *>(r2b + off1);
        const float2 b1 = *reinterpret_cast<const float2 *>(r1b + off1);
        float al0 = staged_alpha_capped(a0.z, a0.w, b0.x, b0.y, c0.w, a0.x - fpx, a0.y - fpy);
        float al1 = staged_alpha_capped(a1.z, a1.w, b1.x, b1.y, c1.w, a1.x - fpx, a1.y - fpy);
        al0 = al0 > kAlphaMin ? al0 : 0.0f;
        al1 = al1 > kAlphaMin ? al1 : 0.0f;
        asm volatile("" : "+v"(al0), "+v"(al1));
        // (as in render_fwd_kernel: a saturated or dead pixel has tl = T = 0 and stays in the mask by itself)
        const float w0 = al0 * T;
        const float tl0 = __builtin_fmaf(-al0, tl, tl);
        const float tT0 = prefix * tl0;
        ar = __builtin_fmaf(c0.x, w0, ar);
        ag = __builtin_fmaf(c0.y, w0, ag);
        ab = __builtin_fmaf(c0.z, w0, ab);
        const unsigned long long s0 = __ballot(tT0 < kTMin);
        tl = tl0;
        T = tT0;
        if (s0 != satmask) {
          if (((s0 & ~satmask) >> lane) & 1ull) { T_fin = tT0; n = base + (off0 >> 4) + 1; tl = 0.0f; T = 0.0f; }
          satmask = s0;
          if (satmask == ~0ull) { live = 0; i = trips; }
        }
        const float w1 = al1 * T;
        const float tl1 = __builtin_fmaf(-al1, tl, tl);
        const float tT1 = prefix * tl1;
        ar = __builtin_fmaf(c1.x, w1, ar);
        ag = __builtin_fmaf(c1.y, w1, ag);
        ab = __builtin_fmaf(c1.z, w1, ab);
        const unsigned long long s1 = __ballot(tT1 < kTMin);
        tl = tl1;
        T = tT1;
        if (s1 != satmask) {
          if (((s1 & ~satmask) >> lane) & 1ull) { T_fin = tT1; n = base + (off1 >> 4) + 1; tl = 0.0f; T = 0.0f; }
          satmask = s1;
          if (satmask == ~0ull) {
            live = 0;
            break;
          }
        }
      }
    }
    if (__syncthreads_and(live <= 0 ? 1 : 0)) break;
  }
  // what the segment leaves behind: T in front of the next one (0 for a pixel that stopped or was dead: dead from here on)
  if (k < m - 1) granule_store(gran_f, fs.epoch, T);
  *part = make_float4(ar, ag, ab, T_fin >= 0.0f ? T_fin : T);
  *stop = T_fin >= 0.0f ? n : (alive ? -1 : -2);
}

#ifndef GS_FWD_WAVES
#define GS_FWD_WAVES 8
#endif
template <bool kPacked>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(GS_FWD_WAVES, 8))) void render_fwd_kernel(const float4 *__restrict__ recs, RawSplats raw,
                                                              const int *__restrict__ sorted,
                                                              const int *__restrict__ ranges, int width, int height,
                                                              int ntx, int num_tiles, float bg,
                                                              int *__restrict__ n_out, float *__restrict__ T_out,
                                                              float *__restrict__ image, float4 *__restrict__ zero,
                                                              long long zero_vec, unsigned short *__restrict__ masks_out,
                                                              const int *__restrict__ order, int *__restrict__ tops_out,
                                                              TileSegments seg, FwdSegments fs) {
  __shared__ float4 s_r0[kBatch + 1], s_r1[kBatch + 1], s_r2[kBatch + 1];  // [kBatch]: the all-zero sentinel record
  __shared__ int s_tile_top;
  __shared__ __attribute__((aligned(16))) unsigned short s_list[16 * kListStride + 2];
  // Optional side job: every workgroup clears its share of `zero` (the gradient rows the backward accumulates into).
  // This kernel leaves most of the HBM bandwidth unused, so the 64 bytes per gaussian ride along for free instead of
  // costing a memset between forward and backward.
  if (zero) {
    const long long per = (zero_vec + gridDim.x - 1) / gridDim.x, lo = per * blockIdx.x, hi = min(lo + per, zero_vec);
    for (long long k = lo + threadIdx.x; k < hi; k += 256) zero[k] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  }
  // the segments of the long lists come first (gs_render.h: FwdSegments): each is a block of its own, and the tiles they
  // belong to have no block in the main grid
  // (the context's forward only: the stand-alone operator on the reference's arrays keeps one block per tile)
  const bool segmented = kPacked && fs.blocks != nullptr;
  if constexpr (kPacked) {
    if (segmented && (int)blockIdx.x < fs.cap) {
      if ((int)blockIdx.x < *fs.count)
        fwd_segment_block<kPacked>(recs, raw, sorted, ranges, width, height, ntx, masks_out, fs, (int)blockIdx.x, s_r0, s_r1, s_r2, s_list);
      return;
    }
  }
  const int tile = ordered_tile(order, segmented ? (int)blockIdx.x - fs.cap : (int)blockIdx.x, num_tiles);
  if (tile >= num_tiles) return;
  if (segmented && fs.rank[tile] >= 0) return;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, row = lane >> 4, j = lane & 15;
#if GS_STAMP
  const unsigned long long st_t0 = GS_NOW(), st_rt0 = __builtin_amdgcn_s_memrealtime();
  unsigned long long st_last = st_t0, st_bar = 0, st_bar1 = 0, st_bar2 = 0, st_stage = 0, st_lists = 0, st_loop = 0, st_trips = 0, st_batches = 0, st_load = 0;
#endif
  if (tid == 0) {
    s_r0[kBatch] = s_r2[kBatch] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    s_r1[kBatch] = sentinel_r1();
    s_tile_top = 0;
  }
  const int tile_x = tile % ntx, tile_y = tile / ntx;
  const int px = tile_x * 16 + (wave & 1) * 8 + (row & 1) * 4 + (j & 3);
  const int py = tile_y * 16 + (wave >> 1) * 8 + (row >> 1) * 4 + (j >> 2);
  const bool inside = px < width && py < height;
  const float fpx = (float)px, fpy = (float)py;
  const float tx0 = (float)(tile_x * 16), ty0 = (float)(tile_y * 16);
  const char *r0b = reinterpret_cast<const char *>(s_r0), *r1b = reinterpret_cast<const char *>(s_r1);
  const char *r2b = reinterpret_cast<const char *>(s_r2);

  const int start = ranges[tile], total = ranges[tile + 1] - start;
  // A saturated pixel keeps T = 0 in the running transmittance, so every later splat blends with weight 0 and the
  // common path needs no per-pixel "done" masking; its real final transmittance and stop index live in T_fin / n.
  float T = inside ? 1.0f : 0.0f, T_fin = -1.0f, ar = 0.0f, ag = 0.0f, ab = 0.0f;
  int n = total;
  unsigned long long satmask = __ballot(!inside);  // lanes whose pixel is saturated or outside the image
  int live = satmask != ~0ull ? 1 : 0;

  static_assert(kSegEntries % kBatch == 0, "a segment boundary is a batch boundary of the forward");
  const bool checkpoints = seg.chk != nullptr && total > kSegSplitMin;  // a long list: the backward may walk it in segments
  for (int base = 0; base < total; base += kBatch) {
    const int count = min(kBatch, total - base);
    // an opaque per-batch copy of the thread index (see render_bwd_kernel): staging and list-building addresses are
    // rebuilt per batch instead of living in registers across the compositing loop
    int t = tid;
    asm volatile("" : "+v"(t));
    if (checkpoints && base > 0 && base % kSegEntries == 0)  // (gs_render.h: TileSegments)
      seg.chk[(size_t)segment_slot(start, base / kSegEntries) * 256 + t] = make_float4(T, ar, ag, ab);
#if GS_STAMP
    ++st_batches;
    GS_LAP(st_lists);  // (prologue of the first batch; nothing between the batches)
#endif
    __syncthreads();
    GS_LAP(st_bar);
    if (t < count) {
      const int g = sorted[start + base + t];
      SplatRec s = load_record<kPacked>(g, recs, raw);
#if GS_STAMP
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // split the staging stamp: loads | block test + LDS stores
      asm volatile("" : "+v"(s.r0.x), "+v"(s.r1.x), "+v"(s.r2.x));
      GS_LAP(st_load);
#endif
      const unsigned int hits = block_hits(s, tx0, ty0);
      if (masks_out) masks_out[start + base + t] = (unsigned short)hits;  // the backward stages the same instances
      stage_record(s);
      // The 0.99 cap goes into the exponent's own clamp (r04): alpha = exp2(min(q, log2 opa)) already has a per-gaussian
      // upper bound on q, so min(0.99, .) in the loop becomes the bound min(log2 opa, log2 0.99), formed here once per
      // (gaussian, tile).  It rides in r2.w, which the loop's 16-byte colour read fetches anyway; the block mask moves to
      // r1.w (the forward's loop reads only r1.xy).  A capped alpha comes out as exp2(kLog2AlphaMax), within two ulps
      // below 0.99f (the backward keeps its own 0.99f: it needs the uncapped value next to the capped one).  NaN stays NaN.
      s.r1.w = __uint_as_float(hits);
      s.r2.w = s.r1.y > kLog2AlphaMax ? kLog2AlphaMax : s.r1.y;
      s_r0[t] = s.r0; s_r1[t] = s.r1; s_r2[t] = s.r2;
    }
    GS_LAP(st_stage);
    __syncthreads();
    GS_LAP(st_bar1);
    if (live > 0) {
      // rows whose 16 pixels are all saturated (or outside) need no list
      const int big = kBatch;
      unsigned short *lists = s_list + (t >> 6) * 4 * kListStride;
      const unsigned int list_lds =
          (unsigned int)(size_t)(__attribute__((address_space(3))) const unsigned short *)(lists + ((t >> 4) & 3) * kListStride);
      const RowCounts rc = build_row_lists<kListStride>(s_r1, lists, count, t >> 6, t & 63,
                                           (unsigned int)(satmask & 0xFFFFull) == 0xFFFFu ? 0 : big,
                                           (unsigned int)((satmask >> 16) & 0xFFFFull) == 0xFFFFu ? 0 : big,
                                           (unsigned int)((satmask >> 32) & 0xFFFFull) == 0xFFFFu ? 0 : big,
                                           (unsigned int)((satmask >> 48) & 0xFFFFull) == 0xFFFFu ? 0 : big, 2);
      const int trips = max(max(rc.c0, rc.c1), max(rc.c2, rc.c3));
#if GS_STAMP
      st_trips += (trips + 1) / 2;
      GS_LAP(st_lists);
#endif
      for (int i = 0; i < trips; i += 2) {
        // two list entries per trip: both records are fetched and both exponentials evaluated before the
        // (sequential) blending; a row past the end of its list reads the sentinel record and blends with alpha 0
        // (two zero-extending 16-bit reads: one 32-bit read costs an and and a shift on the VALU, which is what bounds
        // this loop; r04)
        int off0, off1;
        asm volatile("ds_read_u16 %0, %2\n\tds_read_u16 %1, %2 offset:2\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(off0), "=&v"(off1) : "v"(list_lds + 2 * i) : "memory");
        const float4 a0 = *reinterpret_cast<const float4 *>(r0b + off0), c0 = *reinterpret_cast<const float4 *>(r2b + off0);
        const float2 b0 = *reinterpret_cast<const float2 *>(r1b + off0);
        const float4 a1 = *reinterpret_cast<const float4 *>(r0b + off1), c1 = *reinterpret_cast<const float4 *>(r2b + off1);
        const float2 b1 = *reinterpret_cast<const float2 *>(r1b + off1);
        // (c.w: the exponent's bound, so the colour reads stay 16-byte reads -- a ds_read_b96 costs twice the LDS cycles)
        float al0 = staged_alpha_capped(a0.z, a0.w, b0.x, b0.y, c0.w, a0.x - fpx, a0.y - fpy);  // <= 0.99
        float al1 = staged_alpha_capped(a1.z, a1.w, b1.x, b1.y, c1.w, a1.x - fpx, a1.y - fpy);
        al0 = al0 > kAlphaMin ? al0 : 0.0f;
        al1 = al1 > kAlphaMin ? al1 : 0.0f;
        asm volatile("" : "+v"(al0), "+v"(al1));  // both evaluated (two independent chains) before the sequential blending
        // Invariant: T is either 0 (saturated or outside the image) or >= 1e-4, so "T (1 - alpha) < 1e-4" alone decides
        // whether a pixel is saturated behind this splat, and the compare's lane mask doubles as the saturation
        // bookkeeping.  A saturated pixel has T = 0, hence T (1 - alpha) = 0 < 1e-4: it stays in the mask by itself, and
        // the common path needs no select at all -- only the (rare) trip in which a pixel saturates zeroes its T (r04:
        // two v_cndmask fewer per trip; T (1 - alpha) as one FMA, T - alpha T).
        const float w0 = al0 * T;
        const float tT0 = __builtin_fmaf(-al0, T, T);
        ar = __builtin_fmaf(c0.x, w0, ar);
        ag = __builtin_fmaf(c0.y, w0, ag);
        ab = __builtin_fmaf(c0.z, w0, ab);
        const unsigned long long s0 = __ballot(tT0 < kTMin);  // this splat was still accumulated (render.cu:76-87)
        T = tT0;
        if (s0 != satmask) {  // rare: some pixel saturated with the trip's first splat
          if (((s0 & ~satmask) >> lane) & 1ull) { T_fin = tT0; n = base + (off0 >> 4) + 1; T = 0.0f; }
          satmask = s0;
          // ... the wave's last live one: this trip is its last (the second splat blends with T = 0 everywhere, and the
          // test behind it finds nothing new: without this exit the wave walks the rest of the tile's list for nothing --
          // the dense scenes' forward took twice as long for a day of this round)
          if (satmask == ~0ull) { live = 0; i = trips; }
        }
        const float w1 = al1 * T;
        const float tT1 = __builtin_fmaf(-al1, T, T);
        ar = __builtin_fmaf(c1.x, w1, ar);
        ag = __builtin_fmaf(c1.y, w1, ag);
        ab = __builtin_fmaf(c1.z, w1, ab);
        const unsigned long long s1 = __ballot(tT1 < kTMin);
        T = tT1;
        if (s1 != satmask) {  // rare: ... with its second
          if (((s1 & ~satmask) >> lane) & 1ull) { T_fin = tT1; n = base + (off1 >> 4) + 1; T = 0.0f; }
          satmask = s1;
          if (satmask == ~0ull) {
            live = 0;
            break;
          }
        }
      }
    }
    GS_LAP(st_loop);
    const int all_done = __syncthreads_and(live <= 0 ? 1 : 0);
    GS_LAP(st_bar2);
    if (all_done) break;
  }
  if (tops_out) {
    // the tile's work in the backward = the largest stop index of its pixels (cuda/render_backward.cu:64,74): handed to
    // tile_order_kernel, which deals the backward's tiles heaviest first
    int top = inside ? n : 0;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) top = max(top, __shfl_xor(top, off, 64));
    if (lane == 0) atomicMax(&s_tile_top, top);
    __syncthreads();
    if (tid == 0) tops_out[tile] = s_tile_top;
  }
  if (inside) {
    const int pid = py * width + px;
    const float Tout = T_fin >= 0.0f ? T_fin : T;  // T_fin is set by the splat that saturated the pixel
    n_out[pid] = n;
    T_out[pid] = Tout;
    image[3 * pid + 0] = ar + Tout * bg;
    image[3 * pid + 1] = ag + Tout * bg;
    image[3 * pid + 2] = ab + Tout * bg;
  }
#if GS_STAMP
  if (lane == 0) {
    unsigned long long *o = gs_stamp_fwd + ((size_t)blockIdx.x * 4 + wave) * GS_STAMP_WORDS;
    o[0] = (unsigned long long)tile; o[1] = st_rt0; o[2] = __builtin_amdgcn_s_memrealtime(); o[3] = GS_NOW() - st_t0;
    o[4] = st_bar; o[5] = st_stage; o[6] = st_lists; o[7] = st_loop; o[8] = st_load; o[9] = st_trips; o[10] = st_batches;
    o[11] = 0; o[12] = st_bar1; o[13] = st_bar2; o[14] = 0; o[15] = (unsigned long long)wave;
  }
#endif
}

__device__ __forceinline__ int row_max_int(int v) {  // max over the 16 lanes of a row, in every lane of the row
  v = max(v, __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xF, 0xF, false));
  v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xF, 0xF, false));
  v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x141, 0xF, 0xF, false));
  v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x140, 0xF, 0xF, false));
  return v;
}

template <bool kPacked, bool kRows, int kB>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void render_bwd_kernel(const float4 *__restrict__ recs, RawSplats raw,
                                                              const int *__restrict__ sorted,
                                                              const int *__restrict__ ranges,
                                                              const int *__restrict__ n_px,
                                                              const float *__restrict__ T_px,
                                                              const float *__restrict__ grad_image, int width,
                                                              int height, int ntx, int num_tiles, float bg,
                                                              GradOut out, const unsigned short *__restrict__ masks_in,
                                                              const int *__restrict__ order, TileSegments seg) {
  __shared__ float4 s_r0[kB + 1], s_r1[kB + 1];  // [kB]: the all-zero sentinel record
  // [slot][9]: rgb, S0, Sx, Sy, Sxx, Sxy, Syy.  Doubles on purpose: on gfx950 ds_add_f32 retires about one LANE
  // every three cycles while ds_add_f64 runs at LDS rate (profiles/microbench/lds_atomic_rate: 109 vs 16 cycles for
  // a 36-lane instruction), and the merge across the tile's 16 blocks needs one atomic per trip.
  constexpr int kAcc = 10;  // doubles per slot (nine used): 80 bytes = 5 x the list entry's byte offset
  __shared__ double s_acc[(kB + 1) * kAcc];  // slot kB: the sentinel's (rows past the end of their list add zeros there)
  __shared__ int s_id[kB];
  // row lists | third record array; both are dead once the batch's trips are done, and the flush parks the nine
  // gradient values per gaussian on top of them (kB*36 bytes: the lists and the first slots of s_r2, not the sentinel)
  __shared__ __attribute__((aligned(16))) unsigned char s_mix[16 * kB * 2 + (kB + 1) * 16];
  unsigned short *s_list = reinterpret_cast<unsigned short *>(s_mix);
  float4 *s_r2 = reinterpret_cast<float4 *>(s_mix + 16 * kB * 2);
  float *s_res = reinterpret_cast<float *>(s_mix);
  static_assert(kB * 9 * 4 <= 16 * kB * 2 + kB * 16, "the flush values must not reach the sentinel record");
  __shared__ int s_top;
  // which tile, and which segment of its list (gs_render.h: TileSegments; seg_a = 0, seg_end = -1: the whole list)
  int tile, seg_a = 0, seg_end = -1, chk_slot = -1;
  static_assert(kSegEntries % kB == 0, "a segment is a whole number of backward batches");
  if (seg.chk && (int)blockIdx.x < seg.extra_cap) {  // the further segments come first: each is kSegEntries of work
    const int e = (int)blockIdx.x;
    if (e >= *seg.extra_count) return;
    const int2 ex = seg.extra[e];
    tile = ex.x;
    seg_a = ex.y * kSegEntries;
    seg_end = seg_a + kSegEntries;
    chk_slot = segment_slot(ranges[tile], ex.y + 1);  // the checkpoint at this segment's far boundary
  } else {
    tile = ordered_tile(order, seg.chk ? (int)blockIdx.x - seg.extra_cap : (int)blockIdx.x, num_tiles);
    if (tile >= num_tiles) return;
    if (seg.chk && seg.granted[tile] > 0) { seg_end = kSegEntries; chk_slot = segment_slot(ranges[tile], 1); }
  }
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, row = lane >> 4, j = lane & 15;
#if GS_STAMP
  const unsigned long long st_t0 = GS_NOW(), st_rt0 = __builtin_amdgcn_s_memrealtime();
  unsigned long long st_last = st_t0, st_bar = 0, st_bar1 = 0, st_bar2 = 0, st_bar3 = 0, st_stage = 0, st_lists = 0, st_loop = 0, st_flush = 0, st_trips = 0, st_batches = 0;
  auto st_write = [&]() {
    if (lane == 0) {
      unsigned long long *o = gs_stamp_buf + ((size_t)blockIdx.x * 4 + wave) * GS_STAMP_WORDS;
      o[0] = (unsigned long long)tile; o[1] = st_rt0; o[2] = __builtin_amdgcn_s_memrealtime(); o[3] = GS_NOW() - st_t0;
      o[4] = st_bar; o[5] = st_stage; o[6] = st_lists; o[7] = st_loop; o[8] = st_flush; o[9] = st_trips; o[10] = st_batches;
      unsigned int xcc;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      o[11] = xcc; o[12] = st_bar1; o[13] = st_bar2; o[14] = st_bar3; o[15] = (unsigned long long)wave;
    }
  };
#endif
  const int tile_x = tile % ntx, tile_y = tile / ntx;
  const int px = tile_x * 16 + (wave & 1) * 8 + (row & 1) * 4 + (j & 3);
  const int py = tile_y * 16 + (wave >> 1) * 8 + (row >> 1) * 4 + (j >> 2);
  const bool inside = px < width && py < height;
  const float fpx = (float)px, fpy = (float)py;
  const float tx0 = (float)(tile_x * 16), ty0 = (float)(tile_y * 16);
  const int start = ranges[tile] + seg_a;
  const char *r0b = reinterpret_cast<const char *>(s_r0), *r1b = reinterpret_cast<const char *>(s_r1);
  const char *r2b = reinterpret_cast<const char *>(s_r2);
  char *accb = reinterpret_cast<char *>(s_acc);

  int n = 0;
  float Tf = 0.0f, g0 = 0.0f, g1 = 0.0f, g2 = 0.0f;
  if (inside) {
    const int pid = py * width + px;
    n = n_px[pid];
    Tf = T_px[pid];
    g0 = grad_image[3 * pid]; g1 = grad_image[3 * pid + 1]; g2 = grad_image[3 * pid + 2];
  }
  // Running transmittance, and s = (pixel gradient) . (colour behind the current splat).  The colour behind only ever
  // enters through that dot product, and its back-to-front recurrence c <- c + alpha (colour - c) is linear, so the
  // kernel carries the one number instead of three.  The background is the layer behind everything: with c starting
  // at bg instead of 0, T (colour - c) . grad already contains the reference's - T_final (bg . grad) / (1 - alpha)
  // term (cuda/render_backward.cu:139-151: c' = c + T_final bg / T_next obeys the same recurrence and starts at bg).
  float T = Tf, s = bg * g0 + bg * g1 + bg * g2;
  if (seg_end >= 0) {  // a segment of a split list: entries [seg_a, seg_end) of the tile's list
    if (n > seg_end) {
      // the pixel stops behind this segment: it enters at the far boundary with the forward's checkpoint -- T in front of
      // entry seg_end, and behind it the colour (image - C) / T (the image holds everything, the background included)
      const int pid = py * width + px;
      const float4 ck = seg.chk[(size_t)chk_slot * 256 + tid];
      T = ck.x;
      const float inv = __builtin_amdgcn_rcpf(ck.x);
      s = (g0 * (seg.image[3 * pid] - ck.y) + g1 * (seg.image[3 * pid + 1] - ck.z) + g2 * (seg.image[3 * pid + 2] - ck.w)) * inv;
      n = seg_end - seg_a;
    } else {
      n = max(n - seg_a, 0);  // it stops inside the segment or in front of it: as an unsplit list from here on
    }
  }
  const int row_top_v = row_max_int(n);
  const int rt0 = __builtin_amdgcn_readlane(row_top_v, 0), rt1 = __builtin_amdgcn_readlane(row_top_v, 16);
  const int rt2 = __builtin_amdgcn_readlane(row_top_v, 32), rt3 = __builtin_amdgcn_readlane(row_top_v, 48);
  const int wave_top = max(max(rt0, rt1), max(rt2, rt3));
  if (tid == 0) {
    s_top = 0;
    s_r0[kB] = s_r2[kB] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    s_r1[kB] = sentinel_r1();
  }
  __syncthreads();
  if (lane == 0) atomicMax(&s_top, wave_top);
  __syncthreads();
  const int top = s_top;  // cuda/render_backward.cu:64,74: start at (max n over the tile) - 1
#if GS_STAMP
  GS_LAP(st_stage);
  if (top <= 0) { st_write(); return; }
#endif
  if (top <= 0) return;
  // where this lane's share of the nine row totals goes (see row_moments9), and the lane constants of the sums:
  // pixel position relative to the tile centre, pixel gradient
#if GS_ROWSUM_QUAD == 2
  const int red_idx = row_moments9r_index(lane);
#elif GS_ROWSUM_QUAD
  const int red_idx = row_moments9q_index(lane);
#else
  const int red_idx = row_moments9_index(lane);
#endif
  const bool red_lane = red_idx >= 0;
#if GS_BWD_CAP_AS_FORWARD
  const float alpha_cap = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, __builtin_amdgcn_exp2f(kLog2AlphaMax))));
#endif
  const unsigned int acc_lane = (unsigned int)(size_t)(__attribute__((address_space(3))) char *)accb + (unsigned int)(red_idx * 8);

  // Split staging is for the fused path (packed 48-byte records: three loads that nothing touches until staging time).
  // The raw-array operator builds its record from nine scattered values with arithmetic on them (make_record) and
  // computes the block mask: requested early, that work would make waves 2 and 3 WAIT for the loads in front of the
  // flush's second step (measured: gsplat_render_image_backward 0.38 -> 0.45 ms), so it keeps the r04 staging.
  constexpr bool kSplit = GS_BWD_SPLIT_STAGING != 0 && kPacked;
  // what waves 2 and 3 hold for the NEXT batch: slot (thread - 128)'s list entry, record and block mask
  SplatRec pre = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};
  int pre_g = 0;
  auto prefetch = [&](int next_base, int tt) {
    if constexpr (kSplit) {
      const int slot = tt - 128;
      if (tt >= 128 && next_base >= 0 && slot < min(kB, top - next_base)) {
        pre_g = sorted[start + next_base + slot];
        pre = load_record<kPacked>(pre_g, recs, raw);
        pre.r2.w = __uint_as_float((unsigned int)masks_in[start + next_base + slot]);
        // (nothing here may WAIT for these loads: converting the record in this place holds waves 2 and 3 back from the
        // barrier in front of the flush's second step -- measured +8 us, profiles/r05_ab_split_staging.txt; the values are
        // first touched at staging time, a whole flush later)
      }
    }
  };
  prefetch(((top - 1) / kB) * kB, tid);
  for (int base = ((top - 1) / kB) * kB; base >= 0; base -= kB) {
    const int count = min(kB, top - base);
    // an opaque per-batch copy of the thread index: the LDS addresses of staging and flush derive from it and are
    // recomputed per batch (a few integer operations) instead of being hoisted out of the batch loop, where they
    // would sit in registers across the compositing loop and end up spilled to scratch
    int t = tid;
    asm volatile("" : "+v"(t));
#if GS_STAMP
    ++st_batches;
    GS_LAP(st_flush);
#endif
    __syncthreads();
    GS_LAP(st_bar);
    if constexpr (kSplit) {
      if (t >= 128) {  // waves 2, 3: the records requested during the previous flush go to LDS; they also clear the sums
        const int slot = t - 128;
        if (slot < count) {
          SplatRec s = pre;
          stage_record(s);
          s_r0[slot] = s.r0; s_r1[slot] = s.r1; s_r2[slot] = s.r2;
          s_id[slot] = pre_g;
        }
#if GS_BWD_ACC_CLEAR == 0
        for (int k = slot; k < kAcc * kB; k += 128) s_acc[k] = 0.0;
#endif
      }
#if GS_BWD_ACC_CLEAR == 1
      for (int k = t; k < kAcc * kB; k += 256) s_acc[k] = 0.0;
#elif GS_BWD_ACC_CLEAR == 2
      if (t < 128)  // waves 0, 1 have nothing else to do here
        for (int k = t; k < kAcc * kB; k += 128) s_acc[k] = 0.0;
#endif
    } else {
      if (t < count) {  // count <= kB
        const int g = sorted[start + base + t];
        SplatRec s = load_record<kPacked>(g, recs, raw);
#if GS_ABLATE == 9
        if constexpr (kPacked) {  // one more dependent global round trip: sensitivity to the staging latency
          const float4 extra = recs[3 * (g ^ (int)(s.r0.x == 12345.678f)) + 1];
          if (extra.x == 98765.4f) s.r0.x += 1.0f;
        }
#endif
        // the fused path hands over the forward's block masks; the raw-array operator computes them here
        if constexpr (kPacked) s.r2.w = __uint_as_float((unsigned int)masks_in[start + base + t]);
        else s.r2.w = __uint_as_float(block_hits(s, tx0, ty0));
        stage_record(s);
        s_r0[t] = s.r0; s_r1[t] = s.r1; s_r2[t] = s.r2;
        s_id[t] = g;
      }
      for (int k = t; k < kAcc * kB; k += 256) s_acc[k] = 0.0;
    }
    GS_LAP(st_stage);
    __syncthreads();
    GS_LAP(st_bar1);
    if (base < wave_top) {
      // (list addresses from the opaque index too)
      unsigned short *lists = s_list + (t >> 6) * 4 * kB;
      const unsigned int list_lds =
          (unsigned int)(size_t)(__attribute__((address_space(3))) const unsigned short *)(lists + ((t >> 4) & 3) * kB);
      const RowCounts rc = build_row_lists<kB>(s_r2, lists, count, t >> 6, t & 63, rt0 - base, rt1 - base, rt2 - base, rt3 - base, 1);
      const int trips = max(max(rc.c0, rc.c1), max(rc.c2, rc.c3));
#if GS_STAMP
      st_trips += trips;
      GS_LAP(st_lists);
#endif
      const int n_rel = (n - base) * 16;  // "base + slot < n" on byte offsets
      // The lane constants of the nine sums (pixel position relative to the tile centre, pixel gradient) are rebuilt
      // per batch behind an opaque copy, so that they are not live across staging and flush: held for the whole
      // kernel they push it past the 80-register step and the compiler spills them to scratch (+0.2 GB of traffic).
      float g0b = g0, g1b = g1, g2b = g2;
      asm volatile("" : "+v"(g0b), "+v"(g1b), "+v"(g2b));
#if GS_ROWSUM_QUAD == 2
      const RowsWeights rw = make_rows_weights(t & 63, (float)(((t >> 6) & 1) * 8 + ((t >> 4) & 1) * 4 + (t & 3)) - 7.5f,
                                               (float)((t >> 7) * 8 + ((t >> 5) & 1) * 4 + ((t >> 2) & 3)) - 7.5f, g0b, g1b, g2b);
#elif GS_ROWSUM_QUAD
      // (from the opaque index, like the addresses above: derived from the plain thread index the twelve weights are
      // loop-invariant for the whole kernel, get hoisted above the batch loop and spilled)
      const QuadWeights rw = make_quad_weights(t & 63, (float)(((t >> 6) & 1) * 8 + ((t >> 4) & 1) * 4 + (t & 3)) - 7.5f,
                                               (float)((t >> 7) * 8 + ((t >> 5) & 1) * 4 + ((t >> 2) & 3)) - 7.5f, g0b, g1b, g2b);
#else
      const RowWeights rw = make_row_weights(lane, (float)((wave & 1) * 8 + (row & 1) * 4 + (j & 3)) - 7.5f,
                                             (float)((wave >> 1) * 8 + (row >> 1) * 4 + (j >> 2)) - 7.5f, g0b, g1b, g2b);
#endif
      // The loop is bound by VALU issue at the measured per-instruction costs (tools/valu_cost_model.py: compares,
      // selects and min/max cost 4.4 cycles per wave instruction against 2.9 for a multiply), so it exists twice: when
      // every pixel of the wave needs the whole batch (no pixel stopped inside it: the common case), the per-trip
      // "slot below the pixel's stop index" compare is dropped.
      auto run_trips = [&](auto check_n) {
        constexpr bool kCheckN = decltype(check_n)::value;
        for (int i = trips - 1; i >= 0; --i) {
          // ds_read_u16 zero-extends; read through asm, the compiler would add an "and 0xffff" to every trip
          int off;
          asm volatile("ds_read_u16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(off) : "v"(list_lds + 2 * i) : "memory");
          const float4 a = *reinterpret_cast<const float4 *>(r0b + off), b = *reinterpret_cast<const float4 *>(r1b + off);
#if GS_ABLATE == 4
          const float4 c = make_float4(0.5f, 0.25f, 0.125f, 0.0f);
#else
          const float4 c = *reinterpret_cast<const float4 *>(r2b + off);
          asm volatile("" ::"v"(c.w));  // keep the 16-byte read: as ds_read_b96 it costs twice the LDS cycles (8 vs 4)
#endif
          const float dx = a.x - fpx, dy = a.y - fpy;
          // opa * exp(power): d alpha / d gg . gg (the reference differentiates through the 0.99 cap as if absent)
#if GS_BWD_ALPHA_ASM
          // The exponent chain of staged_alpha (the same operations in the same order: the forward's alpha bit for bit),
          // the exponential and -- in the wait state a transcendental's result needs before a plain VALU instruction may
          // read it -- the first FMA of the colour dot product, as ONE block: between separate asm statements the
          // compiler puts an s_nop per border (three per trip).
          float og, t0;
          {
            float tq, uq;
            asm("v_mul_f32 %[t], %[a2], %[dx]\n\t"
                "v_mul_f32 %[u], %[c2], %[dy]\n\t"
                "v_fmac_f32 %[t], %[b2], %[dy]\n\t"
                "v_fma_f32 %[t], %[t], %[dx], %[lopa]\n\t"
                "v_fmac_f32 %[t], %[u], %[dy]\n\t"
                "v_min_f32 %[t], %[t], %[lopa]\n\t"
                "v_exp_f32 %[og], %[t]\n\t"
                "v_fma_f32 %[t0], %[cx], %[g0], -%[s]"
                : [og] "=&v"(og), [t0] "=&v"(t0), [t] "=&v"(tq), [u] "=&v"(uq)
                : [a2] "v"(a.z), [b2] "v"(a.w), [c2] "v"(b.x), [lopa] "v"(b.y), [dx] "v"(dx), [dy] "v"(dy), [cx] "v"(c.x),
                  [g0] "v"(g0), [s] "v"(s));
          }
#else
          float og = staged_alpha(a.z, a.w, b.x, b.y, dx, dy);  // opa * exp(power), before the 0.99 cap
          const float t0 = __builtin_fmaf(c.x, g0, -s);
#endif
#if GS_EXTRA_FMA > 0
          // slope experiment (tools/experiments/r03_valu_slope.sh): GS_EXTRA_FMA independent plain FMAs per trip whose
          // results die at once -- pure issue slots, no new dependency, no live register
#pragma unroll
          for (int e = 0; e < GS_EXTRA_FMA; ++e) {
            float sink;
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(sink) : "v"(dx), "v"(dy), "v"(a.z));
          }
#endif
          // a row past the end of its list reads the sentinel record (alpha 0).  n is 0 for pixels outside the image,
          // so "inside" needs no separate test.  The 1/255 floor is tested on og itself: min(0.99, og) >= 1/255 says the
          // same (written as "not below", so that a NaN passes as it does through the reference's fminf), and the cap is
          // taken after the select -- one select per trip instead of two (r03: every VALU instruction of this loop costs
          // its full issue slot, tools/experiments/r03_valu_slope.sh).
          const bool valid = kCheckN ? (!(og < kAlphaMin) && (off < n_rel)) : !(og < kAlphaMin);
          if (__ballot(valid) == 0ull) continue;
#if GS_ABLATE == 8
          asm volatile("" ::"v"(og));
          continue;
#endif
          og = valid ? og : 0.0f;
          float alpha;  // fminf() would first canonicalise the selected value (a v_max_f32 og, og): the instruction saved
#if GS_BWD_CAP_AS_FORWARD
          // the cap the FORWARD composed with: it folds min(0.99, .) into the exponent, so a capped alpha is
          // v_exp_f32(kLog2AlphaMax), two ulps below 0.99f -- rebuilding T with the exact 0.99f would drift from the
          // forward's T by 1.5e-5 relative per capped splat; the same value from the same instruction here
          asm("v_min_f32 %0, %2, %1" : "=v"(alpha) : "v"(og), "s"(alpha_cap));
#else
          asm("v_min_f32 %0, 0x3f7d70a4, %1" : "=v"(alpha) : "v"(og));  // min(0.99f, og), kAlphaMax
          static_assert(kAlphaMax == 0.99f, "the literal above is 0.99f");
#endif
          const float inv = __builtin_amdgcn_rcpf(1.0f - alpha);
          T *= inv;                                           // transmittance in front of this splat
          const float aT = alpha * T;
          // t = grad . (colour - colour behind): three FMAs on the carried dot product
          const float t = __builtin_fmaf(c.z, g2, __builtin_fmaf(c.y, g1, t0));
          const float ga = t * T;                             // d/d alpha (cuda/render_backward.cu:139-151)
          s = __builtin_fmaf(alpha, t, s);                    // grad . colour behind the next (nearer) splat
          const float gp = og * ga;                           // d/d power
          // nine raw sums: aT x pixel gradient (d/d rgb) and the six moments of gp about the tile centre; signs, the
          // -1/2 factors, (1 - opa), the shift to the gaussian's centre and 0.5*W / 0.5*H are applied once per gaussian
          // at flush time
#if GS_ABLATE == 2
          asm volatile("" ::"v"(aT), "v"(gp));
#else
#if GS_ROWSUM_QUAD == 2
          unsigned int acc_addr;
          const float red = row_moments9r(aT, gp, rw, (unsigned int)off, acc_lane, acc_addr);
#elif GS_ROWSUM_QUAD
          const float red = row_moments9q(aT, gp, rw);
          const unsigned int acc_addr = acc_lane + __umul24((unsigned int)off, 5u);
#else
          const float red = row_moments9(aT, gp, rw);
          const unsigned int acc_addr = acc_lane + __umul24((unsigned int)off, 5u);
#endif
#if GS_ABLATE == 1
          asm volatile("" ::"v"(red), "v"(acc_addr));
#else
          // Every lane that holds a sum adds it, zero or not (a compare to skip zeros costs more issue cycles than the
          // few extra lanes cost the LDS; rows past their list add zeros to the sentinel slot's accumulators).  32-bit
          // LDS address on purpose: through a generic pointer the compiler forms off * 5 with a 64-bit multiply-add.
          if (red_lane)
            __hip_atomic_fetch_add(reinterpret_cast<__attribute__((address_space(3))) double *>(acc_addr),
                                   (double)red, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#endif
#endif
        }
      };
      // The trip loop issues at a raised priority (r04): a SIMD's vector issue goes to the highest priority first, then
      // to the oldest wave, so the waves inside their loops -- the ones whose instructions bound the kernel -- are served
      // before the waves that stage, build lists or flush next to them (levels 1, 2 and 3 measure the same: 0.2820 ->
      // 0.2745 ms; the same in the forward's loop changed nothing).
#if GS_BWD_PRIO
      __builtin_amdgcn_s_setprio(GS_BWD_PRIO);
#endif
      if (__any(n_rel < count * 16)) run_trips(std::true_type{});
      else run_trips(std::false_type{});
#if GS_BWD_PRIO
      __builtin_amdgcn_s_setprio(0);
#endif
      GS_LAP(st_loop);
    }
    __syncthreads();
    GS_LAP(st_bar2);
    prefetch(base - kB, t);  // waves 2, 3: the next batch's loads are in flight while waves 0, 1 run the flush's first step
    // flush, step 1: one thread per gaussian turns its nine raw sums into the nine gradient values (uniform control
    // flow, the double arithmetic once per gaussian instead of once per lane of a 16-lane group)
    if (t < count) {
      const double *acc = &s_acc[t * kAcc];  // rgb3, S_1, S_cx, S_cy, S_cx2, S_cxcy, S_cy2
      const float4 a = s_r0[t], b = s_r1[t];  // staged conic: see stage_record
      const float opa = b.z;
      // moments of gp about the gaussian's centre from those about the tile centre: d = (X, Y) - (cx, cy)
      const double X = (double)(a.x - (tx0 + 7.5f)), Y = (double)(a.y - (ty0 + 7.5f));
      const double S1 = acc[3], Sx = acc[4], Sy = acc[5];
      const float sx = (float)(X * S1 - Sx), sy = (float)(Y * S1 - Sy);              // sum gp dx, sum gp dy
      const float sxx = (float)(X * (X * S1 - 2.0 * Sx) + acc[6]);                   // sum gp dx^2
      const float sxy = (float)(X * (Y * S1 - Sy) - Y * Sx + acc[7]);                // sum gp dx dy
      const float syy = (float)(Y * (Y * S1 - 2.0 * Sy) + acc[8]);                   // sum gp dy^2
      const float ca = a.z * kConicDiag, cb = a.w * kConicOff, cc = b.x * kConicDiag;
      // cuda/render_backward.cu:170 gates ALL nine adds on any(d/d logit != 0), d/d logit = sum gp * (1 - opa) per
      // thread: a fully opaque gaussian (sigmoid(opacity) == 1) gets no gradient at all, and neither does one whose gp
      // is zero on every pixel of the tile (d/d alpha exactly 0: e.g. a zero pixel gradient, or a colour equal to the
      // colour behind it over a zero background) -- its colour sums may be non-zero and are dropped with the rest.
      // "gp zero everywhere" is read off the six moments (sums of zeros are exact zeros; non-zero gp that cancel in all
      // six at once do not occur).
      const bool any_gp = (S1 != 0.0) | (Sx != 0.0) | (Sy != 0.0) | (acc[6] != 0.0) | (acc[7] != 0.0) | (acc[8] != 0.0);
      const float keep = (opa == 1.0f || !any_gp) ? 0.0f : 1.0f;
      float *res = &s_res[t * 9];
      res[0] = keep * (float)acc[0];
      res[1] = keep * (float)acc[1];
      res[2] = keep * (float)acc[2];
      res[3] = keep * ((float)S1 * (1.0f - opa));                                    // d/d logit (render_backward.cu:154)
      res[4] = keep * (-0.5f * sxx);                                                 // conic00
      res[5] = keep * -sxy;                                                          // conic01
      res[6] = keep * (-0.5f * syy);                                                 // conic11
      res[7] = keep * (-(ca * sx + cb * sy) * (0.5f * (float)width));                // u (render_backward.cu:180-186)
      res[8] = keep * (-(cc * sy + cb * sx) * (0.5f * (float)height));               // v (:181-187)
    }
    GS_LAP(st_flush);
    __syncthreads();
    GS_LAP(st_bar3);
    // step 2: 16 lanes per gaussian -> each wave instruction touches four whole 64-byte rows
    const int k = t & 15;
    if (k < 9) {
#pragma unroll 4
      for (int r = 0; r < (kB + 15) / 16; ++r) {
        const int slot = r * 16 + (t >> 4);
        if (slot >= count) continue;
        const float val = s_res[slot * 9 + k];
        if (!(val != 0.0f)) continue;  // zero (nothing to add) -- NaN still goes out
#if GS_ABLATE == 3
        asm volatile("" ::"v"(val));
        continue;
#endif
        const int g = s_id[slot];
        if constexpr (kRows) {
          atomicAdd(&out.rows[(size_t)g * 16 + k], val);
        } else {
          float *dst = k < 3 ? &out.rgb[3 * (size_t)g + k]
                             : (k == 3 ? &out.opacity[g]
                                       : (k < 7 ? &out.conic[3 * (size_t)g + (k - 4)] : &out.uv[2 * (size_t)g + (k - 7)]));
          atomicAdd(dst, val);
        }
      }
    }
  }
#if GS_STAMP
  GS_LAP(st_flush);
  st_write();
#endif
}

// ---- r06 experiment: the backward with TWO half-batches in flight ("ping-pong").  A batch of the
// kernel above costs four barriers: (0) flush step 2 done -> staging, (1) staging done -> lists + trips, (2) trips done ->
// flush step 1, (3) step 1 done -> step 2; the stamps put 20 % of the wave cycles at those barriers and 24 % in staging
// and flush, during which the trip loop -- the part that bounds the kernel -- does not run in this workgroup.  Here the
// list is walked in halves of 62 slots with two sets of record arrays and accumulators: while all four waves run the
// trips of half p on set X = p & 1, ONE wave (p & 3: the duty rotates) first deals with set 1 - X -- it takes half p-1's
// nine sums per gaussian into registers (one lane per slot), clears them, parks the records of half p+1 (which it
// requested a phase earlier) in their place, converts the sums and sends the nine atomics straight from its registers (no
// second step, no parking of the values in LDS) -- and then joins the trips.  The next duty wave requests the records of
// half p+2 (list entries at the top of the phase, records once its row lists are built: no wait for a dependent round
// trip in front of the trips).  ONE barrier per half = two per 124 slots instead of four, and nothing between them that
// the other three waves wait for.  Fused path only (packed records, gradient rows); same arithmetic in the same order as
// the kernel above per (pixel, gaussian) -- the sums of a gaussian are still merged by LDS atomics in arrival order.
#ifndef GS_BWD_PINGPONG
#define GS_BWD_PINGPONG 0  // launch_render_bwd's default for the fused path; GSPLAT_BWD_PINGPONG=0|1 overrides at run time
#endif
#ifndef GS_PP_DUTY_PRIO
#define GS_PP_DUTY_PRIO 2
#endif
constexpr int kHalf = 62;
static_assert(kSegEntries % kHalf == 0, "a segment is a whole number of half-batches");
static_assert(GS_ROWSUM_QUAD == 2 && GS_BWD_ALPHA_ASM == 1 && GS_BWD_CAP_AS_FORWARD == 1, "the ping-pong kernel exists in the default loop form only");

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void render_bwd_pp_kernel(
    const float4 *__restrict__ recs, const int *__restrict__ sorted, const int *__restrict__ ranges, const int *__restrict__ n_px,
    const float *__restrict__ T_px, const float *__restrict__ grad_image, int width, int height, int ntx, int num_tiles, float bg,
    float *__restrict__ rows_out, const unsigned short *__restrict__ masks_in, const int *__restrict__ order, TileSegments seg) {
  constexpr int kS = kHalf + 1;  // slots of one set: 62 + the all-zero sentinel
  constexpr int kAcc = 10;       // doubles per slot (nine used), as above
  __shared__ float4 s_r0[2 * kS], s_r1[2 * kS], s_r2[2 * kS];
  __shared__ double s_acc[2 * kS * kAcc];
  __shared__ int s_id[2 * kHalf];
  __shared__ __attribute__((aligned(16))) unsigned short s_list[16 * kHalf];  // [wave][row][62]
  __shared__ int s_top;
  int tile, seg_a = 0, seg_end = -1, chk_slot = -1;
  if (seg.chk && (int)blockIdx.x < seg.extra_cap) {
    const int e = (int)blockIdx.x;
    if (e >= *seg.extra_count) return;
    const int2 ex = seg.extra[e];
    tile = ex.x;
    seg_a = ex.y * kSegEntries;
    seg_end = seg_a + kSegEntries;
    chk_slot = segment_slot(ranges[tile], ex.y + 1);
  } else {
    tile = ordered_tile(order, seg.chk ? (int)blockIdx.x - seg.extra_cap : (int)blockIdx.x, num_tiles);
    if (tile >= num_tiles) return;
    if (seg.chk && seg.granted[tile] > 0) { seg_end = kSegEntries; chk_slot = segment_slot(ranges[tile], 1); }
  }
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, row = lane >> 4, j = lane & 15;
#if GS_STAMP
  const unsigned long long st_t0 = GS_NOW(), st_rt0 = __builtin_amdgcn_s_memrealtime();
  unsigned long long st_last = st_t0, st_bar = 0, st_stage = 0, st_lists = 0, st_loop = 0, st_flush = 0, st_trips = 0, st_batches = 0;
  auto st_write = [&]() {
    if (lane == 0) {
      unsigned long long *o = gs_stamp_buf + ((size_t)blockIdx.x * 4 + wave) * GS_STAMP_WORDS;
      o[0] = (unsigned long long)tile; o[1] = st_rt0; o[2] = __builtin_amdgcn_s_memrealtime(); o[3] = GS_NOW() - st_t0;
      o[4] = st_bar; o[5] = st_stage; o[6] = st_lists; o[7] = st_loop; o[8] = st_flush; o[9] = st_trips; o[10] = st_batches;
      unsigned int xcc;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      o[11] = xcc; o[12] = 0; o[13] = 0; o[14] = 0; o[15] = (unsigned long long)wave;
    }
  };
#endif
  const int tile_x = tile % ntx, tile_y = tile / ntx;
  const int px = tile_x * 16 + (wave & 1) * 8 + (row & 1) * 4 + (j & 3);
  const int py = tile_y * 16 + (wave >> 1) * 8 + (row >> 1) * 4 + (j >> 2);
  const bool inside = px < width && py < height;
  const float fpx = (float)px, fpy = (float)py;
  const int start = ranges[tile] + seg_a;

  int n = 0;
  float Tf = 0.0f, g0 = 0.0f, g1 = 0.0f, g2 = 0.0f;
  if (inside) {
    const int pid = py * width + px;
    n = n_px[pid];
    Tf = T_px[pid];
    g0 = grad_image[3 * pid]; g1 = grad_image[3 * pid + 1]; g2 = grad_image[3 * pid + 2];
  }
  float T = Tf, s = bg * g0 + bg * g1 + bg * g2;  // (see render_bwd_kernel: the carried dot product, the background behind all)
  if (seg_end >= 0) {
    if (n > seg_end) {
      const int pid = py * width + px;
      const float4 ck = seg.chk[(size_t)chk_slot * 256 + tid];
      T = ck.x;
      const float inv = __builtin_amdgcn_rcpf(ck.x);
      s = (g0 * (seg.image[3 * pid] - ck.y) + g1 * (seg.image[3 * pid + 1] - ck.z) + g2 * (seg.image[3 * pid + 2] - ck.w)) * inv;
      n = seg_end - seg_a;
    } else {
      n = max(n - seg_a, 0);
    }
  }
  const int row_top_v = row_max_int(n);
  const int rt0 = __builtin_amdgcn_readlane(row_top_v, 0), rt1 = __builtin_amdgcn_readlane(row_top_v, 16);
  const int rt2 = __builtin_amdgcn_readlane(row_top_v, 32), rt3 = __builtin_amdgcn_readlane(row_top_v, 48);
  const int wave_top = max(max(rt0, rt1), max(rt2, rt3));
  if (tid < 2) {  // the two sets' sentinel records
    s_r0[tid * kS + kHalf] = s_r2[tid * kS + kHalf] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    s_r1[tid * kS + kHalf] = sentinel_r1();
  }
  if (tid == 0) s_top = 0;
  for (int k = tid; k < 2 * kS * kAcc; k += 256) s_acc[k] = 0.0;
  __syncthreads();
  if (lane == 0) atomicMax(&s_top, wave_top);
  __syncthreads();
  const int top = s_top;
#if GS_STAMP
  GS_LAP(st_stage);
  if (top <= 0) { st_write(); return; }
#endif
  if (top <= 0) return;
  const int red_idx = row_moments9r_index(lane);
  const bool red_lane = red_idx >= 0;
  const float alpha_cap = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, __builtin_amdgcn_exp2f(kLog2AlphaMax))));
  const unsigned int acc0 = (unsigned int)(size_t)(__attribute__((address_space(3))) char *)reinterpret_cast<char *>(s_acc);

  const int P = (top + kHalf - 1) / kHalf;  // half p covers the entries [(P-1-p) 62, ...): back to front
  // what a duty wave holds for the half it will bring in: slot `lane`'s list entry and block mask (requested a phase
  // earlier).  The 48-byte records themselves never pass through registers: twelve more values per lane, live across a
  // trip loop, are more than the kernel's 64 registers hold, and a spilled load result is first WAITED for -- they go from
  // HBM straight into the set's record arrays (global_load_lds_dwordx4: lane l's 16 bytes land in slot l) while the
  // duty wave runs its own trips, and are put into the loop's form in place at the end of the phase (finish_records).
  int pre_g = 0;
  unsigned int pre_mask = 0;
  // `lo`: the lane index behind an opaque per-phase copy (see render_bwd_kernel: addresses and constants derived from the
  // plain index are loop invariants, get hoisted above the phase loop and spilled -- and a reload from scratch in the duty
  // wave waits for its atomics)
  auto request_entry = [&](int p, int lo) {
    const int b = (P - 1 - p) * kHalf;
    if (lo < min(kHalf, top - b)) {
      pre_mask = (unsigned int)masks_in[start + b + lo];
      pre_g = sorted[start + b + lo];
    }
  };
  typedef __attribute__((address_space(3))) void *lptr_t;
  auto request_records = [&](int p, int lo) {  // half p's records -> set p & 1, asynchronously
    const int b = (P - 1 - p) * kHalf, x = p & 1;
    if (lo < min(kHalf, top - b)) {
      // (as asm: through the builtin the compiler knows a write to LDS is pending and waits for it -- vmcnt(0) -- in front of
      // this wave's next LDS access of any kind, i.e. at once; the wait that counts is finish_records' own)
      const float4 *src = recs + 3 * (size_t)pre_g;
      const unsigned int d0 = (unsigned int)(size_t)(lptr_t)(s_r0 + x * kS), d1 = (unsigned int)(size_t)(lptr_t)(s_r1 + x * kS);
      const unsigned int d2 = (unsigned int)(size_t)(lptr_t)(s_r2 + x * kS);
      unsigned int keep;
      asm volatile("s_mov_b32 %0, m0\n\t"
                   "s_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
                   "s_mov_b32 m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\t"
                   "s_mov_b32 m0, %6\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, off\n\t"
                   "s_mov_b32 m0, %0"
                   : "=&s"(keep)
                   : "v"(src + 0), "v"(src + 1), "v"(src + 2), "s"(d0), "s"(d1), "s"(d2)
                   : "memory");
      s_id[x * kHalf + lo] = pre_g;
    }
  };
  auto finish_records = [&](int p, int lo) {  // once they have landed: the loop's form (stage_record) and the block mask
    const int b = (P - 1 - p) * kHalf, x = p & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lo < min(kHalf, top - b)) {
      SplatRec r;
      r.r0 = s_r0[x * kS + lo]; r.r1 = s_r1[x * kS + lo]; r.r2 = make_float4(0.f, 0.f, 0.f, 0.f);
      stage_record(r);
      s_r0[x * kS + lo] = r.r0; s_r1[x * kS + lo] = r.r1;
      s_r2[x * kS + lo].w = __uint_as_float(pre_mask);
    }
  };
  // one lane per slot: half q's nine sums -> nine gradient values -> nine atomics; in between the set is handed on
  // (sums cleared, the next half's records parked: `then`)
  auto flush_half = [&](int q, int lo, auto then) {
    const int b = (P - 1 - q) * kHalf, x = q & 1, cnt = min(kHalf, top - b);
    const bool mine = lo < cnt;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), bb = a;
    int g = 0;
    if (mine) {
      a = s_r0[x * kS + lo]; bb = s_r1[x * kS + lo];
      g = s_id[x * kHalf + lo];
    }
    // (same wave, LDS in order: the reads above return before the records are overwritten; the sums are read only after
    // the requested records have left their registers -- both at once do not fit the kernel's 64)
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    then();
    if (mine) {
      const double *acc = &s_acc[(x * kS + lo) * kAcc];
      const double a0 = acc[0], a1 = acc[1], a2 = acc[2], S1 = acc[3], Sx = acc[4], Sy = acc[5], Sxx = acc[6], Sxy = acc[7], Syy = acc[8];
      int txo = tile_x, tyo = tile_y, wo = width, ho = height;  // (opaque: see `lo`)
      asm volatile("" : "+s"(txo), "+s"(tyo), "+s"(wo), "+s"(ho));
      const float opa = bb.z;
      const double X = (double)(a.x - ((float)(txo * 16) + 7.5f)), Y = (double)(a.y - ((float)(tyo * 16) + 7.5f));
      const float sx = (float)(X * S1 - Sx), sy = (float)(Y * S1 - Sy);
      const float sxx = (float)(X * (X * S1 - 2.0 * Sx) + Sxx);
      const float sxy = (float)(X * (Y * S1 - Sy) - Y * Sx + Sxy);
      const float syy = (float)(Y * (Y * S1 - 2.0 * Sy) + Syy);
      const float ca = a.z * kConicDiag, cb = a.w * kConicOff, cc = bb.x * kConicDiag;
      const bool any_gp = (S1 != 0.0) | (Sx != 0.0) | (Sy != 0.0) | (Sxx != 0.0) | (Sxy != 0.0) | (Syy != 0.0);
      const float keep = (opa == 1.0f || !any_gp) ? 0.0f : 1.0f;  // cuda/render_backward.cu:170 (see the kernel above)
      // the nine values (and the gaussian) wait in the slot's own -- now dead -- accumulators for the transposed pass below
      float *res = reinterpret_cast<float *>(&s_acc[(x * kS + lo) * kAcc]);
      res[0] = keep * (float)a0;
      res[1] = keep * (float)a1;
      res[2] = keep * (float)a2;
      res[3] = keep * ((float)S1 * (1.0f - opa));
      res[4] = keep * (-0.5f * sxx);
      res[5] = keep * -sxy;
      res[6] = keep * (-0.5f * syy);
      res[7] = keep * (-(ca * sx + cb * sy) * (0.5f * (float)wo));
      res[8] = keep * (-(cc * sy + cb * sx) * (0.5f * (float)ho));
      res[9] = __int_as_float(g);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // 16 lanes per gaussian: each wave instruction adds into four whole 64-byte gradient rows (one lane per VALUE of one
    // gaussian sends nine times as many requests to the L2's atomic units: 1.04 ms instead of 0.28, measured)
    const int k = lo & 15;
#pragma unroll 4
    for (int r = 0; r < (kHalf + 3) / 4; ++r) {
      const int slot = 4 * r + (lo >> 4);
      if (slot < cnt && k < 9) {
        const float *res = reinterpret_cast<const float *>(&s_acc[(x * kS + slot) * kAcc]);
        const float val = res[k];
        const int gg = __float_as_int(res[9]);
        if (val != 0.0f) atomicAdd(rows_out + (size_t)gg * 16 + k, val);  // zero: nothing to add -- NaN still goes out
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (mine) {
      double *acc = &s_acc[(x * kS + lo) * kAcc];
#pragma unroll
      for (int kk = 0; kk < kAcc; ++kk) acc[kk] = 0.0;
    }
  };

  // (every value loaded so far is taken here: a wait for them placed by the compiler inside the trip loop would also wait for
  // the duty wave's atomics and requests, which are younger)
  asm volatile("" : "+v"(T), "+v"(s), "+v"(g0), "+v"(g1), "+v"(g2), "+v"(n));
  // before the first phase: wave 3 brings half 0 in by itself (the one exposed round trip), wave 0 asks for half 1's entries
  if (wave == 3) { request_entry(0, lane); request_records(0, lane); finish_records(0, lane); }
  if (wave == 0 && P > 1) request_entry(1, lane);
  for (int p = 0; p < P; ++p) {
    const int base = (P - 1 - p) * kHalf, count = min(kHalf, top - base), x = p & 1;
#if GS_STAMP
    ++st_batches;
    GS_LAP(st_flush);
#endif
    __syncthreads();
    GS_LAP(st_bar);
    int t = tid;
    asm volatile("" : "+v"(t));
    const int lo = t & 63, wv = __builtin_amdgcn_readfirstlane(t >> 6);
    if (wv == (p & 3)) {  // this phase's duty wave: the other set
#if GS_PP_DUTY_PRIO
      __builtin_amdgcn_s_setprio(GS_PP_DUTY_PRIO);  // (its workgroup's three other waves will wait for this one)
#endif
      if (p > 0) flush_half(p - 1, lo, [&]() { if (p + 1 < P) request_records(p + 1, lo); });
      else if (p + 1 < P) request_records(p + 1, lo);
#if GS_PP_DUTY_PRIO
      __builtin_amdgcn_s_setprio(0);
#endif
    }
    GS_LAP(st_stage);
    const bool asks = wv == ((p + 1) & 3) && p + 2 < P;
    if (asks) request_entry(p + 2, lo);
    const bool active = base < wave_top;
    unsigned int list_lds = 0;
    int trips = 0;
    if (active) {
      unsigned short *lists = s_list + wv * 4 * kHalf;
      list_lds = (unsigned int)(size_t)(__attribute__((address_space(3))) const unsigned short *)(lists + (lo >> 4) * kHalf);
      const RowCounts rc = build_row_lists<kHalf>(s_r2 + x * kS, lists, count, wv, lo, rt0 - base, rt1 - base, rt2 - base, rt3 - base, 1, x * kS);
      trips = max(max(rc.c0, rc.c1), max(rc.c2, rc.c3));
    }
#if GS_STAMP
    st_trips += trips;
    GS_LAP(st_lists);
#endif
    if (active) {
      // (a list entry is the slot's byte offset in the record arrays, the set's first slot included: constant bases)
      const int n_rel = (n - base + x * kS) * 16;
      const char *r0x = reinterpret_cast<const char *>(s_r0), *r1x = reinterpret_cast<const char *>(s_r1);
      const char *r2x = reinterpret_cast<const char *>(s_r2);
      const unsigned int acc_lane = acc0 + (unsigned int)(red_idx * 8);
      float g0b = g0, g1b = g1, g2b = g2;
      asm volatile("" : "+v"(g0b), "+v"(g1b), "+v"(g2b));
      const RowsWeights rw = make_rows_weights(t & 63, (float)(((t >> 6) & 1) * 8 + ((t >> 4) & 1) * 4 + (t & 3)) - 7.5f,
                                               (float)((t >> 7) * 8 + ((t >> 5) & 1) * 4 + ((t >> 2) & 3)) - 7.5f, g0b, g1b, g2b);
      auto run_trips = [&](auto check_n) {
        constexpr bool kCheckN = decltype(check_n)::value;
        for (int i = trips - 1; i >= 0; --i) {
          int off;
          asm volatile("ds_read_u16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(off) : "v"(list_lds + 2 * i) : "memory");
          const float4 a = *reinterpret_cast<const float4 *>(r0x + off), b = *reinterpret_cast<const float4 *>(r1x + off);
          const float4 c = *reinterpret_cast<const float4 *>(r2x + off);
          asm volatile("" ::"v"(c.w));
          const float dx = a.x - fpx, dy = a.y - fpy;
          float og, t0;
          {
            float tq, uq;
            asm("v_mul_f32 %[t], %[a2], %[dx]\n\t"
                "v_mul_f32 %[u], %[c2], %[dy]\n\t"
                "v_fmac_f32 %[t], %[b2], %[dy]\n\t"
                "v_fma_f32 %[t], %[t], %[dx], %[lopa]\n\t"
                "v_fmac_f32 %[t], %[u], %[dy]\n\t"
                "v_min_f32 %[t], %[t], %[lopa]\n\t"
                "v_exp_f32 %[og], %[t]\n\t"
                "v_fma_f32 %[t0], %[cx], %[g0], -%[s]"
                : [og] "=&v"(og), [t0] "=&v"(t0), [t] "=&v"(tq), [u] "=&v"(uq)
                : [a2] "v"(a.z), [b2] "v"(a.w), [c2] "v"(b.x), [lopa] "v"(b.y), [dx] "v"(dx), [dy] "v"(dy), [cx] "v"(c.x),
                  [g0] "v"(g0), [s] "v"(s));
          }
          const bool valid = kCheckN ? (!(og < kAlphaMin) && (off < n_rel)) : !(og < kAlphaMin);
          if (__ballot(valid) == 0ull) continue;
          og = valid ? og : 0.0f;
          float alpha;
          asm("v_min_f32 %0, %2, %1" : "=v"(alpha) : "v"(og), "s"(alpha_cap));
          const float inv = __builtin_amdgcn_rcpf(1.0f - alpha);
          T *= inv;
          const float aT = alpha * T;
          const float tt = __builtin_fmaf(c.z, g2, __builtin_fmaf(c.y, g1, t0));
          const float ga = tt * T;
          s = __builtin_fmaf(alpha, tt, s);
          const float gp = og * ga;
          unsigned int acc_addr;
          const float red = row_moments9r(aT, gp, rw, (unsigned int)off, acc_lane, acc_addr);
          if (red_lane)
            __hip_atomic_fetch_add(reinterpret_cast<__attribute__((address_space(3))) double *>(acc_addr),
                                   (double)red, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
      };
#if GS_BWD_PRIO
      __builtin_amdgcn_s_setprio(GS_BWD_PRIO);
#endif
      if (__any(n_rel < (count + x * kS) * 16)) run_trips(std::true_type{});
      else run_trips(std::false_type{});
#if GS_BWD_PRIO
      __builtin_amdgcn_s_setprio(0);
#endif
    }
    GS_LAP(st_loop);
    if (wv == (p & 3) && p + 1 < P) finish_records(p + 1, lo);
    GS_LAP(st_stage);
  }
  __syncthreads();
  GS_LAP(st_bar);
  if (wave == (P & 3)) flush_half(P - 1, lane, []() {});
#if GS_STAMP
  GS_LAP(st_flush);
  st_write();
#endif
}

// Deals every XCD's run of tiles heaviest first (r04).  The hardware starts workgroups in block order, so with the plain
// map the launch ends on whatever tiles sit at the end of the runs, and its last round lasts as long as the heaviest of
// them; heaviest first, the last round is made of the lightest tiles of the image.  `work[t]`: list length (forward) or
// the largest stop index of the tile's pixels (backward).  One workgroup per XCD run (at most kOrderMaxRun tiles): a
// counting sort on 64 classes of work relative to the run's maximum, all in LDS -- the order inside a class does not
// matter.  order[x * per_xcd + k] = the k-th heaviest tile of run x; padding slots get num_tiles.
__global__ __launch_bounds__(1024) void tile_order_kernel(const int *__restrict__ work, const int *__restrict__ ranges,
                                                          int num_tiles, int *__restrict__ order) {
  constexpr int kClasses = 64, kPer = kOrderMaxRun / 1024;
  __shared__ int s_max, s_count[kClasses], s_base[kClasses];
  const int per_xcd = (num_tiles + 7) >> 3, first = blockIdx.x * per_xcd;
  if (threadIdx.x < kClasses) s_count[threadIdx.x] = 0;
  if (threadIdx.x == 0) s_max = 0;
  __syncthreads();
  int w[kPer], cls[kPer];
#pragma unroll
  for (int k = 0; k < kPer; ++k) {
    const int t = first + threadIdx.x + k * 1024;
    const bool in = threadIdx.x + k * 1024 < per_xcd && t < num_tiles;
    w[k] = in ? (work ? work[t] : ranges[t + 1] - ranges[t]) : -1;
  }
  int m = 0;
#pragma unroll
  for (int k = 0; k < kPer; ++k) m = max(m, w[k]);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = max(m, __shfl_xor(m, off, 64));
  if ((threadIdx.x & 63) == 0) atomicMax(&s_max, m);
  __syncthreads();
  const float scale = (float)kClasses / (float)(s_max + 1);
#pragma unroll
  for (int k = 0; k < kPer; ++k) {
    // class 0 = the heaviest; padding slots (w < 0) behind everything
    cls[k] = w[k] < 0 ? -1 : kClasses - 1 - min(kClasses - 1, (int)((float)w[k] * scale));
    if (cls[k] >= 0) atomicAdd(&s_count[cls[k]], 1);
  }
  __syncthreads();
  if (threadIdx.x < kClasses) {  // exclusive scan of the 64 class counts by one wave
    const int c = s_count[threadIdx.x];
    int incl = c;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int u = __shfl_up(incl, off, 64);
      if ((int)threadIdx.x >= off) incl += u;
    }
    s_base[threadIdx.x] = incl - c;
    s_count[threadIdx.x] = 0;  // becomes the class cursor
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < kPer; ++k)
    if (cls[k] >= 0) order[first + s_base[cls[k]] + atomicAdd(&s_count[cls[k]], 1)] = first + threadIdx.x + k * 1024;
  // padding slots of the run: the tiles of the run come first (their count = the sum of the class counts)
  __syncthreads();
  int total = 0;
  if (threadIdx.x < 64) {
    int c = s_count[threadIdx.x];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off, 64);
    total = c;
    if (threadIdx.x == 0) s_max = total;
  }
  __syncthreads();
  total = s_max;
  for (int k = total + threadIdx.x; k < per_xcd; k += 1024) order[first + k] = num_tiles;
}

int launch_tile_order(const int *work, const int *ranges, int num_tiles, int *order, hipStream_t st) {
  if (((num_tiles + 7) >> 3) > kOrderMaxRun) return GSPLAT_ERR_INVALID_ARG;  // (callers check: no order table then)
  tile_order_kernel<<<8, 1024, 0, st>>>(work, ranges, num_tiles, order);
  GS_LAUNCH_CHECK();
  return GSPLAT_OK;
}
bool tile_order_supported(int num_tiles) { return ((num_tiles + 7) >> 3) <= kOrderMaxRun; }

// r05: which segments the backward walks with a workgroup of their own (gs_render.h: TileSegments), from the forward's
// largest stop index per tile.  ONE workgroup walks the tiles in index order, 1024 at a time, with an exclusive scan of the
// further segments each tile asks for: the table is the same in every run (an atomic counter per tile serialised ~1500
// requests on one L2 line: 13.7 us).  A list whose segments do not fit into the launch's room stays whole, and so does
// every list behind it; `asked` tells the host what would have been needed.  All of a thread's tiles are loaded up front
// (at most kTableChunks = 16 of them: the counting-sort route ends at 16 384 tiles): chunk by chunk the kernel was five
// global round trips long, 12 us behind every forward.
constexpr int kTableChunks = 16;
__device__ __forceinline__ int sum_below(const int *s_wave, int wave) {  // of the 16 per-wave totals, those of the waves in front
  const int4 *v = reinterpret_cast<const int4 *>(s_wave);
  const int4 a = v[0], b = v[1], c = v[2], d = v[3];
  const int w[16] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w, d.x, d.y, d.z, d.w};
  int sum = 0;
#pragma unroll
  for (int k = 0; k < 16; ++k) sum += k < wave ? w[k] : 0;
  return sum;
}
__global__ __launch_bounds__(1024) void tile_segments_kernel(const int *__restrict__ ranges, const int *__restrict__ tops,
                                                             int num_tiles, TileSegments seg) {
  __shared__ __attribute__((aligned(16))) int s_wave[16];
  __shared__ int s_base, s_count, s_max, s_sum;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) s_base = s_count = s_max = s_sum = 0;
  int len[kTableChunks], top[kTableChunks];
#pragma unroll
  for (int c = 0; c < kTableChunks; ++c) {
    const int t = c * 1024 + tid;
    len[c] = top[c] = 0;
    if (t < num_tiles) {
      len[c] = ranges[t + 1] - ranges[t];
      top[c] = tops[t];
    }
  }
  __syncthreads();
  if (seg.stats) {  // how uneven the tiles' work is: decides whether the next forward splits its long lists (gs_fused.hip)
    int mx = 0, sum = 0;
#pragma unroll
    for (int c = 0; c < kTableChunks; ++c) {
      const int reach = min(top[c], len[c]);
      mx = max(mx, reach);
      sum += reach;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      mx = max(mx, __shfl_xor(mx, off, 64));
      sum += __shfl_xor(sum, off, 64);
    }
    if (lane == 0) { atomicMax(&s_max, mx); atomicAdd(&s_sum, sum); }
    __syncthreads();
    if (tid == 0) { seg.stats[0] = tagged_figure(seg.tag, s_max); seg.stats[1] = tagged_figure(seg.tag, s_sum); }
  }
  if (seg.granted == nullptr) return;  // (a render-only context: the figures only)
#pragma unroll
  for (int c = 0; c < kTableChunks; ++c) {
    if (c * 1024 >= num_tiles) break;
    const int t = c * 1024 + tid;
    const int reach = len[c] > kSegSplitMin ? min(top[c], len[c]) : 0;  // (the forward stores checkpoints for exactly these lists)
    const int want = reach > kSegEntries ? (reach + kSegEntries - 1) / kSegEntries - 1 : 0;
    int incl = want;  // inclusive scan over the wave, then over the 16 waves
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int up = __shfl_up(incl, off, 64);
      if (lane >= off) incl += up;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    const int before = s_base + sum_below(s_wave, wave);
    const int pos = before + incl - want;
    int granted = 0;
    if (want > 0 && pos + want <= seg.extra_cap) {
      granted = want;
      for (int k = 1; k <= want; ++k) seg.extra[pos + k - 1] = make_int2(t, k);
      atomicMax(&s_count, pos + want);  // (the lists that fit are a prefix of the tiles: pos only grows)
    }
    if (t < num_tiles) seg.granted[t] = granted;
    __syncthreads();
    if (tid == 1023) s_base = before + incl;
    __syncthreads();
  }
  if (tid == 0) {
    *seg.extra_count = s_count;
    if (seg.asked) *seg.asked = tagged_figure(seg.tag, s_base);
  }
}

int launch_tile_segments(const int *ranges, const int *tops, int num_tiles, const TileSegments &seg, hipStream_t st) {
  tile_segments_kernel<<<1, 1024, 0, st>>>(ranges, tops, num_tiles, seg);
  GS_LAUNCH_CHECK();
  return GSPLAT_OK;
}

// r05: the forward's segment blocks (gs_render.h: FwdSegments), by list length, in front of the forward.  The blocks are
// listed LAYER by layer -- all segments 0, then all segments 1, ... -- so that a segment starts when the segments in front
// of it are through wherever the layers fill the chip (no product pass then, and nothing at all for a segment nobody
// reaches), and side by side with them in the thin layers of the few longest lists.  With the long lists ranked by their
// number of segments m, longest first, layer k is the tiles of rank < L_k (L_k: lists with more than k segments): segment
// (t, k) is block base_k + rank_t, base_k = L_0 + ... + L_(k-1) -- one counting sort of the tiles by m in one workgroup's
// LDS.  A list of more than kFwdMaxLayers segments keeps its one block; if the blocks do not fit into the launch's room
// nothing is split (the next launch's room follows `asked`).
constexpr int kFwdMaxLayers = 128;
__global__ __launch_bounds__(1024) void fwd_segments_table_kernel(const int *__restrict__ ranges, int num_tiles, FwdSegments fs) {
  // [k]: lists of exactly k segments -> lists of more than k segments (suffix sums) | first block of layer k (prefix sums)
  __shared__ int s_more[kFwdMaxLayers + 2], s_base[kFwdMaxLayers + 2], s_cursor[kFwdMaxLayers + 2], s_tmp[kFwdMaxLayers + 2];
  const int tid = threadIdx.x;
  if (tid < kFwdMaxLayers + 2) s_more[tid] = s_cursor[tid] = 0;
  int m[kTableChunks];  // all of the thread's tiles up front (see tile_segments_kernel)
#pragma unroll
  for (int c = 0; c < kTableChunks; ++c) {
    const int t = c * 1024 + tid;
    m[c] = 0;
    if (t < num_tiles) {
      const int len = ranges[t + 1] - ranges[t];
      const int segs = len > kSegSplitMin ? (len + kSegEntries - 1) / kSegEntries : 0;
      m[c] = segs > kFwdMaxLayers ? 0 : segs;
    }
  }
  __syncthreads();
#pragma unroll
  for (int c = 0; c < kTableChunks; ++c)
    if (m[c] > 0) atomicAdd(&s_more[m[c]], 1);
  __syncthreads();
  // L_k = lists with more than k segments: exclusive suffix sums, then base_k = L_0 + .. + L_(k-1): exclusive prefix sums
  // (Hillis-Steele over 129 threads: a thread adding up its 128 predecessors one LDS read after the other took 7 us)
  const bool mine = tid <= kFwdMaxLayers;
  int v = mine ? s_more[tid] : 0;
  const int own = v;
  for (int off = 1; off <= kFwdMaxLayers; off <<= 1) {
    if (mine) s_tmp[tid] = v;
    __syncthreads();
    if (mine && tid + off <= kFwdMaxLayers) v += s_tmp[tid + off];
    __syncthreads();
  }
  const int more = v - own;
  if (mine) s_more[tid] = more;
  v = more;
  for (int off = 1; off <= kFwdMaxLayers; off <<= 1) {
    if (mine) s_tmp[tid] = v;
    __syncthreads();
    if (mine && tid >= off) v += s_tmp[tid - off];
    __syncthreads();
  }
  if (mine) {
    s_base[tid] = v - more;
    fs.base[tid] = v - more;
  }
  __syncthreads();
  const int total = s_base[kFwdMaxLayers];
  const bool fits = total <= fs.cap;
#pragma unroll
  for (int c = 0; c < kTableChunks; ++c) {
    const int t = c * 1024 + tid;
    if (t >= num_tiles) break;
    const int segs = fits ? m[c] : 0;
    int rank = -1;
    if (segs > 0) {
      rank = s_more[segs] + atomicAdd(&s_cursor[segs], 1);  // behind the lists of more segments
      for (int k = 0; k < segs; ++k) fs.blocks[s_base[k] + rank] = make_int2(t, k | (s_more[k] < fs.thin_layer ? 1 << 30 : 0));
    }
    fs.rank[t] = rank;
  }
  if (tid == 0) {
    *fs.count = fits ? total : 0;
    if (fs.asked) *fs.asked = tagged_figure(fs.tag, total);
  }
}

// ... and behind it: one block per tile adds up what the tile's segment blocks left (fixed order: the same image in every
// run), finds the segment the pixel stopped in and writes the forward's per-pixel outputs, the tile's largest stop index
// and the backward's checkpoints {T in front of the boundary, colour in front of it} (gs_render.h: TileSegments).
__global__ __launch_bounds__(256) void fwd_segments_combine_kernel(const int *__restrict__ ranges, int width, int height,
                                                                   int ntx, int num_tiles, float bg, int *__restrict__ n_out,
                                                                   float *__restrict__ T_out, float *__restrict__ image,
                                                                   int *__restrict__ tops_out, FwdSegments fs, TileSegments seg) {
  __shared__ int s_top;
  const int tile = blockIdx.x;
  const int rank = fs.rank[tile];
  if (rank < 0) return;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, row = lane >> 4, j = lane & 15;
  if (tid == 0) s_top = 0;
  __syncthreads();
  const int tile_x = tile % ntx, tile_y = tile / ntx;
  const int px = tile_x * 16 + (wave & 1) * 8 + (row & 1) * 4 + (j & 3);
  const int py = tile_y * 16 + (wave >> 1) * 8 + (row >> 1) * 4 + (j >> 2);
  const bool inside = px < width && py < height;
  const int start = ranges[tile], total = ranges[tile + 1] - start;
  const int m = (total + kSegEntries - 1) / kSegEntries;
  float ar = 0.0f, ag = 0.0f, ab = 0.0f, Tout = 1.0f;
  int n = total;
  // kAhead segments' values are requested before the first is looked at: one at a time the longest list's twenty
  // segments were twenty dependent round trips (28 us behind the forward)
  constexpr int kAhead = 8;
  bool done = false;
  for (int k0 = 0; k0 < m && !done; k0 += kAhead) {
    int st[kAhead];
    float4 p[kAhead];
    float pref[kAhead];
#pragma unroll
    for (int q = 0; q < kAhead; ++q) {
      const int k = min(k0 + q, m - 1);
      const size_t slot = (size_t)(fs.base[k] + rank);
      st[q] = fs.stop[slot * 256 + tid];
      p[q] = fs.part[slot * 256 + tid];
      // T in front of boundary k: what the segment in front left behind (read by the backward only for pixels that pass
      // the boundary alive, for which it is exactly the T segment k started from)
      pref[q] = k > 0 ? __uint_as_float((unsigned int)fs.granules[((size_t)fs.cap + fs.base[k - 1] + rank) * 256 + tid]) : 1.0f;
    }
#pragma unroll
    for (int q = 0; q < kAhead; ++q) {
      const int k = k0 + q;
      if (k >= m || done) break;
      if (k > 0 && seg.chk) seg.chk[(size_t)segment_slot(start, k) * 256 + tid] = make_float4(pref[q], ar, ag, ab);
      if (st[q] == -2) { done = true; break; }  // (a pixel outside the image; inside it a pixel is dead only behind the segment it stopped in)
      ar += p[q].x; ag += p[q].y; ab += p[q].z;
      Tout = p[q].w;
      if (st[q] >= 0) { n = st[q]; done = true; }
    }
  }
  if (tops_out) {
    int top = inside ? n : 0;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) top = max(top, __shfl_xor(top, off, 64));
    if (lane == 0) atomicMax(&s_top, top);
    __syncthreads();
    if (tid == 0) tops_out[tile] = s_top;
  }
  if (inside) {
    const int pid = py * width + px;
    n_out[pid] = n;
    T_out[pid] = Tout;
    image[3 * pid + 0] = ar + Tout * bg;
    image[3 * pid + 1] = ag + Tout * bg;
    image[3 * pid + 2] = ab + Tout * bg;
  }
}

int launch_fwd_segments_table(const int *ranges, int num_tiles, const FwdSegments &fs, hipStream_t st) {
  fwd_segments_table_kernel<<<1, 1024, 0, st>>>(ranges, num_tiles, fs);
  GS_LAUNCH_CHECK();
  return GSPLAT_OK;
}

// host-side launchers shared with gs_fused.hip ------------------------------------------
int launch_render_fwd(const float4 *recs, const RawSplats *raw, const int *sorted, const int *ranges, int width,
                      int height, float bg, int *n_out, float *T_out, float *image, hipStream_t st, float4 *zero,
                      long long zero_vec, unsigned short *masks_out, const int *order, int *tops_out, const TileSegments *segments,
                      const FwdSegments *fwd_segments) {
  const int ntx = (width + 15) / 16, nty = (height + 15) / 16, num_tiles = ntx * nty;
  RawSplats none = {nullptr, nullptr, nullptr, nullptr};
  const TileSegments seg = segments ? *segments : TileSegments{nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr, nullptr};
  const FwdSegments fs = fwd_segments ? *fwd_segments : FwdSegments{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, nullptr, 0, 0};
  const dim3 grid(tile_grid(num_tiles) + (fs.blocks ? fs.cap : 0)), block(256);
  if (recs) {
    render_fwd_kernel<true><<<grid, block, 0, st>>>(recs, none, sorted, ranges, width, height, ntx, num_tiles, bg, n_out, T_out, image, zero, zero_vec, masks_out, order, tops_out, seg, fs);
  } else {
    render_fwd_kernel<false><<<grid, block, 0, st>>>(nullptr, *raw, sorted, ranges, width, height, ntx, num_tiles, bg, n_out, T_out, image, zero, zero_vec, nullptr, order, tops_out, seg, fs);
  }
  GS_LAUNCH_CHECK();
  if (fs.blocks) {
    fwd_segments_combine_kernel<<<num_tiles, 256, 0, st>>>(ranges, width, height, ntx, num_tiles, bg, n_out, T_out, image, tops_out, fs, seg);
    GS_LAUNCH_CHECK();
  }
  return GSPLAT_OK;
}

int launch_render_bwd(const float4 *recs, const RawSplats *raw, const int *sorted, const int *ranges, const int *n_px,
                      const float *T_px, const float *grad_image, int width, int height, float bg, float *rows,
                      float *g_rgb, float *g_opacity, float *g_uv, float *g_conic, hipStream_t st, const unsigned short *masks_in,
                      hipEvent_t ev_start, hipEvent_t ev_stop, const int *order, const TileSegments *segments) {
  const int ntx = (width + 15) / 16, nty = (height + 15) / 16, num_tiles = ntx * nty;
  const TileSegments seg = segments ? *segments : TileSegments{nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr, nullptr};
  // split lists: their further segments are extra blocks in front of the main grid (blocks beyond the count leave at once)
  const dim3 grid(tile_grid(num_tiles) + (seg.chk ? seg.extra_cap : 0)), block(256);
  RawSplats none = {nullptr, nullptr, nullptr, nullptr};
  GradOut out = {rows, g_rgb, g_opacity, g_uv, g_conic};
  const char *pp_env = getenv("GSPLAT_BWD_PINGPONG");  // (read per launch: the tests switch it inside one process)
  const int pingpong = pp_env ? atoi(pp_env) : GS_BWD_PINGPONG;
  if (recs && rows && pingpong) {  // r06 experiment: two half-batches in flight (render_bwd_pp_kernel)
    if (ev_start && ev_stop)
      hipExtLaunchKernelGGL(render_bwd_pp_kernel, grid, block, 0, st, ev_start, ev_stop, 0, recs, sorted, ranges, n_px, T_px,
                            grad_image, width, height, ntx, num_tiles, bg, rows, masks_in, order, seg);
    else
      render_bwd_pp_kernel<<<grid, block, 0, st>>>(recs, sorted, ranges, n_px, T_px, grad_image, width, height, ntx, num_tiles, bg, rows, masks_in, order, seg);
  } else if (recs && rows && ev_start && ev_stop) {
    // timed launch (the context's per-stage timing): the events take the begin and end timestamps of THIS dispatch from
    // its completion signal.  Two hipEventRecord calls around the launch are barrier packets of their own and kept the
    // GPU idle for ~11 us before and ~6 us after the kernel in every step they were on.
    hipExtLaunchKernelGGL((render_bwd_kernel<true, true, GS_BWD_BATCH>), grid, block, 0, st, ev_start, ev_stop, 0, recs, none,
                          sorted, ranges, n_px, T_px, grad_image, width, height, ntx, num_tiles, bg, out, masks_in, order, seg);
  } else if (recs && rows) {
    render_bwd_kernel<true, true, GS_BWD_BATCH><<<grid, block, 0, st>>>(recs, none, sorted, ranges, n_px, T_px, grad_image, width, height, ntx, num_tiles, bg, out, masks_in, order, seg);
  } else if (recs) {
    render_bwd_kernel<true, false, GS_BWD_BATCH><<<grid, block, 0, st>>>(recs, none, sorted, ranges, n_px, T_px, grad_image, width, height, ntx, num_tiles, bg, out, masks_in, order, seg);
  } else if (rows) {  // the reference operator's input arrays, gradient rows out (gsplat_render_image_backward stages them)
    render_bwd_kernel<false, true, GS_BWD_BATCH><<<grid, block, 0, st>>>(nullptr, *raw, sorted, ranges, n_px, T_px, grad_image, width, height, ntx, num_tiles, bg, out, masks_in, order, seg);
  } else {
    render_bwd_kernel<false, false, GS_BWD_BATCH><<<grid, block, 0, st>>>(nullptr, *raw, sorted, ranges, n_px, T_px, grad_image, width, height, ntx, num_tiles, bg, out, masks_in, order, seg);
  }
  GS_LAUNCH_CHECK();
  return GSPLAT_OK;
}

}  // namespace gs

extern "C" {

#if GS_STAMP
// diagnostic builds only: copies the stamp words of the last render_bwd launch (GS_STAMP_WORDS per wave, 4 waves per block)
int gsplat_debug_read_stamps(unsigned long long *dst, size_t words) {
  (void)hipDeviceSynchronize();
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(gs_stamp_buf), words * sizeof(unsigned long long), 0, hipMemcpyDeviceToHost);
}
int gsplat_debug_read_stamps_fwd(unsigned long long *dst, size_t words) {
  (void)hipDeviceSynchronize();
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(gs_stamp_fwd), words * sizeof(unsigned long long), 0, hipMemcpyDeviceToHost);
}
#endif

int gsplat_render_image(const float *uv, const float *opacity, const float *conic, const float *rgb,
                        float background_opacity, const int *sorted_splats, const int *splat_range_by_tile,
                        int image_width, int image_height, int *splats_per_pixel, float *weight_per_pixel,
                        float *image, void *stream) {
  GS_REQUIRE_DEV(uv); GS_REQUIRE_DEV(opacity); GS_REQUIRE_DEV(conic); GS_REQUIRE_DEV(rgb);
  GS_REQUIRE_DEV(sorted_splats); GS_REQUIRE_DEV(splat_range_by_tile); GS_REQUIRE_DEV(splats_per_pixel);
  GS_REQUIRE_DEV(weight_per_pixel); GS_REQUIRE_DEV(image);
  GS_REQUIRE(image_width > 0 && image_height > 0, "image size must be positive");
  gs::RawSplats raw = {uv, opacity, conic, rgb};
  return gs::launch_render_fwd(nullptr, &raw, sorted_splats, splat_range_by_tile, image_width, image_height,
                               background_opacity, splats_per_pixel, weight_per_pixel, image, (hipStream_t)stream, nullptr,
                               0, nullptr, nullptr, nullptr, nullptr, nullptr);
}

}  // extern "C"

namespace {
// rows[g] = {rgb3, opacity1, conic3, uv2, pad7} (what render_bwd_kernel<.., kRows = true> accumulates) added into the
// reference operator's four gradient arrays ("+=", cuda_backward.cuh:116-122)
__global__ __launch_bounds__(256) void rows_to_arrays_kernel(const float4 *__restrict__ rows, long long count,
                                                             float *__restrict__ g_rgb, float *__restrict__ g_opacity,
                                                             float *__restrict__ g_uv, float *__restrict__ g_conic) {
  const long long g = (long long)blockIdx.x * 256 + threadIdx.x;
  if (g >= count) return;
  const float4 a = rows[4 * g], b = rows[4 * g + 1];
  const float c = rows[4 * g + 2].x;
  // a row nobody added to stays +0: skip its nine read-modify-writes (most rows of an over-sized bound, below)
  if (a.x == 0.0f && a.y == 0.0f && a.z == 0.0f && a.w == 0.0f && b.x == 0.0f && b.y == 0.0f && b.z == 0.0f &&
      b.w == 0.0f && c == 0.0f)
    return;
  g_rgb[3 * g] += a.x; g_rgb[3 * g + 1] += a.y; g_rgb[3 * g + 2] += a.z;
  g_opacity[g] += a.w;
  g_conic[3 * g] += b.x; g_conic[3 * g + 1] += b.y; g_conic[3 * g + 2] += b.z;
  g_uv[2 * g] += b.w; g_uv[2 * g + 1] += c;
}

// How many rows of `floats_per_row` floats fit between p and the end of the device allocation p points into: an upper
// bound on the gaussian ids the caller's lists may hold (a larger id would index past the caller's own array).
long long rows_to_allocation_end(const void *p, int floats_per_row) {
  void *base = nullptr;
  size_t size = 0;
  if (hipMemGetAddressRange((hipDeviceptr_t *)&base, &size, (hipDeviceptr_t)p) != hipSuccess) {
    (void)hipGetLastError();
    return -1;
  }
  const long long bytes = (long long)((const char *)base + size - (const char *)p);
  return bytes / (4ll * floats_per_row);
}
}  // namespace

extern "C" {

int gsplat_render_image_backward(const float *uvs, const float *opacity, const float *conic, const float *rgb,
                                 float background_opacity, const int *sorted_splats,
                                 const int *splat_range_by_tile, const int *num_splats_per_pixel,
                                 const float *final_weight_per_pixel, const float *grad_image, int image_width,
                                 int image_height, float *grad_rgb, float *grad_opacity, float *grad_uv,
                                 float *grad_conic, void *stream) {
  GS_REQUIRE_DEV(uvs); GS_REQUIRE_DEV(opacity); GS_REQUIRE_DEV(conic); GS_REQUIRE_DEV(rgb);
  GS_REQUIRE_DEV(sorted_splats); GS_REQUIRE_DEV(splat_range_by_tile); GS_REQUIRE_DEV(num_splats_per_pixel);
  GS_REQUIRE_DEV(final_weight_per_pixel); GS_REQUIRE_DEV(grad_image); GS_REQUIRE_DEV(grad_rgb);
  GS_REQUIRE_DEV(grad_opacity); GS_REQUIRE_DEV(grad_uv); GS_REQUIRE_DEV(grad_conic);
  GS_REQUIRE(image_width > 0 && image_height > 0, "image size must be positive");
  gs::RawSplats raw = {uvs, opacity, conic, rgb};
  hipStream_t st = (hipStream_t)stream;
  // Nine float atomics per (gaussian, tile) into FOUR arrays are four 64-byte atomic requests at the memory side where
  // the fused path's 64-byte gradient row is one (MI355X_MICROARCH.md, global float atomics: ~1.3 TB/s of 64-byte
  // requests chip-wide): at 1e6 gaussians / 1080p that is 8.9 M requests, ~0.4 ms, against 0.33 ms for the whole fused
  // kernel (r04: 0.55 ms for this operator).  So the operator accumulates rows in library scratch and adds them to the
  // caller's arrays in one pass.  The operator is not told the number of gaussians; the rows needed are bounded by the
  // smallest of its per-gaussian arrays (an id beyond it would index past the caller's own allocation).  A bound that
  // is unknown or wasteful (arrays carved out of a much larger allocation) keeps the direct atomics.
  long long bound = -1;
  {
    const struct { const void *p; int w; } arrays[] = {{uvs, 2}, {opacity, 1}, {conic, 3}, {rgb, 3}, {grad_rgb, 3},
                                                       {grad_opacity, 1}, {grad_uv, 2}, {grad_conic, 3}};
    for (const auto &a : arrays) {
      const long long r = rows_to_allocation_end(a.p, a.w);
      if (r < 0) { bound = -1; break; }
      bound = bound < 0 ? r : (r < bound ? r : bound);
    }
  }
  const long long kMaxScratchRows = 4ll << 20;  // 256 MiB of rows: clearing more than that costs what the rows save
  if (bound > 0 && bound <= kMaxScratchRows) {
    gs::ScratchLock lock;
    gs::DeviceBuffer &rows = gs::scratch(gs::SCR_GRADROWS);
    int rc = rows.reserve((size_t)bound * 64);
    if (rc) return rc;
    GS_HIP(hipMemsetAsync(rows.ptr, 0, (size_t)bound * 64, st));
    rc = gs::launch_render_bwd(nullptr, &raw, sorted_splats, splat_range_by_tile, num_splats_per_pixel,
                               final_weight_per_pixel, grad_image, image_width, image_height, background_opacity,
                               rows.as<float>(), nullptr, nullptr, nullptr, nullptr, st, nullptr, nullptr, nullptr, nullptr, nullptr);
    if (rc) return rc;
    rows_to_arrays_kernel<<<gs::div_up(bound, 256), 256, 0, st>>>(rows.as<float4>(), bound, grad_rgb, grad_opacity, grad_uv,
                                                                  grad_conic);
    GS_LAUNCH_CHECK();
    return GSPLAT_OK;
  }
  return gs::launch_render_bwd(nullptr, &raw, sorted_splats, splat_range_by_tile, num_splats_per_pixel,
                               final_weight_per_pixel, grad_image, image_width, image_height, background_opacity,
                               nullptr, grad_rgb, grad_opacity, grad_uv, grad_conic, st, nullptr, nullptr, nullptr, nullptr, nullptr);
}

}  // extern "C"
