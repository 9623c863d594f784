// gs_binning.hip -- tile binning + (tile | depth) radix sort + per-tile ranges.
//
// Semantics follow the reference's get_sorted_gaussian_list (cuda/culling.cu:197-343,
// 386-475): a gaussian is listed in a tile iff the tile lies in its coarse rectangle AND
// its oriented bounding box passes the 4-axis separating-axis test against the closed
// tile AABB; lists are ordered by depth, ties by gaussian id.
//
// Structure (MI355X-first, not the reference's): no (tile, gaussian) candidate-pair buffer
// and no second coarse pass.  One thread per gaussian counts its exact hits, a rocPRIM
// exclusive scan turns counts into offsets, the same thread then emits packed 64-bit keys
// (tile << 32 | order-preserving depth bits) straight to their final slots, and one
// rocPRIM radix sort over only the significant key bits orders them.  Integer-packed keys
// replace the reference's double keys (SURVEY.md 8a hazard 3).
#include <cstring>
#include <rocprim/rocprim.hpp>

#include <algorithm>
#include <cstdlib>

#include "gs_common.h"
#include "gs_math.h"

namespace {

constexpr int kBlock = 256;

// coarse candidate count (what call 1 of the reference protocol reports)
__global__ __launch_bounds__(kBlock) void coarse_count_kernel(const float *__restrict__ uv,
                                                              const float *__restrict__ radius, int ntx, int nty,
                                                              int N, unsigned long long *__restrict__ total) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  unsigned long long mine = 0;
  if (i < N) {
    const gs::TileRect r = gs::coarse_rect(uv[2 * i], uv[2 * i + 1], radius[4 * i], ntx, nty);
    mine = (unsigned long long)(r.x1 - r.x0) * (unsigned long long)(r.y1 - r.y0);
  }
  // wave reduction, then one atomic per wave
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) mine += __shfl_down(mine, off, 64);
  if ((threadIdx.x & 63) == 0 && mine) atomicAdd(total, mine);
}

}  // namespace

namespace gs {

// counts[j] <- exact number of tiles gaussian j is listed in (j = rank ? rank[i] : i for kept i)
__global__ __launch_bounds__(kBlock) void tile_count_kernel(const float *__restrict__ uv,
                                                            const float *__restrict__ radius, int ntx, int nty,
                                                            int N, const unsigned char *__restrict__ mask,
                                                            const int *__restrict__ rank, int *__restrict__ counts,
                                                            unsigned long long *__restrict__ coarse_total) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  unsigned long long coarse = 0;
  if (i < N && (!mask || mask[i])) {
    const int j = rank ? rank[i] : i;
    const float4 rd = reinterpret_cast<const float4 *>(radius)[j];
    const float u = uv[2 * j], v = uv[2 * j + 1];
    const TileRect r = coarse_rect(u, v, rd.x, ntx, nty);
    int hits = 0;
    if (r.x1 > r.x0 && r.y1 > r.y0) {
      coarse = (unsigned long long)(r.x1 - r.x0) * (unsigned long long)(r.y1 - r.y0);
      const Obb o = make_obb(u, v, rd.x, rd.y, rd.z, rd.w);
      const TileRect sp = obb_span(o, r);
      for (int tx = sp.x0; tx < sp.x1; ++tx)
        for (int ty = sp.y0; ty < sp.y1; ++ty) hits += obb_hits_tile(o, tx, ty) ? 1 : 0;
    }
    counts[j] = hits;
  }
  if (coarse_total) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) coarse += __shfl_down(coarse, off, 64);
    if ((threadIdx.x & 63) == 0 && coarse) atomicAdd(coarse_total, coarse);
  }
}

// ------------------------------------------------------------------------------------------------
// Two-level ordering (used by the fused path and by gsplat_get_sorted_gaussian_list):
//   1. every gaussian emits (tile id, payload = depth bits << 32 | gaussian id) for the tiles it is listed in;
//   2. ONE rocPRIM radix sort over only the tile bits (ceil(log2 T) bits, 2 digit passes) groups them by tile;
//   3. one workgroup per tile orders its group by payload with a bitonic sort in LDS and writes the ids.
// Same final order as a global sort on (tile << 32 | depth) with ties broken by gaussian id (payloads are unique
// inside a tile), but the wide 45-bit / 6-pass sort of all S instances is gone.
__global__ __launch_bounds__(kBlock) void tile_emit_payload_kernel(const float *__restrict__ uv,
                                                                   const float *__restrict__ xyz_c,
                                                                   const float *__restrict__ radius, int ntx, int nty,
                                                                   int N, const unsigned char *__restrict__ mask,
                                                                   const int *__restrict__ rank,
                                                                   const int *__restrict__ offsets,
                                                                   const unsigned long long *__restrict__ hitmask,
                                                                   long long capacity,
                                                                   unsigned int *__restrict__ keys,
                                                                   unsigned long long *__restrict__ payload) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= N || (mask && !mask[i])) return;
  const int j = rank ? rank[i] : i;
  int w = offsets[j];
  // `capacity` bounds the writes when the kernel is launched before the host knows the instance total
  const int end = (int)min((long long)offsets[j + 1], capacity);
  if (w >= end) return;
  const float4 rd = reinterpret_cast<const float4 *>(radius)[j];
  const float u = uv[2 * j], v = uv[2 * j + 1];
  const unsigned long long pay = ((unsigned long long)float_sort_bits(xyz_c[3 * j + 2]) << 32) | (unsigned int)j;
  const TileRect r = coarse_rect(u, v, rd.x, ntx, nty);
  const int rh = r.y1 - r.y0;
  if (hitmask && (r.x1 - r.x0) * rh <= 64) {
    // the counting pass already ran the separating-axis tests of this rectangle and left one bit per tile
    // (bit = (tx - x0) * height + (ty - y0)): emit the set bits instead of testing ~6x as many candidates again
    unsigned long long m = hitmask[j];
    while (m != 0ull && w < end) {
      const int b = __builtin_ctzll(m);
      m &= m - 1ull;
      const int tx = r.x0 + b / rh, ty = r.y0 + b % rh;
      keys[w] = (unsigned int)(ty * ntx + tx);
      payload[w] = pay;
      ++w;
    }
    return;
  }
  const Obb o = make_obb(u, v, rd.x, rd.y, rd.z, rd.w);
  const TileRect sp = obb_span(o, r);
  for (int tx = sp.x0; tx < sp.x1; ++tx)
    for (int ty = sp.y0; ty < sp.y1; ++ty)
      if (obb_hits_tile(o, tx, ty) && w < end) {
        keys[w] = (unsigned int)(ty * ntx + tx);
        payload[w] = pay;
        ++w;
      }
}

__global__ __launch_bounds__(kBlock) void tile_ranges32_kernel(const unsigned int *__restrict__ keys, int S,
                                                               int num_tiles, int *__restrict__ ranges,
                                                               int *__restrict__ long_tile_count) {
  const int s = blockIdx.x * kBlock + threadIdx.x;
  if (s == 0) *long_tile_count = 0;  // consumed by the depth-sort kernels that follow on the same stream
  if (s >= S) return;
  const int cur = min((int)keys[s], num_tiles - 1);
  const int prev = s > 0 ? min((int)keys[s - 1], num_tiles - 1) : -1;
  for (int t = prev + 1; t <= cur; ++t) ranges[t] = s;
  if (s == S - 1)
    for (int t = cur + 1; t <= num_tiles; ++t) ranges[t] = S;
}

// Depth order inside every tile: a NORMALISED bitonic network (every comparator puts the minimum at the lower
// index, the first step of each merge pairs i with its mirror image), so virtual +infinity padding above `len`
// never moves and comparators that touch it can simply be skipped: any list length works without padding stores.
// Lists up to kLdsSort entries are sorted in LDS, longer ones in place in global memory by the same workgroup.
constexpr int kLdsSort = 16384;  // 128 KB of LDS: the 16-wave instantiation of tile_depth_sort_kernel

// kWaveLocal: the buffer is LDS.  Thread t of a wave owns comparators whose operands lie in that wave's own
// 128-element span whenever the comparator distance is <= 64, and a wave executes its LDS operations in order, so
// those steps need no workgroup barrier (only a compiler fence); for n2 = 512 that removes 42 of 45 barriers.
// k_first > 2: the caller has already sorted every aligned run of k_first / 2 entries (the merge levels below k_first
// are skipped).
template <bool kWaveLocal, int kThreads = kBlock, typename Buf>
__device__ __forceinline__ void bitonic_sort_block(Buf p, int len, int n2, int tid, int k_first = 2) {
  // a step whose comparators stay inside one wave's span only needs that wave's own earlier LDS writes; a
  // workgroup barrier is required whenever the previous OR the next step crosses waves
  int prev = 1 << 30;  // the loads before the first step were followed by __syncthreads()
  auto sync = [&](int next) {
    if (kWaveLocal && prev <= 64 && next <= 64) {
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
    } else {
      __threadfence_block();
      __syncthreads();
    }
    prev = next;
  };
  prev = 0;  // nothing to order before the very first step
  int lh = 0;  // log2(k / 2)
  while ((2 << lh) < k_first) ++lh;
  for (int k = k_first; k <= n2; k <<= 1, ++lh) {
    {  // mirror step: comparator distance up to k - 1
      sync(k >> 1);
      const int half = k >> 1;
      for (int t = tid; t < (n2 >> 1); t += kThreads) {
        const int base = (t >> lh) << (lh + 1), il = t & (half - 1);
        const int lo = base + il, hi = base + k - 1 - il;
        if (hi < len) {
          const unsigned long long a = p[lo], b = p[hi];
          if (a > b) { p[lo] = b; p[hi] = a; }
        }
      }
    }
    for (int j = k >> 2; j > 0; j >>= 1) {
      sync(j);
      for (int t = tid; t < (n2 >> 1); t += kThreads) {
        const int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1)), hi = lo | j;
        if (hi < len) {
          const unsigned long long a = p[lo], b = p[hi];
          if (a > b) { p[lo] = b; p[hi] = a; }
        }
      }
    }
  }
  __threadfence_block();
  __syncthreads();
}

// Merging the register-sorted runs of a long list in LDS (r03).  `p` holds ceil(len / run) sorted runs of `run` entries
// (the last one shorter); the runs are merged pairwise, level by level.  At each level a thread owns kPer consecutive
// OUTPUT positions: it finds where they start in the two runs of its pair (merge path: a binary search along the
// diagonal i + j = position -- payloads are unique inside a tile, so the split is unique), merges its kPer entries
// sequentially into registers, and after a barrier writes them back in place.  Per level and entry that is ~3 LDS
// reads and one write behind two barriers; the bitonic merge levels this replaces (11 to 14 compare-exchange steps per
// level, each with its own barrier) cost 40 of the 90 us a 16-run list kept its workgroup busy.
template <int kThreads, int kPer, typename Buf>
__device__ __forceinline__ void merge_sorted_runs(Buf p, int len, int run, int tid) {
  for (int w = run; w < len; w <<= 1) {
    unsigned long long out[kPer];
    const int o = tid * kPer;  // first output position of this thread
    if (o < len) {
      const int pair_lo = (o / (2 * w)) * (2 * w);                   // this pair's A = [pair_lo, a_end), B = [a_end, b_end)
      const int a_end = min(pair_lo + w, len), b_end = min(pair_lo + 2 * w, len);
      const int na = a_end - pair_lo, nb = b_end - a_end, d = o - pair_lo;  // d = i + j on the merge path
      int lo = max(0, d - nb), hi = min(d, na);                       // i in [lo, hi]
      while (lo < hi) {  // smallest i with A[i] > B[d - 1 - i]  (i.e. everything before the split is smaller)
        const int i = (lo + hi) >> 1;
        if (p[pair_lo + i] < p[a_end + (d - 1 - i)]) lo = i + 1; else hi = i;
      }
      int i = lo, j = d - lo;
      unsigned long long a = i < na ? p[pair_lo + i] : ~0ull, b = j < nb ? p[a_end + j] : ~0ull;
#pragma unroll
      for (int k = 0; k < kPer; ++k) {
        const bool take_a = a < b;  // an exhausted run reads as ~0: a payload never is (its depth bits would be NaN)
        out[k] = take_a ? a : b;
        if (take_a) { ++i; a = i < na ? p[pair_lo + i] : ~0ull; }
        else { ++j; b = j < nb ? p[a_end + j] : ~0ull; }
      }
    }
    __threadfence_block();
    __syncthreads();
    if (o < len) {
#pragma unroll
      for (int k = 0; k < kPer; ++k)
        if (o + k < len) p[o + k] = out[k];
    }
    __threadfence_block();
    __syncthreads();
  }
}

// One WAVE per tile, keys in registers: lane L holds the E consecutive entries L*E .. L*E+E-1 of the (virtually
// +infinity padded) list.  Comparators whose operands sit in the same lane are plain register compare-exchanges;
// the others fetch the partner lane's entry with a wave shuffle (ds_bpermute: no LDS storage, no bank conflicts) and
// keep the minimum or the maximum according to the lane's side.  No barrier anywhere, no LDS allocation, so 8160
// tiles run at full occupancy; the LDS version of the same network (above) moved 370 KB through LDS per 512-entry
// tile and took 73 us on the benchmark scene.
template <int E>
__device__ __forceinline__ void wave_bitonic_sort(unsigned long long (&v)[E], int lane) {
  constexpr int kTotal = 64 * E;
#pragma unroll
  for (int k = 2; k <= kTotal; k <<= 1) {
    // merge step 1: entry p meets its mirror image p ^ (k - 1)
    if (k <= E) {
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const int o = e ^ (k - 1);
        if (e < o) {
          const unsigned long long a = v[e], b = v[o];
          v[e] = a < b ? a : b;
          v[o] = a < b ? b : a;
        }
      }
    } else {
      const int partner = lane ^ (k / E - 1);
      const bool lower = (lane & (k / E / 2)) == 0;
      unsigned long long got[E];
#pragma unroll
      for (int e = 0; e < E; ++e) got[e] = __shfl(v[E - 1 - e], partner, 64);
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const bool take = lower ? (got[e] < v[e]) : (got[e] > v[e]);
        v[e] = take ? got[e] : v[e];
      }
    }
    // remaining steps of the merge: half cleaners with distance j
#pragma unroll
    for (int j = k >> 2; j > 0; j >>= 1) {
      if (j < E) {
#pragma unroll
        for (int e = 0; e < E; ++e)
          if ((e & j) == 0) {
            const unsigned long long a = v[e], b = v[e | j];
            v[e] = a < b ? a : b;
            v[e | j] = a < b ? b : a;
          }
      } else {
        const int d = j / E;
        const bool lower = (lane & d) == 0;
#pragma unroll
        for (int e = 0; e < E; ++e) {
          const unsigned long long got = __shfl_xor(v[e], d, 64);
          const bool take = lower ? (got < v[e]) : (got > v[e]);
          v[e] = take ? got : v[e];
        }
      }
    }
  }
}

// The same network on keys held as DOUBLES: a compare-exchange is v_min_f64 + v_max_f64 instead of a 64-bit integer
// compare and four selects (the kernel is bound by exactly these instructions).  A payload (order-preserving depth
// bits << 32 | id) with its top bit flipped is, for a positive depth, the bit pattern of a positive normal double
// whose order is the payload's order; tiles holding a key for which that is not true (negative, denormal-range or
// non-finite depth) take the integer workgroup kernel instead.  Padding: +infinity.
__device__ __forceinline__ double key_min(double a, double b) {
  double r;  // through asm: fmin() would first canonicalise both operands
  asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ double key_max(double a, double b) {
  double r;
  asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// Lane exchanges of the network.  Most of them stay inside a row of 16 lanes, where a DPP move on the VALU does what a
// ds_bpermute does through the LDS: the kernel spent a third of the CU's LDS cycles and most of each wave's time in
// 21 dependent bpermute round trips per 512-entry sort (PMC: VALU 27 % busy, LDS 34 %), and the LDS is shared by the
// four SIMDs while the DPP moves are not.  xor 4 and xor 8 are two mirrors in a row (l^7^3, l^15^7).
template <int kCtrl>
__device__ __forceinline__ double dpp_move(double x) {
  const long long b = __double_as_longlong(x);
  int lo = (int)b, hi = (int)(b >> 32);
  lo = __builtin_amdgcn_mov_dpp(lo, kCtrl, 0xF, 0xF, false);
  hi = __builtin_amdgcn_mov_dpp(hi, kCtrl, 0xF, 0xF, false);
  return __longlong_as_double(((long long)hi << 32) | (long long)(unsigned int)lo);
}
template <int D>
__device__ __forceinline__ double from_lane_xor(double x) {  // the value lane ^ D holds, D a power of two
  if constexpr (D == 1) return dpp_move<0xB1>(x);                        // quad_perm [1,0,3,2]
  else if constexpr (D == 2) return dpp_move<0x4E>(x);                   // quad_perm [2,3,0,1]
  else if constexpr (D == 4) return dpp_move<0x1B>(dpp_move<0x141>(x));  // row_half_mirror, then quad_perm [3,2,1,0]
  else if constexpr (D == 8) return dpp_move<0x141>(dpp_move<0x140>(x)); // row_mirror, then row_half_mirror
  else return __shfl_xor(x, D, 64);
}
template <int M>
__device__ __forceinline__ double from_lane_mirror(double x, int lane) {  // the value lane ^ (M - 1) holds
  if constexpr (M == 2) return dpp_move<0xB1>(x);
  else if constexpr (M == 4) return dpp_move<0x1B>(x);
  else if constexpr (M == 8) return dpp_move<0x141>(x);
  else if constexpr (M == 16) return dpp_move<0x140>(x);
  else return __shfl(x, lane ^ (M - 1), 64);
}

// the compare-exchange steps at distances J, J/2, .. 1 (entries; lane L holds entries L*E .. L*E+E-1)
template <int E, int J>
__device__ __forceinline__ void bitonic_clean_f64(double (&v)[E], int lane) {
  if constexpr (J < E) {
#pragma unroll
    for (int e = 0; e < E; ++e)
      if ((e & J) == 0) {
        const double a = v[e], b = v[e | J];
        v[e] = key_min(a, b);
        v[e | J] = key_max(a, b);
      }
  } else {
    constexpr int d = J / E;
    const bool upper = (lane & d) != 0;
#pragma unroll
    for (int e = 0; e < E; ++e) {  // the lower lane keeps the smaller key, the upper lane the larger: one compare, the
      const double got = from_lane_xor<d>(v[e]);  // wave-constant lane pattern folded in on the scalar unit, one select
      v[e] = ((got < v[e]) != upper) ? got : v[e];
    }
  }
  if constexpr (J > 1) bitonic_clean_f64<E, J / 2>(v, lane);
}

// merge level K (sorted runs of K/2 -> K), then the levels above it
template <int E, int K>
__device__ __forceinline__ void bitonic_level_f64(double (&v)[E], int lane) {
  if constexpr (K <= E) {
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const int o = e ^ (K - 1);
      if (e < o) {
        const double a = v[e], b = v[o];
        v[e] = key_min(a, b);
        v[o] = key_max(a, b);
      }
    }
  } else {
    const bool upper = (lane & (K / E / 2)) != 0;
    double got[E];
#pragma unroll
    for (int e = 0; e < E; ++e) got[e] = from_lane_mirror<K / E>(v[E - 1 - e], lane);
#pragma unroll
    for (int e = 0; e < E; ++e) v[e] = ((got[e] < v[e]) != upper) ? got[e] : v[e];
  }
  if constexpr (K >= 4) bitonic_clean_f64<E, K / 4>(v, lane);
  if constexpr (K < 64 * E) bitonic_level_f64<E, 2 * K>(v, lane);
}

template <int E>
__device__ __forceinline__ void wave_bitonic_sort_f64(double (&v)[E], int lane) {
  bitonic_level_f64<E, 2>(v, lane);
}

// Loads `len` (<= 64 E) payloads starting at `start` as doubles (padding: +infinity) and sorts them; lane L ends with
// the entries L*E .. L*E+E-1 of the sorted run.  false: a key has no order-preserving double, nothing was sorted.
template <int E>
__device__ __forceinline__ bool sort_run_in_registers(const unsigned long long *__restrict__ payload, int start, int len,
                                                      int lane, double (&v)[E]) {
  bool ok = true;
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int p = lane * E + e;
    unsigned long long bits = 0x7FF0000000000000ull;  // +infinity
    if (p < len) {
      bits = payload[start + p] ^ 0x8000000000000000ull;
      ok = ok && ((bits >> 52) - 1ull) < 0x7FEull;  // sign clear, exponent field neither 0 nor 0x7FF
    }
    v[e] = __longlong_as_double((long long)bits);
  }
  if (!__all(ok)) return false;
  wave_bitonic_sort_f64<E>(v, lane);
  return true;
}

// one register-sorted run parked in LDS as payloads again; false: a key has no order-preserving double
template <int E>
__device__ __forceinline__ bool sort_run_to_lds(const unsigned long long *__restrict__ payload, int run_len, int lane,
                                                unsigned long long *dst) {
  double v[E];
  if (!sort_run_in_registers<E>(payload, 0, run_len, lane, v)) return false;
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int p = lane * E + e;
    if (p < run_len) dst[p] = (unsigned long long)__double_as_longlong(v[e]) ^ 0x8000000000000000ull;
  }
  return true;
}

// false: the tile holds a key that has no order-preserving double (the caller hands the tile to the integer kernel)
template <int E>
__device__ __forceinline__ bool sort_tile_in_registers(const unsigned long long *__restrict__ payload, int start, int len,
                                                       int lane, int *__restrict__ sorted) {
  double v[E];
  if (!sort_run_in_registers<E>(payload, start, len, lane, v)) return false;
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int p = lane * E + e;
    if (p < len) sorted[start + p] = (int)(unsigned int)((unsigned long long)__double_as_longlong(v[e]) & 0xFFFFFFFFull);
  }
  return true;
}

constexpr int kWaveSortMax = 1024;  // 16 entries per lane; longer lists take the workgroup kernel below

__global__ __launch_bounds__(256) void tile_depth_sort_wave_kernel(const unsigned long long *__restrict__ payload,
                                                                   const int *__restrict__ ranges, int num_tiles,
                                                                   int *__restrict__ sorted, int *__restrict__ long_tiles) {
  __shared__ int s_handed[4], s_base;
  const int wave = threadIdx.x >> 6, tile = blockIdx.x * 4 + wave, lane = threadIdx.x & 63;
  bool done = true;
  if (tile < num_tiles) {
    const int start = ranges[tile], len = ranges[tile + 1] - start;
    if (len > kWaveSortMax) done = false;
    else if (len <= 0) done = true;
    else if (len <= 64) done = sort_tile_in_registers<1>(payload, start, len, lane, sorted);
    else if (len <= 128) done = sort_tile_in_registers<2>(payload, start, len, lane, sorted);
    else if (len <= 256) done = sort_tile_in_registers<4>(payload, start, len, lane, sorted);
    else if (len <= 512) done = sort_tile_in_registers<8>(payload, start, len, lane, sorted);
    else done = sort_tile_in_registers<16>(payload, start, len, lane, sorted);
  }
  // tiles handed to the workgroup kernels: one atomic per workgroup (a dense scene hands over every tile, and
  // thousands of atomics on one address cost more than the sort)
  if (lane == 0) s_handed[wave] = done ? -1 : tile;
  __syncthreads();
  if (threadIdx.x == 0) {
    int n = 0;
    for (int q = 0; q < 4; ++q) n += s_handed[q] >= 0;
    s_base = n ? atomicAdd(&long_tiles[0], n) : 0;
  }
  __syncthreads();
  if (lane == 0 && !done) {
    int slot = s_base;
    for (int q = 0; q < wave; ++q) slot += s_handed[q] >= 0;
    long_tiles[1 + slot] = tile;
  }
}

// Lists the wave kernel handed over: one workgroup each.  Up to kLdsSort entries: every wave first sorts one run of
// kWaveSortMax entries in registers (as the wave kernel does) and parks it in LDS, and only the last one or two merge
// levels of the network run in LDS (11 or 23 steps instead of 66 or 78) -- unless a key has no double form, then the
// whole network runs in LDS on the integer keys.  Longer lists: in place in global memory.
// kWaves = 2: the lists of up to 2 kWaveSortMax entries (128 threads, 16 KB of LDS: ten workgroups per CU -- the
// register sorts are latency bound, so residency is what counts); kWaves = 4: everything longer.
// kRun: entries per register-sorted run (one wave each).  r05: the hand-over class (lists up to 2 kWaveSortMax entries)
// runs FOUR waves over runs of kWaveSortMax / 2 entries instead of two waves over runs of kWaveSortMax: the register
// network of a 512-entry run is 45 stages over 8 entries per lane where the 1024-entry one is 55 over 16 -- less than half
// the dependent chain the workgroup waits for, for one more merge level (GS_SORT_HALF_RUNS=0: the r04 shape).
#ifndef GS_SORT_HALF_RUNS
#define GS_SORT_HALF_RUNS 1
#endif
template <int kWaves, int kRun = kWaveSortMax>
__global__ __launch_bounds__(kWaves * 64) void tile_depth_sort_kernel(unsigned long long *__restrict__ payload,
                                                                      const int *__restrict__ ranges, int num_tiles,
                                                                      int *__restrict__ sorted,
                                                                      const int *__restrict__ long_tiles) {
  constexpr int kThreads = kWaves * 64, kLds = kWaves * kRun;
  constexpr int kClass = kLds / kWaveSortMax;  // 2: the hand-over class; 4, 8, 16: the larger ones
  static_assert(kLds <= kLdsSort && kRun <= kWaveSortMax && kClass * kWaveSortMax == kLds, "LDS buffer");
  __shared__ unsigned long long buf[kLds];
  __shared__ int s_runs_ok;
  // Class 2 works through the wave kernel's hand-over list (long lists AND short ones whose keys have no double
  // form).  The larger instantiations pick their tiles by length from `ranges` alone -- every list above kWaveSortMax
  // entries is handed over anyway -- so they do not depend on the wave kernel and run beside it on streams of their own
  // (SortFork): on a capture with thousands of entries per tile the four kernels used to run one after the other, the
  // last two for a handful of lists each (r03 garden-shaped workload: 43 + 28 + 41 + 64 + 66 us in a row).
  const int count = kClass == 2 ? long_tiles[0] : num_tiles;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int t = blockIdx.x; t < count; t += gridDim.x) {
    const int tile = kClass == 2 ? long_tiles[1 + t] : t;
    const int start = ranges[tile], len = ranges[tile + 1] - start;
    // class K takes the handed-over lists of (K / 2, K] runs of kWaveSortMax entries; 2: every list up to two runs
    // (short lists arrive here when a key has no double form); 16: also everything longer, in place in global memory
    if ((kClass > 2 && len <= kClass / 2 * kWaveSortMax) || (kClass < 16 && len > kLds)) continue;
    int n2 = 1;
    while (n2 < len) n2 <<= 1;
    __syncthreads();  // buf is reused across iterations
    if (len <= kLds) {
      if (tid == 0) s_runs_ok = 1;
      __syncthreads();
      const int run_len = min(kRun, len - wave * kRun);
      if (len > kRun && run_len > 0) {
        // the last run of a list is shorter than the others: the smallest network that holds it
        const unsigned long long *src = payload + start + wave * kRun;
        unsigned long long *dst = buf + wave * kRun;
        bool ok;
        if constexpr (kRun > 512)
          ok = run_len <= 128   ? sort_run_to_lds<2>(src, run_len, lane, dst)
               : run_len <= 256 ? sort_run_to_lds<4>(src, run_len, lane, dst)
               : run_len <= 512 ? sort_run_to_lds<8>(src, run_len, lane, dst)
                                : sort_run_to_lds<kWaveSortMax / 64>(src, run_len, lane, dst);
        else
          ok = run_len <= 128   ? sort_run_to_lds<2>(src, run_len, lane, dst)
               : run_len <= 256 ? sort_run_to_lds<4>(src, run_len, lane, dst)
                                : sort_run_to_lds<8>(src, run_len, lane, dst);
        if (!ok && lane == 0) s_runs_ok = 0;
      }
      __syncthreads();
      if (len > kRun && s_runs_ok) {
        merge_sorted_runs<kThreads, kLds / kThreads>(buf, len, kRun, tid);
      } else {
        __syncthreads();
        for (int i = tid; i < len; i += kThreads) buf[i] = payload[start + i];
        __syncthreads();
        bitonic_sort_block<true, kThreads>(buf, len, n2, tid);
      }
      for (int i = tid; i < len; i += kThreads) sorted[start + i] = (int)(unsigned int)(buf[i] & 0xFFFFFFFFull);
    } else {
      unsigned long long *p = payload + start;
      bitonic_sort_block<false, kThreads>(p, len, n2, tid);
      for (int i = tid; i < len; i += kThreads) sorted[start + i] = (int)(unsigned int)(p[i] & 0xFFFFFFFFull);
    }
  }
}

// per-tile depth order: one wave per tile in registers; the few lists above kWaveSortMax entries are listed in
// `long_tiles` (long_tiles[0] must be 0 on entry) and finished by workgroups of the second kernel
// `longest`: the longest list when the caller knows it (the fused forward reads it with the counts), else < 0
// keys_ok: every depth key is a positive, finite, normal float (the caller's near threshold is positive: cull_gaussians keeps
// z >= near_thresh only), so the wave kernel hands over nothing but lists beyond kWaveSortMax entries -- and when `longest`
// says that there is none, the hand-over kernel is not launched at all (r06: 4.5 us per forward of launch and drain for a
// kernel whose every workgroup left at once -- the headline scene's lists end at ~600 entries).
static int sort_tiles_by_depth(unsigned long long *payload, const int *ranges, int num_tiles, size_t S, int *long_tiles,
                               int *sorted_out, hipStream_t st, long long longest = -1, const SortFork *fork = nullptr,
                               bool keys_ok = false) {
  // tiles the wave kernel may hand over: the long lists and any list with a key that has no double form
  const int max_long = (int)std::min<size_t>((size_t)num_tiles, S);
  // The lists above 2, 4 and 8 runs of kWaveSortMax entries: 256-, 512- and 1024-thread workgroups (4 / 8 / 16 register-
  // sorted runs merged in 32 / 64 / 128 KB of LDS).  With a SortFork they start on side streams behind the placement,
  // beside the wave kernel; without (stand-alone operator), behind it on `st`.
  struct Class { int cls; size_t per; };
  const Class classes[3] = {{0, 2}, {1, 4}, {2, 8}};
  bool forked[3] = {false, false, false};
  // Whatever path leaves this function after a fork -- a failed launch of a later kernel included -- `st` first waits for
  // the side streams: the caller may re-reserve or re-run the forward as soon as it sees the error, and the forked
  // kernels are still writing `payload` / `sorted_out`.
  struct JoinForks {
    const SortFork *fork; hipStream_t st; bool *forked;
    ~JoinForks() {
      for (int k = 0; k < 3; ++k)
        if (forked[k]) { (void)hipStreamWaitEvent(st, fork->ev_join[k], 0); forked[k] = false; }
    }
    int join() {  // the normal exit: the same waits, with their status
      for (int k = 0; k < 3; ++k)
        if (forked[k]) { forked[k] = false; GS_HIP(hipStreamWaitEvent(st, fork->ev_join[k], 0)); }
      return GSPLAT_OK;
    }
  } joiner{fork, st, forked};
  auto launch_class = [&](int k, hipStream_t s) {
    const int cap = (int)std::min<size_t>((size_t)num_tiles, S / (classes[k].per * (size_t)kWaveSortMax));
    if (k == 0) tile_depth_sort_kernel<4><<<std::min(cap, 5 * 256), 256, 0, s>>>(payload, ranges, num_tiles, sorted_out, long_tiles);
    if (k == 1) tile_depth_sort_kernel<8><<<std::min(cap, 2 * 256), 512, 0, s>>>(payload, ranges, num_tiles, sorted_out, long_tiles);
    if (k == 2) tile_depth_sort_kernel<16><<<std::min(cap, 256), 1024, 0, s>>>(payload, ranges, num_tiles, sorted_out, long_tiles);
  };
  auto wanted = [&](int k) {
    const size_t floor_len = classes[k].per * (size_t)kWaveSortMax;
    return max_long > 0 && S / floor_len > 0 && (longest < 0 || (size_t)longest > floor_len);
  };
  if (fork && fork->ready) {
    bool any = false;
    for (int k = 0; k < 3; ++k) any = any || wanted(k);
    if (any) {
      GS_HIP(hipEventRecord(fork->ev_fork, st));
      for (int k = 0; k < 3; ++k)
        if (wanted(k)) {
          GS_HIP(hipStreamWaitEvent(fork->side[k], fork->ev_fork, 0));
          launch_class(k, fork->side[k]);
          const hipError_t launched = hipGetLastError();
          // the join event is recorded even behind a failed launch: whatever did reach the side stream is waited for
          const hipError_t recorded = hipEventRecord(fork->ev_join[k], fork->side[k]);
          forked[k] = recorded == hipSuccess;
          if (launched != hipSuccess) { set_error("tile_depth_sort (side stream): %s", hipGetErrorString(launched)); return GSPLAT_ERR_HIP; }
          GS_HIP(recorded);
        }
    }
  }
  tile_depth_sort_wave_kernel<<<div_up(num_tiles, 4), 256, 0, st>>>(payload, ranges, num_tiles, sorted_out, long_tiles);
  GS_LAUNCH_CHECK();
  if (max_long > 0 && !(keys_ok && longest >= 0 && longest <= (long long)kWaveSortMax)) {
    // workgroups beyond the list's length leave at once; ten (16 KB of LDS) resp. five (32 KB) fit on a CU
#if GS_SORT_HALF_RUNS
    tile_depth_sort_kernel<4, kWaveSortMax / 2><<<std::min(max_long, 8 * 256), 256, 0, st>>>(payload, ranges, num_tiles,
                                                                                             sorted_out, long_tiles);
#else
    tile_depth_sort_kernel<2><<<std::min(max_long, 10 * 256), 128, 0, st>>>(payload, ranges, num_tiles, sorted_out,
                                                                            long_tiles);
#endif
    GS_LAUNCH_CHECK();
    for (int k = 0; k < 3; ++k)
      if (!forked[k] && wanted(k)) {
        launch_class(k, st);
        GS_LAUNCH_CHECK();
      }
  }
  return joiner.join();
}

static int tile_bits(int num_tiles) {
  int b = 1;
  while ((1LL << b) < (long long)num_tiles) ++b;
  return b;
}

// rocPRIM falls back to a merge sort below 2^20 items; Onesweep is faster for these key widths
using OnesweepAlways = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
                                                  rocprim::default_config, 8192>;

size_t binning_temp_bytes(size_t N, size_t S, int num_tiles) {
  size_t b = 0, c = 0;
  (void)rocprim::radix_sort_pairs<OnesweepAlways>(nullptr, b, (unsigned int *)nullptr, (unsigned int *)nullptr,
                                                  (unsigned long long *)nullptr, (unsigned long long *)nullptr,
                                                  S ? S : 1, 0, tile_bits(num_tiles), (hipStream_t)0);
  (void)rocprim::exclusive_scan(nullptr, c, (int *)nullptr, (int *)nullptr, 0, N + 1, rocprim::plus<int>(),
                                (hipStream_t)0);
  size_t d = 0;  // the dense-scene depth pre-sort (see emit_sort_ranges)
  (void)rocprim::radix_sort_pairs<OnesweepAlways>(nullptr, d, (unsigned long long *)nullptr,
                                                  (unsigned long long *)nullptr, (unsigned int *)nullptr,
                                                  (unsigned int *)nullptr, S ? S : 1, 32, 64, (hipStream_t)0);
  b = b > d ? b : d;
  return (b > c ? b : c) + 256;
}

// offsets[0..N] = exclusive scan of counts[0..N] (counts[N] must be 0)
int scan_counts(int N, const int *counts, int *offsets, void *temp, size_t temp_bytes, hipStream_t st) {
  GS_HIP(rocprim::exclusive_scan(temp, temp_bytes, counts, offsets, 0, (size_t)N + 1, rocprim::plus<int>(), st));
  return GSPLAT_OK;
}

__global__ __launch_bounds__(kBlock) void low_words_kernel(const unsigned long long *__restrict__ in, int S,
                                                           int *__restrict__ out) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i < S) out[i] = (int)(unsigned int)(in[i] & 0xFFFFFFFFull);
}

// Average list length per tile above which the depth order is established by a global stable sort on the depth bits
// BEFORE the tile sort (4 more radix passes over all instances) instead of per tile afterwards.  The per-tile network
// is cheaper for short lists (54 us vs 4 x 39 us on the benchmark scene, ~430 per tile) but grows like n log^2 n and
// leaves the register kernel above 1024 entries; dense real scenes (thousands per tile) take the global route.
static long long dense_tile_threshold() {
  static const long long v = [] {
    const char *e = getenv("GSPLAT_DENSE_TILE_AVG");  // average list length above which a radix forward stays radix
    return e ? atoll(e) : 768ll;
  }();
  return v;
}

// ------------------------------------------------------------------------------------------------
// Sparse-scene binning without a global sort.  The per-tile depth sort re-orders every list anyway, so the grouping
// by tile does not have to be stable -- a counting sort with all atomics in LDS does it:
//   count:       done by preprocess_kernel itself (gs_fused.hip): workgroup b counts the tiles its slice of the
//                gaussians hits in an LDS histogram while it runs the hit tests anyway (ds_add_u32 runs at LDS
//                rate; the same count through L2 atomics would cost more than the radix sort) and writes its
//                histogram row table[b][0..T);
//   bin_offsets: one thread per tile turns its column into workgroup-exclusive offsets and a tile total;
//                an exclusive scan of the totals gives the reference's `ranges`;
//   bin_scatter: workgroup b loads ranges[t] + table[b][t] as LDS cursors and places every instance with one
//                returning LDS atomic.
// 12 bytes per instance are written once (the payload), instead of 24 B x 2 x 2 radix passes + histogram + memsets.

template <typename F>
__device__ __forceinline__ void for_each_hit_tile(int j, const float *__restrict__ uv, const float *__restrict__ radius,
                                                  const unsigned long long *__restrict__ hitmask, int ntx, int nty,
                                                  F f) {
  const float4 rd = reinterpret_cast<const float4 *>(radius)[j];
  const float u = uv[2 * j], v = uv[2 * j + 1];
  const TileRect r = coarse_rect(u, v, rd.x, ntx, nty);
  const int rh = r.y1 - r.y0;
  if (r.x1 <= r.x0 || rh <= 0) return;
  if ((r.x1 - r.x0) * rh <= 64) {
    unsigned long long m = hitmask[j];
    const float inv_rh = 1.0f / (float)rh;  // b / rh for b < 64, rh <= 64: exact through the float reciprocal
    while (m != 0ull) {
      const int b = __builtin_ctzll(m);
      m &= m - 1ull;
      const int col = (int)(((float)b + 0.5f) * inv_rh), row = b - col * rh;
      f((r.y0 + row) * ntx + r.x0 + col);
    }
  } else {
    const Obb o = make_obb(u, v, rd.x, rd.y, rd.z, rd.w);
    const TileRect sp = obb_span(o, r);
    for (int tx = sp.x0; tx < sp.x1; ++tx)
      for (int ty = sp.y0; ty < sp.y1; ++ty)
        if (obb_hits_tile(o, tx, ty)) f(ty * ntx + tx);
  }
}

// column t of table -> exclusive prefix over the workgroups; totals[t] = column sum (totals[T] = 0 for the scan).
// One workgroup per 64 tiles: wave w of sixteen owns the rows 16w .. 16w+15 of the table (lane = tile, so every load
// and store is a coalesced 256-byte row segment), keeps its 16 counts in registers and only the wave sums meet in
// LDS.  (Four waves of 64 rows each took twice as long: 64 dependent-latency loads per thread on 68 workgroups.)
__global__ __launch_bounds__(1024) void bin_offsets_kernel(int T, int *__restrict__ table, int *__restrict__ totals,
                                                           int *__restrict__ long_tile_count) {
  constexpr int kRows = kBinBlocks / 16;
  static_assert(kBinBlocks % 16 == 0, "sixteen waves share the table rows");
  __shared__ int s_sum[16][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int t = blockIdx.x * 64 + lane;
  if (blockIdx.x == 0 && threadIdx.x == 0) { totals[T] = 0; *long_tile_count = 0; }
  // r04: a tile's list is laid out XCD-major -- the runs of the 32 workgroups of one XCD (blockIdx % 8: workgroups are
  // dealt round robin) are adjacent, so that their isolated 8-byte stores meet in that XCD's own L2 lines: the binning
  // stage 0.1064 -> 0.1032 ms (profiles/r04_ab_xcd_major_lists.txt; with bin_scatter's streamed inputs read non-temporally
  // on top: 0.1018 -- not kept, see there).  The order inside a tile's list is free: the per-tile depth sort re-orders it.
  static_assert(kBinBlocks == 256, "32 workgroups on each of 8 XCDs");
  auto row_of = [](int pos) { return (pos & 31) * 8 + (pos >> 5); };
  int v[kRows], sum = 0;
  if (t < T) {
#pragma unroll
    for (int k = 0; k < kRows; ++k) { v[k] = table[(size_t)row_of(kRows * w + k) * T + t]; sum += v[k]; }
  }
  s_sum[w][lane] = sum;
  __syncthreads();
  if (t >= T) return;
  int run = 0;
  for (int q = 0; q < w; ++q) run += s_sum[q][lane];
  if (w == 15) totals[t] = run + sum;
#pragma unroll
  for (int k = 0; k < kRows; ++k) { table[(size_t)row_of(kRows * w + k) * T + t] = run; run += v[k]; }
}

#ifndef GS_SCATTER_PREFETCH
#define GS_SCATTER_PREFETCH 1
#endif

// what bin_scatter_kernel needs to publish the forward's host record (pub == nullptr: nothing to publish)
struct RecordSource {
  volatile unsigned long long *pub;
  unsigned long long ticket;
  const int *m_total;
  const unsigned long long *pair_counters;
};

// Placement.  Every workgroup first scans the T tile totals itself (32 KB of L2-resident counts, one wave-shuffle scan
// and one barrier: about a microsecond, 256 times in parallel) instead of reading `ranges` from a scan kernel of one
// workgroup that cost 11 us on the stream (a launch, a chain of dependent loads and stores, nothing to overlap them
// with).  Workgroup 0 stores ranges[0..T] for the kernels behind this one, clamped to `capacity`, the room of the
// instance buffers: those kernels run before the host has seen S (gsplat_rasterize_image), so when S does not fit they
// work on truncated lists inside the buffers, and the host -- which reads the TRUE S from the record the last
// workgroup publishes -- grows the buffers and queues this kernel and the rest again.
__global__ __launch_bounds__(kBinThreads) void bin_scatter_kernel(const float *__restrict__ uv,
                                                                  const float *__restrict__ xyz_c,
                                                                  const float *__restrict__ radius,
                                                                  const unsigned long long *__restrict__ hitmask,
                                                                  const int *__restrict__ rank, int N, int ntx, int nty,
                                                                  const int *__restrict__ table,
                                                                  int *__restrict__ ranges, long long capacity,
                                                                  unsigned long long *__restrict__ payload,
                                                                  RecordSource rec, bool compact_walk) {
  extern __shared__ int s_cur[];
  __shared__ int s_wave[kBinThreads / 64], s_long[kBinThreads / 64];
  const int T = ntx * nty;
  const int lane = threadIdx.x & 63;
  {
    constexpr int kPer = kBinMaxTiles / kBinThreads;  // 16 tiles per thread at most
    const int *totals = table + (size_t)kBinBlocks * T;
    const int w = threadIdx.x >> 6;
    // pass A, tile t = thread + 1024 k (coalesced): the totals go to LDS, this workgroup's offsets stay in registers
    int off_t[kPer];
    {
      int tot[kPer];
#pragma unroll
      for (int k = 0; k < kPer; ++k) {  // all loads of a thread in flight before the first LDS store
        const int t = threadIdx.x + k * kBinThreads;
        tot[k] = t < T ? totals[t] : 0;
        off_t[k] = t < T ? table[(size_t)blockIdx.x * T + t] : 0;
      }
#pragma unroll
      for (int k = 0; k < kPer; ++k) {
        const int t = threadIdx.x + k * kBinThreads;
        if (t < T) s_cur[t] = tot[k];
      }
    }
    __syncthreads();
    // pass B, `per` consecutive tiles per thread: exclusive scan in place
    const int per = (T + kBinThreads - 1) / kBinThreads, lo = threadIdx.x * per;
    int v[kPer], sum = 0, longest = 0;
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
      v[k] = (k < per && lo + k < T) ? s_cur[lo + k] : 0;
      sum += v[k];
      longest = max(longest, v[k]);
    }
    int incl = sum;  // inclusive scan of the thread sums inside the wave, then of the sixteen wave sums
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int u = __shfl_up(incl, off, 64);
      if (lane >= off) incl += u;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) longest = max(longest, __shfl_xor(longest, off, 64));
    if (lane == 63) { s_wave[w] = incl; s_long[w] = longest; }
    __syncthreads();
    int before = 0, S = 0, longest_all = 0;  // the longest list: the host picks the next forward's binning route by it
#pragma unroll
    for (int q = 0; q < kBinThreads / 64; ++q) {
      const int x = s_wave[q];
      before += q < w ? x : 0;
      S += x;
      longest_all = max(longest_all, s_long[q]);
    }
    if (rec.pub && blockIdx.x == kBinBlocks - 1 && threadIdx.x < 64) {  // the forward's host record
      unsigned long long c = rec.pair_counters[threadIdx.x];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off, 64);
      if (threadIdx.x == 0)
        publish_record(rec.pub, rec.ticket, (unsigned int)*rec.m_total, (unsigned int)S, c, (unsigned int)longest_all);
    }
    const int room = (int)min(capacity, 0x7FFFFFFFll);
    int run = before + incl - sum;
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
      if (k < per && lo + k < T) s_cur[lo + k] = min(run, room);
      run += v[k];
    }
    __syncthreads();
    // pass C, coalesced again: ranges for the kernels behind this one, cursors = tile start + this workgroup's offset
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
      const int t = threadIdx.x + k * kBinThreads;
      if (t < T) {
        const int start = s_cur[t];
        if (blockIdx.x == 0) ranges[t] = start;
        s_cur[t] = start + off_t[k];
      }
    }
    if (blockIdx.x == 0 && threadIdx.x == kBinThreads - 1) ranges[T] = min(S, room);
  }
  __syncthreads();
  // This workgroup's gaussians: the chunks preprocess_kernel counted into this workgroup's histogram row (gs_common.h:
  // chunk c of 64 entries belongs to workgroup c % kBinBlocks, wave (c / kBinBlocks) % 16) -- chunks of the compacted
  // slots when that kernel walked those, else chunks of global indices, whose compacted slots are the run
  // [rank[64 c], rank[64 c + 64]) (rank = exclusive scan of the cull mask, rank[N] = M).
  const int M = rank[N];
  const int walk_chunks = bin_chunks(compact_walk ? M : N);
  auto place = [&](int tile, unsigned long long pay) {
    const int pos = atomicAdd(&s_cur[tile], 1);
#if defined(GS_SCATTER_WINDOW)
    // diagnostic build (tools/experiments/r04_scatter_window.sh): every store lands in a 128 KB window that stays in L2,
    // i.e. the kernel without the HBM cost of its isolated 8-byte stores (the lists are garbage: timing only)
    if (pos < capacity) payload[pos & 0x3FFF] = pay;
#else
    if (pos < capacity) payload[pos] = pay;
#endif
  };
  // wave-uniform trip count: rectangles of more than 64 tiles carry no hit mask and repeat the separating-axis tests;
  // one lane walking thousands of tiles alone decided this kernel's duration on scenes with large splats, so the wave
  // takes those one at a time, every lane testing every 64th tile of the rectangle's clipped span (the same functions
  // on the same broadcast inputs as the count in preprocess_kernel: the same instances).
#if GS_SCATTER_PREFETCH
  // r05: the chunk loop as a two-stage pipeline.  A chunk costs two DEPENDENT global round trips before any placement
  // can start (rank[] -> the chunk's slots, then radius / uv / depth / hit mask of those slots), and the kernel runs one
  // workgroup per CU (four waves per SIMD) -- little else hides them.  While chunk k is placed, the five loads of chunk
  // k + 1 and the two rank reads of chunk k + 2 are in flight (94 -> ~106 VGPRs of the 128 this launch shape allows).
  constexpr int kStep = kBinBlocks * (kBinThreads / 64);
  const int c_first = (int)blockIdx.x + kBinBlocks * (int)(threadIdx.x >> 6);
  auto slots_of = [&](int c, int &j, int &hi) {  // chunk c's compacted slots [j - lane, hi)
    if (c >= walk_chunks) { j = 0; hi = 0; return; }
    if (compact_walk) {
      j = c * kBinChunk + lane;
      hi = M;
    } else {
      j = rank[c * kBinChunk] + lane;
      hi = rank[min(c * kBinChunk + kBinChunk, N)];
    }
  };
  struct ChunkData { float4 rd; float u, v, z; unsigned long long hm; };
  auto fetch = [&](int j, int hi, ChunkData &d) {
    d.rd = make_float4(0.0f, 0.0f, 0.0f, 0.0f); d.u = d.v = d.z = 0.0f; d.hm = 0ull;
    if (j < hi) {
      d.rd = reinterpret_cast<const float4 *>(radius)[j];
      d.u = uv[2 * j]; d.v = uv[2 * j + 1];
      d.z = xyz_c[3 * j + 2];
      d.hm = hitmask[j];  // (read for every slot: rectangles above 64 tiles ignore it)
    }
  };
  int j_cur, hi_cur, j_nxt, hi_nxt;
  ChunkData cur, nxt;
  slots_of(c_first, j_cur, hi_cur);
  fetch(j_cur, hi_cur, cur);
  slots_of(c_first + kStep, j_nxt, hi_nxt);
  for (int c = c_first; c < walk_chunks; c += kStep) {
    int j_nn, hi_nn;
    slots_of(c + 2 * kStep, j_nn, hi_nn);  // two chunks ahead: the rank reads
    fetch(j_nxt, hi_nxt, nxt);             // one chunk ahead: the data
    const int j = j_cur, hi = hi_cur;
    const float4 rd = cur.rd;
    const float u = cur.u, v = cur.v;
    unsigned long long pay = 0ull;
    bool big = false;
    Obb bob = {};          // of a lane whose rectangle goes to the cooperative loop: its OBB ...
    TileRect bsp = {0, 0, 0, 0};  // ... and the span of tiles to test
    if (j < hi) {
      pay = ((unsigned long long)float_sort_bits(cur.z) << 32) | (unsigned int)j;
#else
  for (int c = (int)blockIdx.x + kBinBlocks * (int)(threadIdx.x >> 6); c < walk_chunks; c += kBinBlocks * (kBinThreads / 64)) {
    int j, hi;
    if (compact_walk) {
      j = c * kBinChunk + lane;
      hi = M;
    } else {
      j = rank[c * kBinChunk] + lane;
      hi = rank[min(c * kBinChunk + kBinChunk, N)];
    }
    float4 rd = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    float u = 0.0f, v = 0.0f;
    unsigned long long pay = 0ull;
    bool big = false;
    Obb bob = {};          // of a lane whose rectangle goes to the cooperative loop: its OBB ...
    TileRect bsp = {0, 0, 0, 0};  // ... and the span of tiles to test
    if (j < hi) {
      rd = reinterpret_cast<const float4 *>(radius)[j];
      u = uv[2 * j]; v = uv[2 * j + 1];
      pay = ((unsigned long long)float_sort_bits(xyz_c[3 * j + 2]) << 32) | (unsigned int)j;
#endif
      const TileRect r = coarse_rect(u, v, rd.x, ntx, nty);
      const int rh = r.y1 - r.y0;
      if (r.x1 > r.x0 && rh > 0) {
        if ((r.x1 - r.x0) * rh <= 64) {
#if GS_SCATTER_PREFETCH
          unsigned long long m = cur.hm;
#else
          unsigned long long m = hitmask[j];
#endif
          const float inv_rh = 1.0f / (float)rh;  // b / rh for b < 64, rh <= 64: exact through the float reciprocal
          while (m != 0ull) {
            const int b = __builtin_ctzll(m);
            m &= m - 1ull;
            const int col = (int)(((float)b + 0.5f) * inv_rh), row = b - col * rh;
            place((r.y0 + row) * ntx + r.x0 + col, pay);
          }
        } else {
          big = true;
          bob = make_obb(u, v, rd.x, rd.y, rd.z, rd.w);
          bsp = obb_span(bob, r);
        }
      }
    }
    // (the owner built its OBB and span once, with all the other large splats of the wave in the same instructions; the
    // twelve numbers the tile test reads and the span reach the other lanes by v_readlane.  Every lane rebuilding them
    // from six shuffled inputs cost ~150 instructions and seven LDS round trips per large splat.)
    for (unsigned long long todo = __ballot(big); todo != 0ull; todo &= todo - 1ull) {
      const int owner = __builtin_ctzll(todo);
      Obb ob;
      ob.mnx = lane_value(bob.mnx, owner); ob.mxx = lane_value(bob.mxx, owner);
      ob.mny = lane_value(bob.mny, owner); ob.mxy = lane_value(bob.mxy, owner);
      ob.a2x = lane_value(bob.a2x, owner); ob.a2y = lane_value(bob.a2y, owner);
      ob.mn2 = lane_value(bob.mn2, owner); ob.mx2 = lane_value(bob.mx2, owner);
      ob.a3x = lane_value(bob.a3x, owner); ob.a3y = lane_value(bob.a3y, owner);
      ob.mn3 = lane_value(bob.mn3, owner); ob.mx3 = lane_value(bob.mx3, owner);
      const int x0 = lane_value(bsp.x0, owner), x1 = lane_value(bsp.x1, owner);
      const int y0 = lane_value(bsp.y0, owner), y1 = lane_value(bsp.y1, owner);
      const unsigned long long opay =
          ((unsigned long long)(unsigned int)lane_value((int)(pay >> 32), owner) << 32) |
          (unsigned long long)(unsigned int)lane_value((int)(pay & 0xFFFFFFFFull), owner);
      const int sh = y1 - y0, total = (x1 - x0) * sh;
      SpanWalk wk = span_walk(lane, max(sh, 1));
      for (int p = lane; p < total; p += 64) {
        const int tx = x0 + wk.col, ty = y0 + wk.row;
        if (obb_hits_tile(ob, tx, ty)) place(ty * ntx + tx, opay);
        span_step(wk);
      }
    }
#if GS_SCATTER_PREFETCH
    j_cur = j_nxt; hi_cur = hi_nxt; cur = nxt;
    j_nxt = j_nn; hi_nxt = hi_nn;
#endif
  }
}

bool binning_supports_counting_sort(int num_tiles) { return num_tiles <= kBinMaxTiles; }
bool binning_prefers_radix(size_t S, int num_tiles) { return (long long)S > dense_tile_threshold() * (long long)num_tiles; }
// Route of the NEXT forward.  Up to dense_tile_threshold() entries per tile on average: counting sort + per-tile
// sorts, whatever the longest list (a few long lists in a sparse scene are cheaper than two more radix passes over
// everything).  Denser: still the counting sort as long as every list fits the LDS merge (kLdsSort entries: register-
// sorted runs + one or two merge levels -- 0.64 ms against 1.06 ms on the 4 M-gaussian scene); one longer list means
// an in-place global-memory network, and the radix route (depth pre-sort + stable tile sort) is the safer choice.
// After a radix forward the longest list is not known: stay.
bool binning_next_route_is_radix(bool was_counting_sort, size_t S, int num_tiles, long long longest) {
  if (!binning_prefers_radix(S, num_tiles)) return false;
  return was_counting_sort ? longest > kLdsSort : true;
}
size_t binning_table_bytes(int num_tiles) { return ((size_t)kBinBlocks * num_tiles + num_tiles + 2) * sizeof(int); }

// Phase 1 (needs nothing from the host): per-workgroup offsets from the histogram rows preprocess_kernel left in
// `table` (kBinBlocks * T ints, followed by T + 1 totals).
int binning_offsets(int ntx, int nty, int *table, int *long_tiles, hipStream_t st) {
  const int T = ntx * nty;
  bin_offsets_kernel<<<div_up(T, 64), 1024, 0, st>>>(T, table, table + (size_t)kBinBlocks * T, long_tiles);
  GS_LAUNCH_CHECK();
  return GSPLAT_OK;
}

// Phase 2: ranges (clamped to S, the room of the buffers) and placement of at most S payloads, then the per-tile depth
// order.  With `pub` the placement kernel also publishes the forward's host record {M, S, candidate pairs, longest
// list} under `ticket`.
int binning_scatter_and_sort(const float *uv, const float *xyz_c, const float *radius,
                             const unsigned long long *hitmask, const int *rank, int N, int ntx, int nty,
                             const int *table, int *ranges, size_t S, unsigned long long *payload,
                             int *long_tiles, int *sorted_out, long long longest, const int *m_total,
                             const unsigned long long *pair_counters, unsigned long long *pub,
                             unsigned long long ticket, hipStream_t st, const SortFork *fork, bool compact_walk,
                             bool keys_ok) {
  const int T = ntx * nty;
  const RecordSource rec = {pub, ticket, m_total, pair_counters};
  bin_scatter_kernel<<<kBinBlocks, kBinThreads, (size_t)T * sizeof(int), st>>>(uv, xyz_c, radius, hitmask, rank, N, ntx,
                                                                             nty, table, ranges, (long long)S, payload,
                                                                             rec, compact_walk);
  GS_LAUNCH_CHECK();
  return sort_tiles_by_depth(payload, ranges, T, S, long_tiles, sorted_out, st, longest, fork, keys_ok);
}

// The emit step on its own: the fused forward launches it with the buffers' capacity BEFORE it waits for the
// instance total S, so the GPU works through it while the host sleeps on the read-back.
int launch_tile_emit(const float *uv, const float *xyz_c, const float *radius, int ntx, int nty, int N,
                     const unsigned char *mask, const int *rank, const int *offsets,
                     const unsigned long long *hitmask, long long capacity, unsigned int *tkeys,
                     unsigned long long *payload, hipStream_t st) {
  tile_emit_payload_kernel<<<div_up(N, kBlock), kBlock, 0, st>>>(uv, xyz_c, radius, ntx, nty, N, mask, rank, offsets,
                                                                hitmask, capacity, tkeys, payload);
  GS_LAUNCH_CHECK();
  return GSPLAT_OK;
}

// emit -> sort by tile -> ranges -> per-tile depth sort.  pay_a/pay_b: S 64-bit payloads each.
int emit_sort_ranges(const float *uv, const float *xyz_c, const float *radius, int ntx, int nty, int N,
                     const unsigned char *mask, const int *rank, const int *offsets, size_t S, unsigned int *tkeys_a,
                     unsigned int *tkeys_b, unsigned long long *pay_a, unsigned long long *pay_b, int *sorted_out,
                     int *ranges, void *temp, size_t temp_bytes, hipStream_t st, bool already_emitted,
                     const unsigned long long *hitmask) {
  const int num_tiles = ntx * nty;
  if (S == 0) {
    GS_HIP(hipMemsetAsync(ranges, 0, (size_t)(num_tiles + 1) * sizeof(int), st));
    return GSPLAT_OK;
  }
  if (!already_emitted) {
    const int rc = launch_tile_emit(uv, xyz_c, radius, ntx, nty, N, mask, rank, offsets, hitmask, (long long)S, tkeys_a,
                                    pay_a, st);
    if (rc) return rc;
  }
  if ((long long)S > dense_tile_threshold() * (long long)num_tiles) {
    // dense: stable sort by depth bits first (emission order is id order, so equal depths stay id-ordered), then the
    // stable tile sort below leaves every tile's list in (depth, id) order and no per-tile pass is needed
    GS_HIP(rocprim::radix_sort_pairs<OnesweepAlways>(temp, temp_bytes, pay_a, pay_b, tkeys_a, tkeys_b, S, 32, 64, st));
    GS_HIP(rocprim::radix_sort_pairs<OnesweepAlways>(temp, temp_bytes, tkeys_b, tkeys_a, pay_b, pay_a, S, 0,
                                                     tile_bits(num_tiles), st));
    tile_ranges32_kernel<<<div_up((long long)S, kBlock), kBlock, 0, st>>>(tkeys_a, (int)S, num_tiles, ranges,
                                                                        reinterpret_cast<int *>(tkeys_b));
    GS_LAUNCH_CHECK();
    low_words_kernel<<<div_up((long long)S, kBlock), kBlock, 0, st>>>(pay_a, (int)S, sorted_out);
    GS_LAUNCH_CHECK();
    return GSPLAT_OK;
  }
  GS_HIP(rocprim::radix_sort_pairs<OnesweepAlways>(temp, temp_bytes, tkeys_a, tkeys_b, pay_a, pay_b, S, 0,
                                                   tile_bits(num_tiles), st));
  int *long_tiles = reinterpret_cast<int *>(tkeys_a);  // the sort's input keys are dead: reuse them as the list
  tile_ranges32_kernel<<<div_up((long long)S, kBlock), kBlock, 0, st>>>(tkeys_b, (int)S, num_tiles, ranges, long_tiles);
  GS_LAUNCH_CHECK();
  return sort_tiles_by_depth(pay_b, ranges, num_tiles, S, long_tiles, sorted_out, st);
}

}  // namespace gs

extern "C" int gsplat_get_sorted_gaussian_list(const float *uv, const float *xyz, const float *radius, int n_tiles_x,
                                               int n_tiles_y, int N, size_t *sorted_gaussian_count,
                                               int *sorted_gaussians, int *splat_start_end_idx_by_tile_idx,
                                               void *stream) {
  using namespace gs;
  GS_REQUIRE_DEV(uv); GS_REQUIRE_DEV(xyz);
  GS_REQUIRE_DEV(radius);  // read unconditionally by the reference (cuda/culling.cu:210)
  GS_REQUIRE(sorted_gaussian_count != nullptr, "sorted_gaussian_count is null");
  GS_REQUIRE(N >= 0 && n_tiles_x > 0 && n_tiles_y > 0, "bad sizes");
  GS_REQUIRE(((uintptr_t)radius & 15) == 0, "radius must be 16-byte aligned (float4)");
  hipStream_t st = (hipStream_t)stream;
  const int num_tiles = n_tiles_x * n_tiles_y;
  ScratchLock lock;  // library scratch and the pinned count words are process-wide
  int rc = host_words().ensure();
  if (rc) return rc;
  DeviceBuffer &misc = scratch(SCR_MISC);
  rc = misc.reserve(64);
  if (rc) return rc;
  unsigned long long *d_total = misc.as<unsigned long long>();

  if (sorted_gaussians == nullptr) {  // call 1: candidate-pair count only
    GS_HIP(hipMemsetAsync(d_total, 0, sizeof(unsigned long long), st));
    if (N > 0) {
      coarse_count_kernel<<<div_up(N, kBlock), kBlock, 0, st>>>(uv, radius, n_tiles_x, n_tiles_y, N, d_total);
      GS_LAUNCH_CHECK();
    }
    unsigned long long *h = reinterpret_cast<unsigned long long *>(host_words().p);
    GS_HIP(hipMemcpyAsync(h, d_total, sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    GS_HIP(hipStreamSynchronize(st));
    *sorted_gaussian_count = (size_t)h[0];
    return GSPLAT_OK;
  }

  GS_REQUIRE_DEV(sorted_gaussians);
  GS_REQUIRE_DEV(splat_start_end_idx_by_tile_idx);
  if (N == 0) {
    GS_HIP(hipMemsetAsync(splat_start_end_idx_by_tile_idx, 0, (size_t)(num_tiles + 1) * sizeof(int), st));
    return GSPLAT_OK;
  }
  DeviceBuffer &counts = scratch(SCR_COUNTS), &offsets = scratch(SCR_OFFSETS), &tmp = scratch(SCR_TEMP);
  DeviceBuffer &ka = scratch(SCR_KEYS_A), &kb = scratch(SCR_KEYS_B), &pa = scratch(SCR_VALS_B), &pb = scratch(SCR_SPLATS);
  if ((rc = counts.reserve((size_t)(N + 1) * sizeof(int)))) return rc;
  if ((rc = offsets.reserve((size_t)(N + 1) * sizeof(int)))) return rc;
  if ((rc = tmp.reserve(binning_temp_bytes((size_t)N, 1, num_tiles)))) return rc;
  GS_HIP(hipMemsetAsync(counts.as<int>() + N, 0, sizeof(int), st));
  tile_count_kernel<<<div_up(N, kBlock), kBlock, 0, st>>>(uv, radius, n_tiles_x, n_tiles_y, N, nullptr, nullptr,
                                                         counts.as<int>(), nullptr);
  GS_LAUNCH_CHECK();
  if ((rc = scan_counts(N, counts.as<int>(), offsets.as<int>(), tmp.ptr, tmp.bytes, st))) return rc;
  GS_HIP(hipMemcpyAsync(host_words().p, offsets.as<int>() + N, sizeof(int), hipMemcpyDeviceToHost, st));
  GS_HIP(hipStreamSynchronize(st));
  const size_t S = (size_t)host_words().p[0];
  if (S > *sorted_gaussian_count) {
    set_error("%s: %zu instances do not fit the caller's buffer of %zu", __func__, S, *sorted_gaussian_count);
    return GSPLAT_ERR_CAPACITY;
  }
  if ((rc = ka.reserve((S + 1) * sizeof(unsigned int)))) return rc;
  if ((rc = kb.reserve((S + 1) * sizeof(unsigned int)))) return rc;
  if ((rc = pa.reserve((S + 1) * sizeof(unsigned long long)))) return rc;
  if ((rc = pb.reserve((S + 1) * sizeof(unsigned long long)))) return rc;
  if ((rc = tmp.reserve(binning_temp_bytes((size_t)N, S, num_tiles)))) return rc;
  rc = emit_sort_ranges(uv, xyz, radius, n_tiles_x, n_tiles_y, N, nullptr, nullptr, offsets.as<int>(), S,
                        ka.as<unsigned int>(), kb.as<unsigned int>(), pa.as<unsigned long long>(),
                        pb.as<unsigned long long>(), sorted_gaussians, splat_start_end_idx_by_tile_idx, tmp.ptr,
                        tmp.bytes, st, false, nullptr);
  if (rc) return rc;
  // the reference returns only after its blocking read-backs; keep that contract
  GS_HIP(hipStreamSynchronize(st));
  return GSPLAT_OK;
}
