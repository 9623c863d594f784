// gs_pergaussian.hip -- the stand-alone per-gaussian operators of the drop-in surface
// (one thread per gaussian, wave64, 256-thread workgroups).  The math lives in gs_math.h
// and is shared bit-for-bit with the fused preprocess kernels in gs_fused.hip.
#include <cstring>
#include <rocprim/rocprim.hpp>

#include <cmath>

#include "gs_common.h"
#include "gs_math.h"
#include "gs_rows.h"

namespace {

constexpr int kBlock = 256;

__global__ __launch_bounds__(kBlock) void camera_space_kernel(const float *__restrict__ xyz_w,
                                                              const float *__restrict__ view, int N,
                                                              float *__restrict__ xyz_c) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= N) return;
  const gs::Mat34 v = gs::load_view(view);
  float x, y, z;
  gs::camera_space(v, xyz_w[3 * i], xyz_w[3 * i + 1], xyz_w[3 * i + 2], x, y, z);
  xyz_c[3 * i] = x; xyz_c[3 * i + 1] = y; xyz_c[3 * i + 2] = z;
}

__global__ __launch_bounds__(kBlock) void to_screen_kernel(const float *__restrict__ xyz,
                                                           const float *__restrict__ proj, int N, int width,
                                                           int height, float *__restrict__ uv) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= N) return;
  const gs::Mat44 p = gs::load_proj(proj);
  float u, v;
  gs::to_screen(p, xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], width, height, u, v);
  uv[2 * i] = u; uv[2 * i + 1] = v;
}

__global__ __launch_bounds__(kBlock) void cull_kernel(const float *__restrict__ uv, const float *__restrict__ xyz,
                                                      int N, float near_thresh, int padding, int width, int height,
                                                      unsigned char *__restrict__ mask) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= N) return;
  mask[i] = gs::keep(uv[2 * i], uv[2 * i + 1], xyz[3 * i + 2], near_thresh, padding, width, height) ? 1 : 0;
}

__global__ __launch_bounds__(kBlock) void sigma_kernel(const float *__restrict__ q, const float *__restrict__ s,
                                                       int N, float *__restrict__ sigma) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= N) return;
  const float4 qq = reinterpret_cast<const float4 *>(q)[i];
  const gs::RotScale rs = gs::rot_scale(qq.x, qq.y, qq.z, qq.w, s[3 * i], s[3 * i + 1], s[3 * i + 2]);
  float out[6];
  gs::sigma_from(rs, out);
#pragma unroll
  for (int k = 0; k < 6; ++k) sigma[6 * i + k] = out[k];
}

__global__ __launch_bounds__(kBlock) void conic_kernel(const float *__restrict__ xyz, const float *__restrict__ view,
                                                       const float *__restrict__ sigma, float fx, float fy,
                                                       float tan_fovx, float tan_fovy, float mh_dist, int N,
                                                       float *__restrict__ J, float *__restrict__ conic,
                                                       float *__restrict__ radius) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= N) return;
  const gs::Mat34 vw = gs::load_view(view);
  float j[6], s[6], c[3], r[4];
  gs::jacobian(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], fx, fy, tan_fovx, tan_fovy, j);
#pragma unroll
  for (int k = 0; k < 6; ++k) { s[k] = sigma[6 * i + k]; J[6 * i + k] = j[k]; }
  gs::conic_radius(j, s, vw, mh_dist, c, r);
  conic[3 * i] = c[0]; conic[3 * i + 1] = c[1]; conic[3 * i + 2] = c[2];
  if (radius) reinterpret_cast<float4 *>(radius)[i] = make_float4(r[0], r[1], r[2], r[3]);
}

template <int L>
__global__ __launch_bounds__(kBlock) void sh_forward_kernel(const float *__restrict__ xyz,
                                                            const float *__restrict__ sh,
                                                            const float *__restrict__ band0, float cx, float cy,
                                                            float cz, int N, float *__restrict__ rgb) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= N) return;
  constexpr int n = (L + 1) * (L + 1);
  float dx, dy, dz, len;
  gs::view_dir(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], cx, cy, cz, dx, dy, dz, len);
  float out[3];
  gs::sh_to_rgb<L>(sh + (size_t)i * (n - 1) * 3, band0 + 3 * i, dx, dy, dz, out);
  rgb[3 * i] = out[0]; rgb[3 * i + 1] = out[1]; rgb[3 * i + 2] = out[2];
}

// ------------------------------------------------------------------ backward kernels
__global__ __launch_bounds__(kBlock) void to_screen_bwd_kernel(const float *__restrict__ xyz_c,
                                                               const float *__restrict__ proj,
                                                               const float *__restrict__ guv, int N, int width,
                                                               int height, float *__restrict__ g_xyz_c) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= N) return;
  const gs::Mat44 p = gs::load_proj(proj);
  float dx, dy, dz;
  gs::to_screen_bwd(p, xyz_c[3 * i], xyz_c[3 * i + 1], xyz_c[3 * i + 2], guv[2 * i], guv[2 * i + 1], width, height,
                    dx, dy, dz);
  g_xyz_c[3 * i] += dx; g_xyz_c[3 * i + 1] += dy; g_xyz_c[3 * i + 2] += dz;
}

__global__ __launch_bounds__(kBlock) void camera_space_bwd_kernel(const float *__restrict__ view,
                                                                  const float *__restrict__ g_xyz_c, int N,
                                                                  float *__restrict__ g_xyz_w) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= N) return;
  const gs::Mat34 v = gs::load_view(view);
  float dx, dy, dz;
  gs::camera_space_bwd(v, g_xyz_c[3 * i], g_xyz_c[3 * i + 1], g_xyz_c[3 * i + 2], dx, dy, dz);
  g_xyz_w[3 * i] += dx; g_xyz_w[3 * i + 1] += dy; g_xyz_w[3 * i + 2] += dz;
}

__global__ __launch_bounds__(kBlock) void jacobian_bwd_kernel(const float *__restrict__ xyz, float fx, float fy,
                                                              float tan_fovx, float tan_fovy,
                                                              const float *__restrict__ gJ, int N,
                                                              float *__restrict__ g_xyz) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= N) return;
  float dj[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) dj[k] = gJ[6 * i + k];
  float dx, dy, dz;
  gs::jacobian_bwd(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], fx, fy, tan_fovx, tan_fovy, dj, dx, dy, dz);
  if (fabsf(xyz[3 * i + 2]) < 1e-6f) return;  // reference returns before touching the output
  g_xyz[3 * i] += dx; g_xyz[3 * i + 1] += dy; g_xyz[3 * i + 2] += dz;
}

__global__ __launch_bounds__(kBlock) void conic_bwd_kernel(const float *__restrict__ J, const float *__restrict__ sigma,
                                                           const float *__restrict__ view,
                                                           const float *__restrict__ conic,
                                                           const float *__restrict__ gconic, int N,
                                                           float *__restrict__ gJ, float *__restrict__ gsigma) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= N) return;
  const gs::Mat34 vw = gs::load_view(view);
  float j[6], s[6], c[3], dc[3], dJ[6], dS[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) { j[k] = J[6 * i + k]; s[k] = sigma[6 * i + k]; }
#pragma unroll
  for (int k = 0; k < 3; ++k) { c[k] = conic[3 * i + k]; dc[k] = gconic[3 * i + k]; }
  gs::conic_bwd(j, s, vw, c, dc, dJ, dS);
#pragma unroll
  for (int k = 0; k < 6; ++k) { gJ[6 * i + k] += dJ[k]; gsigma[6 * i + k] += dS[k]; }
}

__global__ __launch_bounds__(kBlock) void sigma_bwd_kernel(const float *__restrict__ q, const float *__restrict__ s,
                                                           const float *__restrict__ gsigma, int N,
                                                           float *__restrict__ gq, float *__restrict__ gs_) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= N) return;
  const float4 qq = reinterpret_cast<const float4 *>(q)[i];
  const gs::RotScale rs = gs::rot_scale(qq.x, qq.y, qq.z, qq.w, s[3 * i], s[3 * i + 1], s[3 * i + 2]);
  float g[6], dQ[4], dS[3];
#pragma unroll
  for (int k = 0; k < 6; ++k) g[k] = gsigma[6 * i + k];
  gs::sigma_bwd(rs, g, dQ, dS);
  reinterpret_cast<float4 *>(gq)[i] = make_float4(dQ[0], dQ[1], dQ[2], dQ[3]);
  gs_[3 * i] = dS[0]; gs_[3 * i + 1] = dS[1]; gs_[3 * i + 2] = dS[2];
}

// The coefficient rows (180 B per gaussian at degree 3) and their gradient rows go through LDS as in the fused backward
// (gs_rows.h): a lane reading ITS row touches 64 cache lines per wave instruction and the kernel ran at the cache's tag
// rate (r04: 0.21 ms for 372 MB at 1e6 gaussians); each wave moves its 64 consecutive rows as one linear span instead.
template <int L>
__global__ __launch_bounds__(kBlock) void sh_bwd_kernel(const float *__restrict__ xyz, const float *__restrict__ band0,
                                                        const float *__restrict__ sh, float cx, float cy, float cz,
                                                        const float *__restrict__ grgb, int N,
                                                        float *__restrict__ gsh, float *__restrict__ gband0,
                                                        float *__restrict__ gxyz) {
  constexpr int n = (L + 1) * (L + 1), kRest = (n - 1) * 3;
  __shared__ __attribute__((aligned(16))) float s_sh[kRest > 0 ? kBlock * kRest : 4];
  const int lane = threadIdx.x & 63, wave_first = threadIdx.x - lane;
  const int iw = blockIdx.x * kBlock + wave_first;  // first gaussian of this wave
  if (iw >= N) return;
  const int rows = min(64, N - iw);
  const int i = iw + lane;
  const bool live = lane < rows;
  float *wsh = s_sh + wave_first * kRest;
  if constexpr (kRest > 0) {
    gs::rows_to_lds<kRest>(sh + (size_t)iw * kRest, wsh, rows, lane);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  const int ir = live ? i : iw;  // dead lanes of the last wave recompute row iw and store nothing
  const float gr[3] = {grgb[3 * ir], grgb[3 * ir + 1], grgb[3 * ir + 2]};
  float ox, oy, oz, b0g[3];
  float *row = wsh + (live ? lane : 0) * kRest;  // read as coefficients, overwritten with their gradients
  if (live || kRest == 0)
    gs::sh_bwd<L>(row, band0 + 3 * ir, xyz[3 * ir], xyz[3 * ir + 1], xyz[3 * ir + 2], cx, cy, cz, gr, row, b0g, ox, oy, oz);
  if constexpr (kRest > 0) {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    gs::rows_from_lds<kRest>(gsh + (size_t)iw * kRest, wsh, rows, lane);
  }
  if (!live) return;
  gband0[3 * i] = b0g[0]; gband0[3 * i + 1] = b0g[1]; gband0[3 * i + 2] = b0g[2];
  gxyz[3 * i] += ox; gxyz[3 * i + 1] += oy; gxyz[3 * i + 2] += oz;
}

// ------------------------------------------------------------------ compaction
// compact_masked_array / scatter_masked_array in TWO launches each (the reference host calls them ~8 times per backward and
// ~35 times per optimizer step, cuda/trainer.cu:941-964, 1028-1110; r03 spent five launches per call: memset, mask -> int,
// two kernels of a rocPRIM scan, the row kernel).  Kernel 1: kMaskSlices workgroups, each over one contiguous slice of
// the mask, leave every row's rank INSIDE its slice and the slice's count; kernel 2 (the row kernel) scans the kMaskSlices
// counts in LDS itself and adds a row's slice base.
constexpr int kMaskSlices = 256;
__device__ __host__ inline long long mask_slice_first(long long N, int s) { return N * s / kMaskSlices; }
// the slice that holds row i: the largest s with mask_slice_first(N, s) <= i, i.e. floor(((i + 1) * kMaskSlices - 1) / N)
// -- without the 64-bit division (~100 instructions per element, more than the rest of the row kernel): a float
// estimate, corrected by at most one step either way against the exact slice bounds (a multiply and a shift each)
__device__ inline int mask_slice_of(int N, unsigned int i, float slices_per_row) {
  int s = min(kMaskSlices - 1, (int)(((float)i + 0.5f) * slices_per_row));
  while (mask_slice_first(N, s) > (long long)i) --s;
  while (mask_slice_first(N, s + 1) <= (long long)i) ++s;
  return s;
}

__global__ __launch_bounds__(1024) void mask_slice_ranks_kernel(const unsigned char *__restrict__ mask, int N,
                                                                int *__restrict__ ranks, int *__restrict__ slice_counts) {
  __shared__ int s_wave[16];
  __shared__ int s_run;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const long long lo = mask_slice_first(N, blockIdx.x), hi = mask_slice_first(N, blockIdx.x + 1);
  if (threadIdx.x == 0) s_run = 0;
  __syncthreads();
  for (long long base = lo; base < hi; base += 1024) {
    const long long i = base + threadIdx.x;
    const bool k = i < hi && mask[i] != 0;
    const unsigned long long bal = __ballot(k);
    if (lane == 0) s_wave[w] = __popcll(bal);
    __syncthreads();
    int before = s_run, total = 0;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int c = s_wave[q];
      before += q < w ? c : 0;
      total += c;
    }
    if (i < hi) ranks[i] = before + __popcll(bal & ((1ull << lane) - 1ull));
    __syncthreads();
    if (threadIdx.x == 0) s_run += total;
    __syncthreads();
  }
  if (threadIdx.x == 0) slice_counts[blockIdx.x] = s_run;
}

// exclusive scan of the kMaskSlices slice counts into LDS (s_base[kMaskSlices] = the total); kBlock == kMaskSlices threads
__device__ __forceinline__ void mask_slice_bases(const int *__restrict__ slice_counts, int *s_base) {
  static_assert(kBlock == kMaskSlices, "one thread per slice");
  __shared__ int s_wsum[kBlock / 64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
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
  if (threadIdx.x == kBlock - 1) s_base[kMaskSlices] = before + incl;
  __syncthreads();
}

// A row's compacted slot = its slice's base + its rank inside the slice.  One thread per source element, kRowsPer elements
// per thread (so that the scan of the slice counts is paid once per 2048 elements), 32-bit index arithmetic and -- for the
// strides the reference instantiates (cuda_data.cuh:106-167 call sites: 1, 2, 3, 4, 6 and the SH strides 9, 24, 45) -- a
// compile-time stride: with a run-time 64-bit e / stride the kernel spent its time in the division (r04: 0.5 ms for the
// eight compactions of one backward_pass at 1e6 gaussians, 0.2 ms of it HBM time).
constexpr int kRowsPer = 8;
template <bool kScatter, int kStride>
__global__ __launch_bounds__(kBlock) void masked_rows_kernel(const float *__restrict__ src,
                                                             const unsigned char *__restrict__ mask,
                                                             const int *__restrict__ ranks,
                                                             const int *__restrict__ slice_counts, int N,
                                                             unsigned int total, int stride_rt, float *__restrict__ dst,
                                                             unsigned int room_rows) {
  __shared__ int s_base[kMaskSlices + 1];
  mask_slice_bases(slice_counts, s_base);
  const unsigned int stride = kStride > 0 ? (unsigned int)kStride : (unsigned int)stride_rt;
  const float slices_per_row = (float)kMaskSlices / (float)N;
#pragma unroll
  for (int r = 0; r < kRowsPer; ++r) {
    const unsigned int e = (blockIdx.x * kRowsPer + r) * kBlock + threadIdx.x;
    if (e >= total) return;
    const unsigned int i = e / stride, k = e - i * stride;
    if (!mask[i]) continue;
    const size_t slot = (size_t)(s_base[mask_slice_of(N, i, slices_per_row)] + ranks[i]);
    if (slot >= room_rows) continue;  // the compacted side holds room_rows rows (the caller's count: cuda_data.cuh:106-127)
    if (kScatter) dst[e] = src[slot * stride + k];
    else dst[slot * stride + k] = src[e];
  }
}

// r05: compaction when the compacted slot of every kept row is already known (slots[i] = exclusive scan of the mask:
// what a forward leaves in its context, gsplat_context_last_compaction) -- one launch, one thread per ROW for the narrow
// strides (its slot is looked up once and its 4..24 bytes copied), one thread per element for the wide ones.
template <int kStride>
__global__ __launch_bounds__(kBlock) void ranked_rows_kernel(const float *__restrict__ src,
                                                             const unsigned char *__restrict__ mask,
                                                             const int *__restrict__ slots, int N, int stride_rt,
                                                             float *__restrict__ dst, unsigned int room_rows) {
  const unsigned int stride = kStride > 0 ? (unsigned int)kStride : (unsigned int)stride_rt;
  if constexpr (kStride > 0 && kStride <= 6) {
    const unsigned int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= (unsigned int)N || !mask[i]) return;
    const unsigned int slot = (unsigned int)slots[i];
    if (slot >= room_rows) return;
    const float *s = src + (size_t)i * kStride;
    float *d = dst + (size_t)slot * kStride;
#pragma unroll
    for (int k = 0; k < kStride; ++k) d[k] = s[k];
  } else if constexpr (kStride > 6) {
    // wide rows (the SH strides 9 / 24 / 45): a thread moves one 16-byte piece of a row (rows are 4-byte aligned only:
    // gs::f4u), so a 180-byte row is twelve lanes and a wave instruction moves five whole rows
    constexpr unsigned int kPieces = (kStride + 3) / 4;
    const unsigned int e = blockIdx.x * kBlock + threadIdx.x;
    if (e >= (unsigned int)N * kPieces) return;
    const unsigned int i = e / kPieces, piece = e - i * kPieces;
    if (!mask[i]) return;
    const unsigned int slot = (unsigned int)slots[i];
    if (slot >= room_rows) return;
    const float *s = src + (size_t)i * kStride + 4 * piece;
    float *d = dst + (size_t)slot * kStride + 4 * piece;
    if (4 * piece + 3 < (unsigned int)kStride) {
      *reinterpret_cast<gs::f4u *>(d) = *reinterpret_cast<const gs::f4u *>(s);
    } else {
#pragma unroll
      for (int k = 0; k < 3; ++k)
        if (4 * piece + k < (unsigned int)kStride) d[k] = s[k];
    }
  } else {
    const unsigned int e = blockIdx.x * kBlock + threadIdx.x;
    if (e >= (unsigned int)N * stride) return;
    const unsigned int i = e / stride, k = e - i * stride;
    if (!mask[i]) return;
    const unsigned int slot = (unsigned int)slots[i];
    if (slot < room_rows) dst[(size_t)slot * stride + k] = src[e];
  }
}

// rows[slot] <- i for every set mask entry i (the index list of a compaction), slot < room
__global__ __launch_bounds__(kBlock) void selected_rows_kernel(const unsigned char *__restrict__ mask,
                                                               const int *__restrict__ ranks,
                                                               const int *__restrict__ slice_counts, int N,
                                                               int *__restrict__ rows, unsigned int room) {
  __shared__ int s_base[kMaskSlices + 1];
  mask_slice_bases(slice_counts, s_base);
  const float slices_per_row = (float)kMaskSlices / (float)N;
#pragma unroll
  for (int r = 0; r < kRowsPer; ++r) {
    const unsigned int i = (blockIdx.x * kRowsPer + r) * kBlock + threadIdx.x;
    if (i >= (unsigned int)N) return;
    if (!mask[i]) continue;
    const unsigned int slot = (unsigned int)(s_base[mask_slice_of(N, i, slices_per_row)] + ranks[i]);
    if (slot < room) rows[slot] = (int)i;
  }
}

// dst[0..n) <- value: 16-byte stores, a fixed grid of persistent workgroups striding over the buffer (a fill is pure
// HBM writes: eight 16-byte stores in flight per thread, no tail effects from a grid sized by n)
__global__ __launch_bounds__(kBlock) void fill_f32_kernel(float *__restrict__ dst, size_t n, float value) {
  const size_t tid = (size_t)blockIdx.x * kBlock + threadIdx.x, nthreads = (size_t)gridDim.x * kBlock;
  const size_t head = min(n, (size_t)((16 - ((uintptr_t)dst & 15)) & 15) / 4);  // floats before the first 16-byte boundary
  if (tid < head) dst[tid] = value;
  gs::f4u *v = reinterpret_cast<gs::f4u *>(dst + head);
  const size_t nv = (n - head) / 4;
  const gs::f4u q = {value, value, value, value};
  for (size_t i = tid; i < nv; i += nthreads) v[i] = q;
  const size_t done = head + nv * 4;
  if (tid < n - done) dst[done + tid] = value;
}

template <bool kScatter>
static int launch_masked_rows(const float *src, const unsigned char *mask, const int *ranks, const int *slice_counts, int N,
                              int stride, float *dst, hipStream_t st, unsigned int room_rows = 0xFFFFFFFFu) {
  const long long total = (long long)N * stride;
  if (total > 0xFFFFFFF0ll - (long long)kRowsPer * kBlock) {
    gs::set_error("compact / scatter_masked_array: %d rows of %d elements exceed the 32-bit element index", N, stride);
    return GSPLAT_ERR_INVALID_ARG;
  }
  const dim3 grid(gs::div_up(total, (long long)kRowsPer * kBlock)), block(kBlock);
#define GS_ROWS(S) masked_rows_kernel<kScatter, S><<<grid, block, 0, st>>>(src, mask, ranks, slice_counts, N, (unsigned int)total, stride, dst, room_rows)
  switch (stride) {
    case 1: GS_ROWS(1); break;
    case 2: GS_ROWS(2); break;
    case 3: GS_ROWS(3); break;
    case 4: GS_ROWS(4); break;
    case 6: GS_ROWS(6); break;
    case 9: GS_ROWS(9); break;
    case 24: GS_ROWS(24); break;
    case 45: GS_ROWS(45); break;
    default: GS_ROWS(0); break;
  }
#undef GS_ROWS
  GS_LAUNCH_CHECK();
  return GSPLAT_OK;
}

}  // namespace

namespace gs {
// slice-local ranks of a byte mask + the kMaskSlices slice counts (library scratch); see masked_rows_kernel
static int mask_slice_ranks(const unsigned char *mask, int N, int **ranks, int **slice_counts, hipStream_t st) {
  DeviceBuffer &rk = scratch(SCR_OFFSETS), &cnt = scratch(SCR_COUNTS);
  int rc = rk.reserve((size_t)(N + 1) * sizeof(int));
  if (rc) return rc;
  if ((rc = cnt.reserve((size_t)kMaskSlices * sizeof(int)))) return rc;
  mask_slice_ranks_kernel<<<kMaskSlices, 1024, 0, st>>>(mask, N, rk.as<int>(), cnt.as<int>());
  GS_LAUNCH_CHECK();
  *ranks = rk.as<int>();
  *slice_counts = cnt.as<int>();
  return GSPLAT_OK;
}
}  // namespace gs

extern "C" {

int gsplat_compute_camera_space_points(const float *xyz_w, const float *view, int N, float *xyz_c, void *stream) {
  GS_REQUIRE_DEV(xyz_w); GS_REQUIRE_DEV(view); GS_REQUIRE_DEV(xyz_c);
  GS_REQUIRE(N >= 0, "N < 0");
  if (N == 0) return GSPLAT_OK;
  camera_space_kernel<<<gs::div_up(N, kBlock), kBlock, 0, (hipStream_t)stream>>>(xyz_w, view, N, xyz_c);
  GS_LAUNCH_CHECK();
  return GSPLAT_OK;
}

int gsplat_project_to_screen(const float *xyz, const float *proj, int N, int width, int height, float *uv,
                             void *stream) {
  GS_REQUIRE_DEV(xyz); GS_REQUIRE_DEV(proj); GS_REQUIRE_DEV(uv);
  GS_REQUIRE(N >= 0, "N < 0");
  if (N == 0) return GSPLAT_OK;
  to_screen_kernel<<<gs::div_up(N, kBlock), kBlock, 0, (hipStream_t)stream>>>(xyz, proj, N, width, height, uv);
  GS_LAUNCH_CHECK();
  return GSPLAT_OK;
}

int gsplat_cull_gaussians(const float *uv, const float *xyz, int N, float near_thresh, int padding, int width,
                          int height, unsigned char *mask, void *stream) {
  GS_REQUIRE_DEV(uv); GS_REQUIRE_DEV(xyz); GS_REQUIRE_DEV(mask);
  GS_REQUIRE(N >= 0, "N < 0");
  if (N == 0) return GSPLAT_OK;
  cull_kernel<<<gs::div_up(N, kBlock), kBlock, 0, (hipStream_t)stream>>>(uv, xyz, N, near_thresh, padding, width,
                                                                        height, mask);
  GS_LAUNCH_CHECK();
  return GSPLAT_OK;
}

int gsplat_compute_sigma(const float *quaternion, const float *scale, int N, float *sigma, void *stream) {
  GS_REQUIRE_DEV(quaternion); GS_REQUIRE_DEV(scale); GS_REQUIRE_DEV(sigma);
  GS_REQUIRE(N >= 0, "N < 0");
  GS_REQUIRE(((uintptr_t)quaternion & 15) == 0, "quaternion must be 16-byte aligned");
  if (N == 0) return GSPLAT_OK;
  sigma_kernel<<<gs::div_up(N, kBlock), kBlock, 0, (hipStream_t)stream>>>(quaternion, scale, N, sigma);
  GS_LAUNCH_CHECK();
  return GSPLAT_OK;
}

int gsplat_compute_conic(const float *xyz, const float *view, const float *sigma, float focal_x, float focal_y,
                         float tan_fovx, float tan_fovy, float mh_dist, int N, float *J, float *conic,
                         float *radius, void *stream) {
  GS_REQUIRE_DEV(xyz); GS_REQUIRE_DEV(sigma); GS_REQUIRE_DEV(view); GS_REQUIRE_DEV(J); GS_REQUIRE_DEV(conic);
  // the reference does not assert `radius` (cuda/gaussian.cu:241-245) but always writes it
  GS_REQUIRE_DEV(radius);
  GS_REQUIRE(((uintptr_t)radius & 15) == 0, "radius must be 16-byte aligned (float4)");
  GS_REQUIRE(N >= 0, "N < 0");
  if (N == 0) return GSPLAT_OK;
  conic_kernel<<<gs::div_up(N, kBlock), kBlock, 0, (hipStream_t)stream>>>(xyz, view, sigma, focal_x, focal_y, tan_fovx,
                                                                         tan_fovy, mh_dist, N, J, conic, radius);
  GS_LAUNCH_CHECK();
  return GSPLAT_OK;
}

int gsplat_precompute_spherical_harmonics(const float *xyz, const float *sh_coefficients,
                                          const float *sh_coeffs_band_0, float campos_x, float campos_y,
                                          float campos_z, int l_max, int N, float *rgb, void *stream) {
  GS_REQUIRE_DEV(xyz); GS_REQUIRE_DEV(sh_coeffs_band_0); GS_REQUIRE_DEV(rgb);
  GS_REQUIRE(l_max >= 0 && l_max <= 3, "l_max must be 0..3");
  if (l_max > 0) GS_REQUIRE_DEV(sh_coefficients);
  GS_REQUIRE(N >= 0, "N < 0");
  if (N == 0) return GSPLAT_OK;
  const dim3 g(gs::div_up(N, kBlock)), b(kBlock);
  hipStream_t st = (hipStream_t)stream;
  switch (l_max) {
    case 0: sh_forward_kernel<0><<<g, b, 0, st>>>(xyz, sh_coefficients, sh_coeffs_band_0, campos_x, campos_y, campos_z, N, rgb); break;
    case 1: sh_forward_kernel<1><<<g, b, 0, st>>>(xyz, sh_coefficients, sh_coeffs_band_0, campos_x, campos_y, campos_z, N, rgb); break;
    case 2: sh_forward_kernel<2><<<g, b, 0, st>>>(xyz, sh_coefficients, sh_coeffs_band_0, campos_x, campos_y, campos_z, N, rgb); break;
    default: sh_forward_kernel<3><<<g, b, 0, st>>>(xyz, sh_coefficients, sh_coeffs_band_0, campos_x, campos_y, campos_z, N, rgb); break;
  }
  GS_LAUNCH_CHECK();
  return GSPLAT_OK;
}

int gsplat_project_to_screen_backward(const float *xyz_c, const float *proj, const float *uv_grad_out, int N,
                                      int width, int height, float *xyz_c_grad_in, void *stream) {
  GS_REQUIRE_DEV(xyz_c); GS_REQUIRE_DEV(proj); GS_REQUIRE_DEV(uv_grad_out); GS_REQUIRE_DEV(xyz_c_grad_in);
  GS_REQUIRE(N >= 0, "N < 0");
  if (N == 0) return GSPLAT_OK;
  to_screen_bwd_kernel<<<gs::div_up(N, kBlock), kBlock, 0, (hipStream_t)stream>>>(xyz_c, proj, uv_grad_out, N, width,
                                                                                 height, xyz_c_grad_in);
  GS_LAUNCH_CHECK();
  return GSPLAT_OK;
}

int gsplat_compute_camera_space_points_backward(const float *xyz_w, const float *view, const float *xyz_c_grad_out,
                                                int N, float *xyz_w_grad_in, void *stream) {
  GS_REQUIRE_DEV(xyz_w); GS_REQUIRE_DEV(view); GS_REQUIRE_DEV(xyz_c_grad_out); GS_REQUIRE_DEV(xyz_w_grad_in);
  GS_REQUIRE(N >= 0, "N < 0");
  if (N == 0) return GSPLAT_OK;
  camera_space_bwd_kernel<<<gs::div_up(N, kBlock), kBlock, 0, (hipStream_t)stream>>>(view, xyz_c_grad_out, N,
                                                                                    xyz_w_grad_in);
  GS_LAUNCH_CHECK();
  return GSPLAT_OK;
}

int gsplat_compute_projection_jacobian_backward(const float *xyz_c, float focal_x, float focal_y, float tan_fovx,
                                                float tan_fovy, const float *J_grad_out, int N,
                                                float *xyz_c_grad_in, void *stream) {
  GS_REQUIRE_DEV(xyz_c); GS_REQUIRE_DEV(J_grad_out); GS_REQUIRE_DEV(xyz_c_grad_in);
  GS_REQUIRE(N >= 0, "N < 0");
  if (N == 0) return GSPLAT_OK;
  jacobian_bwd_kernel<<<gs::div_up(N, kBlock), kBlock, 0, (hipStream_t)stream>>>(xyz_c, focal_x, focal_y, tan_fovx,
                                                                                tan_fovy, J_grad_out, N, xyz_c_grad_in);
  GS_LAUNCH_CHECK();
  return GSPLAT_OK;
}

int gsplat_compute_conic_backward(const float *J, const float *sigma, const float *view, const float *conic,
                                  const float *conic_grad_out, int N, float *J_grad_in, float *sigma_grad_in,
                                  void *stream) {
  GS_REQUIRE_DEV(J); GS_REQUIRE_DEV(sigma); GS_REQUIRE_DEV(view); GS_REQUIRE_DEV(conic);
  GS_REQUIRE_DEV(conic_grad_out); GS_REQUIRE_DEV(J_grad_in); GS_REQUIRE_DEV(sigma_grad_in);
  GS_REQUIRE(N >= 0, "N < 0");
  if (N == 0) return GSPLAT_OK;
  conic_bwd_kernel<<<gs::div_up(N, kBlock), kBlock, 0, (hipStream_t)stream>>>(J, sigma, view, conic, conic_grad_out, N,
                                                                             J_grad_in, sigma_grad_in);
  GS_LAUNCH_CHECK();
  return GSPLAT_OK;
}

int gsplat_compute_sigma_backward(const float *quaternion, const float *scale, const float *sigma_grad_out, int N,
                                  float *quaternion_grad_in, float *scale_grad_in, void *stream) {
  GS_REQUIRE_DEV(quaternion); GS_REQUIRE_DEV(scale); GS_REQUIRE_DEV(sigma_grad_out);
  GS_REQUIRE_DEV(quaternion_grad_in); GS_REQUIRE_DEV(scale_grad_in);
  GS_REQUIRE(((uintptr_t)quaternion & 15) == 0 && ((uintptr_t)quaternion_grad_in & 15) == 0,
             "quaternion buffers must be 16-byte aligned");
  GS_REQUIRE(N >= 0, "N < 0");
  if (N == 0) return GSPLAT_OK;
  sigma_bwd_kernel<<<gs::div_up(N, kBlock), kBlock, 0, (hipStream_t)stream>>>(quaternion, scale, sigma_grad_out, N,
                                                                             quaternion_grad_in, scale_grad_in);
  GS_LAUNCH_CHECK();
  return GSPLAT_OK;
}

int gsplat_precompute_spherical_harmonics_backward(const float *xyz_c, const float *rgb_vals,
                                                   const float *sh_coeffs, float campos_x, float campos_y,
                                                   float campos_z, const float *rgb_grad_out, int l_max, int N,
                                                   float *sh_grad_in, float *sh_grad_band_0_in,
                                                   float *xyz_c_grad_in, void *stream) {
  GS_REQUIRE_DEV(xyz_c); GS_REQUIRE_DEV(rgb_vals); GS_REQUIRE_DEV(rgb_grad_out);
  GS_REQUIRE_DEV(sh_grad_band_0_in); GS_REQUIRE_DEV(xyz_c_grad_in);
  GS_REQUIRE(l_max >= 0 && l_max <= 3, "l_max must be 0..3");
  if (l_max > 0) { GS_REQUIRE_DEV(sh_coeffs); GS_REQUIRE_DEV(sh_grad_in); }
  GS_REQUIRE(N >= 0, "N < 0");
  if (N == 0) return GSPLAT_OK;
  const dim3 g(gs::div_up(N, kBlock)), b(kBlock);
  hipStream_t st = (hipStream_t)stream;
  switch (l_max) {
    case 0: sh_bwd_kernel<0><<<g, b, 0, st>>>(xyz_c, rgb_vals, sh_coeffs, campos_x, campos_y, campos_z, rgb_grad_out, N, sh_grad_in, sh_grad_band_0_in, xyz_c_grad_in); break;
    case 1: sh_bwd_kernel<1><<<g, b, 0, st>>>(xyz_c, rgb_vals, sh_coeffs, campos_x, campos_y, campos_z, rgb_grad_out, N, sh_grad_in, sh_grad_band_0_in, xyz_c_grad_in); break;
    case 2: sh_bwd_kernel<2><<<g, b, 0, st>>>(xyz_c, rgb_vals, sh_coeffs, campos_x, campos_y, campos_z, rgb_grad_out, N, sh_grad_in, sh_grad_band_0_in, xyz_c_grad_in); break;
    default: sh_bwd_kernel<3><<<g, b, 0, st>>>(xyz_c, rgb_vals, sh_coeffs, campos_x, campos_y, campos_z, rgb_grad_out, N, sh_grad_in, sh_grad_band_0_in, xyz_c_grad_in); break;
  }
  GS_LAUNCH_CHECK();
  return GSPLAT_OK;
}

int gsplat_compact_masked_array(const float *src, const unsigned char *mask, int N, int stride, float *dst,
                                int *num_selected, void *stream) {
  return gsplat_compact_masked_array_bounded(src, mask, N, stride, dst, N, num_selected, stream);
}

int gsplat_compact_masked_array_bounded(const float *src, const unsigned char *mask, int N, int stride, float *dst,
                                        int dst_rows, int *num_selected, void *stream) {
  GS_REQUIRE(N >= 0 && stride > 0 && dst_rows >= 0, "N < 0, stride <= 0 or dst_rows < 0");
  if (num_selected) *num_selected = 0;
  if (N == 0) return GSPLAT_OK;  // empty input is legal (tests/cuda_data_test.cpp CompactMaskedArrayEmpty)
  GS_REQUIRE_DEV(src); GS_REQUIRE_DEV(mask); GS_REQUIRE_DEV(dst);
  hipStream_t st = (hipStream_t)stream;
  gs::ScratchLock lock;  // library scratch and the pinned count words are process-wide
  int *ranks = nullptr, *slice_counts = nullptr;
  int rc = gs::mask_slice_ranks(mask, N, &ranks, &slice_counts, st);
  if (rc) return rc;
  if ((rc = launch_masked_rows<false>(src, mask, ranks, slice_counts, N, stride, dst, st, (unsigned int)dst_rows))) return rc;
  if (!num_selected) return GSPLAT_OK;  // the caller knows the count (the reference's call sites pass num_culled)
  rc = gs::host_words().ensure();
  if (rc) return rc;
  static_assert(kMaskSlices <= 256, "the pinned words hold the slice counts");
  GS_HIP(hipMemcpyAsync(gs::host_words().p, slice_counts, kMaskSlices * sizeof(int), hipMemcpyDeviceToHost, st));
  GS_HIP(hipStreamSynchronize(st));
  int selected = 0;
  for (int k = 0; k < kMaskSlices; ++k) selected += gs::host_words().p[k];
  *num_selected = selected;
  return GSPLAT_OK;
}

int gsplat_compact_rows_ranked(const float *src, const unsigned char *mask, const int *slots, int N, int stride,
                               float *dst, int dst_rows, void *stream) {
  GS_REQUIRE(N >= 0 && stride > 0 && dst_rows >= 0, "N < 0, stride <= 0 or dst_rows < 0");
  if (N == 0 || dst_rows == 0) return GSPLAT_OK;
  GS_REQUIRE_DEV(src); GS_REQUIRE_DEV(mask); GS_REQUIRE_DEV(slots); GS_REQUIRE_DEV(dst);
  GS_REQUIRE((long long)N * stride < 0xFFFFFFF0ll, "rows x stride exceed the 32-bit element index");
  hipStream_t st = (hipStream_t)stream;
  const unsigned int room = (unsigned int)dst_rows;
#define GS_RANKED(S, THREADS) ranked_rows_kernel<S><<<gs::div_up((long long)(THREADS), kBlock), kBlock, 0, st>>>(src, mask, slots, N, stride, dst, room)
  switch (stride) {
    case 1: GS_RANKED(1, N); break;
    case 2: GS_RANKED(2, N); break;
    case 3: GS_RANKED(3, N); break;
    case 4: GS_RANKED(4, N); break;
    case 6: GS_RANKED(6, N); break;
    case 9: GS_RANKED(9, (long long)N * 3); break;    // one thread per 16-byte piece of a row
    case 24: GS_RANKED(24, (long long)N * 6); break;
    case 45: GS_RANKED(45, (long long)N * 12); break;
    default: GS_RANKED(0, (long long)N * stride); break;
  }
#undef GS_RANKED
  GS_LAUNCH_CHECK();
  return GSPLAT_OK;
}

int gsplat_mask_selected_rows(const unsigned char *mask, int N, int *rows, int rows_cap, void *stream) {
  GS_REQUIRE(N >= 0 && rows_cap >= 0, "negative size");
  if (N == 0 || rows_cap == 0) return GSPLAT_OK;
  GS_REQUIRE_DEV(mask); GS_REQUIRE_DEV(rows);
  hipStream_t st = (hipStream_t)stream;
  gs::ScratchLock lock;
  int *ranks = nullptr, *slice_counts = nullptr;
  int rc = gs::mask_slice_ranks(mask, N, &ranks, &slice_counts, st);
  if (rc) return rc;
  selected_rows_kernel<<<gs::div_up(N, (long long)kRowsPer * kBlock), kBlock, 0, st>>>(mask, ranks, slice_counts, N, rows,
                                                                                      (unsigned int)rows_cap);
  GS_LAUNCH_CHECK();
  return GSPLAT_OK;
}

int gsplat_fill_f32(float *dst, size_t n, float value, void *stream) {
  if (n == 0) return GSPLAT_OK;
  GS_REQUIRE_DEV(dst);
  GS_REQUIRE(((uintptr_t)dst & 3) == 0, "dst must be 4-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  // (one kernel for every value, zero included: hipMemsetAsync reached 3.5 TB/s on the twelve fills of zero_grads)
  const long long blocks = std::min<long long>(8 * 256, (long long)((n / 4 + kBlock - 1) / kBlock) + 1);
  fill_f32_kernel<<<(unsigned int)blocks, kBlock, 0, st>>>(dst, n, value);
  GS_LAUNCH_CHECK();
  return GSPLAT_OK;
}

int gsplat_scatter_masked_array(const float *src, const unsigned char *mask, int N, int stride, float *dst,
                                void *stream) {
  GS_REQUIRE(N >= 0 && stride > 0, "N < 0 or stride <= 0");
  if (N == 0) return GSPLAT_OK;
  GS_REQUIRE_DEV(mask); GS_REQUIRE_DEV(dst);
  if (src == nullptr) return GSPLAT_OK;  // nothing selected (cuda_data.cuh:154-157)
  GS_REQUIRE_DEV(src);
  hipStream_t st = (hipStream_t)stream;
  gs::ScratchLock lock;  // library scratch and the pinned count words are process-wide
  int *ranks = nullptr, *slice_counts = nullptr;
  int rc = gs::mask_slice_ranks(mask, N, &ranks, &slice_counts, st);
  if (rc) return rc;
  return launch_masked_rows<true>(src, mask, ranks, slice_counts, N, stride, dst, st);
}

}  // extern "C"
