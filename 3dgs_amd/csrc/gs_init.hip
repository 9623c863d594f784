// gs_init.hip -- "next" row f3 of SURVEY.md section 8, device part: the initial gaussians from a sparse point cloud.
//
// Semantics: Gaussians::Initialize (src/gaussian.cpp:38-104): for every point the mean distance to its 3 nearest
// neighbours (the 4 smallest distances of an exact kNN query that includes the point itself, first one dropped; 0.01
// when there is no neighbour) becomes the isotropic log-scale; colours become SH band 0, (rgb/255 - 0.5) / C0; opacity
// logit(0.2); identity rotation.  The reference builds a nanoflann kd-tree on the host and queries it under OpenMP.
//
// Here the exact kNN runs on the GPU over an implicit octree: points are quantised to 21 bits per axis on a cubic
// lattice, sorted by 63-bit Morton code (one rocPRIM radix sort), and a cell of any octree level is a contiguous key
// range found by binary search -- no per-level tables, and the cell size adapts to the local density (COLMAP clouds
// vary by orders of magnitude; a uniform grid degenerates on them).  Every thread picks the finest level at which
// its own cell holds at least k+1 points, scans the 3x3x3 block of cells around it and is done when its k-th
// distance is covered by the block (any unscanned point is at least one cell away); otherwise it retries one level
// coarser.  All distances in double, as the reference (coordinates are doubles).
#include <cstring>  // rocPRIM's texture_cache_iterator.hpp uses memset without including it
#include <rocprim/rocprim.hpp>

#include <algorithm>
#include <cmath>

#include "gs_common.h"

namespace {

constexpr int kBlock = 256;
constexpr int kMaxK = 8;       // neighbours (the reference uses 3)

constexpr int kBits = 21;  // lattice bits per axis

struct Lattice {
  double ox, oy, oz, extent, scale;  // scale = 2^21 / extent
};

__device__ __forceinline__ void atomic_min_d(double *a, double v) {
  unsigned long long *p = reinterpret_cast<unsigned long long *>(a), old = *p, assumed;
  do {
    assumed = old;
    if (__longlong_as_double(assumed) <= v) break;
    old = atomicCAS(p, assumed, (unsigned long long)__double_as_longlong(v));
  } while (assumed != old);
}
__device__ __forceinline__ void atomic_max_d(double *a, double v) {
  unsigned long long *p = reinterpret_cast<unsigned long long *>(a), old = *p, assumed;
  do {
    assumed = old;
    if (__longlong_as_double(assumed) >= v) break;
    old = atomicCAS(p, assumed, (unsigned long long)__double_as_longlong(v));
  } while (assumed != old);
}

// bbox[0..2] = min, bbox[3..5] = max (initialised to +-inf by the host); non-finite points are ignored
__global__ __launch_bounds__(kBlock) void bbox_kernel(const double *__restrict__ pts, int N, double *bbox) {
  __shared__ double s[6][kBlock];
  double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < N; i += gridDim.x * kBlock)
    for (int k = 0; k < 3; ++k) {
      const double v = pts[3 * (size_t)i + k];
      if (v == v && fabs(v) < INFINITY) { lo[k] = fmin(lo[k], v); hi[k] = fmax(hi[k], v); }
    }
  for (int k = 0; k < 3; ++k) { s[k][threadIdx.x] = lo[k]; s[3 + k][threadIdx.x] = hi[k]; }
  __syncthreads();
  for (int off = kBlock / 2; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off)
      for (int k = 0; k < 3; ++k) {
        s[k][threadIdx.x] = fmin(s[k][threadIdx.x], s[k][threadIdx.x + off]);
        s[3 + k][threadIdx.x] = fmax(s[3 + k][threadIdx.x], s[3 + k][threadIdx.x + off]);
      }
    __syncthreads();
  }
  if (threadIdx.x < 3) atomic_min_d(&bbox[threadIdx.x], s[threadIdx.x][0]);
  else if (threadIdx.x < 6) atomic_max_d(&bbox[threadIdx.x], s[threadIdx.x][0]);
}

__device__ __forceinline__ unsigned int lattice_coord(double v, double o, double scale) {
  const double c = floor((v - o) * scale);
  const double top = (double)((1u << kBits) - 1u);
  return (c >= 0.0) ? ((c < top) ? (unsigned int)c : (unsigned int)top) : 0u;  // NaN -> 0
}
__device__ __forceinline__ unsigned long long spread3(unsigned long long x) {  // 21 bits -> every third bit
  x &= 0x1FFFFFull;
  x = (x | x << 32) & 0x1F00000000FFFFull;
  x = (x | x << 16) & 0x1F0000FF0000FFull;
  x = (x | x << 8) & 0x100F00F00F00F00Full;
  x = (x | x << 4) & 0x10C30C30C30C30C3ull;
  x = (x | x << 2) & 0x1249249249249249ull;
  return x;
}
__device__ __forceinline__ unsigned long long morton(unsigned int x, unsigned int y, unsigned int z) {
  return spread3(x) | (spread3(y) << 1) | (spread3(z) << 2);
}

__global__ __launch_bounds__(kBlock) void morton_key_kernel(const double *__restrict__ pts, int N, Lattice g,
                                                            unsigned long long *__restrict__ keys,
                                                            int *__restrict__ vals) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= N) return;
  keys[i] = morton(lattice_coord(pts[3 * (size_t)i], g.ox, g.scale), lattice_coord(pts[3 * (size_t)i + 1], g.oy, g.scale),
                   lattice_coord(pts[3 * (size_t)i + 2], g.oz, g.scale));
  vals[i] = i;
}

__global__ __launch_bounds__(kBlock) void gather_kernel(const double *__restrict__ pts, int N,
                                                        const int *__restrict__ vals, double *__restrict__ sorted) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= N) return;
  const int src = vals[i];
  sorted[3 * (size_t)i] = pts[3 * (size_t)src];
  sorted[3 * (size_t)i + 1] = pts[3 * (size_t)src + 1];
  sorted[3 * (size_t)i + 2] = pts[3 * (size_t)src + 2];
}

// first index whose key is >= k
__device__ __forceinline__ int lower_bound(const unsigned long long *__restrict__ keys, int N, unsigned long long k) {
  int lo = 0, hi = N;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (keys[mid] < k) lo = mid + 1; else hi = mid;
  }
  return lo;
}

template <int W>
struct Best {  // the W smallest squared distances seen so far, ascending (W = neighbours + the point itself)
  double d[W];
  __device__ __forceinline__ void init() {
#pragma unroll
    for (int k = 0; k < W; ++k) d[k] = INFINITY;
  }
  __device__ __forceinline__ void insert(double v) {
    if (!(v < d[W - 1])) return;  // NaN distances are never neighbours
    d[W - 1] = v;
#pragma unroll
    for (int k = W - 1; k > 0; --k)
      if (d[k] < d[k - 1]) { const double t = d[k]; d[k] = d[k - 1]; d[k - 1] = t; }
  }
  // drop the smallest (the point itself), average the square roots of the rest that exist (src/gaussian.cpp:82-91)
  __device__ __forceinline__ float mean_distance() const {
    double total = 0.0;
    int count = 0;
#pragma unroll
    for (int k = 1; k < W; ++k)
      if (d[k] < INFINITY) { total += sqrt(d[k]); ++count; }
    return count > 0 ? (float)(total / (double)count) : 0.01f;
  }
};

// One thread per point, in Morton order (neighbouring threads walk the same key ranges).
template <int W>
__global__ __launch_bounds__(kBlock) void knn_octree_kernel(const double *__restrict__ sorted,
                                                            const unsigned long long *__restrict__ keys, int N,
                                                            Lattice g, const int *__restrict__ vals,
                                                            float *__restrict__ mean_dist) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= N) return;
  const double px = sorted[3 * (size_t)i], py = sorted[3 * (size_t)i + 1], pz = sorted[3 * (size_t)i + 2];
  const unsigned int qx = lattice_coord(px, g.ox, g.scale), qy = lattice_coord(py, g.oy, g.scale),
                     qz = lattice_coord(pz, g.oz, g.scale);
  const unsigned long long code = keys[i];
  // finest level (bits per axis) at which the point's own cell holds at least W points; the count shrinks with level
  int lo = 0, hi = kBits;
  while (lo < hi) {
    const int L = (lo + hi + 1) >> 1, sh = 3 * (kBits - L);
    const unsigned long long first = (code >> sh) << sh;
    const int a = lower_bound(keys, N, first);
    const int b = (sh == 0) ? lower_bound(keys, N, first + 1ull)
                            : ((first + (1ull << sh)) >> 63 ? N : lower_bound(keys, N, first + (1ull << sh)));
    if (b - a >= W) lo = L; else hi = L - 1;
  }
  Best<W> best;
  for (int L = lo;; --L) {
    best.init();
    const int drop = kBits - L, sh = 3 * drop;
    const unsigned int cx = qx >> drop, cy = qy >> drop, cz = qz >> drop, top = (1u << L) - 1u;
    for (int dz = -1; dz <= 1; ++dz) {
      if ((dz < 0 && cz == 0) || (dz > 0 && cz == top)) continue;
      for (int dy = -1; dy <= 1; ++dy) {
        if ((dy < 0 && cy == 0) || (dy > 0 && cy == top)) continue;
        for (int dx = -1; dx <= 1; ++dx) {
          if ((dx < 0 && cx == 0) || (dx > 0 && cx == top)) continue;
          const unsigned long long first = morton(cx + dx, cy + dy, cz + dz) << sh;
          const unsigned long long last = first + (sh ? (1ull << sh) : 1ull);
          const int a = lower_bound(keys, N, first);
          const int b = (last >> 63) ? N : lower_bound(keys, N, last);
          for (int q = a; q < b; ++q) {
            const double ddx = sorted[3 * (size_t)q] - px, ddy = sorted[3 * (size_t)q + 1] - py,
                         ddz = sorted[3 * (size_t)q + 2] - pz;
            best.insert(ddx * ddx + ddy * ddy + ddz * ddz);
          }
        }
      }
    }
    if (L == 0) break;  // the block was the whole lattice
    const double reach = g.extent / (double)(1u << L) * (1.0 - 1e-9);  // unscanned points are at least one cell away
    if (best.d[W - 1] <= reach * reach) break;
  }
  mean_dist[vals[i]] = best.mean_distance();
}

__global__ __launch_bounds__(kBlock) void init_attributes_kernel(const double *__restrict__ pts,
                                                                 const unsigned char *__restrict__ colors, int N,
                                                                 const float *__restrict__ mean_dist,
                                                                 float *__restrict__ xyz, float *__restrict__ rgb,
                                                                 float *__restrict__ opacity,
                                                                 float *__restrict__ scale,
                                                                 float *__restrict__ quaternion) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= N) return;
  const float C0 = 0.28209479177387814f;
  for (int k = 0; k < 3; ++k) {
    xyz[3 * (size_t)i + k] = (float)pts[3 * (size_t)i + k];
    rgb[3 * (size_t)i + k] = ((float)colors[3 * (size_t)i + k] / 255.0f - 0.5f) / C0;
  }
  opacity[i] = logf(0.2f) - logf(1.0f - 0.2f);
  const float ls = logf(mean_dist[i]);
  scale[3 * (size_t)i] = scale[3 * (size_t)i + 1] = scale[3 * (size_t)i + 2] = ls;
  quaternion[4 * (size_t)i] = 1.0f;
  quaternion[4 * (size_t)i + 1] = quaternion[4 * (size_t)i + 2] = quaternion[4 * (size_t)i + 3] = 0.0f;
}

}  // namespace

extern "C" {

int gsplat_knn_mean_distance(const double *points_xyz, int N, int k, float *mean_dist, void *stream) {
  GS_REQUIRE(N >= 0, "negative point count");
  GS_REQUIRE(k >= 1 && k <= kMaxK, "k must be in 1..8");
  if (N == 0) return GSPLAT_OK;
  GS_REQUIRE_DEV(points_xyz); GS_REQUIRE_DEV(mean_dist);
  hipStream_t st = (hipStream_t)stream;
  const int want = k + 1;
  gs::ScratchLock lock;  // library scratch and the pinned count words are process-wide
  int rc = gs::host_words().ensure();
  if (rc) return rc;
  gs::DeviceBuffer &misc = gs::scratch(gs::SCR_MISC);
  if ((rc = misc.reserve(256))) return rc;
  double *d_bbox = misc.as<double>();
  double h_bbox[6] = {INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY};
  GS_HIP(hipMemcpyAsync(d_bbox, h_bbox, sizeof(h_bbox), hipMemcpyHostToDevice, st));
  bbox_kernel<<<std::min(1024u, gs::div_up(N, kBlock)), kBlock, 0, st>>>(points_xyz, N, d_bbox);
  GS_LAUNCH_CHECK();
  GS_HIP(hipMemcpyAsync(h_bbox, d_bbox, sizeof(h_bbox), hipMemcpyDeviceToHost, st));
  GS_HIP(hipStreamSynchronize(st));
  Lattice g;
  double extent = 0.0;
  for (int a = 0; a < 3; ++a)
    if (h_bbox[3 + a] >= h_bbox[a]) extent = std::max(extent, h_bbox[3 + a] - h_bbox[a]);
  if (!(extent > 0.0) || !(extent < INFINITY)) extent = 1.0;
  extent *= 1.0 + 1e-9;  // the largest coordinate stays inside the last cell
  g.ox = std::isfinite(h_bbox[0]) ? h_bbox[0] : 0.0;
  g.oy = std::isfinite(h_bbox[1]) ? h_bbox[1] : 0.0;
  g.oz = std::isfinite(h_bbox[2]) ? h_bbox[2] : 0.0;
  g.extent = extent;
  g.scale = (double)(1u << kBits) / extent;

  gs::DeviceBuffer &ka = gs::scratch(gs::SCR_KEYS_A), &kb = gs::scratch(gs::SCR_KEYS_B), &va = gs::scratch(gs::SCR_COUNTS),
                   &vb = gs::scratch(gs::SCR_OFFSETS), &tmp = gs::scratch(gs::SCR_TEMP), &srt = gs::scratch(gs::SCR_SPLATS);
  size_t temp_bytes = 0;
  (void)rocprim::radix_sort_pairs(nullptr, temp_bytes, (unsigned long long *)nullptr, (unsigned long long *)nullptr,
                                  (int *)nullptr, (int *)nullptr, (size_t)N, 0, 3 * kBits, st);
  if ((rc = ka.reserve((size_t)N * 8)) || (rc = kb.reserve((size_t)N * 8)) || (rc = va.reserve((size_t)N * 4)) ||
      (rc = vb.reserve((size_t)N * 4)) || (rc = tmp.reserve(temp_bytes + 256)) || (rc = srt.reserve((size_t)N * 24)))
    return rc;
  const unsigned int blocks = gs::div_up(N, kBlock);
  morton_key_kernel<<<blocks, kBlock, 0, st>>>(points_xyz, N, g, ka.as<unsigned long long>(), va.as<int>());
  GS_LAUNCH_CHECK();
  GS_HIP(rocprim::radix_sort_pairs(tmp.ptr, temp_bytes, ka.as<unsigned long long>(), kb.as<unsigned long long>(),
                                   va.as<int>(), vb.as<int>(), (size_t)N, 0, 3 * kBits, st));
  gather_kernel<<<blocks, kBlock, 0, st>>>(points_xyz, N, vb.as<int>(), srt.as<double>());
  GS_LAUNCH_CHECK();
#define GS_KNN_CASE(W)                                                                                             \
  case W:                                                                                                          \
    knn_octree_kernel<W><<<blocks, kBlock, 0, st>>>(srt.as<double>(), kb.as<unsigned long long>(), N, g, vb.as<int>(), \
                                                   mean_dist);                                                     \
    break;
  switch (want) { GS_KNN_CASE(2) GS_KNN_CASE(3) GS_KNN_CASE(4) GS_KNN_CASE(5) GS_KNN_CASE(6) GS_KNN_CASE(7) GS_KNN_CASE(8) GS_KNN_CASE(9) }
#undef GS_KNN_CASE
  GS_LAUNCH_CHECK();
  return GSPLAT_OK;
}

int gsplat_initialize_gaussians(const double *points_xyz, const unsigned char *points_rgb, int N, float *xyz, float *rgb,
                                float *opacity, float *scale, float *quaternion, void *stream) {
  GS_REQUIRE(N >= 0, "negative point count");
  if (N == 0) return GSPLAT_OK;
  GS_REQUIRE_DEV(points_xyz); GS_REQUIRE_DEV(points_rgb); GS_REQUIRE_DEV(xyz); GS_REQUIRE_DEV(rgb);
  GS_REQUIRE_DEV(opacity); GS_REQUIRE_DEV(scale); GS_REQUIRE_DEV(quaternion);
  gs::ScratchLock lock;
  gs::DeviceBuffer &md = gs::scratch(gs::SCR_LOSS_MU);  // N mean distances
  int rc = md.reserve((size_t)N * sizeof(float));
  if (rc) return rc;
  if ((rc = gsplat_knn_mean_distance(points_xyz, N, 3, md.as<float>(), stream))) return rc;
  init_attributes_kernel<<<gs::div_up(N, kBlock), kBlock, 0, (hipStream_t)stream>>>(
      points_xyz, points_rgb, N, md.as<float>(), xyz, rgb, opacity, scale, quaternion);
  GS_LAUNCH_CHECK();
  return GSPLAT_OK;
}

}  // extern "C"
