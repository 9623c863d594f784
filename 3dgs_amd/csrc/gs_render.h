// gs_render.h -- device helpers shared by the compositing kernels (gs_render.hip) and the
// fused preprocess kernel (gs_fused.hip): the 48-byte "splat record" a tile stages in LDS,
// wave64 cross-lane reductions on DPP, and the XCD-aware block -> tile map.
#pragma once
#include <hip/hip_runtime.h>

namespace gs {

constexpr float kAlphaMin = 0.00392156862f;  // 1/255, cuda/render.cu:74
constexpr float kAlphaMax = 0.99f;           // cuda/render.cu:73
constexpr float kTMin = 0.0001f;             // cuda/render.cu:77

// Everything the compositing loops need about one gaussian, as three float4:
//   r0 = {u, v, conic00, conic01}
//   r1 = {conic11, sigmoid(opacity), hx, hy}   hx/hy: half extents of the axis-aligned box that
//                                              contains every pixel centre with alpha >= 1/255
//   r2 = {r, g, b, <quadrant hit bits, filled per tile>}
struct SplatRec { float4 r0, r1, r2; };

__device__ __forceinline__ float sigmoid_fast(float logit) { return 1.0f / (1.0f + __expf(-logit)); }

// Conservative footprint: alpha = min(.99, opa*exp(p)) can reach 1/255 only where
// p >= -tau, tau = ln(255*opa), i.e. inside the ellipse 0.5 d^T C d <= tau, whose bounding
// box has half widths sqrt(2 tau C^-1_xx), sqrt(2 tau C^-1_yy).  Degenerate / NaN conics
// get an infinite box (always visited), opacities that can never reach 1/255 an empty one.
__device__ __forceinline__ void footprint(float a, float b, float c, float opa, float &hx, float &hy) {
  const float det = a * c - b * b;
  if (!(opa * 255.0f >= 0.999f)) {
    hx = hy = -INFINITY;
    if (opa != opa) hx = hy = INFINITY;
    return;
  }
  if (!(det > 0.0f) || !(a > 0.0f) || !(c > 0.0f) || !(det < INFINITY)) {
    hx = hy = INFINITY;
    return;
  }
  const float tau2 = 2.0f * fmaxf(0.0f, logf(255.0f * opa)) + 1e-3f;
  hx = sqrtf(tau2 * c / det) * 1.0005f + 0.01f;
  hy = sqrtf(tau2 * a / det) * 1.0005f + 0.01f;
}

__device__ __forceinline__ SplatRec make_record(float u, float v, float a, float b, float c, float logit, float r,
                                                float g, float bl) {
  SplatRec s;
  const float opa = sigmoid_fast(logit);
  float hx, hy;
  footprint(a, b, c, opa, hx, hy);
  s.r0 = make_float4(u, v, a, b);
  s.r1 = make_float4(c, opa, hx, hy);
  s.r2 = make_float4(r, g, bl, 0.0f);
  return s;
}

// 4-bit mask: bit q set when the footprint box may touch 8x8 quadrant q of the tile whose
// first pixel is (x0, y0).  Written so that any NaN makes the test pass.
__device__ __forceinline__ unsigned int quadrant_hits(const SplatRec &s, float x0, float y0) {
  const float u = s.r0.x, v = s.r0.y, hx = s.r1.z, hy = s.r1.w;
  const float lo_x = u - hx, hi_x = u + hx, lo_y = v - hy, hi_y = v + hy;
  const bool xl = !(hi_x < x0) && !(lo_x > x0 + 7.0f);
  const bool xr = !(hi_x < x0 + 8.0f) && !(lo_x > x0 + 15.0f);
  const bool yt = !(hi_y < y0) && !(lo_y > y0 + 7.0f);
  const bool yb = !(hi_y < y0 + 8.0f) && !(lo_y > y0 + 15.0f);
  return (xl && yt ? 1u : 0u) | (xr && yt ? 2u : 0u) | (xl && yb ? 4u : 0u) | (xr && yb ? 8u : 0u);
}

// exponent of the gaussian at offset (dx, dy) = (u - px, v - py); 2 FMAs on purpose
__device__ __forceinline__ float gauss_power(float a, float b, float c, float dx, float dy) {
  const float ax = a * dx;
  const float bx = b * dx;
  float t = ax * dx;
  t = __builtin_fmaf(c * dy, dy, t);
  return __builtin_fmaf(-bx, dy, -0.5f * t);
}

// ---- wave64 reduction of NINE values at once.
// A plain butterfly costs 6 DPP adds per value (54).  Here the values are folded pairwise while the lane groups
// halve: v_permlane32_swap exchanges the upper half of one register with the lower half of another, so ONE swap +
// ONE add turns two registers into one that holds the half-wave sums of both values (gfx950 only); the same with
// v_permlane16_swap across 16-lane rows; the last four steps stay inside a row on DPP.  28 VALU instead of 54.
// Result: in a lane of row r (= lane >> 4), q0 holds the total of value {0,2,1,3}[r], q1 of value {4,6,5,7}[r],
// and q2 (row 0 only) of value 8.
typedef unsigned int gs_u2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float fold32(float a, float b) {
  const gs_u2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  return __uint_as_float(r.x) + __uint_as_float(r.y);
}
__device__ __forceinline__ float fold16(float a, float b) {
  const gs_u2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  return __uint_as_float(r.x) + __uint_as_float(r.y);
}
template <int kCtrl>
__device__ __forceinline__ float dpp_add(float v) {
  const int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(v), kCtrl, 0xF, 0xF, false);
  return v + __int_as_float(moved);
}
__device__ __forceinline__ float row_sum(float v) {  // every lane of a 16-lane row ends with the row total
  v = dpp_add<0xB1>(v);   // quad_perm [1,0,3,2]
  v = dpp_add<0x4E>(v);   // quad_perm [2,3,0,1]
  v = dpp_add<0x141>(v);  // row_half_mirror
  v = dpp_add<0x140>(v);  // row_mirror
  return v;
}
struct Sum9 { float q0, q1, q2; };
__device__ __forceinline__ Sum9 wave_sum9(float v0, float v1, float v2, float v3, float v4, float v5, float v6,
                                          float v7, float v8) {
  const float r0 = fold32(v0, v1), r1 = fold32(v2, v3), r2 = fold32(v4, v5), r3 = fold32(v6, v7);
  const float r4 = fold32(v8, 0.0f);
  Sum9 s;
  s.q0 = row_sum(fold16(r0, r1));
  s.q1 = row_sum(fold16(r2, r3));
  s.q2 = row_sum(fold16(r4, 0.0f));
  return s;
}
// which of the nine values a lane's q0 / q1 holds
__device__ __forceinline__ int sum9_index_q0(int lane) { return ((lane >> 4) & 1) * 2 + (lane >> 5); }
__device__ __forceinline__ int sum9_index_q1(int lane) { return 4 + ((lane >> 4) & 1) * 2 + (lane >> 5); }
__device__ __forceinline__ int wave_max_int(int v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = max(v, __shfl_xor(v, off, 64));
  return v;
}

// Blocks are dealt round-robin over the 8 XCDs; give each XCD a contiguous run of tiles so
// neighbouring tiles (which share gaussians) hit the same L2.  Returns >= num_tiles for the
// padding blocks of the last round.
__device__ __forceinline__ int block_to_tile(int block, int num_tiles) {
  const int per_xcd = (num_tiles + 7) >> 3;
  return (block & 7) * per_xcd + (block >> 3);
}
static inline int tile_grid(int num_tiles) { return ((num_tiles + 7) >> 3) * 8; }

}  // namespace gs
