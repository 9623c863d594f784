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
//   r2 = {r, g, b, <16 block-hit bits, filled per tile>}
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

// 16-bit mask: bit (4*by + bx) set when the footprint box may touch the 4x4 pixel block (bx, by) of the tile whose
// first pixel is (x0, y0).  Written so that any NaN makes the test pass.
__device__ __forceinline__ unsigned int subblock_hits(const SplatRec &s, float x0, float y0) {
  const float u = s.r0.x, v = s.r0.y, hx = s.r1.z, hy = s.r1.w;
  const float lo_x = u - hx, hi_x = u + hx, lo_y = v - hy, hi_y = v + hy;
  unsigned int xm = 0u, ym = 0u;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float o = 4.0f * (float)k;
    xm |= (!(hi_x < x0 + o) && !(lo_x > x0 + o + 3.0f)) ? (1u << k) : 0u;
    ym |= (!(hi_y < y0 + o) && !(lo_y > y0 + o + 3.0f)) ? (1u << k) : 0u;
  }
  const unsigned int ys = (ym | (ym << 3) | (ym << 6) | (ym << 9)) & 0x1111u;  // bit k -> bit 4k
  return ys * xm;
}

// exponent of the gaussian at offset (dx, dy) = (u - px, v - py); 2 FMAs on purpose
__device__ __forceinline__ float gauss_power(float a, float b, float c, float dx, float dy) {
  const float ax = a * dx;
  const float bx = b * dx;
  float t = ax * dx;
  t = __builtin_fmaf(c * dy, dy, t);
  return __builtin_fmaf(-bx, dy, -0.5f * t);
}

// ---- DPP helpers
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
// ---- row-level (16 lanes) reduction of NINE values at once, for kernels where every 16-lane row works on its own
// (gaussian, 4x4 pixel block) pair.  Transposed butterfly: the two stages that pair lanes 4 and 8 apart fold two
// registers into one with TWO full-rate DPP adds (the second writes only the banks = 4-lane groups that keep the
// other value, via bank_mask), the two stages inside a quad use a select pair + one DPP add.  21 VALU, no
// v_permlane, and the nine row totals end up in nine different lanes of ONE register, so one ds_add_f32 adds
// all of them.  Lane j of a row (bits b3 b2 b1 b0): b1 == 0 -> total of value 4*b0 + 2*b3 + b2; b1 == 1 -> value 8.
// The block schedules its own DPP hazards (a VGPR written by VALU may be read by a DPP source operand only two
// instructions later): the leading s_nop covers the inputs, the ordering covers the rest.
__device__ __forceinline__ float row_sum9(float v0, float v1, float v2, float v3, float v4, float v5, float v6,
                                          float v7, float v8) {
  float r0, r1, r2, r3, r4, q0, q1, q2, snd, kp, m0, out;
  const unsigned long long odd = 0xAAAAAAAAAAAAAAAAull, bit1 = 0xCCCCCCCCCCCCCCCCull;
  asm volatile(
      "s_nop 1\n\t"
      // stage A: lanes 4 apart (lane bit 2): banks 0,2 keep the even value, banks 1,3 the odd one
      "v_add_f32_dpp %[r0], %[v0], %[v0] row_shl:4 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
      "v_add_f32_dpp %[r0], %[v1], %[v1] row_shr:4 row_mask:0xf bank_mask:0xa bound_ctrl:0\n\t"
      "v_add_f32_dpp %[r1], %[v2], %[v2] row_shl:4 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
      "v_add_f32_dpp %[r1], %[v3], %[v3] row_shr:4 row_mask:0xf bank_mask:0xa bound_ctrl:0\n\t"
      "v_add_f32_dpp %[r2], %[v4], %[v4] row_shl:4 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
      "v_add_f32_dpp %[r2], %[v5], %[v5] row_shr:4 row_mask:0xf bank_mask:0xa bound_ctrl:0\n\t"
      "v_add_f32_dpp %[r3], %[v6], %[v6] row_shl:4 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
      "v_add_f32_dpp %[r3], %[v7], %[v7] row_shr:4 row_mask:0xf bank_mask:0xa bound_ctrl:0\n\t"
      "v_add_f32_dpp %[r4], %[v8], %[v8] row_shl:4 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
      "v_add_f32_dpp %[r4], %[v8], %[v8] row_shr:4 row_mask:0xf bank_mask:0xa bound_ctrl:0\n\t"
      // stage B: lanes 8 apart (lane bit 3): banks 0,1 keep the even register, banks 2,3 the odd one
      "v_add_f32_dpp %[q0], %[r0], %[r0] row_shl:8 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
      "v_add_f32_dpp %[q0], %[r1], %[r1] row_shr:8 row_mask:0xf bank_mask:0xc bound_ctrl:0\n\t"
      "v_add_f32_dpp %[q1], %[r2], %[r2] row_shl:8 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
      "v_add_f32_dpp %[q1], %[r3], %[r3] row_shr:8 row_mask:0xf bank_mask:0xc bound_ctrl:0\n\t"
      "v_add_f32_dpp %[q2], %[r4], %[r4] row_shl:8 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
      "v_add_f32_dpp %[q2], %[r4], %[r4] row_shr:8 row_mask:0xf bank_mask:0xc bound_ctrl:0\n\t"
      // stage C: lane bit 0: even lanes keep q0, odd lanes q1; q2 is summed in place
      "v_cndmask_b32_e64 %[snd], %[q1], %[q0], %[odd]\n\t"
      "v_cndmask_b32_e64 %[kp], %[q0], %[q1], %[odd]\n\t"
      "s_nop 0\n\t"
      "v_add_f32_dpp %[r0], %[q2], %[q2] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %[m0], %[snd], %[kp] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      // stage D: lane bit 1: lanes with the bit clear keep m0 (values 0..7), the others value 8
      "v_cndmask_b32_e64 %[snd], %[r0], %[m0], %[bit1]\n\t"
      "v_cndmask_b32_e64 %[kp], %[m0], %[r0], %[bit1]\n\t"
      "s_nop 1\n\t"
      "v_add_f32_dpp %[out], %[snd], %[kp] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      : [r0] "=&v"(r0), [r1] "=&v"(r1), [r2] "=&v"(r2), [r3] "=&v"(r3), [r4] "=&v"(r4), [q0] "=&v"(q0),
        [q1] "=&v"(q1), [q2] "=&v"(q2), [snd] "=&v"(snd), [kp] "=&v"(kp), [m0] "=&v"(m0), [out] "=&v"(out)
      : [v0] "v"(v0), [v1] "v"(v1), [v2] "v"(v2), [v3] "v"(v3), [v4] "v"(v4), [v5] "v"(v5), [v6] "v"(v6),
        [v7] "v"(v7), [v8] "v"(v8), [odd] "s"(odd), [bit1] "s"(bit1));
  return out;
}
__device__ __host__ __forceinline__ int row_sum9_index(int lane) {
  return (lane & 2) ? 8 : 4 * (lane & 1) + 2 * ((lane >> 3) & 1) + ((lane >> 2) & 1);
}
__device__ __host__ __forceinline__ bool row_sum9_active(int lane) { return !(lane & 2) || (lane & 15) == 2; }

// Blocks are dealt round-robin over the 8 XCDs; give each XCD a contiguous run of tiles so
// neighbouring tiles (which share gaussians) hit the same L2.  Returns >= num_tiles for the
// padding blocks of the last round.
__device__ __forceinline__ int block_to_tile(int block, int num_tiles) {
  const int per_xcd = (num_tiles + 7) >> 3;
  return (block & 7) * per_xcd + (block >> 3);
}
static inline int tile_grid(int num_tiles) { return ((num_tiles + 7) >> 3) * 8; }

}  // namespace gs
