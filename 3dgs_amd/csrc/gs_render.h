// gs_render.h -- device helpers shared by the compositing kernels (gs_render.hip) and the
// fused preprocess kernel (gs_fused.hip): the 48-byte "splat record" a tile stages in LDS,
// wave64 cross-lane reductions on DPP, and the XCD-aware block -> tile map.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

namespace gs {

constexpr float kAlphaMin = 0.00392156862f;  // 1/255, cuda/render.cu:74
constexpr float kAlphaMax = 0.99f;           // cuda/render.cu:73
constexpr float kTMin = 0.0001f;             // cuda/render.cu:77
// log2(0.99) = -0.01449957, minus two ulps of a value near one under exp2: the forward's cap in the exponent domain
// (render_fwd_kernel staging) -- v_exp_f32 of it is at most 0.99f
constexpr float kLog2AlphaMax = -0.01449975f;

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
// Returns tau2 = 2 tau (+ margin), the level of the quadratic form a dx^2 + 2 b dx dy + c dy^2 at that ellipse.
__device__ __forceinline__ float footprint(float a, float b, float c, float opa, float &hx, float &hy) {
  const float det = a * c - b * b;
  if (!(opa * 255.0f >= 0.999f)) {
    hx = hy = -INFINITY;
    if (opa != opa) hx = hy = INFINITY;
    return 0.0f;
  }
  if (!(det > 0.0f) || !(a > 0.0f) || !(c > 0.0f) || !(det < INFINITY)) {
    hx = hy = INFINITY;
    return INFINITY;
  }
  const float tau2 = 2.0f * fmaxf(0.0f, logf(255.0f * opa)) + 1e-3f;
  hx = sqrtf(tau2 * c / det) * 1.0005f + 0.01f;
  hy = sqrtf(tau2 * a / det) * 1.0005f + 0.01f;
  return tau2;
}

__device__ __forceinline__ SplatRec make_record(float u, float v, float a, float b, float c, float logit, float r,
                                                float g, float bl) {
  SplatRec s;
  const float opa = sigmoid_fast(logit);
  float hx, hy;
  const float tau2 = footprint(a, b, c, opa, hx, hy);
  s.r0 = make_float4(u, v, a, b);
  s.r1 = make_float4(c, opa, hx, hy);
  s.r2 = make_float4(r, g, bl, tau2);
  return s;
}

// 16-bit mask: bit (4*by + bx) set when the ellipse alpha >= 1/255 may touch the 4x4 pixel block (bx, by) of the tile
// whose first pixel is (x0, y0) -- the ellipse itself, not its bounding box.  For each of the tile's four strips of pixel rows (dy in [Y0, Y0+3]
// about the centre) the ellipse q(dx, dy) = a dx^2 + 2 b dx dy + c dy^2 <= tau2 covers the dx-interval
// [min_y L(y), max_y R(y)], L/R(y) = -(b/a) y -/+ sqrt((tau2 - (det/a) y^2) / a); R is concave with its maximum at the
// dy of the ellipse's rightmost point, -(b/c) hx, L convex with its minimum at +(b/c) hx, so each bound is one
// evaluation at that dy clamped into the strip: two square roots per strip, about 200 instructions per (gaussian,
// tile) -- once, the forward hands the masks to the backward.  On the benchmark scene the box admits 17.4 M
// (gaussian, block) pairs, the ellipse 14.3 M: 17 % fewer loop trips in both compositing kernels.  Degenerate conics
// (infinite box) visit every block; what a NaN centre visits does not matter (its alpha is NaN on every pixel: no splat).
__device__ __forceinline__ unsigned int block_hits(const SplatRec &s, float x0, float y0) {
  const float a = s.r0.z, b = s.r0.w, c = s.r1.x, hx = s.r1.z, hy = s.r1.w, tau2 = s.r2.w;
  if (!(hx < INFINITY) || !(hy < INFINITY)) return 0xFFFFu;  // degenerate conic or NaN: always visited
  if (!(hx > 0.0f)) return 0u;                                // opacity can never reach 1/255
  const float inv_a = 1.0f / a, boa = b * inv_a;
  const float kappa = c - b * boa;    // det / a
  const float ystar = -(b / c) * hx;  // dy of the rightmost point; the leftmost one sits at -ystar
  const float U = s.r0.x - x0, V = s.r0.y - y0;  // centre relative to the tile's first pixel
  unsigned int out = 0u;
#pragma unroll 1  // one strip at a time: unrolled, the four strips' temporaries push the callers past their VGPR step
  for (int l = 0; l < 4; ++l) {
    const float Y0 = 4.0f * (float)l - V, Y1 = Y0 + 3.0f;
    const float ylo = fmaxf(Y0, -hy), yhi = fminf(Y1, hy);
    const float yr = __builtin_amdgcn_fmed3f(ystar, ylo, yhi), yl = __builtin_amdgcn_fmed3f(-ystar, ylo, yhi);
    const float wr = __builtin_sqrtf(fmaxf(0.0f, (tau2 - kappa * yr * yr) * inv_a));
    const float wl = __builtin_sqrtf(fmaxf(0.0f, (tau2 - kappa * yl * yl) * inv_a));
    const float R = wr - boa * yr, L = -wl - boa * yl;
    const float Rm = R + (0.01f + 5e-4f * fabsf(R)), Lm = L - (0.01f + 5e-4f * fabsf(L));
    // blocks k with 4k - U <= Rm and 4k + 3 - U >= Lm: k in [ceil((Lm + U - 3) / 4), floor((Rm + U) / 4)]
    const int k_hi = min(3, (int)floorf((Rm + U) * 0.25f)), k_lo = max(0, (int)ceilf((Lm + U - 3.0f) * 0.25f));
    unsigned int row = k_hi >= k_lo ? ((2u << k_hi) - (1u << k_lo)) : 0u;
    if (ylo > yhi) row = 0u;  // the strip lies above or below the ellipse
    out |= row << (4 * l);
  }
  return out;
}

// ---- staged (LDS) form of a record.  The compositing loops evaluate opa * exp(min(0, power)) as one base-2
// exponential with the opacity folded into the exponent,
//   log2(opa * exp(power)) = a2 dx^2 + b2 dx dy + c2 dy^2 + log2(opa),   (a2, b2, c2) = -log2(e) * (a/2, b, c/2),
// which takes four VALU instructions fewer per (pixel, gaussian) than power -> * log2(e) -> exp -> * opa.
// r0 = (u, v, a2, b2), r1 = (c2, log2 opa, opa, hy), r2 = (rgb, hit mask).  Both directions use the same form, so
// the backward recomputes the forward's alpha bit for bit.
constexpr float kLog2e = 1.44269504088896340736f;

__device__ __forceinline__ void stage_record(SplatRec &s) {
  const float opa = s.r1.y;
  s.r0.z *= -0.5f * kLog2e;
  s.r0.w *= -kLog2e;
  s.r1.x *= -0.5f * kLog2e;
  s.r1.y = __log2f(opa);  // -inf for opacity 0: alpha 0
  s.r1.z = opa;
}

// the all-zero sentinel record a list is padded with: alpha = 2^-inf = 0 at every pixel
__device__ __forceinline__ float4 sentinel_r1() { return make_float4(0.0f, -INFINITY, 0.0f, 0.0f); }

// log2 of the unclamped alpha; min(.., log2 opa) is the reference's min(0, power)
__device__ __forceinline__ float log2_alpha(float a2, float b2, float c2, float lopa, float dx, float dy) {
  float t = a2 * dx;
  t = __builtin_fmaf(b2, dy, t);
  float q;  // three-operand form on purpose: as v_fmac the compiler first copies lopa (still needed for the min)
  asm("v_fma_f32 %0, %1, %2, %3" : "=v"(q) : "v"(t), "v"(dx), "v"(lopa));
  q = __builtin_fmaf(c2 * dy, dy, q);
  float r;  // fminf() would first canonicalise the loaded lopa (one more VALU instruction per evaluation)
  asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(q), "v"(lopa));
  return r;
}

// the forward's form: the clamp is min(log2 opa, log2 0.99) (staged per gaussian), so the result is the CAPPED alpha
__device__ __forceinline__ float staged_alpha_capped(float a2, float b2, float c2, float lopa, float lbound, float dx, float dy) {
  float t = a2 * dx;
  t = __builtin_fmaf(b2, dy, t);
  float q = __builtin_fmaf(t, dx, lopa);  // (lopa dies here: a plain v_fmac, no copy -- and no asm, whose borders cost a wait state)
  q = __builtin_fmaf(c2 * dy, dy, q);
  float r;
  asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(q), "v"(lbound));
  return __builtin_amdgcn_exp2f(r);
}

// the conic entries back from their staged form (flush of the backward): a = a2 * kConicDiag, b = b2 * kConicOff
constexpr float kConicDiag = -2.0f / kLog2e, kConicOff = -1.0f / kLog2e;

// opa * exp(min(0, power)) from a staged record
__device__ __forceinline__ float staged_alpha(float a2, float b2, float c2, float lopa, float dx, float dy) {
  return __builtin_amdgcn_exp2f(log2_alpha(a2, b2, c2, lopa, dx, dy));
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

// ---- the backward's nine row sums, products included.  Every sum is (a per-pixel factor) x (a lane constant):
//   rgb_c = sum aT * grad_c(pixel),   S_m = sum gp * m(cx, cy),  m in {1, cx, cy, cx^2, cx cy, cy^2},
// with (cx, cy) the pixel's position relative to the TILE centre (the moments about the gaussian's own centre follow
// per gaussian at flush time: dx = X - cx with X = u - tile centre).  So the first butterfly stage needs no separate
// products: with the partner lane l ^ 7 (row_half_mirror, which also swaps the two banks of a pair) a register of the
// stage is  F * W_own + dpp(F) * W_partner  = one v_mul + one v_fmac_dpp, W_* loop-invariant per lane.
// Ten instructions instead of nine products + ten adds.  Later stages as in row_sum9.
struct RowWeights {  // banks 0,2 (lane bit 2 clear) | banks 1,3
  float a_own, a_par;  // grad0 | grad1          x aT
  float b_own, b_par;  // grad2 | (unused)       x aT
  float c_own, c_par;  // 1     | cx             x gp
  float d_own, d_par;  // cy    | cx^2           x gp
  float e_own, e_par;  // cx cy | cy^2           x gp
};

// lane = lane in the wave, (cx, cy) and grad[3] this lane's pixel; *_par are the same quantities of lane ^ 7
__device__ __forceinline__ RowWeights make_row_weights(int lane, float cx, float cy, float g0, float g1, float g2) {
  const bool odd = lane & 4;
  auto partner = [](float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141 /* row_half_mirror */, 0xF, 0xF, false));
  };
  RowWeights w;
  w.a_own = odd ? g1 : g0;  w.a_par = partner(odd ? g0 : g1);  // the partner sits in the other kind of bank
  w.b_own = g2;             w.b_par = partner(g2);
  w.c_own = odd ? cx : 1.0f;       w.c_par = partner(odd ? 1.0f : cx);
  w.d_own = odd ? cx * cx : cy;    w.d_par = partner(odd ? cy : cx * cx);
  w.e_own = odd ? cy * cy : cx * cy;  w.e_par = partner(odd ? cx * cy : cy * cy);
  return w;
}

__device__ __forceinline__ float row_moments9(float aT, float gp, const RowWeights &w) {
  float r0, r1, r2, r3, r4;  // few temporaries on purpose: the backward sits at a VGPR-occupancy step
  const unsigned long long odd = 0xAAAAAAAAAAAAAAAAull, bit1 = 0xCCCCCCCCCCCCCCCCull;
  asm volatile(
      // stage A (lane bit 2, partner l ^ 7): the five plain products first, they double as the wait states between
      // the instruction that wrote gp and its first DPP read
      "v_mul_f32 %[r0], %[aT], %[a_own]\n\t"
      "v_mul_f32 %[r1], %[aT], %[b_own]\n\t"
      "v_mul_f32 %[r2], %[gp], %[c_own]\n\t"
      "v_mul_f32 %[r3], %[gp], %[d_own]\n\t"
      "v_mul_f32 %[r4], %[gp], %[e_own]\n\t"
      "v_fmac_f32_dpp %[r0], %[aT], %[a_par] row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f32_dpp %[r1], %[aT], %[b_par] row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f32_dpp %[r2], %[gp], %[c_par] row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f32_dpp %[r3], %[gp], %[d_par] row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f32_dpp %[r4], %[gp], %[e_par] row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      // stage B: lanes 8 apart (lane bit 3), in place: banks 0,1 keep the even register, banks 2,3 take the odd one
      // (row_shl:8 with bound_ctrl:0 adds 0 in banks 2,3); the (cx cy | cy^2) register pairs with itself: one rotate
      "v_add_f32_dpp %[r0], %[r0], %[r0] row_shl:8 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
      "v_add_f32_dpp %[r0], %[r1], %[r1] row_shr:8 row_mask:0xf bank_mask:0xc bound_ctrl:0\n\t"
      "v_add_f32_dpp %[r2], %[r2], %[r2] row_shl:8 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
      "v_add_f32_dpp %[r2], %[r3], %[r3] row_shr:8 row_mask:0xf bank_mask:0xc bound_ctrl:0\n\t"
      "v_add_f32_dpp %[r4], %[r4], %[r4] row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
      // stage C: lane bit 0: even lanes keep r0, odd lanes r2; r4 is summed in place
      "v_cndmask_b32_e64 %[r1], %[r2], %[r0], %[odd]\n\t"
      "v_cndmask_b32_e64 %[r3], %[r0], %[r2], %[odd]\n\t"
      "s_nop 0\n\t"
      "v_add_f32_dpp %[r4], %[r4], %[r4] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %[r1], %[r1], %[r3] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      // stage D: lane bit 1: lanes with the bit clear keep r1 (values 0..7), the others the r4 pair
      "v_cndmask_b32_e64 %[r0], %[r4], %[r1], %[bit1]\n\t"
      "v_cndmask_b32_e64 %[r2], %[r1], %[r4], %[bit1]\n\t"
      "s_nop 1\n\t"
      "v_add_f32_dpp %[r0], %[r0], %[r2] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      : [r0] "=&v"(r0), [r1] "=&v"(r1), [r2] "=&v"(r2), [r3] "=&v"(r3), [r4] "=&v"(r4)
      : [aT] "v"(aT), [gp] "v"(gp), [a_own] "v"(w.a_own), [a_par] "v"(w.a_par), [b_own] "v"(w.b_own),
        [b_par] "v"(w.b_par), [c_own] "v"(w.c_own), [c_par] "v"(w.c_par), [d_own] "v"(w.d_own), [d_par] "v"(w.d_par),
        [e_own] "v"(w.e_own), [e_par] "v"(w.e_par), [odd] "s"(odd), [bit1] "s"(bit1));
  return r0;
}
// which of the nine sums a lane of `out` holds: 0..2 rgb, 3 S_1, 4 S_cx, 5 S_cy, 6 S_cx2, 7 S_cxcy, 8 S_cy2; -1: none
__device__ __host__ __forceinline__ int row_moments9_index(int lane) {
  const int b0 = lane & 1, b1 = (lane >> 1) & 1, b2 = (lane >> 2) & 1, b3 = (lane >> 3) & 1;
  if (b1) return (b0 == 0 && b3 == 0) ? 7 + b2 : -1;  // the (cx cy | cy^2) register: lanes 2 and 6 of the row
  const int a = 4 * b0 + 2 * b3 + b2;                  // A0..A7 = rgb0 rgb1 rgb2 (dup) S_1 S_cx S_cy S_cx2
  return a < 3 ? a : (a == 3 ? -1 : a - 1);
}

// ---- r04: the same nine row sums with the first TWO butterfly stages folded into the products.  A quad = the four
// lanes of one pixel row of the 4x4 block.  Three registers, each the quad-partial of up to four sums (one per lane of
// the quad), built as  F * W_0 + qp1(F) * W_1 + qp2(F) * W_2 + qp3(F) * W_3  (qp_k = the quad permutation lane -> lane ^ k,
// W_k = the weight of the sum THIS lane is responsible for, at the pixel of lane ^ k: loop-invariant lane constants):
//   A from aT: lanes 0..2 of the quad -> rgb0, rgb1, rgb2        (lane 3: weight 0)
//   B from gp: lanes 0..3             -> S_1, S_cx, S_cy, S_cx2
//   C from gp: lanes 0..1             -> S_cxcy, S_cy2           (lanes 2, 3: weight 0)
// = 3 v_mul + 9 v_fmac_dpp; then the four quads (= DPP banks) of the row are summed with the masked-bank pairs of
// row_sum9: lanes 4 apart fold A and B into one register (banks 0,2: A, banks 1,3: B; 2 instructions) and C onto itself (1),
// lanes 8 apart fold that pair (banks 0,1: the A / B totals; banks 2,3: C's; 2 instructions).  17 VALU + one wait state
// instead of 22 + three; bank 0 ends with the rgb totals, bank 1 with S_1 S_cx S_cy S_cx2, lanes 8, 9 with S_cxcy S_cy2.
struct QuadWeights { float a[4], b[4], c[4]; };

// lane = lane in the wave, (cx, cy) and grad[3] this lane's pixel (cx relative to the tile centre: the lanes of a quad are
// four pixels in a row, lane ^ k sits (lane ^ k) - lane pixels to the right)
__device__ __forceinline__ QuadWeights make_quad_weights(int lane, float cx, float cy, float g0, float g1, float g2) {
  const int p = lane & 3;
  QuadWeights w;
  auto qp = [](float v, auto ctrl) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), decltype(ctrl)::value, 0xF, 0xF, false));
  };
  // rgb: the lane that will RECEIVE through qp_k is lane ^ k, responsible for channel p ^ k: this lane offers that channel
  auto offer = [&](int k) { const int ch = p ^ k; return ch == 0 ? g0 : ch == 1 ? g1 : ch == 2 ? g2 : 0.0f; };
  w.a[0] = offer(0);
  w.a[1] = qp(offer(1), std::integral_constant<int, 0xB1>{});  // quad_perm [1,0,3,2]
  w.a[2] = qp(offer(2), std::integral_constant<int, 0x4E>{});  // quad_perm [2,3,0,1]
  w.a[3] = qp(offer(3), std::integral_constant<int, 0x1B>{});  // quad_perm [3,2,1,0]
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float cxk = cx + (float)((p ^ k) - p);  // pixel of lane ^ k (same row: cy is shared)
    w.b[k] = p == 0 ? 1.0f : p == 1 ? cxk : p == 2 ? cy : cxk * cxk;
    w.c[k] = p == 0 ? cxk * cy : p == 1 ? cy * cy : 0.0f;
  }
  return w;
}

__device__ __forceinline__ float row_moments9q(float aT, float gp, const QuadWeights &w) {
  float A, B, C;
  asm volatile(
      // the three plain products first: they are also the wait states between the instruction that wrote gp and its
      // first DPP read
      "v_mul_f32 %[A], %[aT], %[a0]\n\t"
      "v_mul_f32 %[B], %[gp], %[b0]\n\t"
      "v_mul_f32 %[C], %[gp], %[c0]\n\t"
      "v_fmac_f32_dpp %[C], %[gp], %[c1] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f32_dpp %[C], %[gp], %[c2] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f32_dpp %[C], %[gp], %[c3] quad_perm:[3,2,1,0] row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f32_dpp %[A], %[aT], %[a1] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f32_dpp %[A], %[aT], %[a2] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f32_dpp %[A], %[aT], %[a3] quad_perm:[3,2,1,0] row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f32_dpp %[B], %[gp], %[b1] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f32_dpp %[B], %[gp], %[b2] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f32_dpp %[B], %[gp], %[b3] quad_perm:[3,2,1,0] row_mask:0xf bank_mask:0xf\n\t"
      // quads 4 lanes apart: C onto itself (banks 0 and 2 end with the pair sums), then A | B into A (banks 0,2 | 1,3)
      "v_add_f32_dpp %[A], %[A], %[A] row_shl:4 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
      "v_add_f32_dpp %[A], %[B], %[B] row_shr:4 row_mask:0xf bank_mask:0xa bound_ctrl:0\n\t"
      "v_add_f32_dpp %[C], %[C], %[C] row_shl:4 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
      // quads 8 lanes apart: banks 0,1 keep the A | B totals, banks 2,3 take C's (bank 2: the total, bank 3: unused).
      // (a VGPR written by VALU may be read through DPP two wait states later: C's fold and the s_nop cover A)
      "s_nop 0\n\t"
      "v_add_f32_dpp %[A], %[A], %[A] row_shl:8 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
      "v_add_f32_dpp %[A], %[C], %[C] row_shr:8 row_mask:0xf bank_mask:0xc bound_ctrl:0\n\t"
      : [A] "=&v"(A), [B] "=&v"(B), [C] "=&v"(C)
      : [aT] "v"(aT), [gp] "v"(gp), [a0] "v"(w.a[0]), [a1] "v"(w.a[1]), [a2] "v"(w.a[2]), [a3] "v"(w.a[3]),
        [b0] "v"(w.b[0]), [b1] "v"(w.b[1]), [b2] "v"(w.b[2]), [b3] "v"(w.b[3]), [c0] "v"(w.c[0]), [c1] "v"(w.c[1]),
        [c2] "v"(w.c[2]), [c3] "v"(w.c[3]));
  return A;
}
// which of the nine sums a lane holds after row_moments9q (0..2 rgb, 3 S_1, 4 S_cx, 5 S_cy, 6 S_cx2, 7 S_cxcy, 8 S_cy2; -1: none)
__device__ __host__ __forceinline__ int row_moments9q_index(int lane) {
  const int j = lane & 15;
  return j < 3 ? j : (j >= 4 && j < 8) ? j - 1 : j == 8 ? 7 : j == 9 ? 8 : -1;
}

// ---- r04 (second half): the same nine row sums in 14 VALU.  The four lanes of a quad are four pixels of ONE pixel row,
// so inside a quad cy is a constant: the quad-partials of S_cy, S_cxcy and S_cy2 are cy x S_1, cy x S_cx and cy^2 x S_1 of
// the same quad.  The quad stage therefore builds only two registers,
//   A from aT: lanes 0..2 of the quad -> rgb0, rgb1, rgb2           (lane 3: weight 0)
//   B from gp: lanes 0..3             -> S_1, S_cx, S_cx2, S_1 (again)
// (2 v_mul + 6 v_fmac_dpp), one more product forms  Q = B x (cy, cy, 0, cy^2)  = the quad-partials of S_cy, S_cxcy, -,
// S_cy2, and the four quads of the row are summed as in row_moments9q (A | B into banks 0,2 | 1,3, Q onto itself, then
// the halves: 5 masked DPP adds).  Bank 0 ends with the rgb totals, bank 1 with S_1 S_cx S_cx2 (S_1), bank 2 with S_cy
// S_cxcy - S_cy2.  The wait state the last fold needs is filled by the caller's own instruction: the LDS address of the
// atomic that follows (addr = off * 5 + acc_lane), which the loop needs anyway.
struct RowsWeights { float a[4], b[4], q; };

__device__ __forceinline__ RowsWeights make_rows_weights(int lane, float cx, float cy, float g0, float g1, float g2) {
  const int p = lane & 3;
  RowsWeights w;
  auto qp = [](float v, auto ctrl) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), decltype(ctrl)::value, 0xF, 0xF, false));
  };
  auto offer = [&](int k) { const int ch = p ^ k; return ch == 0 ? g0 : ch == 1 ? g1 : ch == 2 ? g2 : 0.0f; };
  w.a[0] = offer(0);
  w.a[1] = qp(offer(1), std::integral_constant<int, 0xB1>{});  // quad_perm [1,0,3,2]
  w.a[2] = qp(offer(2), std::integral_constant<int, 0x4E>{});  // quad_perm [2,3,0,1]
  w.a[3] = qp(offer(3), std::integral_constant<int, 0x1B>{});  // quad_perm [3,2,1,0]
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float cxk = cx + (float)((p ^ k) - p);  // pixel of lane ^ k (same pixel row)
    w.b[k] = p == 1 ? cxk : p == 2 ? cxk * cxk : 1.0f;
  }
  w.q = p == 3 ? cy * cy : p == 2 ? 0.0f : cy;
  return w;
}

// returns the register of the totals; `addr` <- off * 5 + acc_lane (the caller's atomic address, computed in the wait state)
__device__ __forceinline__ float row_moments9r(float aT, float gp, const RowsWeights &w, unsigned int off, unsigned int acc_lane,
                                               unsigned int &addr) {
  float A, B, Q;
  asm volatile(
      // the two plain products first: the wait states between the instructions that wrote aT / gp and their first DPP read
      "v_mul_f32 %[A], %[aT], %[a0]\n\t"
      "v_mul_f32 %[B], %[gp], %[b0]\n\t"
      "v_fmac_f32_dpp %[A], %[aT], %[a1] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f32_dpp %[B], %[gp], %[b1] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f32_dpp %[A], %[aT], %[a2] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f32_dpp %[B], %[gp], %[b2] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f32_dpp %[A], %[aT], %[a3] quad_perm:[3,2,1,0] row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f32_dpp %[B], %[gp], %[b3] quad_perm:[3,2,1,0] row_mask:0xf bank_mask:0xf\n\t"
      "v_mul_f32 %[Q], %[B], %[wq]\n\t"
      // quads 4 lanes apart: A | B into A (banks 0,2 | 1,3), Q onto itself (banks 0 and 2 end with the pair sums)
      "v_add_f32_dpp %[A], %[A], %[A] row_shl:4 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
      "v_add_f32_dpp %[A], %[B], %[B] row_shr:4 row_mask:0xf bank_mask:0xa bound_ctrl:0\n\t"
      "v_add_f32_dpp %[Q], %[Q], %[Q] row_shl:4 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
      // (a VGPR written by VALU may be read through DPP two wait states later: Q's fold and the address cover A)
      "v_mad_u32_u24 %[addr], %[off], 5, %[acc]\n\t"
      // quads 8 lanes apart: banks 0,1 keep the A | B totals, banks 2,3 take Q's (bank 2: the total, bank 3: unused)
      "v_add_f32_dpp %[A], %[A], %[A] row_shl:8 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
      "v_add_f32_dpp %[A], %[Q], %[Q] row_shr:8 row_mask:0xf bank_mask:0xc bound_ctrl:0\n\t"
      : [A] "=&v"(A), [B] "=&v"(B), [Q] "=&v"(Q), [addr] "=&v"(addr)
      : [aT] "v"(aT), [gp] "v"(gp), [a0] "v"(w.a[0]), [a1] "v"(w.a[1]), [a2] "v"(w.a[2]), [a3] "v"(w.a[3]),
        [b0] "v"(w.b[0]), [b1] "v"(w.b[1]), [b2] "v"(w.b[2]), [b3] "v"(w.b[3]), [wq] "v"(w.q), [off] "v"(off),
        [acc] "v"(acc_lane));
  return A;
}
// which of the nine sums a lane holds after row_moments9r (0..2 rgb, 3 S_1, 4 S_cx, 5 S_cy, 6 S_cx2, 7 S_cxcy, 8 S_cy2; -1: none)
__device__ __host__ __forceinline__ int row_moments9r_index(int lane) {
  const int j = lane & 15;
  return j < 3 ? j : j == 4 ? 3 : j == 5 ? 4 : j == 6 ? 6 : j == 8 ? 5 : j == 9 ? 7 : j == 11 ? 8 : -1;
}

// Blocks are dealt round-robin over the 8 XCDs; give each XCD a contiguous run of tiles so
// neighbouring tiles (which share gaussians) hit the same L2.  Returns >= num_tiles for the
// padding blocks of the last round.
__device__ __forceinline__ int block_to_tile(int block, int num_tiles) {
  const int per_xcd = (num_tiles + 7) >> 3;
  return (block & 7) * per_xcd + (block >> 3);
}
static inline int tile_grid(int num_tiles) { return ((num_tiles + 7) >> 3) * 8; }
// r04: the same map through an optional order table (tile_order_kernel, gs_render.hip): slot s of the table holds the tile
// that the block mapped to slot s should take -- every XCD's run of tiles re-ordered heaviest first, so that the
// launch's last round is made of its lightest tiles (entries >= num_tiles: padding)
__device__ __forceinline__ int ordered_tile(const int *__restrict__ order, int block, int num_tiles) {
  const int slot = block_to_tile(block, num_tiles);
  return order ? order[slot] : slot;
}
constexpr int kOrderMaxRun = 2048;  // tiles per XCD run the order kernel handles (16384 tiles: the counting-sort limit)

// r05 -- long tile lists, split for the BACKWARD.  One workgroup walks one tile's list batch after batch: a list of ten
// thousand entries is eighty batches in a row, and on a trained capture (lists of 1100 entries on average, 9800 at most)
// that ONE chain was the duration of the whole launch -- render_bwd 0.72 ms with the VALU 19 % busy.  The back-to-front
// recurrence only needs, at a boundary b of the list, the transmittance in front of entry b and the colour composited
// behind it; the forward passes both through anyway.  So for tiles whose list exceeds kSegSplitMin entries the forward
// stores, at every kSegEntries-th entry it reaches, a per-pixel checkpoint {T_b, C_b} (C_b: the colour accumulated from
// the entries in front of b, without the background), and the backward gives every segment [k kSegEntries,
// (k+1) kSegEntries) that some pixel of the tile reaches a workgroup of its own: a pixel that stops behind the segment
// starts it from T_b and the colour (image - C_b) / T_b behind it, a pixel that stops inside or in front of it from its
// final transmittance as before.  The gradient rows are added with atomics either way.
//   Checkpoint slots need no table: the boundary k >= 1 of the tile whose list starts at instance r has the slot
// r / kSegEntries + k (boundaries are kSegEntries instances apart inside a list and further apart across lists, so no two
// share a slot; the pool has room / kSegEntries + 2 slots).  Which segments exist is decided BEHIND the forward, from the
// tiles' largest stop indices (tile_segments_kernel): segment 0 of every tile is the block the tile has always had, the
// further segments are extra blocks in front of the main grid, listed in `extra`; granted[t] = how many tile t got
// (0: its block walks the whole list, also when the launch's room for extra blocks was used up).
#ifndef GS_SEG_ENTRIES
#define GS_SEG_ENTRIES 496
#endif
constexpr int kSegEntries = GS_SEG_ENTRIES;  // 2 forward batches = 4 backward batches
#ifndef GS_SEG_SPLIT_MIN
#define GS_SEG_SPLIT_MIN (3 * GS_SEG_ENTRIES)
#endif
constexpr int kSegSplitMin = GS_SEG_SPLIT_MIN;  // lists up to here stay whole
struct TileSegments {
  int *granted;            // [num_tiles]: further segments of the tile in `extra` (written behind the forward)
  int2 *extra;             // [extra_cap]: (tile, k) of the segments k >= 1
  int *extra_count;        // [1]
  float4 *chk;             // [slots][256]: one checkpoint per boundary, per pixel of the tile (thread order)
  const float *image;      // the forward's image (the backward's view of the colour behind a boundary)
  int extra_cap;           // room in `extra`; a multiple of 8 (the main blocks keep their XCDs)
  // Host-visible figures of this forward, each ONE 8-byte word {the forward's ticket << 32 | value} (gs_common.h: the
  // record's form), in the slot of the ticket's parity: the host takes a slot only when it carries the ticket of a forward
  // whose kernels it knows to have completed (r06, ADVICE r05: the plain ints of r05 were read while the kernels that
  // write them could still be running, and the forward's split decision followed host / GPU timing).
  unsigned long long *asked;  // [1]: the segments the tiles asked for (sizes a later launch's room)
  unsigned long long *stats;  // [2]: the largest stop index of any tile | the sum of the tiles' largest stop indices
  unsigned int tag;           // low half of the forward's ticket
};
__host__ __device__ inline unsigned long long tagged_figure(unsigned int tag, int value) {
  return ((unsigned long long)tag << 32) | (unsigned int)value;
}
__host__ __device__ inline int segment_slot(int list_start, int boundary) { return list_start / kSegEntries + boundary; }

// r05 -- the same long lists in the FORWARD.  The front-to-back recurrence is serial in the transmittance only: a
// segment k of a list can composite on its own once it knows P_k, the T in front of its first entry.  Every segment of a
// long list is a block of its own (fwd_segment_block), and the blocks are dispatched LAYER by layer: all segments 0,
// then all segments 1, ...
//  * Where a layer fills the chip, the layer behind it starts when it is through: a block finds the final T of the block
//    in front of it (F_(k-1) = P_k) already published, composites (phase C), and publishes its own.  A segment no pixel
//    reaches finds only zeros and leaves at once.  This is the unsplit forward's work, dealt in equal pieces.
//  * In the thin layers of the few longest lists the blocks run side by side.  A block that does not find F_(k-1) first
//    multiplies up the transmittance t_k of ITS entries per pixel (phase A: alpha evaluation only, no colour, no stop
//    logic: ~70 % of a compositing pass) and publishes it, then collects P_k = F_q t_(q+1) ... t_(k-1) from the nearest
//    finished block q in front of it and the products behind that, and composites.
// Published values are 8-byte {launch epoch, value} granules per pixel, written and polled with agent-scope atomics: the
// data is its own flag.  Phase C runs with T = P_k * t, t the running product inside the segment -- bit for bit the
// sequence phase A multiplied up -- so F_k = P_k t_k = P_(k+1) exactly, whichever way a block obtained it: a pixel that
// has not stopped in segment k (every T >= 1e-4) is live in segment k + 1, one that has stopped (T < 1e-4 at some entry)
// is dead there, whatever the rounding -- no pixel is lost or composited twice.  Segment 0 (P = 1) is the unsplit
// arithmetic.  A block only ever waits for blocks with smaller indices (workgroups are dispatched in index order) and
// not for ever: when its poll budget is used up it multiplies the product up itself from entry 0 (same values), so
// every wave reaches its end whatever the dispatch order.  A one-block-per-tile kernel behind the forward
// (fwd_segments_combine_kernel) adds the segments' colours in order, finds the segment in which the pixel stopped and
// writes image / T / stop index / the backward's checkpoints: sums in a fixed order, the same image in every run.
struct FwdSegments {
  int *rank;                     // [num_tiles]: the tile's rank among the long lists (most segments first), -1: one block
  int *base;                     // [129]: first block of layer k; segment (t, k) is block (= storage slot) base[k] + rank[t]
  int2 *blocks;                  // [cap]: (tile, k | thin-layer flag << 30) by block
  int *count;                    // [1]: blocks in use
  unsigned long long *granules;  // [2][cap][256]: {epoch, value} per block and pixel: t_k | the T behind segment k
  float4 *part;                  // [cap][256]: the segment's colour and the T behind it (its final T if the pixel stopped)
  int *stop;                     // [cap][256]: the stop index if the pixel stopped in the segment, -1 live, -2 dead
  int cap;                       // room for segment blocks; a multiple of 8 (the main blocks keep their XCDs)
  unsigned int epoch;            // of this launch; never 0
  unsigned long long *asked;     // [1], host-visible, {ticket << 32 | value} (see TileSegments): segment blocks the lists asked for
  unsigned int tag;              // low half of the forward's ticket
  int *fallbacks;                // [1], device: segment blocks whose polls ran out and which multiplied the product up themselves
  int poll_budget;               // polls (~1 us each) before a segment block multiplies the product up itself
  int thin_layer;                // layers of fewer blocks run their lists' segments side by side (phase A)
};
// r06: 128 polls ~ 0.13 ms, a dozen segments' worth of compositing -- a block whose predecessor is not even
// resident (the dispatch order it relies on is an assumption) recomputes after that instead of after 4 ms (r05: 4096),
// and every such block is counted (gsplat_context_get_counters out[9]).
constexpr int kFwdPollBudget = 128, kFwdThinLayerDefault = 512;  // (gsplat_context_set_segment_options changes them)

}  // namespace gs
