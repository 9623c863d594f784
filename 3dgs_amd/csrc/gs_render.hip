// gs_render.hip -- per-tile alpha compositing, forward and backward, for gfx950.
//
// Semantics: render_image / render_image_backward of the reference (cuda/render.cu:6-135,
// cuda/render_backward.cu:11-258) -- front-to-back blending of each 16x16 tile's depth-
// sorted list with the 1/255 alpha floor, the 0.99 cap and the T < 1e-4 stop; the backward
// walks the list back to front rebuilding T and the colour behind each splat.
//
// Design (not the reference's one-warp-per-tile, 8-pixels-per-lane layout):
//   * one 256-thread workgroup per tile = four wave64s, each owning one 8x8 pixel quadrant,
//     one pixel per lane;
//   * the tile's list is consumed in batches of 256 entries: every thread gathers ONE
//     gaussian (three 16-byte loads of a 48-byte record, or the reference's four arrays),
//     evaluates sigmoid(opacity) and a conservative footprint box once, and parks it in LDS;
//   * each wave ballots the batch's quadrant-hit bits into a 64-bit scalar mask and only
//     visits gaussians whose footprint can reach its quadrant (skipping is exact: a skipped
//     gaussian has alpha < 1/255 on every pixel of the quadrant); visited gaussians are read
//     from LDS at a wave-uniform address (broadcast);
//   * forward: a wave stops as soon as all 64 pixels are saturated, the workgroup when all
//     four waves have;
//   * backward: nine partial sums per (wave, gaussian) are reduced across the wave on DPP,
//     merged across the four waves in LDS, and flushed once per batch to HBM as whole
//     64-byte gradient rows (or into the reference's four gradient arrays).
#include "gs_common.h"
#include "gs_render.h"

namespace gs {

constexpr int kBatch = 256;

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

// ------------------------------------------------------------------------------ forward
template <bool kPacked>
__global__ __launch_bounds__(256) void render_fwd_kernel(const float4 *__restrict__ recs, RawSplats raw,
                                                         const int *__restrict__ sorted,
                                                         const int *__restrict__ ranges, int width, int height,
                                                         int ntx, int num_tiles, float bg,
                                                         int *__restrict__ n_out, float *__restrict__ T_out,
                                                         float *__restrict__ image) {
  __shared__ float4 s_r0[kBatch], s_r1[kBatch], s_r2[kBatch];
  __shared__ unsigned char s_list[4 * 64];
  const int tile = block_to_tile(blockIdx.x, num_tiles);
  if (tile >= num_tiles) return;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int tile_x = tile % ntx, tile_y = tile / ntx;
  const int px = tile_x * 16 + (wave & 1) * 8 + (lane & 7);
  const int py = tile_y * 16 + (wave >> 1) * 8 + (lane >> 3);
  const bool inside = px < width && py < height;
  const float fpx = (float)px, fpy = (float)py;
  const float tx0 = (float)(tile_x * 16), ty0 = (float)(tile_y * 16);

  const int start = ranges[tile], total = ranges[tile + 1] - start;
  // A saturated pixel keeps T = 0 in the running transmittance, so every later splat blends with weight 0 and the
  // common path needs no per-pixel "done" masking; its real final transmittance and stop index live in T_fin / n.
  float T = inside ? 1.0f : 0.0f, T_fin = -1.0f, ar = 0.0f, ag = 0.0f, ab = 0.0f;
  int n = total;
  unsigned long long satmask = __ballot(!inside);  // lanes whose pixel is saturated or outside the image
  int live = satmask != ~0ull ? 1 : 0;             // wave-uniform: some pixel of the quadrant is still unsaturated

  for (int base = 0; base < total; base += kBatch) {
    const int count = min(kBatch, total - base);
    __syncthreads();
    if (tid < count) {
      const int g = sorted[start + base + tid];
      SplatRec s = load_record<kPacked>(g, recs, raw);
      s.r2.w = __uint_as_float(quadrant_hits(s, tx0, ty0));
      s_r0[tid] = s.r0; s_r1[tid] = s.r1; s_r2[tid] = s.r2;
    }
    __syncthreads();
    for (int sb = 0; sb < count && live > 0; sb += 64) {
      const int slot_l = sb + lane;
      const unsigned int bits = slot_l < count ? __float_as_uint(s_r2[slot_l].w) : 0u;
      const bool hit = (bits >> wave) & 1u;
      const unsigned long long m = __ballot(hit);
      const int cnt = __popcll(m);
      if (cnt == 0) continue;
      // The scalar unit, not the VALU, limits this loop (profiles/: ~17 SALU per visit when the hit mask is
      // popped bit by bit), so the hit slots are compacted once per 64 entries with lane-parallel work: lane t ends
      // up holding the t-th hit slot and each visit fetches it with one v_readlane.
      const int pos = __builtin_amdgcn_mbcnt_hi((unsigned int)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)m, 0u));
      if (hit) s_list[wave * 64 + pos] = (unsigned char)lane;
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      const int my_hit = sb + (int)s_list[wave * 64 + lane];
      for (int t = 0; t < cnt; t += 2) {
        // two visited gaussians per trip: both records are fetched from LDS and both exponentials evaluated before
        // the (sequential) blending
        const int slot0 = __builtin_amdgcn_readlane(my_hit, t);
        const bool two = t + 1 < cnt;
        const int slot1 = __builtin_amdgcn_readlane(my_hit, two ? t + 1 : t);
        const float4 a0 = s_r0[slot0], c0 = s_r2[slot0];
        const float2 b0 = *reinterpret_cast<const float2 *>(&s_r1[slot0]);
        const float4 a1 = s_r0[slot1], c1 = s_r2[slot1];
        const float2 b1 = *reinterpret_cast<const float2 *>(&s_r1[slot1]);
        asm volatile("" ::"v"(c0.w), "v"(c1.w));  // keep 16-byte reads (ds_read_b96 costs twice the LDS cycles)
        const float p0 = fminf(0.0f, gauss_power(a0.z, a0.w, b0.x, a0.x - fpx, a0.y - fpy));
        const float p1 = fminf(0.0f, gauss_power(a1.z, a1.w, b1.x, a1.x - fpx, a1.y - fpy));
        float al0 = fminf(kAlphaMax, b0.y * __expf(p0));
        float al1 = fminf(kAlphaMax, b1.y * __expf(p1));
        al0 = al0 > kAlphaMin ? al0 : 0.0f;
        al1 = (al1 > kAlphaMin && two) ? al1 : 0.0f;
        // Invariant: T is either 0 (saturated or outside the image) or >= 1e-4, so "T * (1 - alpha) < 1e-4" alone
        // decides the next T; the compare's lane mask doubles as the saturation bookkeeping (one scalar compare per
        // visit, the per-pixel records are only touched in the rare branch).
        const float w0 = al0 * T;
        const float tT0 = T * (1.0f - al0);
        ar = __builtin_fmaf(c0.x, w0, ar);
        ag = __builtin_fmaf(c0.y, w0, ag);
        ab = __builtin_fmaf(c0.z, w0, ab);
        const unsigned long long s0 = __ballot(tT0 < kTMin);  // this splat was still accumulated (render.cu:76-87)
        T = tT0 < kTMin ? 0.0f : tT0;
        const float w1 = al1 * T;
        const float tT1 = T * (1.0f - al1);
        ar = __builtin_fmaf(c1.x, w1, ar);
        ag = __builtin_fmaf(c1.y, w1, ag);
        ab = __builtin_fmaf(c1.z, w1, ab);
        const unsigned long long s1 = __ballot(tT1 < kTMin);
        T = tT1 < kTMin ? 0.0f : tT1;
        if (s1 != satmask) {  // rare: some pixel saturated in this trip
          const unsigned long long bit = 1ull << lane;
          if ((s0 & ~satmask) & bit) { T_fin = tT0; n = base + slot0 + 1; }
          if ((s1 & ~s0) & bit) { T_fin = tT1; n = base + slot1 + 1; }
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
  if (inside) {
    const int pid = py * width + px;
    const float Tout = T_fin >= 0.0f ? T_fin : T;  // T_fin is set by the splat that saturated the pixel
    n_out[pid] = n;
    T_out[pid] = Tout;
    image[3 * pid + 0] = ar + Tout * bg;
    image[3 * pid + 1] = ag + Tout * bg;
    image[3 * pid + 2] = ab + Tout * bg;
  }
}

// ------------------------------------------------------------------------------ backward
struct GradOut {         // either whole rows ...
  float *rows;           // [M,16]: rgb3 opacity1 conic3 uv2, 7 pad
  // ... or the reference operator's four arrays
  float *rgb, *opacity, *uv, *conic;
};

template <bool kPacked, bool kRows>
__global__ __launch_bounds__(256) void render_bwd_kernel(const float4 *__restrict__ recs, RawSplats raw,
                                                         const int *__restrict__ sorted,
                                                         const int *__restrict__ ranges,
                                                         const int *__restrict__ n_px,
                                                         const float *__restrict__ T_px,
                                                         const float *__restrict__ grad_image, int width, int height,
                                                         int ntx, int num_tiles, float bg, GradOut out) {
  __shared__ float4 s_r0[kBatch], s_r1[kBatch], s_r2[kBatch];
  __shared__ float s_acc[kBatch * 9];
  __shared__ int s_id[kBatch];
  __shared__ int s_top;
  const int tile = block_to_tile(blockIdx.x, num_tiles);
  if (tile >= num_tiles) return;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int tile_x = tile % ntx, tile_y = tile / ntx;
  const int px = tile_x * 16 + (wave & 1) * 8 + (lane & 7);
  const int py = tile_y * 16 + (wave >> 1) * 8 + (lane >> 3);
  const bool inside = px < width && py < height;
  const float fpx = (float)px, fpy = (float)py;
  const float tx0 = (float)(tile_x * 16), ty0 = (float)(tile_y * 16);
  const int start = ranges[tile];

  int n = 0;
  float Tf = 0.0f, g0 = 0.0f, g1 = 0.0f, g2 = 0.0f;
  if (inside) {
    const int pid = py * width + px;
    n = n_px[pid];
    Tf = T_px[pid];
    g0 = grad_image[3 * pid]; g1 = grad_image[3 * pid + 1]; g2 = grad_image[3 * pid + 2];
  }
  float T = Tf, c0 = 0.0f, c1 = 0.0f, c2 = 0.0f;  // running transmittance, colour behind the current splat
  const float tfb = Tf * (bg * g0 + bg * g1 + bg * g2);  // T_final * (background . grad)
  const int wave_top = wave_max_int(n);
  if (tid == 0) s_top = 0;
  __syncthreads();
  if (lane == 0) atomicMax(&s_top, wave_top);
  __syncthreads();
  const int top = s_top;  // cuda/render_backward.cu:64,74: start at (max n over the tile) - 1
  if (top <= 0) return;
  // where this lane's share of the nine wave totals goes (see wave_sum9)
  const bool row_leader = (lane & 15) == 0;
  const int acc_q0 = sum9_index_q0(lane) * kBatch, acc_q1 = sum9_index_q1(lane) * kBatch;

  for (int base = ((top - 1) / kBatch) * kBatch; base >= 0; base -= kBatch) {
    const int count = min(kBatch, top - base);
    __syncthreads();
    if (tid < count) {
      const int g = sorted[start + base + tid];
      SplatRec s = load_record<kPacked>(g, recs, raw);
      s.r2.w = __uint_as_float(quadrant_hits(s, tx0, ty0));
      s_r0[tid] = s.r0; s_r1[tid] = s.r1; s_r2[tid] = s.r2;
      s_id[tid] = g;
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) s_acc[k * kBatch + tid] = 0.0f;
    __syncthreads();
    for (int sb = ((count - 1) >> 6) << 6; sb >= 0; sb -= 64) {
      if (base + sb >= wave_top) continue;
      const int slot_l = sb + lane;
      const unsigned int bits =
          (slot_l < count && base + slot_l < wave_top) ? __float_as_uint(s_r2[slot_l].w) : 0u;
      unsigned long long m = __ballot((bits >> wave) & 1u);
      while (m != 0ull) {
        const int j = 63 - __builtin_clzll(m);
        m &= ~(1ull << j);
        const int slot = sb + j;
        const float4 a = s_r0[slot], b = s_r1[slot], c = s_r2[slot];
        const float dx = a.x - fpx, dy = a.y - fpy;
        const float power = fminf(0.0f, gauss_power(a.z, a.w, b.x, dx, dy));
        float gg = __expf(power);
        const float opa = b.y;
        float alpha = fminf(kAlphaMax, opa * gg);
        // n is 0 for pixels outside the image, so "inside" needs no separate test
        const bool valid = (alpha >= kAlphaMin) && (base + slot < n);
        if (__ballot(valid) == 0ull) continue;
        // one select, two multiplies: selects on SGPR masks are ~8x the issue cost of a multiply on this chip
        const float vf = valid ? 1.0f : 0.0f;
        alpha *= vf;
        gg *= vf;
        const float inv = __builtin_amdgcn_rcpf(1.0f - alpha);
        T *= inv;                                           // transmittance in front of this splat
        const float aT = alpha * T;
        const float v0 = aT * g0, v1 = aT * g1, v2 = aT * g2;  // d/d rgb
        const float d0 = c.x - c0, d1 = c.y - c1, d2 = c.z - c2;
        float ga = __builtin_fmaf(d0, g0, __builtin_fmaf(d1, g1, d2 * g2));
        ga = __builtin_fmaf(ga, T, -(tfb * inv));           // d/d alpha (cuda/render_backward.cu:139-151)
        c0 = __builtin_fmaf(alpha, d0, c0);                 // colour behind the next (nearer) splat
        c1 = __builtin_fmaf(alpha, d1, c1);
        c2 = __builtin_fmaf(alpha, d2, c2);
        const float gp = gg * (ga * opa);                   // d/d power
        // cuda/render_backward.cu:170 gates on any(d/d logit != 0), d/d logit = gp * (1 - opa)
        if (opa == 1.0f || __ballot(gp != 0.0f) == 0ull) continue;
        const float gpx = gp * dx, gpy = gp * dy;
        // nine raw sums; signs, the -1/2 factors, (1 - opa) and the 0.5*W / 0.5*H are applied once per gaussian
        // at flush time:  S0 = sum gp, Sx, Sy, Sxx, Sxy, Syy
        const Sum9 r = wave_sum9(v0, v1, v2, gp, gpx, gpy, gpx * dx, gpx * dy, gpy * dy);
        if (row_leader) {
          atomicAdd(&s_acc[acc_q0 + slot], r.q0);
          atomicAdd(&s_acc[acc_q1 + slot], r.q1);
          if (lane == 0) atomicAdd(&s_acc[8 * kBatch + slot], r.q2);
        }
      }
    }
    __syncthreads();
    // flush: 16 lanes per gaussian -> each wave instruction touches four whole 64-byte rows.
    // s_acc rows: 0..2 rgb, 3 S0, 4 Sx, 5 Sy, 6 Sxx, 7 Sxy, 8 Syy
    const int k = tid & 15;
    if (k < 9) {
#pragma unroll 4
      for (int r = 0; r < 16; ++r) {
        const int slot = r * 16 + (tid >> 4);
        if (slot >= count) continue;
        float val;
        if (k < 3) {
          val = s_acc[k * kBatch + slot];
        } else if (k == 3) {
          val = s_acc[3 * kBatch + slot] * (1.0f - s_r1[slot].y);              // d/d logit (render_backward.cu:154)
        } else if (k == 4) {
          val = -0.5f * s_acc[6 * kBatch + slot];                               // conic00
        } else if (k == 5) {
          val = -s_acc[7 * kBatch + slot];                                      // conic01
        } else if (k == 6) {
          val = -0.5f * s_acc[8 * kBatch + slot];                               // conic11
        } else {
          const float sx = s_acc[4 * kBatch + slot], sy = s_acc[5 * kBatch + slot];
          const float4 a = s_r0[slot];
          val = (k == 7) ? -(a.z * sx + a.w * sy) * (0.5f * (float)width)       // u (render_backward.cu:180-186)
                         : -(s_r1[slot].x * sy + a.w * sx) * (0.5f * (float)height);  // v (:181-187)
        }
        if (val == 0.0f) continue;
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
}

// host-side launchers shared with gs_fused.hip ------------------------------------------
int launch_render_fwd(const float4 *recs, const RawSplats *raw, const int *sorted, const int *ranges, int width,
                      int height, float bg, int *n_out, float *T_out, float *image, hipStream_t st) {
  const int ntx = (width + 15) / 16, nty = (height + 15) / 16, num_tiles = ntx * nty;
  const dim3 grid(tile_grid(num_tiles)), block(256);
  RawSplats none = {nullptr, nullptr, nullptr, nullptr};
  if (recs)
    render_fwd_kernel<true><<<grid, block, 0, st>>>(recs, none, sorted, ranges, width, height, ntx, num_tiles, bg, n_out, T_out, image);
  else
    render_fwd_kernel<false><<<grid, block, 0, st>>>(nullptr, *raw, sorted, ranges, width, height, ntx, num_tiles, bg, n_out, T_out, image);
  GS_LAUNCH_CHECK();
  return GSPLAT_OK;
}

int launch_render_bwd(const float4 *recs, const RawSplats *raw, const int *sorted, const int *ranges, const int *n_px,
                      const float *T_px, const float *grad_image, int width, int height, float bg, float *rows,
                      float *g_rgb, float *g_opacity, float *g_uv, float *g_conic, hipStream_t st) {
  const int ntx = (width + 15) / 16, nty = (height + 15) / 16, num_tiles = ntx * nty;
  const dim3 grid(tile_grid(num_tiles)), block(256);
  RawSplats none = {nullptr, nullptr, nullptr, nullptr};
  GradOut out = {rows, g_rgb, g_opacity, g_uv, g_conic};
  if (recs && rows)
    render_bwd_kernel<true, true><<<grid, block, 0, st>>>(recs, none, sorted, ranges, n_px, T_px, grad_image, width, height, ntx, num_tiles, bg, out);
  else if (recs)
    render_bwd_kernel<true, false><<<grid, block, 0, st>>>(recs, none, sorted, ranges, n_px, T_px, grad_image, width, height, ntx, num_tiles, bg, out);
  else
    render_bwd_kernel<false, false><<<grid, block, 0, st>>>(nullptr, *raw, sorted, ranges, n_px, T_px, grad_image, width, height, ntx, num_tiles, bg, out);
  GS_LAUNCH_CHECK();
  return GSPLAT_OK;
}

}  // namespace gs

extern "C" {

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
                               background_opacity, splats_per_pixel, weight_per_pixel, image, (hipStream_t)stream);
}

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
  return gs::launch_render_bwd(nullptr, &raw, sorted_splats, splat_range_by_tile, num_splats_per_pixel,
                               final_weight_per_pixel, grad_image, image_width, image_height, background_opacity,
                               nullptr, grad_rgb, grad_opacity, grad_uv, grad_conic, (hipStream_t)stream);
}

}  // extern "C"
