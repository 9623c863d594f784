// gs_density.hip -- operators of the "next" row f4 that the reference declares in its public headers and tests:
// clone_gaussians / split_gaussians (include/gsplat_cuda/adaptive_density.cuh, cuda/adaptive_density.cu:12-206) and
// compute_morton_codes (include/gsplat_cuda/cuda_forward.cuh:174-188, cuda/culling.cu:14-63, 345-359), plus the
// data-parallel pieces of the policy that drives them (the masks of TrainerImpl::adaptive_density_step, add_sh_band's
// re-layout, sort_gaussians' gather); the schedule itself lives in 3dgs_amd/trainer.py.
//
// split_gaussians draws its two new centres from N(xyz, R diag(exp(scale))^2 R^T).  The reference seeds cuRAND from
// time(NULL) per call, so its positions are not reproducible; here the normals come from a counter-based generator
// (splitmix64 of (seed, gaussian, sample, axis) + Box-Muller), so a seed reproduces a split bit for bit.
#include "gs_common.h"

namespace {

constexpr int kBlock = 256;

// cuda/culling.cu:19-37.  NOTE: the masks of the third and fourth step are the reference's, not the textbook
// 3-way bit-spread constants; bit-exact parity with its Morton order (and its test, tests/cuda_forward_test.cpp:964-972)
// requires exactly these.
__device__ __forceinline__ unsigned long long ref_spread_bits(unsigned long long n) {
  n &= 0x1FFFFFull;
  n = (n | (n << 32)) & 0x1F000000FFFFull;
  n = (n | (n << 16)) & 0x1F0000FF0000FFull;
  n = (n | (n << 8)) & 0x100F807C0F807C0Full;
  n = (n | (n << 4)) & 0x1084210842108421ull;
  n = (n | (n << 2)) & 0x1249249249249249ull;
  return n;
}

// (uint64_t)float as the reference's device code does it: truncation, negative and NaN -> 0, huge -> saturate
__device__ __forceinline__ unsigned long long to_u64(float v) {
  if (!(v > 0.0f)) return 0ull;
  if (v >= 18446744073709551616.0f) return ~0ull;
  return (unsigned long long)v;
}

__global__ __launch_bounds__(kBlock) void morton_codes_kernel(int N, const float *__restrict__ xyz, float x_max,
                                                              float y_max, float z_max, float x_min, float y_min,
                                                              float z_min, unsigned long long *__restrict__ codes) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= N) return;
  const float kMax = 2097151.0f;  // (1 << 21) - 1 converted to float, as uint32 / float does
  const float x = xyz[3 * (size_t)i], y = xyz[3 * (size_t)i + 1], z = xyz[3 * (size_t)i + 2];
  const unsigned long long xq = to_u64((x - x_min) * (kMax / (x_max - x_min)));
  const unsigned long long yq = to_u64((y - y_min) * (kMax / (y_max - y_min)));
  const unsigned long long zq = to_u64((z - z_min) * (kMax / (z_max - z_min)));
  codes[i] = (ref_spread_bits(zq) << 2) | (ref_spread_bits(yq) << 1) | ref_spread_bits(xq);
}

struct Attr {  // GaussianParameters arrays
  const float *xyz, *rgb, *op, *scale, *quat, *sh;
};
struct AttrOut {
  float *xyz, *rgb, *op, *scale, *quat, *sh;
};

__device__ __forceinline__ void copy_rest(const Attr &in, const AttrOut &out, size_t src, size_t dst, int num_sh_coef) {
  for (int k = 0; k < 3; ++k) out.rgb[3 * dst + k] = in.rgb[3 * src + k];
  out.op[dst] = in.op[src];
  for (int k = 0; k < 4; ++k) out.quat[4 * dst + k] = in.quat[4 * src + k];
  const size_t w = (size_t)num_sh_coef * 3;
  for (size_t k = 0; k < w; ++k) out.sh[dst * w + k] = in.sh[src * w + k];
}

__global__ __launch_bounds__(kBlock) void clone_kernel(int N, int num_sh_coef, const unsigned char *__restrict__ mask,
                                                       const int *__restrict__ write_ids, Attr in, AttrOut out) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= N || !mask[i]) return;
  const size_t dst = (size_t)write_ids[i];
  for (int k = 0; k < 3; ++k) {
    out.xyz[3 * dst + k] = in.xyz[3 * (size_t)i + k];
    out.scale[3 * dst + k] = in.scale[3 * (size_t)i + k];
  }
  copy_rest(in, out, (size_t)i, dst, num_sh_coef);
}

__device__ __forceinline__ unsigned long long splitmix64(unsigned long long x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
// standard normal number `counter` of stream `seed`
__device__ __forceinline__ float normal_sample(unsigned long long seed, unsigned long long counter) {
  const unsigned long long bits = splitmix64(splitmix64(seed) ^ (counter * 0xD1342543DE82EF95ull + 1ull));
  const float u1 = (float)((bits >> 40) + 1ull) * (1.0f / 16777216.0f);          // (0, 1]
  const float u2 = (float)((bits >> 16) & 0xFFFFFFull) * (1.0f / 16777216.0f);   // [0, 1)
  return sqrtf(-2.0f * logf(u1)) * cosf(6.28318530717958647692f * u2);
}

__global__ __launch_bounds__(kBlock) void split_kernel(int N, float scale_factor, int num_sh_coef,
                                                       const unsigned char *__restrict__ mask,
                                                       const int *__restrict__ write_ids, Attr in, AttrOut out,
                                                       unsigned long long seed) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= N || !mask[i]) return;
  const size_t base = (size_t)write_ids[i] * 2;
  const float ex = expf(in.scale[3 * (size_t)i]), ey = expf(in.scale[3 * (size_t)i + 1]),
              ez = expf(in.scale[3 * (size_t)i + 2]);
  const float q0 = in.quat[4 * (size_t)i], q1 = in.quat[4 * (size_t)i + 1], q2 = in.quat[4 * (size_t)i + 2],
              q3 = in.quat[4 * (size_t)i + 3];
  const float inv = rsqrtf(q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3);
  const float w = q0 * inv, x = q1 * inv, y = q2 * inv, z = q3 * inv;
  const float r00 = 1.0f - 2.0f * (y * y + z * z), r01 = 2.0f * (x * y - w * z), r02 = 2.0f * (x * z + w * y);
  const float r10 = 2.0f * (x * y + w * z), r11 = 1.0f - 2.0f * (x * x + z * z), r12 = 2.0f * (y * z - w * x);
  const float r20 = 2.0f * (x * z - w * y), r21 = 2.0f * (y * z + w * x), r22 = 1.0f - 2.0f * (x * x + y * y);
  for (int j = 0; j < 2; ++j) {
    const unsigned long long c = ((unsigned long long)i * 2ull + (unsigned long long)j) * 3ull;
    const float vx = normal_sample(seed, c) * ex, vy = normal_sample(seed, c + 1) * ey,
                vz = normal_sample(seed, c + 2) * ez;
    const size_t dst = base + j;
    out.xyz[3 * dst] = in.xyz[3 * (size_t)i] + (vx * r00 + vy * r01 + vz * r02);
    out.xyz[3 * dst + 1] = in.xyz[3 * (size_t)i + 1] + (vx * r10 + vy * r11 + vz * r12);
    out.xyz[3 * dst + 2] = in.xyz[3 * (size_t)i + 2] + (vx * r20 + vy * r21 + vz * r22);
    out.scale[3 * dst] = logf(ex / scale_factor);
    out.scale[3 * dst + 1] = logf(ey / scale_factor);
    out.scale[3 * dst + 2] = logf(ez / scale_factor);
    copy_rest(in, out, (size_t)i, dst, num_sh_coef);
  }
}

// ---- policy half of adaptive density control (TrainerImpl::adaptive_density_step, cuda/trainer.cu:416-575):
// one pass computes the average screen-space gradient, the largest extent, and the prune / clone / split / keep
// masks of every gaussian (the reference runs six thrust transforms and three counts).
__global__ __launch_bounds__(kBlock) void density_masks_kernel(int N, const float *__restrict__ opacity,
                                                               const float *__restrict__ scale,
                                                               const float *__restrict__ uv_grad_accum,
                                                               const int *__restrict__ grad_accum_dur,
                                                               float op_threshold, float max_scale,
                                                               float grad_threshold, float clone_scale_threshold,
                                                               unsigned char *__restrict__ prune,
                                                               unsigned char *__restrict__ clone,
                                                               unsigned char *__restrict__ split,
                                                               unsigned char *__restrict__ keep,
                                                               int *__restrict__ counts) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  bool p = false, c = false, s = false;
  if (i < N) {
    const int dur = grad_accum_dur[i];
    const float avg = dur == 0 ? 0.0f : uv_grad_accum[i] / (float)dur;               // ComputeAvgGrad
    const float max_s = fmaxf(expf(scale[3 * (size_t)i]), fmaxf(expf(scale[3 * (size_t)i + 1]),
                                                                 expf(scale[3 * (size_t)i + 2])));  // ComputeScaleMax
    // IdentifyPrune (cuda/trainer.cu:438-468): low opacity always goes; an oversized gaussian survives only if it is
    // about to be split or cloned (the 1.6 is the reference's literal, not config.split_scale_factor)
    if (opacity[i] < op_threshold) p = true;
    else if (avg > grad_threshold && (max_s / 1.6f) <= max_scale) p = false;
    else p = max_s > max_scale;
    c = !p && avg > grad_threshold && max_s <= clone_scale_threshold;                 // IdentifyClone
    s = !p && avg > grad_threshold && max_s > clone_scale_threshold;                  // IdentifySplit
    prune[i] = p; clone[i] = c; split[i] = s;
    keep[i] = !(p || s);                                                              // CombineMasks
  }
  const unsigned long long bp = __ballot(p), bc = __ballot(c), bs = __ballot(s);
  if ((threadIdx.x & 63) == 0) {
    if (bp) atomicAdd(&counts[0], __popcll(bp));
    if (bc) atomicAdd(&counts[1], __popcll(bc));
    if (bs) atomicAdd(&counts[2], __popcll(bs));
  }
}

// add_sh_band (cuda/trainer.cu:363-413): coefficient k of gaussian g moves from g*old+k to g*new+k, new ones are 0
__global__ __launch_bounds__(kBlock) void expand_sh_kernel(long long total_new, int old_w, int new_w,
                                                           const float *__restrict__ in, float *__restrict__ out) {
  const long long e = (long long)blockIdx.x * kBlock + threadIdx.x;
  if (e >= total_new) return;
  const long long g = e / new_w;
  const int k = (int)(e - g * new_w);
  out[e] = k < old_w ? in[g * old_w + k] : 0.0f;
}

// sort_gaussians' gather (cuda/trainer.cu:793-851): out row i = in row order[i]
__global__ __launch_bounds__(kBlock) void gather_rows_kernel(long long total, int stride, const int *__restrict__ order,
                                                             const float *__restrict__ in, float *__restrict__ out) {
  const long long e = (long long)blockIdx.x * kBlock + threadIdx.x;
  if (e >= total) return;
  const long long r = e / stride;
  out[e] = in[(long long)order[r] * stride + (e - r * stride)];
}

int check_attr(int N, int num_sh_coef, const unsigned char *mask, const int *write_ids, const Attr &in,
               const AttrOut &out, const char *fn) {
  if (N < 0 || num_sh_coef < 0) { gs::set_error("%s: invalid argument: negative size", fn); return GSPLAT_ERR_INVALID_ARG; }
  if (N == 0) return GSPLAT_OK;
  int st;
  // the reference checks exactly these two (cuda/adaptive_density.cu:171-172); the attribute arrays are checked too
  if ((st = gs::check_device_ptr(mask, "mask", fn)) || (st = gs::check_device_ptr(write_ids, "write_ids", fn))) return st;
  const void *ptrs[] = {in.xyz, in.rgb, in.op, in.scale, in.quat, out.xyz, out.rgb, out.op, out.scale, out.quat};
  const char *names[] = {"xyz_in", "rgb_in", "op_in", "scale_in", "quat_in", "xyz_out", "rgb_out", "op_out", "scale_out", "quat_out"};
  for (int k = 0; k < 10; ++k)
    if ((st = gs::check_device_ptr(ptrs[k], names[k], fn))) return st;
  if (num_sh_coef > 0 && ((st = gs::check_device_ptr(in.sh, "sh_in", fn)) || (st = gs::check_device_ptr(out.sh, "sh_out", fn))))
    return st;
  return GSPLAT_OK;
}

}  // namespace

extern "C" {

int gsplat_compute_morton_codes(int N, const float *xyz, float x_max, float y_max, float z_max, float x_min,
                                float y_min, float z_min, unsigned long long *codes, void *stream) {
  GS_REQUIRE(N >= 0, "negative size");
  if (N == 0) return GSPLAT_OK;
  GS_REQUIRE_DEV(xyz); GS_REQUIRE_DEV(codes);
  morton_codes_kernel<<<gs::div_up(N, kBlock), kBlock, 0, (hipStream_t)stream>>>(N, xyz, x_max, y_max, z_max, x_min,
                                                                               y_min, z_min, codes);
  GS_LAUNCH_CHECK();
  return GSPLAT_OK;
}

int gsplat_clone_gaussians(int N, int num_sh_coef, const unsigned char *mask, const int *write_ids, const float *xyz_in,
                           const float *rgb_in, const float *op_in, const float *scale_in, const float *quat_in,
                           const float *sh_in, float *xyz_out, float *rgb_out, float *op_out, float *scale_out,
                           float *quat_out, float *sh_out, void *stream) {
  const Attr in = {xyz_in, rgb_in, op_in, scale_in, quat_in, sh_in};
  const AttrOut out = {xyz_out, rgb_out, op_out, scale_out, quat_out, sh_out};
  const int rc = check_attr(N, num_sh_coef, mask, write_ids, in, out, __func__);
  if (rc || N == 0) return rc;
  clone_kernel<<<gs::div_up(N, kBlock), kBlock, 0, (hipStream_t)stream>>>(N, num_sh_coef, mask, write_ids, in, out);
  GS_LAUNCH_CHECK();
  return GSPLAT_OK;
}

int gsplat_split_gaussians(int N, float scale_factor, int num_sh_coef, const unsigned char *mask, const int *write_ids,
                           const float *xyz_in, const float *rgb_in, const float *op_in, const float *scale_in,
                           const float *quat_in, const float *sh_in, float *xyz_out, float *rgb_out, float *op_out,
                           float *scale_out, float *quat_out, float *sh_out, unsigned long long seed, void *stream) {
  const Attr in = {xyz_in, rgb_in, op_in, scale_in, quat_in, sh_in};
  const AttrOut out = {xyz_out, rgb_out, op_out, scale_out, quat_out, sh_out};
  const int rc = check_attr(N, num_sh_coef, mask, write_ids, in, out, __func__);
  if (rc || N == 0) return rc;
  GS_REQUIRE(scale_factor > 0.0f, "scale_factor must be positive");
  split_kernel<<<gs::div_up(N, kBlock), kBlock, 0, (hipStream_t)stream>>>(N, scale_factor, num_sh_coef, mask, write_ids,
                                                                        in, out, seed);
  GS_LAUNCH_CHECK();
  return GSPLAT_OK;
}

int gsplat_density_masks(int N, const float *opacity, const float *scale, const float *uv_grad_accum,
                         const int *grad_accum_dur, float op_threshold, float max_scale, float uv_grad_threshold,
                         float clone_scale_threshold, unsigned char *prune_mask, unsigned char *clone_mask,
                         unsigned char *split_mask, unsigned char *keep_mask, int *counts, void *stream) {
  GS_REQUIRE(N >= 0, "negative size");
  GS_REQUIRE_DEV(counts);
  hipStream_t st = (hipStream_t)stream;
  GS_HIP(hipMemsetAsync(counts, 0, 3 * sizeof(int), st));
  if (N == 0) return GSPLAT_OK;
  GS_REQUIRE_DEV(opacity); GS_REQUIRE_DEV(scale); GS_REQUIRE_DEV(uv_grad_accum); GS_REQUIRE_DEV(grad_accum_dur);
  GS_REQUIRE_DEV(prune_mask); GS_REQUIRE_DEV(clone_mask); GS_REQUIRE_DEV(split_mask); GS_REQUIRE_DEV(keep_mask);
  density_masks_kernel<<<gs::div_up(N, kBlock), kBlock, 0, st>>>(N, opacity, scale, uv_grad_accum, grad_accum_dur,
                                                               op_threshold, max_scale, uv_grad_threshold,
                                                               clone_scale_threshold, prune_mask, clone_mask, split_mask,
                                                               keep_mask, counts);
  GS_LAUNCH_CHECK();
  return GSPLAT_OK;
}

int gsplat_expand_sh(int N, int l_max_old, const float *sh_in, float *sh_out, void *stream) {
  GS_REQUIRE(N >= 0 && l_max_old >= 0 && l_max_old <= 2, "expand_sh grows SH degree 0..2 by one band");
  if (N == 0) return GSPLAT_OK;
  GS_REQUIRE_DEV(sh_out);
  const int old_w = ((l_max_old + 1) * (l_max_old + 1) - 1) * 3, new_w = ((l_max_old + 2) * (l_max_old + 2) - 1) * 3;
  if (old_w > 0) GS_REQUIRE_DEV(sh_in);
  GS_REQUIRE(sh_in != sh_out, "expand_sh is not in place");
  const long long total = (long long)N * new_w;
  expand_sh_kernel<<<gs::div_up(total, kBlock), kBlock, 0, (hipStream_t)stream>>>(total, old_w, new_w, sh_in, sh_out);
  GS_LAUNCH_CHECK();
  return GSPLAT_OK;
}

int gsplat_gather_rows(int N, int stride, const int *order, const float *in, float *out, void *stream) {
  GS_REQUIRE(N >= 0 && stride >= 0, "negative size");
  if (N == 0 || stride == 0) return GSPLAT_OK;
  GS_REQUIRE_DEV(order); GS_REQUIRE_DEV(in); GS_REQUIRE_DEV(out);
  GS_REQUIRE(in != out, "gather_rows is not in place");
  const long long total = (long long)N * stride;
  gather_rows_kernel<<<gs::div_up(total, kBlock), kBlock, 0, (hipStream_t)stream>>>(total, stride, order, in, out);
  GS_LAUNCH_CHECK();
  return GSPLAT_OK;
}

}  // extern "C"
