// gs_pergaussian.hip -- the stand-alone per-gaussian operators of the drop-in surface
// (one thread per gaussian, wave64, 256-thread workgroups).  The math lives in gs_math.h
// and is shared bit-for-bit with the fused preprocess kernels in gs_fused.hip.
#include <cstring>
#include <rocprim/rocprim.hpp>

#include "gs_common.h"
#include "gs_math.h"

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

template <int L>
__global__ __launch_bounds__(kBlock) void sh_bwd_kernel(const float *__restrict__ xyz, const float *__restrict__ band0,
                                                        const float *__restrict__ sh, float cx, float cy, float cz,
                                                        const float *__restrict__ grgb, int N,
                                                        float *__restrict__ gsh, float *__restrict__ gband0,
                                                        float *__restrict__ gxyz) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= N) return;
  constexpr int n = (L + 1) * (L + 1);
  const float gr[3] = {grgb[3 * i], grgb[3 * i + 1], grgb[3 * i + 2]};
  float ox, oy, oz, b0g[3];
  gs::sh_bwd<L>(sh + (size_t)i * (n - 1) * 3, band0 + 3 * i, xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], cx, cy, cz,
                gr, gsh + (size_t)i * (n - 1) * 3, b0g, ox, oy, oz);
  gband0[3 * i] = b0g[0]; gband0[3 * i + 1] = b0g[1]; gband0[3 * i + 2] = b0g[2];
  gxyz[3 * i] += ox; gxyz[3 * i + 1] += oy; gxyz[3 * i + 2] += oz;
}

// ------------------------------------------------------------------ compaction
__global__ __launch_bounds__(kBlock) void mask_to_int_kernel(const unsigned char *__restrict__ mask, int N,
                                                             int *__restrict__ flags) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i < N) flags[i] = mask[i] ? 1 : 0;
}

// one thread per destination element; rank[i] = exclusive scan of the mask
__global__ __launch_bounds__(kBlock) void compact_rows_kernel(const float *__restrict__ src,
                                                              const unsigned char *__restrict__ mask,
                                                              const int *__restrict__ rank, long long total,
                                                              int stride, float *__restrict__ dst) {
  const long long e = (long long)blockIdx.x * kBlock + threadIdx.x;
  if (e >= total) return;
  const int i = (int)(e / stride), k = (int)(e % stride);
  if (mask[i]) dst[(size_t)rank[i] * stride + k] = src[e];
}

__global__ __launch_bounds__(kBlock) void scatter_rows_kernel(const float *__restrict__ src,
                                                              const unsigned char *__restrict__ mask,
                                                              const int *__restrict__ rank, long long total,
                                                              int stride, float *__restrict__ dst) {
  const long long e = (long long)blockIdx.x * kBlock + threadIdx.x;
  if (e >= total) return;
  const int i = (int)(e / stride), k = (int)(e % stride);
  if (mask[i]) dst[e] = src[(size_t)rank[i] * stride + k];
}

}  // namespace

namespace gs {
// exclusive scan of a byte mask into int ranks (+ total at ranks[N]); shared with gs_fused.hip
int mask_ranks(const unsigned char *mask, int N, int *ranks /*N+1*/, hipStream_t st) {
  if (N <= 0) return GSPLAT_OK;
  DeviceBuffer &flags = scratch(SCR_COUNTS);
  int rc = flags.reserve((size_t)(N + 1) * sizeof(int));
  if (rc) return rc;
  GS_HIP(hipMemsetAsync(flags.as<int>() + N, 0, sizeof(int), st));
  mask_to_int_kernel<<<div_up(N, kBlock), kBlock, 0, st>>>(mask, N, flags.as<int>());
  GS_LAUNCH_CHECK();
  size_t tmp_bytes = 0;
  GS_HIP(rocprim::exclusive_scan(nullptr, tmp_bytes, flags.as<int>(), ranks, 0, (size_t)N + 1, rocprim::plus<int>(), st));
  DeviceBuffer &tmp = scratch(SCR_TEMP);
  rc = tmp.reserve(tmp_bytes);
  if (rc) return rc;
  GS_HIP(rocprim::exclusive_scan(tmp.ptr, tmp_bytes, flags.as<int>(), ranks, 0, (size_t)N + 1, rocprim::plus<int>(), st));
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
  GS_REQUIRE(N >= 0 && stride > 0, "N < 0 or stride <= 0");
  if (num_selected) *num_selected = 0;
  if (N == 0) return GSPLAT_OK;  // empty input is legal (tests/cuda_data_test.cpp CompactMaskedArrayEmpty)
  GS_REQUIRE_DEV(src); GS_REQUIRE_DEV(mask); GS_REQUIRE_DEV(dst);
  hipStream_t st = (hipStream_t)stream;
  gs::ScratchLock lock;  // library scratch and the pinned count words are process-wide
  gs::DeviceBuffer &ranks = gs::scratch(gs::SCR_OFFSETS);
  int rc = ranks.reserve((size_t)(N + 1) * sizeof(int));
  if (rc) return rc;
  rc = gs::mask_ranks(mask, N, ranks.as<int>(), st);
  if (rc) return rc;
  const long long total = (long long)N * stride;
  compact_rows_kernel<<<gs::div_up(total, kBlock), kBlock, 0, st>>>(src, mask, ranks.as<int>(), total, stride, dst);
  GS_LAUNCH_CHECK();
  if (!num_selected) return GSPLAT_OK;  // the caller knows the count (the reference's call sites pass num_culled)
  rc = gs::host_words().ensure();
  if (rc) return rc;
  GS_HIP(hipMemcpyAsync(gs::host_words().p, ranks.as<int>() + N, sizeof(int), hipMemcpyDeviceToHost, st));
  GS_HIP(hipStreamSynchronize(st));
  *num_selected = gs::host_words().p[0];
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
  gs::DeviceBuffer &ranks = gs::scratch(gs::SCR_OFFSETS);
  int rc = ranks.reserve((size_t)(N + 1) * sizeof(int));
  if (rc) return rc;
  rc = gs::mask_ranks(mask, N, ranks.as<int>(), st);
  if (rc) return rc;
  const long long total = (long long)N * stride;
  scatter_rows_kernel<<<gs::div_up(total, kBlock), kBlock, 0, st>>>(src, mask, ranks.as<int>(), total, stride, dst);
  GS_LAUNCH_CHECK();
  return GSPLAT_OK;
}

}  // extern "C"
