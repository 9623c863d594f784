// cuda_forward.cuh -- drop-in for the reference header of the same name
// (AndrewBoessen/3DGS include/gsplat_cuda/cuda_forward.cuh:26-131).  Same free functions,
// same parameter lists and default stream; every body is one call into the MI355X library
// (include/gsplat_hip.h).  All of the reference header's functions are here: the rasterizer
// operators, and fused_loss / compute_psnr / compute_morton_codes at the
// end of the file.  Only accumulate_gradients is absent: the reference declares it and never
// defines it.
#pragma once
#include "hip_compat.h"

inline constexpr int TILE_SIZE_FWD = 16;

inline void compute_conic(float *const xyz, const float *view, float *const sigma, const float focal_x,
                          const float focal_y, const float tan_fovx, const float tan_fovy, const float mh_dist,
                          const int N, float *J, float *conic, float4 *radius, cudaStream_t stream = 0) {
  gsplat_shim::require_ok(gsplat_compute_conic(xyz, view, sigma, focal_x, focal_y, tan_fovx, tan_fovy, mh_dist, N, J,
                                               conic, reinterpret_cast<float *>(radius), stream),
                          "compute_conic");
}

inline void compute_sigma(float *const quaternion, float *const scale, const int N, float *sigma,
                          cudaStream_t stream = 0) {
  gsplat_shim::require_ok(gsplat_compute_sigma(quaternion, scale, N, sigma, stream), "compute_sigma");
}

inline void compute_camera_space_points(float *const xyz_w, const float *view, const int N, float *xyz_c,
                                        cudaStream_t stream = 0) {
  gsplat_shim::require_ok(gsplat_compute_camera_space_points(xyz_w, view, N, xyz_c, stream),
                          "compute_camera_space_points");
}

inline void project_to_screen(float *const xyz, const float *proj, const int N, const int width, const int height,
                              float *uv, cudaStream_t stream = 0) {
  gsplat_shim::require_ok(gsplat_project_to_screen(xyz, proj, N, width, height, uv, stream), "project_to_screen");
}

inline void cull_gaussians(float *const uv, float *const xyz, const int N, const float near_thresh, const int padding,
                           const int width, const int height, bool *mask, cudaStream_t stream = 0) {
  static_assert(sizeof(bool) == 1, "mask is one byte per gaussian");
  gsplat_shim::require_ok(gsplat_cull_gaussians(uv, xyz, N, near_thresh, padding, width, height,
                                                reinterpret_cast<unsigned char *>(mask), stream),
                          "cull_gaussians");
}

inline void get_sorted_gaussian_list(const float *uv, const float *xyz, const float4 *radius, const int n_tiles_x,
                                     const int n_tiles_y, const int N, size_t &sorted_gaussian_bytes,
                                     int *sorted_gaussians, int *splat_start_end_idx_by_tile_idx,
                                     cudaStream_t stream = 0) {
  gsplat_shim::require_ok(
      gsplat_get_sorted_gaussian_list(uv, xyz, reinterpret_cast<const float *>(radius), n_tiles_x, n_tiles_y, N,
                                      &sorted_gaussian_bytes, sorted_gaussians, splat_start_end_idx_by_tile_idx, stream),
      "get_sorted_gaussian_list");
}

inline void precompute_spherical_harmonics(const float *xyz, const float *sh_coefficients,
                                           const float *sh_coeffs_band_0, const float3 campos, const int l_max,
                                           const int N, float *rgb, cudaStream_t stream = 0) {
  gsplat_shim::require_ok(gsplat_precompute_spherical_harmonics(xyz, sh_coefficients, sh_coeffs_band_0, campos.x,
                                                                campos.y, campos.z, l_max, N, rgb, stream),
                          "precompute_spherical_harmonics");
}

inline void render_image(const float *uv, const float *opacity, const float *conic, const float *rgb,
                         const float background_opacity, const int *sorted_splats, const int *splat_range_by_tile,
                         const int image_width, const int image_height, int *splats_per_pixel,
                         float *weight_per_pixel, float *image, cudaStream_t stream = 0) {
  gsplat_shim::require_ok(gsplat_render_image(uv, opacity, conic, rgb, background_opacity, sorted_splats,
                                              splat_range_by_tile, image_width, image_height, splats_per_pixel,
                                              weight_per_pixel, image, stream),
                          "render_image");
}

// "next" row f1: loss + metric (reference declarations: include/gsplat_cuda/cuda_forward.cuh:144-156)
inline float fused_loss(const float *predicted_data, const float *gt_data, int rows, int cols, const float ssim_weight,
                        float *image_grad, cudaStream_t stream = 0) {
  float loss = 0.0f;
  gsplat_shim::require_ok(gsplat_fused_loss(predicted_data, gt_data, rows, cols, ssim_weight, image_grad, &loss, stream),
                          "fused_loss");
  return loss;
}

inline float compute_psnr(const float *predicted_data, const float *gt_data, int rows, int cols,
                          cudaStream_t stream = 0) {
  float psnr = 0.0f;
  gsplat_shim::require_ok(gsplat_compute_psnr(predicted_data, gt_data, rows, cols, &psnr, stream), "compute_psnr");
  return psnr;
}

// "next" row f4 operator (reference declaration: include/gsplat_cuda/cuda_forward.cuh:186-188)
inline void compute_morton_codes(const int N, const float *d_xyz, const float x_max, const float y_max,
                                 const float z_max, const float x_min, const float y_min, const float z_min,
                                 uint64_t *codes, cudaStream_t stream = 0) {
  static_assert(sizeof(uint64_t) == sizeof(unsigned long long), "64-bit codes");
  gsplat_shim::require_ok(gsplat_compute_morton_codes(N, d_xyz, x_max, y_max, z_max, x_min, y_min, z_min,
                                                      reinterpret_cast<unsigned long long *>(codes), stream),
                          "compute_morton_codes");
}
