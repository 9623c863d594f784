// cuda_backward.cuh -- drop-in for the reference header of the same name
// (AndrewBoessen/3DGS include/gsplat_cuda/cuda_backward.cuh:21-123): the seven backward
// operators with the reference's parameter lists, each forwarding to include/gsplat_hip.h.
// Output semantics ("=" vs "+=") are those of the reference and are listed in gsplat_hip.h.
#pragma once
#include "hip_compat.h"

inline constexpr int TILE_SIZE_BWD = 16;

inline void project_to_screen_backward(const float *const xyz_c, const float *const proj,
                                       const float *const uv_grad_out, const int N, const int width, const int height,
                                       float *xyz_c_grad_in, cudaStream_t stream = 0) {
  gsplat_shim::require_ok(
      gsplat_project_to_screen_backward(xyz_c, proj, uv_grad_out, N, width, height, xyz_c_grad_in, stream),
      "project_to_screen_backward");
}

inline void compute_camera_space_points_backward(const float *const xyz_w, const float *const view,
                                                 const float *const xyz_c_grad_out, const int N, float *xyz_w_grad_in,
                                                 cudaStream_t stream = 0) {
  gsplat_shim::require_ok(
      gsplat_compute_camera_space_points_backward(xyz_w, view, xyz_c_grad_out, N, xyz_w_grad_in, stream),
      "compute_camera_space_points_backward");
}

inline void compute_projection_jacobian_backward(const float *const xyz_c, const float focal_x, const float focal_y,
                                                 const float tan_fovx, const float tan_fovy,
                                                 const float *const J_grad_out, const int N, float *xyz_c_grad_in,
                                                 cudaStream_t stream = 0) {
  gsplat_shim::require_ok(gsplat_compute_projection_jacobian_backward(xyz_c, focal_x, focal_y, tan_fovx, tan_fovy,
                                                                      J_grad_out, N, xyz_c_grad_in, stream),
                          "compute_projection_jacobian_backward");
}

inline void compute_conic_backward(const float *const J, const float *const sigma, const float *const view,
                                   const float *const conic, const float *const conic_grad_out, const int N,
                                   float *J_grad_in, float *sigma_grad_in, cudaStream_t stream = 0) {
  gsplat_shim::require_ok(
      gsplat_compute_conic_backward(J, sigma, view, conic, conic_grad_out, N, J_grad_in, sigma_grad_in, stream),
      "compute_conic_backward");
}

inline void compute_sigma_backward(const float *const quaternion, const float *const scale,
                                   const float *const sigma_grad_out, const int N, float *quaternion_grad_in,
                                   float *scale_grad_in, cudaStream_t stream = 0) {
  gsplat_shim::require_ok(
      gsplat_compute_sigma_backward(quaternion, scale, sigma_grad_out, N, quaternion_grad_in, scale_grad_in, stream),
      "compute_sigma_backward");
}

inline void precompute_spherical_harmonics_backward(const float *const xyz_c, const float *const rgb_vals,
                                                    const float *const sh_coeffs, const float3 campos,
                                                    const float *const rgb_grad_out, const int l_max, const int N,
                                                    float *sh_grad_in, float *sh_grad_band_0_in, float *xyz_c_grad_in,
                                                    cudaStream_t stream = 0) {
  gsplat_shim::require_ok(gsplat_precompute_spherical_harmonics_backward(
                              xyz_c, rgb_vals, sh_coeffs, campos.x, campos.y, campos.z, rgb_grad_out, l_max, N,
                              sh_grad_in, sh_grad_band_0_in, xyz_c_grad_in, stream),
                          "precompute_spherical_harmonics_backward");
}

inline void render_image_backward(const float *const uvs, const float *const opacity, const float *const conic,
                                  const float *const rgb, const float background_opacity,
                                  const int *const sorted_splats, const int *const splat_range_by_tile,
                                  const int *const num_splats_per_pixel, const float *const final_weight_per_pixel,
                                  const float *const grad_image, const int image_width, const int image_height,
                                  float *grad_rgb, float *grad_opacity, float *grad_uv, float *grad_conic,
                                  cudaStream_t stream = 0) {
  gsplat_shim::require_ok(
      gsplat_render_image_backward(uvs, opacity, conic, rgb, background_opacity, sorted_splats, splat_range_by_tile,
                                   num_splats_per_pixel, final_weight_per_pixel, grad_image, image_width, image_height,
                                   grad_rgb, grad_opacity, grad_uv, grad_conic, stream),
      "render_image_backward");
}
