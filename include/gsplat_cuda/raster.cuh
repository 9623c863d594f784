// raster.cuh -- source-compatible shim for the reference's per-view forward
// (reference: include/gsplat_cuda/raster.cuh:22-24, cuda/raster.cu:12-136), on top of gsplat_rasterize_image.
//
// rasterize_image is a template on the three HOST types it only reads a few members of, so that it compiles both in
// the reference tree (Camera / Image from dataloader/colmap.hpp, ConfigParameters from gsplat/utils.hpp: Eigen based)
// and in a tree without Eigen:
//   camera.width, camera.height, camera.params[0], camera.params[1]     (cuda/raster.cu:15-16,92-93)
//   image.CamPos()[0..2]                                                 (cuda/raster.cu:76)
//   config.near_thresh, config.cull_mask_padding, config.mh_dist         (cuda/raster.cu:33,100)
// camera_parameters.d_view / d_proj must already hold this view's matrices, as in the reference
// (cuda/trainer.cu:1310-1331).  Like the reference it prints to stderr and exits when nothing is in view.
//
// The fused forward writes its outputs into blocks of the library's pool that belong to a gsplat_context; the
// reference's callers read vectors owned by pass_data, so ownership of the thirteen blocks moves there
// (gsplat_context_detach_forward_outputs: no copy -- r04 copied 0.2 GB per call) and the context takes replacements from
// the pool at its next forward: a fresh ForwardPassData per iteration allocates nothing and copies nothing.
#pragma once

#include "cuda_data.cuh"

namespace gsplat_shim {
// one workspace per (capacity, image size); the reference's rasterize_image is called from a single training thread
inline gsplat_context *context_for(int capacity, int width, int height) {
  static gsplat_context *ctx = nullptr;
  static int cap = 0, w = 0, h = 0;
  if (!ctx || capacity > cap || width > w || height > h) {
    if (ctx) (void)gsplat_context_destroy(ctx);
    cap = capacity > cap ? capacity : cap; w = width > w ? width : w; h = height > h ? height : h;
    require_ok(gsplat_context_create(&ctx, cap, w, h), "rasterize_image (workspace)");
  }
  return ctx;
}

}  // namespace gsplat_shim

template <class CameraT, class ImageT, class ConfigT>
void rasterize_image(const int num_gaussians, const CameraT &camera, const ImageT &image, const ConfigT &config,
                     CameraParameters &camera_parameters, GaussianParameters &gaussians, ForwardPassData &pass_data,
                     const float bg_color, const int l_max) {
  const int width = (int)camera.width, height = (int)camera.height;
  const int capacity = (int)(gaussians.d_opacity.size() > (size_t)num_gaussians ? gaussians.d_opacity.size() : (size_t)num_gaussians);
  gsplat_context *ctx = gsplat_shim::context_for(capacity, width, height);
  gsplat_shim::last_context() = ctx;
  const auto campos = image.CamPos();
  gsplat_camera cam;
  cam.width = width; cam.height = height;
  cam.focal_x = (float)camera.params[0]; cam.focal_y = (float)camera.params[1];
  cam.campos[0] = (float)campos[0]; cam.campos[1] = (float)campos[1]; cam.campos[2] = (float)campos[2];
  cam.view = thrust::raw_pointer_cast(camera_parameters.d_view.data());
  cam.proj = thrust::raw_pointer_cast(camera_parameters.d_proj.data());
  gsplat_gaussians g;
  g.num_gaussians = num_gaussians;
  g.xyz = thrust::raw_pointer_cast(gaussians.d_xyz.data()); g.rgb = thrust::raw_pointer_cast(gaussians.d_rgb.data());
  g.sh = l_max > 0 ? thrust::raw_pointer_cast(gaussians.d_sh.data()) : nullptr;
  g.opacity = thrust::raw_pointer_cast(gaussians.d_opacity.data()); g.scale = thrust::raw_pointer_cast(gaussians.d_scale.data());
  g.quaternion = thrust::raw_pointer_cast(gaussians.d_quaternion.data());
  gsplat_raster_config rc;
  rc.near_thresh = (float)config.near_thresh; rc.mh_dist = (float)config.mh_dist; rc.cull_mask_padding = (int)config.cull_mask_padding;
  gsplat_forward_view v;
  const int status = gsplat_rasterize_image(ctx, &g, &cam, &rc, bg_color, l_max, &v, 0);
  if (status == GSPLAT_ERR_NO_VISIBLE) {
    std::fprintf(stderr, "Error no Gaussians in view for image\n");  // cuda/raster.cu:38-41
    std::exit(EXIT_FAILURE);
  }
  gsplat_shim::require_ok(status, "rasterize_image");
  const size_t N = (size_t)num_gaussians, M = v.num_culled, P = (size_t)width * height;
  const size_t T = (size_t)((width + 15) / 16) * ((height + 15) / 16);
  pass_data.num_culled = M;
  // r05: ForwardPassData takes over the forward's OWN output arrays (blocks of the library's pool) instead of copying
  // them: the context replaces them from the pool at its next forward -- with the very blocks this pass_data returns
  // when the host destroys it at the end of the iteration (cuda/trainer.cu:1295).  r04 copied 0.2 GB here.
  gsplat_shim::require_ok(gsplat_context_detach_forward_outputs(ctx), "rasterize_image (hand-over)");
  pass_data.d_mask.adopt(v.mask, N);
  pass_data.d_uv.adopt(v.uv, N * 2);
  pass_data.d_xyz_c.adopt(v.xyz_c, N * 3);
  pass_data.d_sigma.adopt(v.sigma, M * 6);
  pass_data.d_conic.adopt(v.conic, M * 3);
  pass_data.d_J.adopt(v.J, M * 6);
  pass_data.d_precomputed_rgb.adopt(v.precomputed_rgb, M * 3);
  pass_data.d_radius.adopt(v.radius, M);
  pass_data.d_sorted_gaussians.adopt(v.sorted_gaussians, v.num_splats);
  pass_data.d_splat_start_end_idx_by_tile_idx.adopt(v.splat_start_end_idx_by_tile_idx, T + 1);
  pass_data.d_image_buffer.adopt(v.image, P * 3);
  pass_data.d_weight_per_pixel.adopt(v.weight_per_pixel, P);
  pass_data.d_splats_per_pixel.adopt(v.splats_per_pixel, P);
}
