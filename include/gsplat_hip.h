/*
 * gsplat_hip.h -- C ABI of libgsplat_hip.so, the MI355X (gfx950) differentiable
 * gaussian-splat rasterizer.
 *
 * Drop-in boundary.  Every gsplat_<op> below replaces the free function <op> that the
 * reference declares in include/gsplat_cuda/cuda_forward.cuh:26-131 and
 * include/gsplat_cuda/cuda_backward.cuh:21-123 (AndrewBoessen/3DGS) and takes the same
 * raw device pointers and scalars in the same order.  ABI adaptations, all mechanical:
 *   float3 campos        -> three floats
 *   float4* radius       -> float* (4 floats per gaussian, 16-byte aligned)
 *   bool* mask           -> unsigned char* (one byte per gaussian, 0/1)
 *   size_t& count        -> size_t*
 *   cudaStream_t stream  -> void* (a hipStream_t; NULL = the null stream)
 *   void / exit(1)       -> int status (0 = GSPLAT_OK, negative = error; nothing exits)
 * include/gsplat_cuda/cuda_forward.cuh and cuda_backward.cuh in this repository are C++
 * shims with the reference's exact signatures (and its print-and-exit error behaviour)
 * that forward to these symbols, so a host written against the reference's headers
 * compiles unchanged.
 *
 * Ownership: the caller owns every buffer it passes.  Outputs documented "+=" are
 * accumulated into and must be pre-zeroed by the caller, exactly as in the reference
 * (cuda/trainer.cu:247-261).  Operators may use library-owned scratch memory that is
 * grown on demand and released by gsplat_release_scratch().
 *
 * Threading: the stand-alone operators are written for one host thread per device at a time (the reference's trainer
 * thread).  A gsplat_context is used by one thread at a time; different contexts may be driven by different threads
 * of one process on the same stream (several in-process ranks, 3dgs_amd/dist.py ThreadGroup): the entry points that
 * touch the library-owned scratch (gsplat_fused_loss, gsplat_compute_psnr, gsplat_compact/scatter_masked_array,
 * gsplat_get_sorted_gaussian_list, gsplat_initialize_gaussians, gsplat_knn_mean_distance) serialise on a lock.
 */
#ifndef GSPLAT_HIP_H
#define GSPLAT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GSPLAT_OK 0
#define GSPLAT_ERR_NULL_POINTER (-1)   /* a required pointer is NULL            (checks.cuh:17-21)  */
#define GSPLAT_ERR_NOT_DEVICE (-2)     /* pointer is not device memory          (checks.cuh:23-38)  */
#define GSPLAT_ERR_INVALID_ARG (-3)    /* negative size, l_max outside 0..3 ... (raster.cu:58-60)   */
#define GSPLAT_ERR_HIP (-4)            /* a HIP runtime call or kernel launch failed                 */
#define GSPLAT_ERR_NO_VISIBLE (-5)     /* nothing survives culling              (raster.cu:38-41)   */
#define GSPLAT_ERR_CAPACITY (-6)       /* context capacity exceeded                                   */

#define GSPLAT_TILE_SIZE 16 /* TILE_SIZE_FWD / TILE_SIZE_BWD, cuda_forward.cuh:8, cuda_backward.cuh:8 */

/* Human-readable text for the last error raised on the calling thread. */
const char *gsplat_last_error(void);
/* Library ABI version (bumped when a signature changes); bindings compare it with the GSPLAT_ABI_VERSION they were
 * written against, so a stale prebuilt library fails at load time, not with a wrong argument list. */
#define GSPLAT_ABI_VERSION 8
int gsplat_abi_version(void);
/* What the loaded binary was built from: sha256 (first 16 hex digits) over the kernel sources (3dgs_amd/csrc: Makefile,
 * *.h, *.hip) at link time, and the extra compiler flags of a diagnostic build ("" for the product build).  A loader
 * that did not build the library itself compares the hash with the sources it sees (3dgs_amd/_lib.py), and stored
 * profiles (profiles/traffic.json) are tied to it. */
const char *gsplat_source_hash(void);
const char *gsplat_build_flags(void);
/* Frees library-owned scratch memory of the current device. */
int gsplat_release_scratch(void);

/* Device block pool.  The reference host creates and destroys ~20 thrust::device_vectors per training iteration (a
 * fresh ForwardPassData, cuda/trainer.cu:1295; one vector per compact_masked_array, cuda/trainer.cu:941-964,1028-1044):
 * with the runtime's allocator every one of them is a hipMalloc + a hipFree (which synchronises the device) -- 4 of the
 * 6.5 ms the unmodified reference-host iteration took at 1e6 gaussians.  The vectors the drop-in headers create
 * (include/gsplat_cuda/cuda_data.cuh: gsplat_shim::device_array) draw from this pool instead: a freed block is kept and
 * handed to the next request of its size class (three mantissa bits: <= 12.5 % slack), so the steady state allocates
 * nothing.  Blocks are reused in STREAM ORDER: a block freed while work of stream s still reads it is first re-used by
 * work queued on s afterwards; a request from another stream is ordered behind s (r06: the _on forms below).
 * gsplat_pool_release hipFree()s the idle blocks; gsplat_pool_bytes reports idle (idle_only != 0) or idle + live bytes. */
int gsplat_pool_alloc(void **ptr, size_t bytes);
int gsplat_pool_free(void *ptr);
/* r06: the stream-aware forms.  gsplat_pool_free_on(ptr, s): work queued on stream `s` so far may still touch the block.
 * gsplat_pool_alloc_on(&ptr, bytes, t): the block will next be touched by work queued on stream `t`; when it takes a
 * block that another stream returned, `t` is first ordered behind that stream (an event recorded on it now -- later than
 * the return, hence covering it -- and awaited by `t`; a device synchronisation if that stream no longer exists); blocks
 * returned by `t` itself are preferred and cost nothing.  The plain forms above are the NULL-stream forms
 * (cuda/trainer.cu: everything but the image upload runs on the NULL stream; its transfer_stream, :1257-1259, is the
 * reason the pool needs the ordering).  gsplat_pool_cross_stream_reuses: how often the ordering was needed. */
int gsplat_pool_alloc_on(void **ptr, size_t bytes, void *stream);
int gsplat_pool_free_on(void *ptr, void *stream);
unsigned long long gsplat_pool_cross_stream_reuses(void);
int gsplat_pool_release(void);
/* Frees idle blocks of the CURRENT device (largest first) while more than keep_bytes of them are cached; other devices
 * are neither synchronised nor touched.  What gsplat_context_destroy calls (keep_bytes = 1 GiB). */
int gsplat_pool_trim(size_t keep_bytes);
size_t gsplat_pool_bytes(int idle_only);

/* ---------------------------------------------------------------- forward operators --- */

/* replaces compute_camera_space_points  (cuda_forward.cuh:50-51, cuda/projection.cu:100-114)
 * xyz_c[N,3] = view[0:3,0:4] * [xyz_w;1], view row-major 4x4 */
int gsplat_compute_camera_space_points(const float *xyz_w, const float *view, int N, float *xyz_c, void *stream);

/* replaces project_to_screen  (cuda_forward.cuh:63-64, cuda/projection.cu:116-130) -> uv[N,2] */
int gsplat_project_to_screen(const float *xyz, const float *proj, int N, int width, int height, float *uv,
                             void *stream);

/* replaces cull_gaussians  (cuda_forward.cuh:78-79, cuda/culling.cu:361-375); mask: 1 = keep */
int gsplat_cull_gaussians(const float *uv, const float *xyz, int N, float near_thresh, int padding, int width,
                          int height, unsigned char *mask, void *stream);

/* replaces compute_sigma  (cuda_forward.cuh:39, cuda/gaussian.cu:220-235)
 * quaternion[N,4] in (w,x,y,z) order, scale[N,3] log-scales -> sigma[N,6] = [xx,xy,xz,yy,yz,zz] */
int gsplat_compute_sigma(const float *quaternion, const float *scale, int N, float *sigma, void *stream);

/* replaces compute_conic  (cuda_forward.cuh:26-28, cuda/gaussian.cu:237-262)
 * -> J[N,6], conic[N,3] = [c00,c01,c11], radius[N,4] = {r_major, r_minor, sin(theta), cos(theta)} */
int gsplat_compute_conic(const float *xyz, const float *view, const float *sigma, float focal_x, float focal_y,
                         float tan_fovx, float tan_fovy, float mh_dist, int N, float *J, float *conic,
                         float *radius, void *stream);

/* replaces get_sorted_gaussian_list  (cuda_forward.cuh:95-97, cuda/culling.cu:386-475)
 * Two-call protocol, as in the reference:
 *   sorted_gaussians == NULL : *sorted_gaussian_count <- number of coarse (tile, gaussian)
 *                              candidate pairs; the caller sizes sorted_gaussians with it.
 *   sorted_gaussians != NULL : sorted_gaussians[0..S) <- gaussian ids ordered by (tile, depth),
 *                              ties by gaussian id; splat_start_end_idx_by_tile_idx[0..T] <-
 *                              start offsets (entry T = S).  Blocks the host, as the reference does. */
int gsplat_get_sorted_gaussian_list(const float *uv, const float *xyz, const float *radius, int n_tiles_x,
                                    int n_tiles_y, int N, size_t *sorted_gaussian_count, int *sorted_gaussians,
                                    int *splat_start_end_idx_by_tile_idx, void *stream);

/* replaces precompute_spherical_harmonics  (cuda_forward.cuh:111-113, cuda/spherical_harmonics.cu:62-94)
 * sh_coefficients[N,(l_max+1)^2-1,3] (may be NULL when l_max == 0), band 0 in sh_coeffs_band_0[N,3];
 * rgb = 0.5 + sum coeff * Y(dir), dir = normalize(xyz - campos); no clamp */
int gsplat_precompute_spherical_harmonics(const float *xyz, const float *sh_coefficients,
                                          const float *sh_coeffs_band_0, float campos_x, float campos_y,
                                          float campos_z, int l_max, int N, float *rgb, void *stream);

/* replaces render_image  (cuda_forward.cuh:131-134, cuda/render.cu:110-135)
 * -> splats_per_pixel[H,W], weight_per_pixel[H,W] (final transmittance), image[H,W,3] */
int gsplat_render_image(const float *uv, const float *opacity, const float *conic, const float *rgb,
                        float background_opacity, const int *sorted_splats, const int *splat_range_by_tile,
                        int image_width, int image_height, int *splats_per_pixel, float *weight_per_pixel,
                        float *image, void *stream);

/* --------------------------------------------------------------- backward operators --- */

/* replaces project_to_screen_backward  (cuda_backward.cuh:21-23); xyz_c_grad_in += */
int gsplat_project_to_screen_backward(const float *xyz_c, const float *proj, const float *uv_grad_out, int N,
                                      int width, int height, float *xyz_c_grad_in, void *stream);

/* replaces compute_camera_space_points_backward  (cuda_backward.cuh:34-36); xyz_w_grad_in += */
int gsplat_compute_camera_space_points_backward(const float *xyz_w, const float *view, const float *xyz_c_grad_out,
                                                int N, float *xyz_w_grad_in, void *stream);

/* replaces compute_projection_jacobian_backward  (cuda_backward.cuh:47-49); xyz_c_grad_in += */
int gsplat_compute_projection_jacobian_backward(const float *xyz_c, float focal_x, float focal_y, float tan_fovx,
                                                float tan_fovy, const float *J_grad_out, int N,
                                                float *xyz_c_grad_in, void *stream);

/* replaces compute_conic_backward  (cuda_backward.cuh:61-63); J_grad_in +=, sigma_grad_in +=
 * (off-diagonal sigma entries receive the sum of both symmetric positions) */
int gsplat_compute_conic_backward(const float *J, const float *sigma, const float *view, const float *conic,
                                  const float *conic_grad_out, int N, float *J_grad_in, float *sigma_grad_in,
                                  void *stream);

/* replaces compute_sigma_backward  (cuda_backward.cuh:75-76); quaternion_grad_in =, scale_grad_in = */
int gsplat_compute_sigma_backward(const float *quaternion, const float *scale, const float *sigma_grad_out, int N,
                                  float *quaternion_grad_in, float *scale_grad_in, void *stream);

/* replaces precompute_spherical_harmonics_backward  (cuda_backward.cuh:90-94)
 * sh_grad_in =, sh_grad_band_0_in =, xyz_c_grad_in += */
int gsplat_precompute_spherical_harmonics_backward(const float *xyz_c, const float *rgb_vals,
                                                   const float *sh_coeffs, float campos_x, float campos_y,
                                                   float campos_z, const float *rgb_grad_out, int l_max, int N,
                                                   float *sh_grad_in, float *sh_grad_band_0_in,
                                                   float *xyz_c_grad_in, void *stream);

/* replaces render_image_backward  (cuda_backward.cuh:116-122); all four outputs += (pre-zeroed by the caller);
 * grad_uv carries the reference's extra 0.5*W / 0.5*H factor (cuda/render_backward.cu:186-187) */
int gsplat_render_image_backward(const float *uvs, const float *opacity, const float *conic, const float *rgb,
                                 float background_opacity, const int *sorted_splats,
                                 const int *splat_range_by_tile, const int *num_splats_per_pixel,
                                 const float *final_weight_per_pixel, const float *grad_image, int image_width,
                                 int image_height, float *grad_rgb, float *grad_opacity, float *grad_uv,
                                 float *grad_conic, void *stream);

/* ---------------------------------------------------- "next" rows: loss, metric, optimizer --- */

/* replaces fused_loss  (cuda_forward.cuh:146-147, cuda/loss.cu:430-471): L1 + SSIM loss and dL/dimage.
 * image_grad[H,W,3] is overwritten.  If loss_out != NULL the mean loss is read back (blocks the host, as the
 * reference's return value does); pass NULL to stay asynchronous. */
int gsplat_fused_loss(const float *predicted_data, const float *gt_data, int rows, int cols, float ssim_weight,
                      float *image_grad, float *loss_out, void *stream);

/* replaces compute_psnr  (cuda_forward.cuh:158, cuda/loss.cu:510-525); blocks the host */
int gsplat_compute_psnr(const float *predicted_data, const float *gt_data, int rows, int cols, float *psnr_out,
                        void *stream);

/* replaces adam_step  (include/gsplat_cuda/optimizer.cuh:27-29, cuda/optimizer.cu:31-44) on N*S elements */
int gsplat_adam_step(float *params, const float *param_grads, float *exp_avg, float *exp_avg_sq, float lr, float b1,
                     float b2, float eps, float bias1, float bias2, int N, int S, void *stream);

/* One Adam parameter group: global-order parameter and moment arrays [N,stride] and its learning rate. */
#define GSPLAT_MAX_ADAM_GROUPS 8
typedef struct gsplat_adam_group {
  float *param;             /* [N,stride] device, updated in place                                  */
  float *exp_avg;           /* [N,stride] device, first moment                                      */
  float *exp_avg_sq;        /* [N,stride] device, second moment                                     */
  const float *grad;        /* [M,stride] device, compacted order (gsplat_optimizer_step only)      */
  int stride;
  int packed_column;        /* first column in a packed row (gsplat_optimizer_step_packed only)     */
  float lr;
} gsplat_adam_group;

/* replaces TrainerImpl::optimizer_step  (cuda/trainer.cu:1027-1158): the reference compacts parameters and both
 * moments of every group by the view's mask, runs adam_step on the compacted copies and scatters them back
 * (~35 thrust passes).  Here one kernel updates the visible rows in place through compact_to_global
 * (gsplat_forward_view.compact_to_global, length num_culled); rows the view did not see keep their moments.
 * If uv_grad_accum / grad_accum_dur are non-NULL the densification statistics of trainer.cu:1136-1157 are
 * updated too: uv_grad_accum[i] += |grad_uv[r]|, grad_accum_dur[i] += 1. */
int gsplat_optimizer_step(const int *compact_to_global, int num_culled, const gsplat_adam_group *groups, int n_groups,
                          float b1, float b2, float eps, float bias1, float bias2, const float *grad_uv,
                          float *uv_grad_accum, int *grad_accum_dur, void *stream);

/* The spherical-harmonics group of gsplat_optimizer_step without its gradient array.  The SH gradients of ONE view are an
 * outer product (cuda/spherical_harmonics_backward.cu:168-209): d/d sh[k][c] = Y_{k+1}(direction) * grad_precompute_rgb[c],
 * direction = normalised (xyz - camera position) -- the values the backward would have stored in grad_sh, bit for bit
 * (the same basis function, the same product).  For the visible rows (compact_to_global, num_culled) this entry point
 * rebuilds them from the 24 bytes they are made of and applies the same in-place Adam update to sh / exp_avg /
 * exp_avg_sq [N, (l_max+1)^2 - 1, 3]; a backward that was given grad_sh == NULL plus this call replace a backward that
 * writes grad_sh and an optimizer group that reads it back (360 bytes per visible gaussian and step at l_max 3).
 * xyz [N,3] must still hold the positions the backward saw: call it BEFORE the step that updates the xyz group.
 * grad_precompute_rgb [num_culled,3]: the intermediate gradient of that name (gsplat_gradients). */
int gsplat_optimizer_step_sh_factored(const int *compact_to_global, int num_culled, int l_max, float *sh, float *exp_avg,
                                      float *exp_avg_sq, float lr, float b1, float b2, float eps, float bias1,
                                      float bias2, const float *xyz, float cam_x, float cam_y, float cam_z,
                                      const float *grad_precompute_rgb, void *stream);

/* r05 -- the colour groups of a W-view step without their gradient arrays: the band-0 group (rgb[N,3]) and the SH group
 * (sh[N,(l_max+1)^2-1,3]) are updated in place from what the split exchange delivers -- rgb_all = W blocks of
 * `rank_stride` floats, block r = rank r's g_rgb[N,3] in global order followed by its camera position[3] -- as
 * grad[k][c] = sum over the views r (in rank order) of g_rgb^r[c] * Y_k(direction from camera r to the gaussian): the
 * sums gsplat_unpack_gradients_split writes into the packed rows, formed in the same order with the same operations, so
 * parameters and moments are bit-identical to gsplat_unpack_gradients_split + gsplat_optimizer_step_packed -- without
 * writing and re-reading packed[N, 12 + 3 n] (240 B per gaussian at SH degree 3, each way).  A row is updated when
 * common[i*12 + 11], the number of views that saw the gaussian, is positive (the reference's per-view masks,
 * cuda/trainer.cu:1028-1085).  Must run BEFORE the xyz group moves (the directions are rebuilt from xyz).  The other
 * four groups: gsplat_optimizer_step_packed on common[N,12] itself (width 12, columns xyz 0, opacity 3, scale 4,
 * quaternion 7).  sh may be NULL at l_max == 0. */
int gsplat_optimizer_step_sh_views(int l_max, int num_gaussians, int world_size, const float *xyz, const float *rgb_all,
                                   size_t rank_stride, const float *common, float *rgb, float *rgb_exp_avg,
                                   float *rgb_exp_avg_sq, float lr_rgb, float *sh, float *sh_exp_avg,
                                   float *sh_exp_avg_sq, float lr_sh, float b1, float b2, float eps, float bias1,
                                   float bias2, void *stream);

/* Multi-view variant: gradients are rows of the all-reduced packed layout
 * (gsplat_pack_gradients_global / gsplat_unpack_gradients_factored, row width `width`, last column = number of
 * views that saw the gaussian); rows with a zero count are skipped, which is the union of the per-view masks.
 * Densification statistics of a W-view step (trainer.cu:1136-1157 once per view): when uv_grad_accum / grad_accum_dur
 * are non-NULL, uv_grad_accum[i] += uv_norm_sum[i] (the all-reduced gsplat_pack_uv_grad_norm column: the sum over the
 * step's views of |grad_uv|) and grad_accum_dur[i] += the row's view count. */
int gsplat_optimizer_step_packed(const float *packed, int num_gaussians, int width, const gsplat_adam_group *groups,
                                 int n_groups, float b1, float b2, float eps, float bias1, float bias2,
                                 const float *uv_norm_sum, float *uv_grad_accum, int *grad_accum_dur, void *stream);

/* replaces Gaussians::Initialize  (src/gaussian.cpp:38-104; host kd-tree + OpenMP in the reference): the initial
 * gaussians of a sparse point cloud, computed on the GPU.  points_xyz [N,3] doubles and points_rgb [N,3] bytes are
 * DEVICE arrays (COLMAP's Point3D fields, include/gsplat_host.h reads them); outputs are the GaussianParameters
 * arrays: xyz [N,3], rgb [N,3] = (rgb/255 - 0.5)/C0, opacity [N] = logit(0.2), scale [N,3] = log(mean distance to the
 * 3 nearest neighbours, 0.01 if there is none), quaternion [N,4] = (1,0,0,0).  Blocks the host (two small
 * read-backs size the grid). */
int gsplat_initialize_gaussians(const double *points_xyz, const unsigned char *points_rgb, int N, float *xyz, float *rgb,
                                float *opacity, float *scale, float *quaternion, void *stream);
/* the exact k-nearest-neighbour statistic on its own: mean_dist[i] = mean distance of point i to its k (1..8)
 * nearest other points (duplicates count as neighbours at distance 0, as in the reference's kd-tree query) */
int gsplat_knn_mean_distance(const double *points_xyz, int N, int k, float *mean_dist, void *stream);

/* replaces compute_morton_codes  (cuda_forward.cuh:174-188, cuda/culling.cu:14-63): 63-bit Morton code of each
 * position inside the given box, bit-exact with the reference (including its bit-spread masks). */
int gsplat_compute_morton_codes(int N, const float *xyz, float x_max, float y_max, float z_max, float x_min,
                                float y_min, float z_min, unsigned long long *codes, void *stream);

/* replaces clone_gaussians  (adaptive_density.cuh:9-31, cuda/adaptive_density.cu:12-66): gaussians with mask[i] set
 * are copied to row write_ids[i] (exclusive scan of the mask) of the output arrays.  mask is one byte per gaussian. */
int gsplat_clone_gaussians(int N, int num_sh_coef, const unsigned char *mask, const int *write_ids, const float *xyz_in,
                           const float *rgb_in, const float *op_in, const float *scale_in, const float *quat_in,
                           const float *sh_in, float *xyz_out, float *rgb_out, float *op_out, float *scale_out,
                           float *quat_out, float *sh_out, void *stream);

/* replaces split_gaussians  (adaptive_density.cuh:33-57, cuda/adaptive_density.cu:68-164): every masked gaussian
 * yields two gaussians at rows 2*write_ids[i] and 2*write_ids[i]+1: centres drawn from N(xyz, R diag(exp(scale))^2 R^T),
 * scales log(exp(scale)/scale_factor), everything else copied.  `seed` makes the draw reproducible (the reference
 * seeds from time(NULL); its C++ shim passes that). */
int gsplat_split_gaussians(int N, float scale_factor, int num_sh_coef, const unsigned char *mask, const int *write_ids,
                           const float *xyz_in, const float *rgb_in, const float *op_in, const float *scale_in,
                           const float *quat_in, const float *sh_in, float *xyz_out, float *rgb_out, float *op_out,
                           float *scale_out, float *quat_out, float *sh_out, unsigned long long seed, void *stream);

/* The data-parallel half of TrainerImpl::adaptive_density_step (cuda/trainer.cu:416-575): per gaussian the average
 * screen-space gradient uv_grad_accum / grad_accum_dur, the largest extent max(exp(scale)), and the reference's
 * IdentifyPrune / IdentifyClone / IdentifySplit / CombineMasks decisions.  counts[3] (device) receives the number of
 * pruned, cloned and split gaussians.  Masks are one byte per gaussian. */
int gsplat_density_masks(int N, const float *opacity, const float *scale, const float *uv_grad_accum,
                         const int *grad_accum_dur, float op_threshold, float max_scale, float uv_grad_threshold,
                         float clone_scale_threshold, unsigned char *prune_mask, unsigned char *clone_mask,
                         unsigned char *split_mask, unsigned char *keep_mask, int *counts, void *stream);

/* add_sh_band's re-layout (cuda/trainer.cu:363-413): sh_in [N,(l+1)^2-1,3] -> sh_out [N,(l+2)^2-1,3], new
 * coefficients zero.  l_max_old == 0 just zero-fills the first band. */
int gsplat_expand_sh(int N, int l_max_old, const float *sh_in, float *sh_out, void *stream);

/* sort_gaussians' gather (cuda/trainer.cu:793-851): out[i, :] = in[order[i], :] for rows of `stride` floats. */
int gsplat_gather_rows(int N, int stride, const int *order, const float *in, float *out, void *stream);

/* ------------------------------------------------------------- compaction templates --- */

/* replaces compact_masked_array<STRIDE>  (cuda_data.cuh:106-127): stable compaction of src[N,stride] by mask[N]
 * into dst (room for N*stride floats, or for the selected rows when the caller knows their number, as the reference's
 * callers do: they pass num_culled).  num_selected != NULL: *num_selected <- rows kept (a host value: blocks the host);
 * NULL: no read-back, the call stays asynchronous. */
int gsplat_compact_masked_array(const float *src, const unsigned char *mask, int N, int stride, float *dst,
                                int *num_selected, void *stream);
/* The same with the room of dst stated (r05): rows whose compacted slot is >= dst_rows are NOT written.  The reference's
 * template trusts its caller's num_culled (cuda_data.cuh:106-127) and so do the callers here (no count read-back), but a
 * count that is too small then drops rows instead of overwriting whatever follows dst. */
int gsplat_compact_masked_array_bounded(const float *src, const unsigned char *mask, int N, int stride, float *dst,
                                        int dst_rows, int *num_selected, void *stream);

/* replaces scatter_masked_array<STRIDE>  (cuda_data.cuh:150-167): dst[i] <- src[rank(i)] where mask[i] */
int gsplat_scatter_masked_array(const float *src, const unsigned char *mask, int N, int stride, float *dst,
                                void *stream);

/* --------------------------------------------------- fused per-view pass (additive API) --- */
/*
 * gsplat_rasterize_image / gsplat_backward_pass are the device-resident equivalents of
 * rasterize_image (raster.cuh:22-24, cuda/raster.cu:12-136) and of the operator chain in
 * TrainerImpl::backward_pass (cuda/trainer.cu:941-1012).  They run on a context that owns
 * a persistent workspace (no per-call allocation), fuse the per-gaussian operators into
 * one preprocess kernel each way, and need one host read-back per forward (M and S).
 * Results are identical in layout to ForwardPassData / GaussianGradients: per-gaussian
 * buffers are in compacted (post-cull) order.
 */
typedef struct gsplat_context gsplat_context;

typedef struct gsplat_camera {
  int width, height;
  float focal_x, focal_y;  /* camera.params[0], params[1]                         */
  float campos[3];         /* Image::CamPos()                                     */
  const float *view;       /* device, 16 floats row-major [R|t]  (trainer.cu:1321-1331) */
  const float *proj;       /* device, 16 floats row-major        (trainer.cu:1310-1318) */
} gsplat_camera;

typedef struct gsplat_gaussians {      /* GaussianParameters, cuda_data.cuh:11-16; all device pointers */
  int num_gaussians;
  const float *xyz;        /* [N,3] */
  const float *rgb;        /* [N,3]  SH band 0 */
  const float *sh;         /* [N,(l_max+1)^2-1,3], may be NULL when l_max == 0 */
  const float *opacity;    /* [N]   logits */
  const float *scale;      /* [N,3] log-scales */
  const float *quaternion; /* [N,4] (w,x,y,z) */
} gsplat_gaussians;

typedef struct gsplat_raster_config {  /* the ConfigParameters fields the path reads, raster.cu:33,100 */
  float near_thresh, mh_dist;
  int cull_mask_padding;
} gsplat_raster_config;

typedef struct gsplat_forward_view {   /* ForwardPassData, cuda_data.cuh:70-86: pointers INTO the context */
  size_t num_culled;                   /* M */
  size_t num_pairs;                    /* coarse candidates (what call 1 of get_sorted_gaussian_list reports) */
  size_t num_splats;                   /* S */
  const unsigned char *mask;           /* [N] */
  const float *uv, *xyz_c;             /* [N,2], [N,3] uncompacted */
  const int *compact_to_global;        /* [M] */
  const float *sigma, *conic, *J, *precomputed_rgb, *radius; /* [M,6] [M,3] [M,6] [M,3] [M,4] */
  const float *uv_selected, *xyz_c_selected;                 /* [M,2] [M,3] */
  const int *sorted_gaussians;         /* [S] compacted ids */
  const int *splat_start_end_idx_by_tile_idx; /* [T+1] */
  const float *image, *weight_per_pixel;      /* [H,W,3] [H,W] */
  const int *splats_per_pixel;                /* [H,W] */
} gsplat_forward_view;

typedef struct gsplat_gradients {      /* GaussianGradients (leaf part), compacted order [M,...], caller-owned */
  /* grad_sh may be NULL (ABI 5): a single-view optimizer can rebuild the SH gradients from grad_precompute_rgb and the
   * viewing direction (gsplat_optimizer_step_sh_factored); the backward then skips their 12 (l_max+1)^2 - 12 bytes. */
  float *grad_xyz, *grad_rgb, *grad_sh, *grad_opacity, *grad_scale, *grad_quaternion;
  /* optional intermediates (may be NULL): */
  float *grad_conic, *grad_uv, *grad_J, *grad_sigma, *grad_xyz_c, *grad_precompute_rgb;
} gsplat_gradients;

/* r06: the optimizer state the per-gaussian backward needs to apply the masked Adam step itself
 * (gsplat_backward_gaussians_adam).  Groups in this order: 0 xyz, 1 rgb (band 0), 2 sh, 3 opacity, 4 scale, 5 quaternion;
 * moments in GLOBAL order next to the parameters, laid out like them (quaternion moments 16-byte aligned); group 2 is
 * ignored when l_max == 0. */
typedef struct gsplat_adam_fused {
  float *exp_avg[6], *exp_avg_sq[6];
  float lr[6];                       /* per group (cuda/trainer.cu:1049-1071) */
  float b1, b2, eps, bias1, bias2;   /* include/gsplat_cuda/optimizer.cuh:9-11; bias_k = 1 - b_k^(iteration + 1) */
  float *uv_grad_accum;              /* [N] += |grad_uv| of the visible gaussians; may be NULL */
  int *grad_accum_dur;               /* [N] += 1 for the visible gaussians; may be NULL */
  int mode;                          /* 0: all six groups in the kernel.  1: band 0, opacity, scale, quaternion and the
                                      * statistics in the kernel; it stores grad_xyz and grad_precompute_rgb (`out` must hold
                                      * them) and the caller runs gsplat_optimizer_step_sh_factored and then
                                      * gsplat_optimizer_step on the xyz group alone -- the positions must not move before
                                      * the SH group has taken its directions from them.  2: all six groups, in TWO kernels:
                                      * the SH group's step in a kernel of its own in front, which reads the coefficient rows
                                      * once -- for the update and for the sums over them that the position gradient needs
                                      * (handed on in a [M,3] array of the context) -- then the per-gaussian backward with the
                                      * five small groups, which no longer reads the SH rows at all.  Same bits as modes 0, 1 */
} gsplat_adam_fused;

int gsplat_context_create(gsplat_context **out, int max_gaussians, int max_width, int max_height);
int gsplat_context_destroy(gsplat_context *ctx);
/* bytes of device memory currently held by the context */
size_t gsplat_context_bytes(const gsplat_context *ctx);

int gsplat_rasterize_image(gsplat_context *ctx, const gsplat_gaussians *gaussians, const gsplat_camera *camera,
                           const gsplat_raster_config *config, float bg_color, int l_max,
                           gsplat_forward_view *out, void *stream);

/* Consumes the state left in ctx by the last gsplat_rasterize_image.  All leaf gradients are
 * overwritten ("=" on a zeroed buffer, i.e. the state after zero_grads + backward_pass). */
int gsplat_backward_pass(gsplat_context *ctx, const gsplat_gaussians *gaussians, const gsplat_camera *camera,
                         const float *grad_image, float bg_color, int l_max, const gsplat_gradients *out,
                         void *stream);
/* r06 -- single-GPU training: gsplat_backward_gaussians and the optimizer step of TrainerImpl::optimizer_step
 * (cuda/trainer.cu:1027-1158) in ONE pass over the visible gaussians.  The kernel differentiates a gaussian and applies
 * adam_kernel's update (cuda/optimizer.cu:6-29: NaN gradient -> 0, bias-corrected moments) to its rows of all six
 * parameter groups IN PLACE -- through the pointers of `gaussians`, which this entry point writes -- and of their moments,
 * the SH group from the factored gradient (gsplat_optimizer_step_sh_factored's product), and adds |grad_uv| / 1 to the
 * densification statistics.  Call it where gsplat_backward_gaussians would be called (after gsplat_backward_render).
 * Parameters, moments and statistics come out bit-identical to gsplat_backward_gaussians (grad_sh NULL) +
 * gsplat_optimizer_step_sh_factored + gsplat_optimizer_step; what is saved is the gradients' round trip through HBM and
 * the second and third read of parameters the backward has just read (~370 of ~2030 bytes per visible gaussian at
 * l_max 3).  `out` may be NULL; when given, its arrays are filled as gsplat_backward_gaussians fills them (except grad_sh,
 * which this form never stores). */
int gsplat_backward_gaussians_adam(gsplat_context *ctx, const gsplat_gaussians *gaussians, const gsplat_camera *cam,
                                   int l_max, const gsplat_adam_fused *opt, const gsplat_gradients *out, void *stream);

/* The same backward in two calls, for hosts that overlap a gradient exchange with it (3dgs_amd/dist.py):
 * gsplat_backward_render runs the compositing backward (render_image_backward) and, if rgb_global != NULL, leaves this
 * view's dL/d(precomputed rgb) in GLOBAL gaussian order in rgb_global[N,3] (zero where culled) -- final at that point,
 * so an all-gather of it can run while gsplat_backward_gaussians executes the per-gaussian operator chain. */
int gsplat_backward_render(gsplat_context *ctx, const float *grad_image, float bg_color, float *rgb_global,
                           void *stream);
int gsplat_backward_gaussians(gsplat_context *ctx, const gsplat_gaussians *gaussians, const gsplat_camera *camera,
                              int l_max, const gsplat_gradients *out, void *stream);
/* The per-gaussian chain for the gaussians with GLOBAL index in [first_gaussian, end_gaussian) only (their compacted
 * slots are contiguous: the compaction keeps the order).  A view-sharded step calls it chunk by chunk and starts the
 * all-reduce of one chunk's twelve common columns (gsplat_pack_gradients_split_range) while the next chunk is computed;
 * the union of the chunks is exactly gsplat_backward_gaussians. */
int gsplat_backward_gaussians_range(gsplat_context *ctx, const gsplat_gaussians *gaussians, const gsplat_camera *camera,
                                    int l_max, const gsplat_gradients *out, int first_gaussian, int end_gaussian,
                                    void *stream);

/* r05 -- the view-sharded step without compacted gradient arrays and without a pack pass (3dgs_amd/dist.py, exchange
 * "split").  What the exchange sums per gaussian is common[N,12] = {grad_xyz 3, grad_opacity, grad_scale 3,
 * grad_quaternion 4, 1.0 if this view saw the gaussian} in GLOBAL gaussian order, plus (training) uv_norm[N] =
 * |grad_uv|; what it gathers is this view's g_rgb[N,3].
 *   gsplat_backward_render_split  = gsplat_backward_render that, in the pass that scatters g_rgb to rgb_global, also
 *                                   clears the rows of `common` / `uv_norm` of the gaussians this view culled;
 *   gsplat_backward_gaussians_split = the per-gaussian operator chain (cuda/trainer.cu:979-1012) whose twelve leaf
 *                                   values go straight into common[compact_to_global[j]] (and |grad_uv| into uv_norm);
 *                                   the SH and band-0 gradients are not stored at all: gsplat_optimizer_step_sh_views
 *                                   rebuilds them from the gathered g_rgb.  [first, end): a range of global indices as in
 *                                   gsplat_backward_gaussians_range.
 * Values are bit-identical to gsplat_backward_gaussians + gsplat_pack_gradients_split (+ gsplat_pack_uv_grad_norm). */
int gsplat_backward_render_split(gsplat_context *ctx, const float *grad_image, float bg_color, float *rgb_global,
                                 float *common, float *uv_norm, void *stream);
int gsplat_backward_gaussians_split(gsplat_context *ctx, const gsplat_gaussians *gaussians, const gsplat_camera *camera,
                                    int l_max, float *common, float *uv_norm, int first_gaussian, int end_gaussian,
                                    void *stream);

/* Binning route of the fused forward.  0 (default): automatic -- the LDS counting sort + per-tile depth sort, or, when
 * the previous forward had more than ~768 list entries per tile (dense real scenes) or the tile grid exceeds 16384
 * tiles, stable radix sorts on (depth bits, tile).  1 / 2 force one route.  Both produce identical lists. */
int gsplat_context_set_binning_route(gsplat_context *ctx, int route);

/* Testing / tuning hook for the forward's long-list segments (every value < 0 keeps what is set).
 * poll_budget: how often a segment's workgroup polls for the transmittance published by the workgroups in front of it
 * before it multiplies that product up itself (default 4096; 1 makes every workgroup that is not handed its value at once
 * take that path -- same values, more work).  thin_layer_blocks: layers of fewer segment workgroups run their lists'
 * segments side by side, each publishing its own transmittance product first (default 512; 0: none do, every workgroup
 * waits for the one in front).  gate: the forward splits its long lists when the previous forward's longest chain exceeded
 * gate x the work per resident workgroup (default 3, or GSPLAT_FWD_SEGMENTS_GATE; 0: always).  Results do not depend on
 * any of the three. */
int gsplat_context_set_segment_options(gsplat_context *ctx, int poll_budget, int thin_layer_blocks, float gate);

/* Measurement hook: when enabled, every stage of the two fused passes is bracketed by HIP events on the
 * caller's stream.  Stage ids: 0 project+cull+scan, 1 preprocess+scan, 2 emit + tile sort + ranges + per-tile depth sort, 3 reserved,
 * 4 compositing forward, 5 gradient-row memset, 6 compositing backward, 7 per-gaussian backward.
 * gsplat_context_get_timing synchronises the device, writes the per-stage sum of milliseconds and the number
 * of samples since timing was (re)enabled, and returns the number of stages. */
/* Render-only use (serving): forwards stop writing what only a backward or a training host reads -- Sigma, J, conic,
 * the evaluated SH colour (those four pointers of gsplat_forward_view come back NULL) and the compositing kernels'
 * block masks -- and the backward entry points answer "no forward pass recorded" until it is switched off again.
 * Image, per-pixel counts / transmittance and the sorted lists are unchanged. */
/* r05 -- zero-copy hand-over of a forward's outputs.  The thirteen arrays of the reference's ForwardPassData
 * (cuda_data.cuh:70-86: mask, uv, xyz_c, sigma, conic, J, precomputed_rgb, radius, sorted_gaussians,
 * splat_start_end_idx_by_tile_idx, image, weight_per_pixel, splats_per_pixel -- the pointers the last gsplat_forward_view
 * reported) are blocks of the library's pool; after this call the CALLER owns them and returns each with
 * gsplat_pool_free, and the context takes fresh blocks of the same sizes from the pool at its next forward (which finds
 * the ones the caller has returned in the meantime: no allocator call in the steady state of a training loop).  The
 * rasterize_image shim (include/gsplat_cuda/raster.cuh) uses it instead of thirteen device-to-device copies.  The
 * context's fused backward is unavailable for that forward afterwards (the stand-alone operators take the arrays). */
int gsplat_context_detach_forward_outputs(gsplat_context *ctx);
/* The compaction the last completed forward of `ctx` already computed for its cull mask: *mask = the mask array that
 * forward wrote (the pointer gsplat_forward_view reported, whoever owns the block now), *slots[i] = compacted slot of
 * gaussian i where mask[i] is set (its exclusive scan; the entries of culled gaussians are unspecified),
 * *compact_to_global[j] = gaussian of slot j, for N = *num_gaussians and M = *num_culled.  All of it is valid until the
 * context's NEXT forward and describes the mask's contents as that forward wrote them.  NULL pointers when there is none.
 * The drop-in compact_masked_array uses it when it is handed that very mask array (cuda/trainer.cu:941-964, 1028-1044:
 * the host compacts ~25 arrays per iteration by pass_data.d_mask): gsplat_compact_rows_ranked then needs one launch and
 * no scan of the mask. */
int gsplat_context_last_compaction(gsplat_context *ctx, const unsigned char **mask, const int **slots,
                                   const int **compact_to_global, int *num_gaussians, int *num_culled);
/* compact_masked_array with the slots given: dst[slots[i], :] <- src[i, :] for every i with mask[i] set and
 * slots[i] < dst_rows.  One launch. */
int gsplat_compact_rows_ranked(const float *src, const unsigned char *mask, const int *slots, int N, int stride,
                               float *dst, int dst_rows, void *stream);
/* dst[0..n) <- value on `stream`, asynchronously (hipMemsetAsync for 0.0f).  What the drop-in headers route the host's
 * thrust::fill_n calls on float device vectors to (cuda_data.cuh: zero_grads' twelve fills, cuda/trainer.cu:247-261, each
 * of which otherwise ends in a stream synchronisation inside thrust). */
int gsplat_fill_f32(float *dst, size_t n, float value, void *stream);
/* rows[j] <- index of the j-th set entry of mask[0..N) for j < rows_cap (the row list of a compaction; what the lazy
 * SH compaction of the drop-in headers gathers through).  No read-back. */
int gsplat_mask_selected_rows(const unsigned char *mask, int N, int *rows, int rows_cap, void *stream);
int gsplat_context_set_render_only(gsplat_context *ctx, int enabled);
/* Lean forward (training through the fused entry points): gsplat_backward_pass recomputes Sigma, J and the conic from
 * the parameters and never reads the evaluated SH colour, so a caller that does not look at those four arrays of
 * ForwardPassData (cuda_data.cuh:70-86) can switch their stores off: the four pointers of gsplat_forward_view come back
 * NULL, and so do the uncompacted `uv` / `xyz_c` (the compacted uv_selected / xyz_c_selected stay); the per-gaussian
 * forward then writes 104 instead of 176 bytes per visible gaussian and the cull 5 instead of 25 per gaussian.  Image, lists, radii and all
 * gradients are unchanged.  Off by default: the reference's own backward_pass (cuda/trainer.cu:941-1012) hands these
 * arrays to the stand-alone backward operators. */
int gsplat_context_set_lean_forward(gsplat_context *ctx, int enabled);
/* r06: how the per-gaussian forward is launched (GSPLAT_PRE_SPLIT in the environment sets the default of new contexts).
 * 0 (default): the single fused kernel.  1: two kernels one behind the other on `stream` -- the SH colour as a stream
 * of its own (SH rows through LDS as linear spans), then geometry / record / tile count with the colour read back.
 * 2: the same two kernels side by side -- the colour kernel on a low-priority stream of the context, forked behind the
 * cull and joined in front of the binning kernels; it writes the colour straight into the 48-byte records.  The reference
 * launches compute_rgb_from_sh and the covariance chain separately too (cuda/raster.cu:78-100).  Every output is
 * bit-identical in all modes; on MI355X modes 1 and 2 measure 25 and 40 us SLOWER than mode 0 at 1e6 gaussians
 * (profiles/r06_ab_preprocess_split.txt) -- they are kept as the measured answer to "split it", not as a recommendation. */
int gsplat_context_set_preprocess_split(gsplat_context *ctx, int mode);
/* What the forwards of this context did so far: out[0] forwards completed, out[1] forwards whose speculatively queued
 * tail (placement, per-tile sorts, compositing: queued before the host has seen the counts, from the previous forward's
 * figures) had to be redone because the instances outgrew the buffers or the longest list needed a sort kernel that
 * was not queued, out[2] forwards that walked the compacted slots, out[3] growths of the instance buffers, out[4]
 * compositing backwards that took their tiles heaviest first (scenes whose longest tile list was more than three times
 * the average in the forward before), out[5] compositing backwards that walked lists of more than 1488 entries as
 * segments of 496 with a workgroup each (the forward before had such a list; the forward stores a per-pixel checkpoint
 * at every segment boundary it reaches), out[6] forwards in which every segment of such a list was a workgroup of its
 * own (r06: counted once per forward, a redone tail included), out[7] / out[8] the largest stop index of any tile and the
 * sum of the tiles' largest stop indices in the last forward that published them (contexts that have seen a list beyond
 * 1488 entries).  The forward splits when the figures of the forward TWO back say max * 2048 > 3 * sum: those are the
 * newest ones the host knows to be complete without synchronising, so the decision -- and with it the bits of the image
 * -- never follows host / GPU timing (r06).  out[9] segment workgroups of split forwards whose polls for the workgroup
 * in front ran out and which multiplied the transmittance product up themselves (results identical; every one of them
 * is work done twice).  Writes min(n, 10) values and returns 10. */
int gsplat_context_get_counters(gsplat_context *ctx, long long *out, int n);
int gsplat_context_set_timing(gsplat_context *ctx, int enabled);
/* The same for a subset of the stages (bit k of stage_mask = stage k; 0 switches timing off).  Every timed stage
 * costs two event records per call, about 0.7 % of a 1 ms step each: bench.py times only stage 6 inside its timed
 * region and all stages in a separate pass. */
int gsplat_context_set_timing_stages(gsplat_context *ctx, unsigned int stage_mask);
int gsplat_context_get_timing(gsplat_context *ctx, double *stage_ms_sum, long long *stage_count, int max_stages);

/* View-sharded training support: scatter compacted per-view gradients into one global-order
 * row-major buffer packed[N, 12 + 3*n_coeffs] = [xyz3 | band0 3 | sh 3(n_coeffs-1) | opacity1 | scale3 | quat4 |
 * visible1], zero where culled, ready for one RCCL all-reduce. */
int gsplat_pack_gradients_global(gsplat_context *ctx, const gsplat_gradients *grads, int l_max, int num_gaussians,
                                 float *packed, void *stream);
int gsplat_packed_gradient_width(int l_max);

/* Smaller exchange payload for the same result.  Every SH-coefficient gradient of one view is g_rgb[3] x Y_k(dir),
 * so a rank only has to ship g_rgb: factored[N, 12 + 3*world] = [xyz3 | opacity1 | scale3 | quat4 | visible1 |
 * g_rgb of rank 0 | ... | g_rgb of rank world-1] (a rank fills only its own slot; the SUM all-reduce gathers the
 * slots).  gsplat_unpack_gradients_factored rebuilds the global-order packed[N, 12+3*n_coeffs] buffer from the
 * reduced rows, the gaussian positions and all ranks' camera positions campos_all[world,3] (device pointer).
 * gsplat_backward_pass must have been given grad_precompute_rgb. */
int gsplat_factored_gradient_width(int world_size);
int gsplat_pack_gradients_factored(gsplat_context *ctx, const gsplat_gradients *grads, int num_gaussians, int rank,
                                   int world_size, float *factored, void *stream);
int gsplat_unpack_gradients_factored(const float *xyz, const float *campos_all, const float *factored, int l_max,
                                     int num_gaussians, int world_size, float *packed, void *stream);


/* The same factorisation as two collectives with less traffic: common[N,12] = [xyz3 | opacity1 | scale3 | quat4 |
 * visible1] is SUM all-reduced, and every rank's g_rgb[N,3] (+ its camera position in row N) is all-gathered into
 * rgb_all = world blocks of rank_stride floats (block r = rank r's [N,3] g_rgb followed by campos[3]).  At 8 ranks
 * and SH degree 3 a rank moves 2*(7/8)*48 MB + 7*12 MB = 168 MB per step instead of 252 MB (one factored all-reduce)
 * or 420 MB (full rows).  rgb may be NULL when gsplat_backward_render already produced it.  The unpack may be issued in
 * two halves that write disjoint columns of `packed`: common == NULL rebuilds only the SH columns (it can run while the
 * all-reduce of `common` is still in flight), rgb_all == NULL places only the twelve reduced columns. */
int gsplat_pack_gradients_split(gsplat_context *ctx, const gsplat_gradients *grads, int num_gaussians, float *common,
                                float *rgb, void *stream);
/* Rows [first_gaussian, end_gaussian) of `common` (and of `rgb` when given) only: the chunked exchange. */
int gsplat_pack_gradients_split_range(gsplat_context *ctx, const gsplat_gradients *grads, int num_gaussians,
                                      int first_gaussian, int end_gaussian, float *common, float *rgb, void *stream);
int gsplat_unpack_gradients_split(const float *xyz, const float *common, const float *rgb_all, size_t rank_stride,
                                  int l_max, int num_gaussians, int world_size, float *packed, void *stream);

/* The per-view densification statistic in global gaussian order: uv_norm[i] = |grad_uv| of this view's backward
 * (PositionalGradientNorm, cuda/trainer.cu:1143-1148) or 0 where the view culled the gaussian.  SUM all-reduced next
 * to the gradients it feeds gsplat_optimizer_step_packed, so that density control runs on a view-sharded step.
 * grads->grad_uv must have been requested from the backward. */
int gsplat_pack_uv_grad_norm(gsplat_context *ctx, const gsplat_gradients *grads, int num_gaussians, float *uv_norm,
                             void *stream);

#ifdef __cplusplus
}
#endif
#endif /* GSPLAT_HIP_H */
