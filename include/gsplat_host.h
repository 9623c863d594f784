/* gsplat_host.h -- C ABI of libgsplat_host.so: the dependency-free dataset plumbing either side of the rasterizer
 * (SURVEY.md section 8f, row f3).  Pure host code (g++), no Eigen / yaml-cpp / nanoflann.
 *
 * Replaces, for a host written in C or C++ (or any FFI):
 *   ReadCamerasBinary / ReadImagesBinary / ReadPoints3DBinary   src/colmap.cpp:41-196  (include/dataloader/colmap.hpp)
 *   Image::CamPos, computeMaxDiagonal                           src/colmap.cpp:35-39, 198-236
 *   parseConfig                                                 src/utils.cpp:17-87
 *   save_ply                                                    src/utils.cpp:89-175
 * Gaussians::Initialize (src/gaussian.cpp:38-104) runs on the GPU: gsplat_initialize_gaussians in gsplat_hip.h.
 *
 * Readers follow a two-call pattern: call with the output arrays NULL to get the counts, then with arrays of at
 * least that capacity.  Every function returns GSPLAT_HOST_OK or a negative code; gsplat_host_last_error() has text.
 */
#ifndef GSPLAT_HOST_H
#define GSPLAT_HOST_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define GSPLAT_HOST_OK 0
#define GSPLAT_HOST_ERR_NULL -1
#define GSPLAT_HOST_ERR_INVALID_ARG -3
#define GSPLAT_HOST_ERR_CAPACITY -6
#define GSPLAT_HOST_ERR_IO -7          /* file cannot be opened / written */
#define GSPLAT_HOST_ERR_PARSE -8       /* truncated or malformed content, missing config key */
#define GSPLAT_HOST_ERR_UNSUPPORTED -9 /* camera model other than SIMPLE_PINHOLE / PINHOLE (src/colmap.cpp:70-73) */

const char *gsplat_host_last_error(void);

typedef struct gsplat_colmap_camera { /* Camera, include/dataloader/colmap.hpp */
  int id, model_id, width, height, num_params;
  double params[12];
} gsplat_colmap_camera;

typedef struct gsplat_colmap_image { /* Image */
  int id, camera_id;
  double qvec[4]; /* w x y z */
  double tvec[3];
  char name[1024]; /* img_root_dir + "images[_<downsample>]/" + file name */
  uint64_t first_point2d, num_points2d; /* slice of the xys / point3d_ids arrays */
} gsplat_colmap_image;

typedef struct gsplat_colmap_point3d { /* Point3D */
  uint64_t id;
  double xyz[3];
  unsigned char rgb[3];
  double error;
  uint64_t first_track, track_length; /* slice of the image_ids / point2d_idxs arrays */
} gsplat_colmap_point3d;

const char *gsplat_colmap_model_name(int model_id); /* NULL for unknown ids */

/* params, width and height are divided by downsample_factor as the reference does (src/colmap.cpp:89-94) */
int gsplat_colmap_read_cameras(const char *path, int downsample_factor, gsplat_colmap_camera *out, size_t capacity,
                               size_t *count);
int gsplat_colmap_read_images(const char *path, const char *img_root_dir, int downsample_factor,
                              gsplat_colmap_image *out, size_t capacity, size_t *count, double *xys /* [P,2] */,
                              int64_t *point3d_ids /* [P] */, size_t points_capacity, size_t *points_count);
int gsplat_colmap_read_points3d(const char *path, gsplat_colmap_point3d *out, size_t capacity, size_t *count,
                                int *image_ids, int *point2d_idxs, size_t track_capacity, size_t *track_count);

/* world-space camera centre -R^T t  (Image::CamPos) and the row-major rotation matrix of qvec */
void gsplat_qvec_to_rotmat(const double qvec[4], double rot[9]);
void gsplat_camera_position(const double qvec[4], const double tvec[3], double out[3]);
/* largest distance of a camera centre from the mean centre (computeMaxDiagonal); 0 for n == 0 */
int gsplat_scene_extent(const double *qvecs /* [n,4] */, const double *tvecs /* [n,3] */, size_t n, double *out);

typedef struct gsplat_config { /* ConfigParameters, include/gsplat/utils.hpp:10-70 */
  char dataset_path[1024], output_dir[1024];
  int downsample_factor, print_interval, num_iters;
  double ssim_frac;
  int test_eval_interval, test_split_ratio;
  double initial_opacity;
  int initial_scale_num_neighbors;
  double initial_scale_factor, max_initial_scale;
  double near_thresh, mh_dist;
  int cull_mask_padding;
  double base_lr, xyz_lr_multiplier_init, xyz_lr_multiplier_final, quat_lr_multiplier, scale_lr_multiplier,
      opacity_lr_multiplier, rgb_lr_multiplier, sh_lr_multiplier;
  int use_background, use_background_end;
  int reset_opacity_interval;
  double reset_opacity_value;
  int reset_opacity_start, reset_opacity_end;
  int use_sh_precompute, max_sh_band, add_sh_band_interval;
  int use_split, use_clone, use_delete;
  int adaptive_control_start, adaptive_control_end, adaptive_control_interval, max_gaussians;
  double delete_opacity_threshold, uv_grad_threshold, split_scale_factor;
} gsplat_config;

/* Flat "key: value" YAML subset (comments, quoted strings, true/false, numbers): all the reference's config files use.
 * Every key of ConfigParameters is required (missing -> GSPLAT_HOST_ERR_PARSE), unknown keys are ignored. */
int gsplat_parse_config(const char *path, gsplat_config *out);

/* Binary little-endian PLY in the reference's property order (x y z nx ny nz f_dc_0..2 f_rest_* opacity scale_0..2
 * rot_0..3).  quaternion is [N,4] (w,x,y,z) as on the device; like the reference the file stores x,y,z,w.
 * sh may be NULL when sh_floats == 0. */
int gsplat_save_ply(const char *path, size_t num_gaussians, int sh_floats, const float *xyz, const float *rgb,
                    const float *sh, const float *opacity, const float *scale, const float *quaternion);

#ifdef __cplusplus
}
#endif
#endif /* GSPLAT_HOST_H */
