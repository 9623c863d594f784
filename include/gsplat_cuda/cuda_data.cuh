// cuda_data.cuh -- source-compatible shim for the reference's device-buffer structs and compaction templates
// (reference: include/gsplat_cuda/cuda_data.cuh:11-167, constructors cuda/data.cu:9-107).  The structs keep the
// reference's member names and capacities (they ARE the interface its trainer and tests use); rocThrust provides
// thrust::device_vector.  compact_masked_array / scatter_masked_array forward to libgsplat_hip.so instead of the
// reference's thrust::copy_if / scatter compositions.
#pragma once

#include <thrust/device_vector.h>

#include <cstdio>
#include <cstdlib>
#include <exception>

#include "hip_compat.h"

namespace gsplat_shim {
template <typename F> inline void alloc_or_exit(const char *what, F &&f) {
  try {
    f();
  } catch (const std::exception &e) {  // cuda/data.cu: print and exit on allocation failure
    std::fprintf(stderr, "CUDA Memory Allocation Error (%s): %s\n", what, e.what());
    std::exit(EXIT_FAILURE);
  }
}
}  // namespace gsplat_shim

struct GaussianParameters {  // SH is allocated at its full capacity (15 coefficients) and packed at the current band
  thrust::device_vector<float> d_xyz, d_rgb, d_sh, d_opacity, d_scale, d_quaternion;
  explicit GaussianParameters(size_t n) {
    gsplat_shim::alloc_or_exit("GaussianParameters", [&] {
      d_xyz.resize(n * 3); d_rgb.resize(n * 3); d_sh.resize(n * 45); d_opacity.resize(n); d_scale.resize(n * 3);
      d_quaternion.resize(n * 4);
    });
  }
};

struct OptimizerParameters {  // Adam moments, zero-initialised
  thrust::device_vector<float> m_grad_xyz, m_grad_rgb, m_grad_sh, m_grad_opacity, m_grad_scale, m_grad_quaternion;
  thrust::device_vector<float> v_grad_xyz, v_grad_rgb, v_grad_sh, v_grad_opacity, v_grad_scale, v_grad_quaternion;
  explicit OptimizerParameters(size_t n) {
    gsplat_shim::alloc_or_exit("OptimizerParameters", [&] {
      m_grad_xyz.assign(n * 3, 0.f); m_grad_rgb.assign(n * 3, 0.f); m_grad_sh.assign(n * 45, 0.f);
      m_grad_opacity.assign(n, 0.f); m_grad_scale.assign(n * 3, 0.f); m_grad_quaternion.assign(n * 4, 0.f);
      v_grad_xyz.assign(n * 3, 0.f); v_grad_rgb.assign(n * 3, 0.f); v_grad_sh.assign(n * 45, 0.f);
      v_grad_opacity.assign(n, 0.f); v_grad_scale.assign(n * 3, 0.f); v_grad_quaternion.assign(n * 4, 0.f);
    });
  }
};

struct GaussianGradients {  // compacted order; the second line are the intermediates of the backward chain
  thrust::device_vector<float> d_grad_xyz, d_grad_rgb, d_grad_sh, d_grad_opacity, d_grad_scale, d_grad_quaternion;
  thrust::device_vector<float> d_grad_conic, d_grad_uv, d_grad_J, d_grad_sigma, d_grad_xyz_c, d_grad_precompute_rgb;
  explicit GaussianGradients(size_t n) {
    gsplat_shim::alloc_or_exit("GaussianGradients", [&] {
      d_grad_xyz.resize(n * 3); d_grad_rgb.resize(n * 3); d_grad_sh.resize(n * 45); d_grad_opacity.resize(n);
      d_grad_scale.resize(n * 3); d_grad_quaternion.resize(n * 4);
      d_grad_conic.resize(n * 3); d_grad_uv.resize(n * 2); d_grad_J.resize(n * 6); d_grad_sigma.resize(n * 6);
      d_grad_xyz_c.resize(n * 3); d_grad_precompute_rgb.resize(n * 3);
    });
  }
};

struct GradientAccumulators {  // density-control statistics, zero-initialised
  thrust::device_vector<float> d_uv_grad_accum;
  thrust::device_vector<int> d_grad_accum_dur;
  explicit GradientAccumulators(size_t n) {
    gsplat_shim::alloc_or_exit("GradientAccumulators", [&] { d_uv_grad_accum.assign(n, 0.f); d_grad_accum_dur.assign(n, 0); });
  }
};

struct CameraParameters {  // row-major 4x4 view and projection matrices
  thrust::device_vector<float> d_view, d_proj;
  CameraParameters() {
    gsplat_shim::alloc_or_exit("CudaDataManager", [&] { d_view.resize(16); d_proj.resize(16); });
  }
};

struct CudaDataManager {  // owns every persistent device buffer of a training run
  const size_t max_gaussians;
  GaussianParameters gaussians;
  OptimizerParameters optimizer;
  GaussianGradients gradients;
  GradientAccumulators accumulators;
  CameraParameters camera;
  explicit CudaDataManager(size_t n) : max_gaussians(n), gaussians(n), optimizer(n), gradients(n), accumulators(n), camera() {}
};

struct ForwardPassData {  // per-view outputs of rasterize_image, saved for the backward pass
  size_t num_culled = 0;
  thrust::device_vector<float> d_sigma, d_conic, d_J, d_precomputed_rgb;  // [num_culled, 6 | 3 | 6 | 3]
  thrust::device_vector<float> d_uv, d_xyz_c;                             // [N, 2 | 3], uncompacted
  thrust::device_vector<bool> d_mask;                                     // [N]
  thrust::device_vector<float4> d_radius;                                 // [num_culled]
  thrust::device_vector<int> d_sorted_gaussians, d_splat_start_end_idx_by_tile_idx;
  thrust::device_vector<float> d_image_buffer, d_weight_per_pixel;
  thrust::device_vector<int> d_splats_per_pixel;
};

// compact_masked_array<STRIDE>(source, mask, num_culled): the rows of `source` whose mask entry is set, in order.
template <int STRIDE, typename T, typename MaskType>
thrust::device_vector<T> compact_masked_array(const thrust::device_vector<T> &d_source,
                                              const thrust::device_vector<MaskType> &d_mask, int num_culled) {
  static_assert(sizeof(T) == 4, "rows are made of 4-byte elements (float / int)");
  static_assert(sizeof(MaskType) == 1, "the mask is one byte per row (bool)");
  thrust::device_vector<T> d_selected((size_t)num_culled * STRIDE);
  int selected = 0;
  gsplat_shim::require_ok(
      gsplat_compact_masked_array(reinterpret_cast<const float *>(thrust::raw_pointer_cast(d_source.data())),
                                  reinterpret_cast<const unsigned char *>(thrust::raw_pointer_cast(d_mask.data())),
                                  (int)d_mask.size(), STRIDE,
                                  reinterpret_cast<float *>(thrust::raw_pointer_cast(d_selected.data())), &selected, 0),
      "compact_masked_array");
  if (selected != num_culled) {
    std::fprintf(stderr, "compact_masked_array: num_culled = %d but the mask selects %d rows\n", num_culled, selected);
    std::exit(EXIT_FAILURE);
  }
  return d_selected;
}

// scatter_masked_array<STRIDE>(compacted, mask, destination): the inverse; rows whose mask entry is clear are untouched.
template <int STRIDE, typename T, typename MaskType>
void scatter_masked_array(const thrust::device_vector<T> &d_compacted, const thrust::device_vector<MaskType> &d_mask,
                          thrust::device_vector<T> &d_destination) {
  static_assert(sizeof(T) == 4 && sizeof(MaskType) == 1, "4-byte elements, one-byte mask");
  if (d_compacted.size() / STRIDE == 0) return;
  gsplat_shim::require_ok(
      gsplat_scatter_masked_array(reinterpret_cast<const float *>(thrust::raw_pointer_cast(d_compacted.data())),
                                  reinterpret_cast<const unsigned char *>(thrust::raw_pointer_cast(d_mask.data())),
                                  (int)d_mask.size(), STRIDE,
                                  reinterpret_cast<float *>(thrust::raw_pointer_cast(d_destination.data())), 0),
      "scatter_masked_array");
}
