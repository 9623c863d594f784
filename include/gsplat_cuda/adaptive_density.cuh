// adaptive_density.cuh -- source-compatible shim for the reference's clone / split operators
// (reference: include/gsplat_cuda/adaptive_density.cuh:9-57), forwarding to libgsplat_hip.so.
#pragma once

#include <ctime>

#include "hip_compat.h"

inline void clone_gaussians(const int N, const int num_sh_coef, const bool *mask, const int *write_ids,
                            const float *xyz_in, const float *rgb_in, const float *op_in, const float *scale_in,
                            const float *quat_in, const float *sh_in, float *xyz_out, float *rgb_out, float *op_out,
                            float *scale_out, float *quat_out, float *sh_out, cudaStream_t stream = 0) {
  static_assert(sizeof(bool) == 1, "mask is one byte per gaussian");
  gsplat_shim::require_ok(gsplat_clone_gaussians(N, num_sh_coef, reinterpret_cast<const unsigned char *>(mask), write_ids,
                                                 xyz_in, rgb_in, op_in, scale_in, quat_in, sh_in, xyz_out, rgb_out, op_out,
                                                 scale_out, quat_out, sh_out, stream),
                          "clone_gaussians");
}

inline void split_gaussians(const int N, const float scale_factor, const int num_sh_coef, const bool *mask,
                            const int *write_ids, const float *xyz_in, const float *rgb_in, const float *op_in,
                            const float *scale_in, const float *quat_in, const float *sh_in, float *xyz_out,
                            float *rgb_out, float *op_out, float *scale_out, float *quat_out, float *sh_out,
                            cudaStream_t stream = 0) {
  // the reference seeds its generator from the wall clock on every call (cuda/adaptive_density.cu:199)
  gsplat_shim::require_ok(
      gsplat_split_gaussians(N, scale_factor, num_sh_coef, reinterpret_cast<const unsigned char *>(mask), write_ids, xyz_in,
                             rgb_in, op_in, scale_in, quat_in, sh_in, xyz_out, rgb_out, op_out, scale_out, quat_out, sh_out,
                             (unsigned long long)std::time(nullptr), stream),
      "split_gaussians");
}
