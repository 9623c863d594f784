// optimizer.cuh -- source-compatible shim for the reference's Adam entry point
// (reference: include/gsplat_cuda/optimizer.cuh:9-29), forwarding to libgsplat_hip.so.
#pragma once

#include "hip_compat.h"

inline constexpr float B1 = 0.9f;
inline constexpr float B2 = 0.999f;
inline constexpr float EPS = 1e-8f;

inline void adam_step(float *params, float *const param_grads, float *exp_avg, float *exp_avg_sq, const float lr,
                      const float b1, const float b2, const float eps, const float bias1, const float bias2,
                      const int N, const int S, cudaStream_t stream = 0) {
  gsplat_shim::require_ok(
      gsplat_adam_step(params, param_grads, exp_avg, exp_avg_sq, lr, b1, b2, eps, bias1, bias2, N, S, stream),
      "adam_step");
}
