// hip_compat.h -- lets a host translation unit written against the reference's headers
// (which include <cuda_runtime.h> and use cudaStream_t / float3 / float4) compile with hipcc.
// Not a CUDA shim for kernels: the only things aliased are the two types that appear in the
// operator signatures.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#include "../gsplat_hip.h"

typedef hipStream_t cudaStream_t;

namespace gsplat_shim {
// The reference's operators do not return a status: a bad pointer prints to stderr and exits
// (cuda/checks.cuh:17-38).  The shims keep that contract on top of the status-returning C ABI.
inline void require_ok(int status, const char *op) {
  if (status != GSPLAT_OK) {
    std::fprintf(stderr, "Assertion failed in %s: %s\n", op, gsplat_last_error());
    std::exit(EXIT_FAILURE);
  }
}
}  // namespace gsplat_shim
