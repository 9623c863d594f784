// gs_rows.h -- array-of-rows data between global memory and LDS (shared by the fused per-gaussian backward in
// gs_fused.hip and the stand-alone spherical-harmonics backward in gs_pergaussian.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace gs {

// ---- staging of array-of-rows data through LDS.  A lane reading ITS row of kRest floats (180 B at SH degree 3)
// touches 64 different cache lines per wave instruction and runs at the cache's tag rate, not at HBM rate; the
// wave's 64 rows as one linear span touch 8 lines per instruction.
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));  // a 16-byte access at 4-byte alignment

// A wave's `rows` (<= 64) consecutive rows of kRest floats between global memory and its LDS staging area, as one
// linear span of 16-byte accesses.  A full wave issues all its loads before the first LDS store (11.25 per lane at
// degree 3), so the whole 11.5 KB is in flight at once.
template <int kRest>
__device__ __forceinline__ void rows_to_lds(const float *__restrict__ src, float *wsh, int rows, int lane) {
  constexpr int kVec = 16 * kRest, kFull = kVec / 64, kTail = kVec % 64;  // float4s of a full wave's span
  if (rows == 64) {
    float4 v[kFull + 1];
#pragma unroll
    for (int t = 0; t < kFull; ++t) v[t] = __builtin_bit_cast(float4, *reinterpret_cast<const f4u *>(src + 4 * (lane + 64 * t)));
    if (kTail > 0 && lane < kTail) v[kFull] = __builtin_bit_cast(float4, *reinterpret_cast<const f4u *>(src + 4 * (lane + 64 * kFull)));
#pragma unroll
    for (int t = 0; t < kFull; ++t) *reinterpret_cast<float4 *>(wsh + 4 * (lane + 64 * t)) = v[t];
    if (kTail > 0 && lane < kTail) *reinterpret_cast<float4 *>(wsh + 4 * (lane + 64 * kFull)) = v[kFull];
  } else {
    const int total = rows * kRest;
    for (int e = lane * 4; e + 3 < total; e += 256)
      *reinterpret_cast<float4 *>(wsh + e) = __builtin_bit_cast(float4, *reinterpret_cast<const f4u *>(src + e));
    for (int e = (total & ~3) + lane; e < total; e += 64) wsh[e] = src[e];
  }
}
template <int kRest>
__device__ __forceinline__ void rows_from_lds(float *__restrict__ dst, const float *wsh, int rows, int lane) {
  const int total = rows * kRest;
  for (int e = lane * 4; e + 3 < total; e += 256)
    *reinterpret_cast<f4u *>(dst + e) = __builtin_bit_cast(f4u, *reinterpret_cast<const float4 *>(wsh + e));
  for (int e = (total & ~3) + lane; e < total; e += 64) dst[e] = wsh[e];
}


}  // namespace gs
