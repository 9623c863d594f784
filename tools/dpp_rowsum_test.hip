// Unit test for gs::row_sum9 (row-level transposed reduction on masked DPP adds): run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include "../3dgs_amd/csrc/gs_render.h"

__global__ void k(const float *in, float *out) {
  const int lane = threadIdx.x;
  float v[9];
  for (int i = 0; i < 9; ++i) v[i] = in[i * 64 + lane];
  out[lane] = gs::row_sum9(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], v[8]);
}

int main() {
  float h[9 * 64], *d_in, *d_out, o[64];
  for (int i = 0; i < 9 * 64; ++i) h[i] = (float)((i * 37) % 101) * 0.25f - 7.0f;
  hipMalloc(&d_in, sizeof(h)); hipMalloc(&d_out, sizeof(o));
  hipMemcpy(d_in, h, sizeof(h), hipMemcpyHostToDevice);
  k<<<1, 64>>>(d_in, d_out);
  hipMemcpy(o, d_out, sizeof(o), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int lane = 0; lane < 64; ++lane) {
    const int row = lane >> 4, idx = gs::row_sum9_index(lane);
    float ref = 0.0f;
    for (int j = 0; j < 16; ++j) ref += h[idx * 64 + row * 16 + j];
    if (gs::row_sum9_active(lane) && std::fabs(ref - o[lane]) > 1e-4f) {
      std::printf("lane %d idx %d got %g want %g\n", lane, idx, o[lane], ref);
      ++bad;
    }
  }
  std::printf(bad ? "row_sum9: %d mismatches\n" : "row_sum9: ok\n", bad);
  return bad != 0;
}
