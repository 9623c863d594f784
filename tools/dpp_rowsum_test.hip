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

// row_moments9: aT and gp per lane, weights = arbitrary lane "constants"; every active lane against a host sum
__global__ void km(const float *in, float *out) {
  const int lane = threadIdx.x;
  const float aT = in[lane], gp = in[64 + lane], cx = in[128 + lane], cy = in[192 + lane];
  const float g0 = in[256 + lane], g1 = in[320 + lane], g2 = in[384 + lane];
  const gs::RowWeights w = gs::make_row_weights(lane, cx, cy, g0, g1, g2);
  out[lane] = gs::row_moments9(aT, gp, w);
}

// row_moments9q (r04: two butterfly stages folded into the products): cx / cy are the REAL pixel offsets of the lane
// (the quad's four lanes are four pixels in a row -- make_quad_weights derives the neighbours' cx from that)
__global__ void kq(const float *in, float *out) {
  const int lane = threadIdx.x;
  const float aT = in[lane], gp = in[64 + lane], cx = in[128 + lane], cy = in[192 + lane];
  const float g0 = in[256 + lane], g1 = in[320 + lane], g2 = in[384 + lane];
  const gs::QuadWeights w = gs::make_quad_weights(lane, cx, cy, g0, g1, g2);
  out[lane] = gs::row_moments9q(aT, gp, w);
}

// row_moments9r (r04: cy is a constant inside a quad, three of the moments come from the quad-partials of S_1 and S_cx);
// out[64 + lane]: the LDS address the block computes in its wait state (off * 5 + acc_lane)
__global__ void kr(const float *in, float *out) {
  const int lane = threadIdx.x;
  const float aT = in[lane], gp = in[64 + lane], cx = in[128 + lane], cy = in[192 + lane];
  const float g0 = in[256 + lane], g1 = in[320 + lane], g2 = in[384 + lane];
  const gs::RowsWeights w = gs::make_rows_weights(lane, cx, cy, g0, g1, g2);
  unsigned int addr;
  out[lane] = gs::row_moments9r(aT, gp, w, 16u * (unsigned int)(lane >> 4), 1000u + (unsigned int)lane, addr);
  out[64 + lane] = (float)addr;
}

static int test_moments(const float *h, float *d_in, float *d_out, int quad = 0) {
  float o[128];
  if (quad == 2) kr<<<1, 64>>>(d_in, d_out);
  else if (quad) kq<<<1, 64>>>(d_in, d_out);
  else km<<<1, 64>>>(d_in, d_out);
  hipMemcpy(o, d_out, sizeof(o), hipMemcpyDeviceToHost);
  int bad = 0, active = 0;
  for (int lane = 0; lane < 64; ++lane) {
    const int row = lane >> 4, idx = quad == 2 ? gs::row_moments9r_index(lane)
                                                : quad ? gs::row_moments9q_index(lane) : gs::row_moments9_index(lane);
    if (quad == 2 && o[64 + lane] != (float)(80 * row + 1000 + lane)) {
      std::printf("row_moments9r lane %d: address %g want %d\n", lane, o[64 + lane], 80 * row + 1000 + lane);
      ++bad;
    }
    if (idx < 0) continue;
    ++active;
    double ref = 0.0;
    for (int j = 0; j < 16; ++j) {
      const int l = row * 16 + j;
      const double aT = h[l], gp = h[64 + l], cx = h[128 + l], cy = h[192 + l];
      const double term[9] = {aT * h[256 + l], aT * h[320 + l], aT * h[384 + l], gp, gp * cx, gp * cy, gp * cx * cx,
                              gp * cx * cy, gp * cy * cy};
      ref += term[idx];
    }
    if (std::fabs(ref - o[lane]) > 1e-3 * (1.0 + std::fabs(ref))) {
      std::printf("moments lane %d idx %d got %g want %g\n", lane, idx, o[lane], ref);
      ++bad;
    }
  }
  if (active != 36) { std::printf("row_moments9: %d active lanes, want 36\n", active); ++bad; }
  if (quad == 2) std::printf(bad ? "row_moments9r: %d mismatches\n" : "row_moments9r: ok\n", bad);
  else if (quad) std::printf(bad ? "row_moments9q: %d mismatches\n" : "row_moments9q: ok\n", bad);
  else std::printf(bad ? "row_moments9: %d mismatches\n" : "row_moments9: ok\n", bad);
  return bad;
}

int main() {
  float h[9 * 64], *d_in, *d_out, o[64];  // (d_out: 128 floats, the r form also returns its address)
  for (int i = 0; i < 9 * 64; ++i) h[i] = (float)((i * 37) % 101) * 0.25f - 7.0f;
  hipMalloc(&d_in, sizeof(h)); hipMalloc(&d_out, 2 * sizeof(o));
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
  bad += test_moments(h, d_in, d_out);
  // the quad form needs a real pixel grid in (cx, cy): lane j of a row = pixel (j & 3, j >> 2) of a 4x4 block
  for (int lane = 0; lane < 64; ++lane) {
    const int row = lane >> 4, j = lane & 15;
    h[128 + lane] = (float)((row & 1) * 4 + (j & 3)) - 7.5f;
    h[192 + lane] = (float)((row >> 1) * 4 + (j >> 2)) - 7.5f;
  }
  hipMemcpy(d_in, h, sizeof(h), hipMemcpyHostToDevice);
  bad += test_moments(h, d_in, d_out, 1);
  bad += test_moments(h, d_in, d_out, 2);
  return bad != 0;
}
