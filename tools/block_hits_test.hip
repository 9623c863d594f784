// Property test for gs::block_hits: the mask must contain every 4x4 block that holds a pixel with alpha >= 1/255
// (conservative), for random centres, sizes, anisotropies, orientations and opacities; also reports how tight it is.
// Run on the GPU box (tests/test_rowsum_gpu.py).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>
#include "../3dgs_amd/csrc/gs_render.h"

__global__ void k(const float *in, int n, unsigned int *out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float *p = in + 6 * i;  // u, v, a, b, c, logit
  gs::SplatRec s = gs::make_record(p[0], p[1], p[2], p[3], p[4], p[5], 0.5f, 0.5f, 0.5f);
  const unsigned int mask = gs::block_hits(s, 32.0f, 48.0f);
  gs::stage_record(s);
  unsigned int exact = 0u;
  for (int y = 0; y < 16; ++y)
    for (int x = 0; x < 16; ++x) {
      const float al = fminf(gs::kAlphaMax, gs::staged_alpha(s.r0.z, s.r0.w, s.r1.x, s.r1.y, s.r0.x - (32.0f + x), s.r0.y - (48.0f + y)));
      if (al >= gs::kAlphaMin) exact |= 1u << (4 * (y >> 2) + (x >> 2));
    }
  out[2 * i] = mask;
  out[2 * i + 1] = exact;
}

int main() {
  const int n = 1 << 18;
  std::mt19937 rng(7);
  std::uniform_real_distribution<float> U(0.0f, 1.0f);
  std::vector<float> h(6 * (size_t)n);
  for (int i = 0; i < n; ++i) {
    const float th = 3.14159265f * U(rng);
    const float s1 = std::exp(std::log(0.05f) + U(rng) * std::log(40000.0f)), s2 = s1 * std::exp(-U(rng) * std::log(1000.0f));  // 0.05 .. 2000 px, up to 1000:1
    const float cs = std::cos(th), sn = std::sin(th);
    // covariance R diag(s1^2, s2^2) R^T; conic = inverse
    const float A = cs * cs * s1 * s1 + sn * sn * s2 * s2, B = cs * sn * (s1 * s1 - s2 * s2), C = sn * sn * s1 * s1 + cs * cs * s2 * s2;
    const float det = A * C - B * B;
    h[6 * i + 0] = 32.0f + (U(rng) * 3.0f - 1.0f) * 16.0f * (i % 8 == 0 ? 20.0f : 1.0f);  // centre near the tile, some far away
    h[6 * i + 1] = 48.0f + (U(rng) * 3.0f - 1.0f) * 16.0f * (i % 8 == 0 ? 20.0f : 1.0f);
    h[6 * i + 2] = C / det; h[6 * i + 3] = -B / det; h[6 * i + 4] = A / det;
    h[6 * i + 5] = -6.0f + 14.0f * U(rng);  // logit of the opacity
  }
  float *d_in; unsigned int *d_out;
  hipMalloc(&d_in, h.size() * sizeof(float)); hipMalloc(&d_out, 2 * (size_t)n * sizeof(unsigned int));
  hipMemcpy(d_in, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice);
  k<<<n / 256, 256>>>(d_in, n, d_out);
  std::vector<unsigned int> o(2 * (size_t)n);
  hipMemcpy(o.data(), d_out, o.size() * sizeof(unsigned int), hipMemcpyDeviceToHost);
  long long missed = 0, admitted = 0, needed = 0;
  for (int i = 0; i < n; ++i) {
    const unsigned int mask = o[2 * i], exact = o[2 * i + 1];
    if (exact & ~mask) {
      if (missed < 5) std::printf("gaussian %d: mask %04x misses %04x (u %g v %g conic %g %g %g logit %g)\n", i, mask, exact & ~mask,
                                  h[6 * i], h[6 * i + 1], h[6 * i + 2], h[6 * i + 3], h[6 * i + 4], h[6 * i + 5]);
      ++missed;
    }
    admitted += __builtin_popcount(mask); needed += __builtin_popcount(exact);
  }
  std::printf("blocks admitted %lld, blocks with a valid pixel %lld (%.3f x)\n", admitted, needed, (double)admitted / (double)needed);
  std::printf(missed ? "block_hits: %lld gaussians with a missed block\n" : "block_hits: ok\n", missed);
  return missed != 0;
}
