// LDS atomic issue cost on gfx950: ds_add_f32 / ds_add_u32 / ds_write_b32 with N active lanes per instruction
// (conflict-free addresses), 8 waves per SIMD resident.  Prints cycles per wave instruction per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s\n", hipGetErrorString(e)); return 1; } } while (0)

template <int kMode>
__global__ __launch_bounds__(256) void k(float *out, int iters, unsigned long long active) {
  __shared__ float s[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) s[i] = 0.0f;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool on = (active >> lane) & 1ull;
  float *p = &s[wave * 1024 + lane];
  if (on) {
    for (int it = 0; it < iters; ++it) {
      float *q = p + ((it & 7) << 6);
      if (kMode == 0) atomicAdd(q, 1.0f);
      if (kMode == 1) atomicAdd(reinterpret_cast<unsigned int *>(q), 1u);
      if (kMode == 2) *reinterpret_cast<volatile float *>(q) = (float)it;
      if (kMode == 3) atomicAdd(reinterpret_cast<unsigned long long *>(s) + (q - s) / 2 + (lane & 1) * 0, (unsigned long long)it);
      if (kMode == 4) atomicAdd(reinterpret_cast<double *>(s) + (q - s) / 2, 1.0);
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = s[0] + s[1024];
}

int main() {
  float *d; CK(hipMalloc(&d, 1 << 20));
  const int blocks = 256 * 8, iters = 4000;
  const char *names[5] = {"ds_add_f32", "ds_add_u32", "ds_write_b32", "ds_add_u64", "ds_add_f64"};
  const unsigned long long pats[5] = {~0ull, 0x0000FFFFFFFFFFFFull & 0xFFFFFFFFFull /*36*/, 0x1111111111111111ull /*16*/,
                                      0x0001000100010001ull /*4*/, 1ull};
  const int cnt[5] = {64, 36, 16, 4, 1};
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int m = 0; m < 5; ++m)
    for (int p = 0; p < 5; ++p) {
      for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(a));
        if (m == 0) k<0><<<blocks, 256>>>(d, iters, pats[p]);
        if (m == 1) k<1><<<blocks, 256>>>(d, iters, pats[p]);
        if (m == 2) k<2><<<blocks, 256>>>(d, iters, pats[p]);
        if (m == 3) k<3><<<blocks, 256>>>(d, iters, pats[p]);
        if (m == 4) k<4><<<blocks, 256>>>(d, iters, pats[p]);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
      }
      float ms; CK(hipEventElapsedTime(&ms, a, b));
      // wave instructions per CU = blocks/256 CUs * 4 waves * iters
      const double per_cu = (double)blocks / 256.0 * 4.0 * iters;
      printf("%-13s %2d lanes  %.3f ms  %.2f cycles per wave instruction per CU (2.4 GHz)\n", names[m], cnt[p], ms,
             ms * 1e-3 * 2.4e9 / per_cu);
    }
  return 0;
}
