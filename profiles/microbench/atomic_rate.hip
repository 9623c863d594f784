// Microbenchmark: integer atomics on a small table of counters (tile histogram / cursor pattern of the binning).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void hist_noret(const int *__restrict__ tile, int n, int per, int *cnt) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  for (int k = 0; k < per; ++k) atomicAdd(&cnt[tile[i * per + k]], 1);
}
__global__ void cursor_ret(const int *__restrict__ tile, int n, int per, int *cnt, int *out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  for (int k = 0; k < per; ++k) {
    const int t = tile[i * per + k];
    const int pos = atomicAdd(&cnt[t], 1);
    out[t * 1024 + (pos & 1023)] = i;
  }
}
int main() {
  const int n = 1000000, per = 4, T = 8160;
  std::vector<int> h((size_t)n * per);
  unsigned s = 12345;
  for (auto &v : h) { s = s * 1664525u + 1013904223u; v = (s >> 8) % T; }
  int *tile, *cnt, *out;
  (void)hipMalloc(&tile, h.size() * 4); (void)hipMalloc(&cnt, T * 4); (void)hipMalloc(&out, (size_t)T * 1024 * 4);
  (void)hipMemcpy(tile, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    float ms;
    (void)hipMemset(cnt, 0, T * 4);
    (void)hipEventRecord(e0);
    hist_noret<<<(n + 255) / 256, 256>>>(tile, n, per, cnt);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1);
    printf("no-return int atomics : %d adds on %d counters: %.1f us  (%.2f G atomics/s)\n", n * per, T, ms * 1e3, n * per / ms / 1e6);
    (void)hipMemset(cnt, 0, T * 4);
    (void)hipEventRecord(e0);
    cursor_ret<<<(n + 255) / 256, 256>>>(tile, n, per, cnt, out);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1);
    printf("returning int atomics : %d adds + scatter store: %.1f us  (%.2f G atomics/s)\n", n * per, ms * 1e3, n * per / ms / 1e6);
  }
  return 0;
}
