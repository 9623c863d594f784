// Do isolated 8-byte stores get cheaper when the addresses of all concurrently running workgroups fall into a small
// region (so that partially written 32-byte sectors could merge in an XCD's L2 before they leave it)?
// 3.5 M stores of 8 bytes at pseudo-random 8-byte slots of a region of R bytes (R = 1 MB .. 32 MB), every slot written
// once per pass when R = 28 MB.  Prints time per launch; run under rocprofv3 --pmc WRITE_SIZE for the bytes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void scatter(unsigned long long *buf, unsigned long long slots, unsigned long long n, unsigned long long mul) {
  const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  // a bijection on [0, slots) when slots is a power of two and mul is odd: every slot of the region is written
  const unsigned long long pos = (i * mul + 12345ull) & (slots - 1);
  buf[pos] = i;
}
__global__ void linear(unsigned long long *buf, unsigned long long n) {
  const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) buf[i] = i;
}
int main() {
  const unsigned long long n = 4ull << 20;  // 4 Mi stores of 8 bytes = 32 MiB of payload
  unsigned long long *buf;
  hipMalloc(&buf, 64ull << 20);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (unsigned long long region = 1ull << 20; region <= 32ull << 20; region <<= 1) {
    const unsigned long long slots = region / 8;
    for (int rep = 0; rep < 3; ++rep) scatter<<<(unsigned)(n / 256), 256>>>(buf, slots, n, 2654435761ull);
    hipEventRecord(a);
    for (int rep = 0; rep < 20; ++rep) scatter<<<(unsigned)(n / 256), 256>>>(buf, slots, n, 2654435761ull);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("random 8-byte stores into a %2llu MiB region: %.1f us per 4 Mi stores\n", region >> 20, ms / 20 * 1e3);
  }
  hipEventRecord(a);
  for (int rep = 0; rep < 20; ++rep) linear<<<(unsigned)(n / 256), 256>>>(buf, n);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  printf("linear 8-byte stores (32 MiB): %.1f us\n", ms / 20 * 1e3);
  return 0;
}
