// Second microbenchmark: selects / compares / cross-lane reads (cycles per wave64 instruction per SIMD, 4 waves/SIMD)
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
#define A8 "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
template <int KIND>
__global__ __launch_bounds__(256) void k(float *out, int iters, unsigned long long msk) {
  float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  float b = 1.0001f;
  unsigned long long sm = msk;
  for (int i = 0; i < iters; ++i) {
    if (KIND == 0) {  // v_cndmask_b32_e64 with an SGPR-pair mask
      REP8(asm volatile("v_cndmask_b32_e64 %0, %0, %8, %9\n v_cndmask_b32_e64 %1, %1, %8, %9\n v_cndmask_b32_e64 %2, %2, %8, %9\n v_cndmask_b32_e64 %3, %3, %8, %9\n"
                        "v_cndmask_b32_e64 %4, %4, %8, %9\n v_cndmask_b32_e64 %5, %5, %8, %9\n v_cndmask_b32_e64 %6, %6, %8, %9\n v_cndmask_b32_e64 %7, %7, %8, %9\n"
                        : A8 : "v"(b), "s"(sm));)
    } else if (KIND == 1) {  // v_cmp_gt_f32 -> vcc then v_cndmask (pair)
      REP8(asm volatile("v_cmp_gt_f32 vcc, %0, %8\n v_cndmask_b32 %0, %0, %8, vcc\n v_cmp_gt_f32 vcc, %1, %8\n v_cndmask_b32 %1, %1, %8, vcc\n"
                        "v_cmp_gt_f32 vcc, %2, %8\n v_cndmask_b32 %2, %2, %8, vcc\n v_cmp_gt_f32 vcc, %3, %8\n v_cndmask_b32 %3, %3, %8, vcc\n"
                        : A8 : "v"(b) : "vcc");)
    } else if (KIND == 2) {  // v_cmp only (to sgpr pair)
      REP8(asm volatile("v_cmp_gt_f32 vcc, %0, %8\n v_cmp_gt_f32 vcc, %1, %8\n v_cmp_gt_f32 vcc, %2, %8\n v_cmp_gt_f32 vcc, %3, %8\n"
                        "v_cmp_gt_f32 vcc, %4, %8\n v_cmp_gt_f32 vcc, %5, %8\n v_cmp_gt_f32 vcc, %6, %8\n v_cmp_gt_f32 vcc, %7, %8\n"
                        : A8 : "v"(b) : "vcc");)
    } else if (KIND == 3) {  // v_min_f32
      REP8(asm volatile("v_min_f32 %0, %0, %8\n v_min_f32 %1, %1, %8\n v_min_f32 %2, %2, %8\n v_min_f32 %3, %3, %8\n"
                        "v_min_f32 %4, %4, %8\n v_min_f32 %5, %5, %8\n v_min_f32 %6, %6, %8\n v_min_f32 %7, %7, %8\n"
                        : A8 : "v"(b));)
    } else if (KIND == 4) {  // v_readlane_b32
      int s0;
      REP8(asm volatile("v_readlane_b32 %8, %0, 3\n v_readlane_b32 %8, %1, 4\n v_readlane_b32 %8, %2, 5\n v_readlane_b32 %8, %3, 6\n"
                        "v_readlane_b32 %8, %4, 7\n v_readlane_b32 %8, %5, 8\n v_readlane_b32 %8, %6, 9\n v_readlane_b32 %8, %7, 10\n"
                        : A8, "=s"(s0));)
    } else if (KIND == 5) {  // v_fma with an SGPR operand
      REP8(asm volatile("v_fma_f32 %0, %0, %8, %0\n v_fma_f32 %1, %1, %8, %1\n v_fma_f32 %2, %2, %8, %2\n v_fma_f32 %3, %3, %8, %3\n"
                        "v_fma_f32 %4, %4, %8, %4\n v_fma_f32 %5, %5, %8, %5\n v_fma_f32 %6, %6, %8, %6\n v_fma_f32 %7, %7, %8, %7\n"
                        : A8 : "s"((float)1.0001f));)
    } else if (KIND == 6) {  // v_add_f32 row_mirror dpp
      REP8(asm volatile("s_nop 1\n v_add_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %1, %1 row_mirror row_mask:0xf bank_mask:0xf\n"
                        "v_add_f32_dpp %2, %2, %2 row_mirror row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %3, %3, %3 row_mirror row_mask:0xf bank_mask:0xf\n"
                        "v_add_f32_dpp %4, %4, %4 row_mirror row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %5, %5, %5 row_mirror row_mask:0xf bank_mask:0xf\n"
                        "v_add_f32_dpp %6, %6, %6 row_mirror row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %7, %7, %7 row_mirror row_mask:0xf bank_mask:0xf\n"
                        : A8);)
    } else if (KIND == 7) {  // v_mov_b32
      REP8(asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %4\n"
                        "v_mov_b32 %4, %5\n v_mov_b32 %5, %6\n v_mov_b32 %6, %7\n v_mov_b32 %7, %0\n"
                        : A8);)
    } else if (KIND == 8) {  // exec-masked update: s_mov exec + v_mul under mask + restore
      REP8(asm volatile("s_mov_b64 s[10:11], exec\n s_and_b64 exec, exec, %8\n v_mul_f32 %0, %0, %9\n v_mul_f32 %1, %1, %9\n v_mul_f32 %2, %2, %9\n v_mul_f32 %3, %3, %9\n s_mov_b64 exec, s[10:11]\n"
                        "s_mov_b64 s[10:11], exec\n s_and_b64 exec, exec, %8\n v_mul_f32 %4, %4, %9\n v_mul_f32 %5, %5, %9\n v_mul_f32 %6, %6, %9\n v_mul_f32 %7, %7, %9\n s_mov_b64 exec, s[10:11]\n"
                        : A8 : "s"(sm), "v"(b) : "s10", "s11");)
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
template <int KIND>
void run(const char *name, double instr_per_rep) {
  hipDeviceProp_t prop;
  (void)hipGetDeviceProperties(&prop, 0);
  const int bpc = 4, blocks = prop.multiProcessorCount * bpc, iters = 2000;
  float *out;
  (void)hipMalloc(&out, (size_t)blocks * 256 * 4);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<KIND><<<blocks, 256>>>(out, 10, 0x5555555555555555ull);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  k<KIND><<<blocks, 256>>>(out, iters, 0x5555555555555555ull);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double n = (double)bpc * iters * 8 * instr_per_rep;
  printf("%-34s %.3f ms -> %.2f cycles per instruction per SIMD (at 2.4 GHz nominal)\n", name, ms, ms * 1e-3 * 2.4e9 / n);
  (void)hipFree(out);
}
int main() {
  run<0>("v_cndmask_b32_e64 (sgpr mask)", 8);
  run<1>("v_cmp_gt_f32 + v_cndmask (vcc)", 8);
  run<2>("v_cmp_gt_f32 -> vcc", 8);
  run<3>("v_min_f32", 8);
  run<4>("v_readlane_b32", 8);
  run<5>("v_fma_f32 with sgpr operand", 8);
  run<6>("v_add_f32_dpp row_mirror", 8);
  run<7>("v_mov_b32", 8);
  run<8>("exec-masked 4x v_mul (+3 salu)", 8);
  return 0;
}
