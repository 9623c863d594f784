// Third microbenchmark: relative issue cost of common VALU ops in ONE binary, interleaved (8 waves/SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
#define A8 "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
#define OP8(op) op " %0, %0, %8\n " op " %1, %1, %8\n " op " %2, %2, %8\n " op " %3, %3, %8\n " op " %4, %4, %8\n " op " %5, %5, %8\n " op " %6, %6, %8\n " op " %7, %7, %8\n"
template <int KIND>
__global__ __launch_bounds__(256) void k(float *out, int iters, float sb) {
  float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  float b = 1.0001f;
  for (int i = 0; i < iters; ++i) {
    if (KIND == 0) { REP8(asm volatile(OP8("v_mul_f32") : A8 : "v"(b));) }
    if (KIND == 1) { REP8(asm volatile(OP8("v_add_f32") : A8 : "v"(b));) }
    if (KIND == 2) { REP8(asm volatile(OP8("v_sub_f32") : A8 : "v"(b));) }
    if (KIND == 3) { REP8(asm volatile(OP8("v_min_f32") : A8 : "v"(b));) }
    if (KIND == 4) { REP8(asm volatile(OP8("v_max_f32") : A8 : "v"(b));) }
    if (KIND == 5) { REP8(asm volatile("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8\n" : A8 : "v"(b));) }
    if (KIND == 6) { REP8(asm volatile("v_fmac_f32 %0, %8, %8\n v_fmac_f32 %1, %8, %8\n v_fmac_f32 %2, %8, %8\n v_fmac_f32 %3, %8, %8\n v_fmac_f32 %4, %8, %8\n v_fmac_f32 %5, %8, %8\n v_fmac_f32 %6, %8, %8\n v_fmac_f32 %7, %8, %8\n" : A8 : "v"(b));) }
    if (KIND == 7) { REP8(asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %4\n v_mov_b32 %4, %5\n v_mov_b32 %5, %6\n v_mov_b32 %6, %7\n v_mov_b32 %7, %0\n" : A8);) }
    if (KIND == 8) { REP8(asm volatile("v_cmp_gt_f32 vcc, %0, %8\n v_cmp_gt_f32 vcc, %1, %8\n v_cmp_gt_f32 vcc, %2, %8\n v_cmp_gt_f32 vcc, %3, %8\n v_cmp_gt_f32 vcc, %4, %8\n v_cmp_gt_f32 vcc, %5, %8\n v_cmp_gt_f32 vcc, %6, %8\n v_cmp_gt_f32 vcc, %7, %8\n" : A8 : "v"(b) : "vcc");) }
    if (KIND == 9) { REP8(asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n" : A8 : "v"(b));) }
    if (KIND == 10) { REP8(asm volatile(OP8("v_mul_f32") : A8 : "s"(sb));) }           // SGPR operand
    if (KIND == 11) { REP8(asm volatile("v_mul_f32 %0, 0.5, %0\n v_mul_f32 %1, 0.5, %1\n v_mul_f32 %2, 0.5, %2\n v_mul_f32 %3, 0.5, %3\n v_mul_f32 %4, 0.5, %4\n v_mul_f32 %5, 0.5, %5\n v_mul_f32 %6, 0.5, %6\n v_mul_f32 %7, 0.5, %7\n" : A8);) }  // inline constant
    if (KIND == 12) { REP8(asm volatile("v_mul_f32 %0, 0x3fb8aa3b, %0\n v_mul_f32 %1, 0x3fb8aa3b, %1\n v_mul_f32 %2, 0x3fb8aa3b, %2\n v_mul_f32 %3, 0x3fb8aa3b, %3\n v_mul_f32 %4, 0x3fb8aa3b, %4\n v_mul_f32 %5, 0x3fb8aa3b, %5\n v_mul_f32 %6, 0x3fb8aa3b, %6\n v_mul_f32 %7, 0x3fb8aa3b, %7\n" : A8);) }  // 32-bit literal
    if (KIND == 13) { REP8(asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n" : A8);) }
    if (KIND == 14) { REP8(asm volatile("v_med3_f32 %0, %0, %8, %8\n v_med3_f32 %1, %1, %8, %8\n v_med3_f32 %2, %2, %8, %8\n v_med3_f32 %3, %3, %8, %8\n v_med3_f32 %4, %4, %8, %8\n v_med3_f32 %5, %5, %8, %8\n v_med3_f32 %6, %6, %8, %8\n v_med3_f32 %7, %7, %8, %8\n" : A8 : "v"(b));) }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
static float *g_out;
template <int KIND> float run1() {
  const int blocks = 256 * 8, iters = 1000;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0);
  k<KIND><<<blocks, 256>>>(g_out, iters, 1.0001f);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  return ms;
}
int main() {
  (void)hipMalloc(&g_out, 256 * 8 * 256 * 4);
  const char *names[] = {"v_mul_f32", "v_add_f32", "v_sub_f32", "v_min_f32", "v_max_f32", "v_fma_f32", "v_fmac_f32", "v_mov_b32",
                         "v_cmp_gt_f32", "v_cndmask_b32 (vcc)", "v_mul_f32 sgpr src", "v_mul_f32 inline const", "v_mul_f32 literal", "v_exp_f32", "v_med3_f32"};
  float best[15];
  for (int i = 0; i < 15; ++i) best[i] = 1e9f;
  for (int rep = 0; rep < 4; ++rep) {
    float t[15] = {run1<0>(), run1<1>(), run1<2>(), run1<3>(), run1<4>(), run1<5>(), run1<6>(), run1<7>(), run1<8>(), run1<9>(), run1<10>(), run1<11>(), run1<12>(), run1<13>(), run1<14>()};
    if (rep) for (int i = 0; i < 15; ++i) best[i] = t[i] < best[i] ? t[i] : best[i];
  }
  for (int i = 0; i < 15; ++i) printf("%-24s %.3f ms   %.2fx v_mul\n", names[i], best[i], best[i] / best[0]);
  return 0;
}
