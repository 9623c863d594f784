// gs_loss.hip -- "next" rows f1 / f2 of SURVEY.md section 8: the fused L1 + SSIM loss that produces the
// grad_image the rasterizer backward consumes, the PSNR metric, and the element-wise Adam step.
//
// Semantics: fused_loss / compute_psnr (cuda/loss.cu:58-525) and adam_step (cuda/optimizer.cu:6-44) of the
// reference: 11-tap separable Gaussian window, clamped borders for the statistics, zero padding for the adjoint
// convolution, gradient scaled by 1/(H*W*3), NaN gradients treated as 0 by Adam.
//
// Structure: one 256-thread workgroup per 16x16 pixel tile; the 26x26 halo tile of both images (all three
// interleaved channels) is loaded once into LDS with coalesced reads, the horizontal pass writes a 26x16 strip per
// channel, the vertical pass finishes in registers.  The loss is reduced per wave on DPP and added to one of 256
// spread counters (a single hot atomic would serialise 32k adds), summed by the host when it asks for the value.
#include "gs_common.h"
#include "gs_render.h"

namespace {

constexpr int kT = 16, kHalo = 5, kS = kT + 2 * kHalo;  // 26
constexpr int kSpread = 256;

// 11-tap window (cuda/loss.cu:12-17); a constexpr table so that the fully unrolled taps become literal operands
// (a v_fma with an SGPR operand issues 1.7x slower than one with a literal on this chip, profiles/microbench)
constexpr float cGauss[11] = {0.001028380123898387f,  0.0075987582094967365f, 0.036000773310661316f,
                              0.10936068743467331f,   0.21300552785396576f,   0.26601171493530273f,
                              0.21300552785396576f,   0.10936068743467331f,   0.036000773310661316f,
                              0.0075987582094967365f, 0.001028380123898387f};

__device__ __forceinline__ float wave_sum(float v) {
  v = gs::row_sum(v);
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}

__global__ __launch_bounds__(256) void loss_forward_kernel(int H, int W, float ssim_weight,
                                                           const float *__restrict__ pred,
                                                           const float *__restrict__ gt, float *__restrict__ acc,
                                                           float *__restrict__ dm_mu, float *__restrict__ dm_s1,
                                                           float *__restrict__ dm_s12) {
  __shared__ float sT[kS * kS * 6];       // [y][x][pred rgb | gt rgb]
  __shared__ float sH[kS * kT * 5];       // horizontal pass of one channel: [y][x][5 stats]
  const int tid = threadIdx.x, lx = tid & 15, ly = tid >> 4;
  const int x0 = blockIdx.x * kT, y0 = blockIdx.y * kT;
  const float C1 = (0.01f * 1.0f) * (0.01f * 1.0f), C2 = (0.03f * 1.0f) * (0.03f * 1.0f);
  for (int t = tid; t < kS * kS; t += 256) {
    const int sy = t / kS, sx = t % kS;
    const int gy = min(max(y0 + sy - kHalo, 0), H - 1), gx = min(max(x0 + sx - kHalo, 0), W - 1);
    const size_t g = ((size_t)gy * W + gx) * 3;
    sT[t * 6 + 0] = pred[g]; sT[t * 6 + 1] = pred[g + 1]; sT[t * 6 + 2] = pred[g + 2];
    sT[t * 6 + 3] = gt[g]; sT[t * 6 + 4] = gt[g + 1]; sT[t * 6 + 5] = gt[g + 2];
  }
  __syncthreads();
  const int px = x0 + lx, py = y0 + ly;
  const bool inside = px < W && py < H;
  float loss = 0.0f;
  for (int c = 0; c < 3; ++c) {
    for (int o = tid; o < kS * kT; o += 256) {  // horizontal pass: 26 rows x 16 columns
      const int ry = o / kT, rx = o % kT + kHalo;
      float sX = 0, sX2 = 0, sY = 0, sY2 = 0, sXY = 0;
#pragma unroll
      for (int d = 1; d <= kHalo; ++d) {
        const float w = cGauss[kHalo - d];
        const float Xl = sT[(ry * kS + rx - d) * 6 + c], Yl = sT[(ry * kS + rx - d) * 6 + 3 + c];
        const float Xr = sT[(ry * kS + rx + d) * 6 + c], Yr = sT[(ry * kS + rx + d) * 6 + 3 + c];
        sX += (Xl + Xr) * w; sX2 += (Xl * Xl + Xr * Xr) * w; sY += (Yl + Yr) * w; sY2 += (Yl * Yl + Yr * Yr) * w;
        sXY += (Xl * Yl + Xr * Yr) * w;
      }
      const float wc = cGauss[kHalo], Xc = sT[(ry * kS + rx) * 6 + c], Yc = sT[(ry * kS + rx) * 6 + 3 + c];
      sX += Xc * wc; sX2 += Xc * Xc * wc; sY += Yc * wc; sY2 += Yc * Yc * wc; sXY += Xc * Yc * wc;
      float *h = &sH[o * 5];
      h[0] = sX; h[1] = sX2; h[2] = sY; h[3] = sY2; h[4] = sXY;
    }
    __syncthreads();
    if (inside) {  // vertical pass + SSIM
      float o0 = 0, o1 = 0, o2 = 0, o3 = 0, o4 = 0;
      const int cy = ly + kHalo;
#pragma unroll
      for (int d = 1; d <= kHalo; ++d) {
        const float w = cGauss[kHalo - d];
        const float *t = &sH[((cy - d) * kT + lx) * 5], *b = &sH[((cy + d) * kT + lx) * 5];
        o0 += (t[0] + b[0]) * w; o1 += (t[1] + b[1]) * w; o2 += (t[2] + b[2]) * w; o3 += (t[3] + b[3]) * w;
        o4 += (t[4] + b[4]) * w;
      }
      const float wc = cGauss[kHalo];
      const float *ct = &sH[(cy * kT + lx) * 5];
      o0 += ct[0] * wc; o1 += ct[1] * wc; o2 += ct[2] * wc; o3 += ct[3] * wc; o4 += ct[4] * wc;
      const float mu1 = o0, mu2 = o2, mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2;
      const float s1 = o1 - mu1_sq, s2 = o3 - mu2_sq, s12 = o4 - mu1 * mu2;
      const float A = mu1_sq + mu2_sq + C1, B = s1 + s2 + C2, Cc = 2.f * mu1 * mu2 + C1, D = 2.f * s12 + C2;
      // the reference's six divisions (cuda/loss.cu:214-236) share two reciprocals: 1/A and 1/B
      const float iA = __builtin_amdgcn_rcpf(A), iB = __builtin_amdgcn_rcpf(B), iAB = iA * iB;
      const float ssim = Cc * D * iAB;
      const int ti = ((ly + kHalo) * kS + lx + kHalo) * 6;
      const float l1 = fabsf(sT[ti + c] - sT[ti + 3 + c]);
      loss += (1.0f - ssim_weight) * l1 + ssim_weight * (1.0f - ssim);
      const float two_mu1_ssim = 2.f * mu1 * ssim;
      const float d_mu1 = 2.f * mu2 * (D - Cc) * iAB - two_mu1_ssim * iA + two_mu1_ssim * iB;
      const size_t id = ((size_t)py * W + px) * 3 + c;
      dm_mu[id] = -ssim_weight * d_mu1;
      dm_s1[id] = ssim_weight * (ssim * iB);
      dm_s12[id] = -ssim_weight * (2.f * Cc * iAB);
    }
    __syncthreads();
  }
  loss = wave_sum(loss);
  if ((tid & 63) == 0) atomicAdd(&acc[(blockIdx.y * gridDim.x + blockIdx.x) * 4 + (tid >> 6) & (kSpread - 1)], loss);
}

__global__ __launch_bounds__(256) void loss_backward_kernel(int H, int W, float ssim_weight,
                                                            const float *__restrict__ pred,
                                                            const float *__restrict__ gt,
                                                            const float *__restrict__ dm_mu,
                                                            const float *__restrict__ dm_s1,
                                                            const float *__restrict__ dm_s12,
                                                            float *__restrict__ image_grad) {
  __shared__ float sD[kS * kS * 9];   // [y][x][channel][3 maps], zero outside the image
  __shared__ float sV[kS * kT * 3];   // horizontal pass of one channel
  const int tid = threadIdx.x, lx = tid & 15, ly = tid >> 4;
  const int x0 = blockIdx.x * kT, y0 = blockIdx.y * kT;
  for (int t = tid; t < kS * kS; t += 256) {
    const int sy = t / kS, sx = t % kS;
    const int gy = y0 + sy - kHalo, gx = x0 + sx - kHalo;
    const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;
    const size_t g = in ? ((size_t)gy * W + gx) * 3 : 0;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      sD[t * 9 + c * 3 + 0] = in ? dm_mu[g + c] : 0.0f;
      sD[t * 9 + c * 3 + 1] = in ? dm_s1[g + c] : 0.0f;
      sD[t * 9 + c * 3 + 2] = in ? dm_s12[g + c] : 0.0f;
    }
  }
  __syncthreads();
  const int px = x0 + lx, py = y0 + ly;
  const bool inside = px < W && py < H;
  const float grad_scale = 1.0f / (float)(H * W * 3);
  for (int c = 0; c < 3; ++c) {
    for (int o = tid; o < kS * kT; o += 256) {
      const int ry = o / kT, rx = o % kT + kHalo;
      float a0 = 0, a1 = 0, a2 = 0;
#pragma unroll
      for (int d = 1; d <= kHalo; ++d) {
        const float w = cGauss[kHalo - d];
        const float *l = &sD[(ry * kS + rx - d) * 9 + c * 3], *r = &sD[(ry * kS + rx + d) * 9 + c * 3];
        a0 += (l[0] + r[0]) * w; a1 += (l[1] + r[1]) * w; a2 += (l[2] + r[2]) * w;
      }
      const float *m = &sD[(ry * kS + rx) * 9 + c * 3];
      const float wc = cGauss[kHalo];
      a0 += m[0] * wc; a1 += m[1] * wc; a2 += m[2] * wc;
      sV[o * 3] = a0; sV[o * 3 + 1] = a1; sV[o * 3 + 2] = a2;
    }
    __syncthreads();
    if (inside) {
      float s0 = 0, s1 = 0, s2 = 0;
      const int cy = ly + kHalo;
#pragma unroll
      for (int d = 1; d <= kHalo; ++d) {
        const float w = cGauss[kHalo - d];
        const float *t = &sV[((cy - d) * kT + lx) * 3], *b = &sV[((cy + d) * kT + lx) * 3];
        s0 += (t[0] + b[0]) * w; s1 += (t[1] + b[1]) * w; s2 += (t[2] + b[2]) * w;
      }
      const float *ct = &sV[(cy * kT + lx) * 3];
      const float wc = cGauss[kHalo];
      s0 += ct[0] * wc; s1 += ct[1] * wc; s2 += ct[2] * wc;
      const size_t id = ((size_t)py * W + px) * 3 + c;
      const float p1 = pred[id], p2 = gt[id];
      const float ssim_g = s0 + (2.f * p1) * s1 + p2 * s2;
      const float l1_g = (1.0f - ssim_weight) * ((p1 > p2) ? 1.0f : -1.0f);
      image_grad[id] = (ssim_g + l1_g) * grad_scale;
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void mse_kernel(long long n, const float *__restrict__ pred,
                                                  const float *__restrict__ gt, float *__restrict__ acc) {
  float s = 0.0f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const float d = pred[i] - gt[i];
    s += d * d;
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) atomicAdd(&acc[(blockIdx.x * 4 + (threadIdx.x >> 6)) & (kSpread - 1)], s);
}

__global__ __launch_bounds__(256) void adam_kernel(long long n, float *__restrict__ param,
                                                   const float *__restrict__ grad, float *__restrict__ m,
                                                   float *__restrict__ v, float lr, float b1, float b2, float eps,
                                                   float bias1, float bias2) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float g = grad[i];
  if (g != g) g = 0.0f;
  const float mi = b1 * m[i] + (1.0f - b1) * g;
  const float vi = b2 * v[i] + (1.0f - b2) * g * g;
  const float m_hat = mi / bias1, v_hat = vi / bias2;
  param[i] += -lr * m_hat / (sqrtf(v_hat) + eps);
  m[i] = mi;
  v[i] = vi;
}

// ---- masked in-place optimizer step (replaces the ~35 compact/scatter calls of TrainerImpl::optimizer_step,
// cuda/trainer.cu:1027-1158: only gaussians visible in the view are touched, everything else keeps its moments).
struct GroupTable {
  gsplat_adam_group g[GSPLAT_MAX_ADAM_GROUPS];
  int start[GSPLAT_MAX_ADAM_GROUPS + 1];      // prefix of strides
  int blk_start[GSPLAT_MAX_ADAM_GROUPS + 1];  // prefix of workgroups per group (filled by group_blocks)
  int n;
};

__device__ __forceinline__ void adam_update(float *param, float *m, float *v, float g, float lr, float b1, float b2,
                                            float eps, float bias1, float bias2) {
  if (g != g) g = 0.0f;
  const float mi = b1 * *m + (1.0f - b1) * g;
  const float vi = b2 * *v + (1.0f - b2) * g * g;
  const float m_hat = mi / bias1, v_hat = vi / bias2;
  *param += -lr * m_hat / (sqrtf(v_hat) + eps);
  *m = mi;
  *v = vi;
}

// kPacked = false: grads are compacted [M,stride] arrays and c2g maps compacted -> global rows.
// kPacked = true : grads are rows of the packed global layout; a row is live when its last column (views that saw
//                  the gaussian) is positive.
// Workgroups are dealt to the parameter groups in order (blk_start[k] .. blk_start[k+1]), so the group -- its
// pointers, stride and learning rate -- is uniform per workgroup and lives in scalar registers; thread i of a group
// owns element i of that group's [rows, stride] gradient array, i.e. the gradient read is perfectly coalesced.
template <bool kPacked>
__global__ __launch_bounds__(256) void optimizer_step_kernel(int rows, GroupTable t, const int *__restrict__ c2g,
                                                             const float *__restrict__ packed, int width, float b1,
                                                             float b2, float eps, float bias1, float bias2,
                                                             const float *__restrict__ grad_uv,
                                                             float *__restrict__ uv_accum, int *__restrict__ accum_dur) {
  int k = 0;
#pragma unroll
  for (int q = 1; q < GSPLAT_MAX_ADAM_GROUPS; ++q) k += (q < t.n && (int)blockIdx.x >= t.blk_start[q]) ? 1 : 0;
  const gsplat_adam_group G = t.g[k];
  const unsigned int i = (blockIdx.x - (unsigned int)t.blk_start[k]) * 256u + threadIdx.x;
  const unsigned int stride = (unsigned int)G.stride;
  if (i >= (unsigned int)rows * stride) return;
  const unsigned int r = i / stride, c = i - r * stride;
  long long row;
  if (kPacked) {
    row = r;
    if (!(packed[(size_t)r * width + width - 1] > 0.0f)) return;
  } else {
    row = c2g[r];
  }
  const float g = kPacked ? packed[(size_t)r * width + G.packed_column + c] : G.grad[i];
  const long long o = row * stride + c;
  adam_update(&G.param[o], &G.exp_avg[o], &G.exp_avg_sq[o], g, G.lr, b1, b2, eps, bias1, bias2);
  if (k == 0 && c == 0) {  // densification statistics, cuda/trainer.cu:1136-1157
    if (kPacked) {  // grad_uv = the all-reduced sum over the step's views of |grad_uv| [N]; one count per view
      if (uv_accum) uv_accum[row] += grad_uv[r];
      if (accum_dur) accum_dur[row] += (int)packed[(size_t)r * width + width - 1];
    } else {
      if (uv_accum) {
        const float u = grad_uv[2 * r], v = grad_uv[2 * r + 1];
        uv_accum[row] += sqrtf(u * u + v * v);
      }
      if (accum_dur) accum_dur[row] += 1;
    }
  }
}

int build_group_table(const gsplat_adam_group *groups, int n_groups, bool packed, GroupTable *t, const char *fn) {
  if (!groups || n_groups < 1 || n_groups > GSPLAT_MAX_ADAM_GROUPS) {
    gs::set_error("%s: invalid argument: need 1..%d parameter groups", fn, GSPLAT_MAX_ADAM_GROUPS);
    return GSPLAT_ERR_INVALID_ARG;
  }
  t->n = n_groups;
  t->start[0] = 0;
  for (int k = 0; k < n_groups; ++k) {
    const gsplat_adam_group &g = groups[k];
    if (g.stride < 1 || (packed && g.packed_column < 0)) {
      gs::set_error("%s: invalid argument: group %d has a bad stride/column", fn, k);
      return GSPLAT_ERR_INVALID_ARG;
    }
    int st;
    if ((st = gs::check_device_ptr(g.param, "group.param", fn)) || (st = gs::check_device_ptr(g.exp_avg, "group.exp_avg", fn)) ||
        (st = gs::check_device_ptr(g.exp_avg_sq, "group.exp_avg_sq", fn)))
      return st;
    if (!packed && (st = gs::check_device_ptr(g.grad, "group.grad", fn))) return st;
    t->g[k] = g;
    t->start[k + 1] = t->start[k] + g.stride;
  }
  for (int k = n_groups; k < GSPLAT_MAX_ADAM_GROUPS; ++k) t->start[k + 1] = t->start[n_groups];
  return GSPLAT_OK;
}

// workgroups of 256 threads per group for `rows` rows; returns the total
static long long group_blocks(GroupTable *t, long long rows) {
  long long run = 0;
  for (int k = 0; k <= GSPLAT_MAX_ADAM_GROUPS; ++k) {
    t->blk_start[k] = (int)run;
    if (k < t->n) run += (rows * t->g[k].stride + 255) / 256;
  }
  return run;
}

int read_spread_sum(float *d_acc, hipStream_t st, double *out) {
  int rc = gs::host_words().ensure();
  if (rc) return rc;
  static float *h = nullptr;
  if (!h && hipHostMalloc((void **)&h, kSpread * sizeof(float), hipHostMallocDefault) != hipSuccess) {
    gs::set_error("hipHostMalloc failed");
    return GSPLAT_ERR_HIP;
  }
  GS_HIP(hipMemcpyAsync(h, d_acc, kSpread * sizeof(float), hipMemcpyDeviceToHost, st));
  GS_HIP(hipStreamSynchronize(st));
  double s = 0.0;
  for (int k = 0; k < kSpread; ++k) s += (double)h[k];
  *out = s;
  return GSPLAT_OK;
}

}  // namespace

extern "C" {

int gsplat_fused_loss(const float *predicted_data, const float *gt_data, int rows, int cols, float ssim_weight,
                      float *image_grad, float *loss_out, void *stream) {
  GS_REQUIRE_DEV(predicted_data); GS_REQUIRE_DEV(gt_data); GS_REQUIRE_DEV(image_grad);
  GS_REQUIRE(rows > 0 && cols > 0, "image size must be positive");
  hipStream_t st = (hipStream_t)stream;
  const size_t bytes = (size_t)rows * cols * 3 * sizeof(float);
  gs::DeviceBuffer &mu = gs::scratch(gs::SCR_LOSS_MU), &s1 = gs::scratch(gs::SCR_LOSS_S1),
                   &s12 = gs::scratch(gs::SCR_LOSS_S12), &acc = gs::scratch(gs::SCR_LOSS_ACC);
  int rc;
  if ((rc = mu.reserve(bytes)) || (rc = s1.reserve(bytes)) || (rc = s12.reserve(bytes)) ||
      (rc = acc.reserve(kSpread * sizeof(float))))
    return rc;
  GS_HIP(hipMemsetAsync(acc.ptr, 0, kSpread * sizeof(float), st));
  const dim3 grid((cols + kT - 1) / kT, (rows + kT - 1) / kT), block(256);
  loss_forward_kernel<<<grid, block, 0, st>>>(rows, cols, ssim_weight, predicted_data, gt_data, acc.as<float>(),
                                              mu.as<float>(), s1.as<float>(), s12.as<float>());
  GS_LAUNCH_CHECK();
  loss_backward_kernel<<<grid, block, 0, st>>>(rows, cols, ssim_weight, predicted_data, gt_data, mu.as<float>(),
                                               s1.as<float>(), s12.as<float>(), image_grad);
  GS_LAUNCH_CHECK();
  if (loss_out) {  // the reference returns the value, i.e. blocks (cuda/loss.cu:468-470)
    double total;
    if ((rc = read_spread_sum(acc.as<float>(), st, &total))) return rc;
    *loss_out = (float)(total / (double)((size_t)rows * cols * 3));
  }
  return GSPLAT_OK;
}

int gsplat_compute_psnr(const float *predicted_data, const float *gt_data, int rows, int cols, float *psnr_out,
                        void *stream) {
  GS_REQUIRE_DEV(predicted_data); GS_REQUIRE_DEV(gt_data);
  GS_REQUIRE(psnr_out != nullptr, "psnr_out is null");
  GS_REQUIRE(rows > 0 && cols > 0, "image size must be positive");
  hipStream_t st = (hipStream_t)stream;
  gs::DeviceBuffer &acc = gs::scratch(gs::SCR_LOSS_ACC);
  int rc = acc.reserve(kSpread * sizeof(float));
  if (rc) return rc;
  GS_HIP(hipMemsetAsync(acc.ptr, 0, kSpread * sizeof(float), st));
  const long long n = (long long)rows * cols * 3;
  mse_kernel<<<2048, 256, 0, st>>>(n, predicted_data, gt_data, acc.as<float>());
  GS_LAUNCH_CHECK();
  double total;
  if ((rc = read_spread_sum(acc.as<float>(), st, &total))) return rc;
  const float mse = (float)(total / (double)n);
  *psnr_out = mse == 0.0f ? 100.0f : 10.0f * log10f(1.0f / mse);  // cuda/loss.cu:520-524
  return GSPLAT_OK;
}

int gsplat_adam_step(float *params, const float *param_grads, float *exp_avg, float *exp_avg_sq, float lr, float b1,
                     float b2, float eps, float bias1, float bias2, int N, int S, void *stream) {
  GS_REQUIRE_DEV(params); GS_REQUIRE_DEV(param_grads); GS_REQUIRE_DEV(exp_avg); GS_REQUIRE_DEV(exp_avg_sq);
  GS_REQUIRE(N >= 0 && S >= 0, "negative size");
  const long long n = (long long)N * S;
  if (n == 0) return GSPLAT_OK;
  adam_kernel<<<gs::div_up(n, 256), 256, 0, (hipStream_t)stream>>>(n, params, param_grads, exp_avg, exp_avg_sq, lr, b1,
                                                                  b2, eps, bias1, bias2);
  GS_LAUNCH_CHECK();
  return GSPLAT_OK;
}

int gsplat_optimizer_step(const int *compact_to_global, int num_culled, const gsplat_adam_group *groups, int n_groups,
                          float b1, float b2, float eps, float bias1, float bias2, const float *grad_uv,
                          float *uv_grad_accum, int *grad_accum_dur, void *stream) {
  GS_REQUIRE(num_culled >= 0, "negative gaussian count");
  GroupTable t;
  int rc = build_group_table(groups, n_groups, false, &t, __func__);
  if (rc) return rc;
  if (num_culled == 0) return GSPLAT_OK;
  GS_REQUIRE_DEV(compact_to_global);
  if (uv_grad_accum) { GS_REQUIRE_DEV(uv_grad_accum); GS_REQUIRE_DEV(grad_uv); }
  if (grad_accum_dur) GS_REQUIRE_DEV(grad_accum_dur);
  GS_REQUIRE((long long)num_culled * t.start[t.n] < (1ll << 31), "too many parameters for one launch");
  const long long blocks = group_blocks(&t, num_culled);
  optimizer_step_kernel<false><<<(unsigned int)blocks, 256, 0, (hipStream_t)stream>>>(
      num_culled, t, compact_to_global, nullptr, 0, b1, b2, eps, bias1, bias2, grad_uv, uv_grad_accum, grad_accum_dur);
  GS_LAUNCH_CHECK();
  return GSPLAT_OK;
}

int gsplat_optimizer_step_packed(const float *packed, int num_gaussians, int width, const gsplat_adam_group *groups,
                                 int n_groups, float b1, float b2, float eps, float bias1, float bias2,
                                 const float *uv_norm_sum, float *uv_grad_accum, int *grad_accum_dur, void *stream) {
  GS_REQUIRE(num_gaussians >= 0, "negative gaussian count");
  GroupTable t;
  int rc = build_group_table(groups, n_groups, true, &t, __func__);
  if (rc) return rc;
  for (int k = 0; k < n_groups; ++k)
    GS_REQUIRE(groups[k].packed_column + groups[k].stride <= width - 1, "group columns exceed the packed row");
  if (num_gaussians == 0) return GSPLAT_OK;
  GS_REQUIRE_DEV(packed);
  if (uv_grad_accum) { GS_REQUIRE_DEV(uv_grad_accum); GS_REQUIRE_DEV(uv_norm_sum); }
  if (grad_accum_dur) GS_REQUIRE_DEV(grad_accum_dur);
  GS_REQUIRE((long long)num_gaussians * t.start[t.n] < (1ll << 31), "too many parameters for one launch");
  const long long blocks = group_blocks(&t, num_gaussians);
  optimizer_step_kernel<true><<<(unsigned int)blocks, 256, 0, (hipStream_t)stream>>>(
      num_gaussians, t, nullptr, packed, width, b1, b2, eps, bias1, bias2, uv_norm_sum, uv_grad_accum, grad_accum_dur);
  GS_LAUNCH_CHECK();
  return GSPLAT_OK;
}

}  // extern "C"
