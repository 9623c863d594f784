// gs_loss.hip -- "next" rows f1 / f2 of SURVEY.md section 8: the fused L1 + SSIM loss that produces the
// grad_image the rasterizer backward consumes, the PSNR metric, and the element-wise Adam step.
//
// Semantics: fused_loss / compute_psnr (cuda/loss.cu:58-525) and adam_step (cuda/optimizer.cu:6-44) of the
// reference: 11-tap separable Gaussian window, clamped borders for the statistics, zero padding for the adjoint
// convolution, gradient scaled by 1/(H*W*3), NaN gradients treated as 0 by Adam.
//
// Structure: one 256-thread workgroup per 32x16 pixel tile.  The 42x26 halo tile is staged once into LDS as channel
// planes (thread = one interleaved column of one image, 26 row loads in flight, scalar row clamps), then per channel
// the horizontal pass computes four adjacent outputs per thread from three ds_read_b128 + one ds_read_b64 per plane
// (3.5 LDS words per output instead of 11) and writes stat planes, and the vertical pass computes two adjacent rows
// per thread from twelve rows of those planes (two stats per ds_read_b64).  Pitches and the lane -> (row, quad) maps
// are chosen with the LDS bank rules of MI355X_MICROARCH.md so that every wide access is conflict-free (forward) or
// two-way at worst (backward, whose nine planes leave no room for the wider pitch at three workgroups per CU).
// Results of the three channels stay in registers and leave as three contiguous floats per pixel.  Workgroups are
// dealt to the eight XCDs round-robin by the hardware; tile_of() gives each XCD one contiguous band of the image,
// so the halo rows and columns shared by neighbouring tiles hit that XCD's L2 instead of being fetched 2.1 times
// over the fabric (FETCH_SIZE 251 MB -> see profiles/).  The loss is reduced per wave on DPP and added to one of
// 256 spread counters (a single hot atomic would serialise the adds), summed by the host when it asks for the value.
#include "gs_common.h"
#include "gs_render.h"
#include "gs_math.h"

namespace {

constexpr int kTW = 32, kTH = 16, kHalo = 5, kSW = kTW + 2 * kHalo, kSH = kTH + 2 * kHalo;  // 42 x 26 halo tile
constexpr int kPitchF = 48;        // staged row pitch, forward: rows 48 words apart + row-fastest lanes = no conflicts
constexpr int kPitchB = 44;        // backward: the smallest 16-byte aligned pitch that holds 42 columns
constexpr int kPair = 68, kSingle = 36;  // row pitches of a two-stat interleaved plane / a one-stat plane
constexpr int kCols = kSW * 3;     // interleaved floats in one halo row of one image
constexpr int kSpread = 256;
static_assert(2 * kCols <= 256 && kSH <= 32 && kTW * (kTH / 2) == 256, "tile shape vs 256 threads");

// 11-tap window (cuda/loss.cu:12-17); a constexpr table so that the fully unrolled taps become literal operands
// (a v_fma with an SGPR operand issues 1.7x slower than one with a literal on this chip, profiles/microbench).
// The convolutions use explicit fmaf: the library is built with -ffp-contract=off, and mul + add would double the
// VALU instructions of kernels that are VALU-issue bound.
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

// XCD-aware tile order: workgroup id -> XCD id % 8 (hardware), so XCD x takes tiles [x * per, (x + 1) * per)
__device__ __forceinline__ bool tile_of(int ntx, int nty, int &tx, int &ty, int &t) {
  const int n = ntx * nty, per = (n + 7) >> 3;
  t = (int)(blockIdx.x & 7u) * per + (int)(blockIdx.x >> 3);
  if ((int)(blockIdx.x >> 3) >= per || t >= n) return false;
  ty = t / ntx;
  tx = t - ty * ntx;
  return true;
}

// LDS accesses that must keep their width: left to itself the compiler re-slices neighbouring reads into
// ds_read_b96 / ds_read2_b64 / ds_read2_b32 forms, which the LDS serves at half the rate of b128 / b64.
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef const volatile __attribute__((address_space(3))) v4f *lds4;
typedef const volatile __attribute__((address_space(3))) v2f *lds2;

// fourteen consecutive words of a staged row: three ds_read_b128 and one ds_read_b64
__device__ __forceinline__ void load_row14(const float *p, float (&v)[14]) {
  const lds4 q = (lds4)(__attribute__((address_space(3))) const float *)p;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const v4f t = q[k];
    v[4 * k] = t.x; v[4 * k + 1] = t.y; v[4 * k + 2] = t.z; v[4 * k + 3] = t.w;
  }
  const v2f t = *(lds2)(q + 3);
  v[12] = t.x; v[13] = t.y;
}

// two vertically adjacent 11-tap outputs of both stats of an interleaved plane (twelve ds_read_b64)
__device__ __forceinline__ void column_pair(const float *col, float (&a)[2], float (&b)[2]) {
  const lds2 q = (lds2)(__attribute__((address_space(3))) const float *)col;
  v2f v[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) v[k] = q[k * (kPair / 2)];
  a[0] = a[1] = b[0] = b[1] = 0.0f;
#pragma unroll
  for (int k = 0; k < 11; ++k) {
    a[0] = fmaf(cGauss[k], v[k].x, a[0]); a[1] = fmaf(cGauss[k], v[k + 1].x, a[1]);
    b[0] = fmaf(cGauss[k], v[k].y, b[0]); b[1] = fmaf(cGauss[k], v[k + 1].y, b[1]);
  }
}

__device__ __forceinline__ void column_single(const float *col, float (&a)[2]) {
  float v[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) v[k] = col[k * kSingle];
  a[0] = a[1] = 0.0f;
#pragma unroll
  for (int k = 0; k < 11; ++k) { a[0] = fmaf(cGauss[k], v[k], a[0]); a[1] = fmaf(cGauss[k], v[k + 1], a[1]); }
}

__global__ __launch_bounds__(256) void loss_forward_kernel(int H, int W, int ntx, int nty, float ssim_weight,
                                                           const float *__restrict__ pred,
                                                           const float *__restrict__ gt, float *__restrict__ acc,
                                                           float *__restrict__ dm_mu, float *__restrict__ dm_s1,
                                                           float *__restrict__ dm_s12) {
  __shared__ __attribute__((aligned(16))) float sR[6 * kSH * kPitchF];  // planes: pred r g b, gt r g b
  __shared__ __attribute__((aligned(16))) float sP[2 * kSH * kPair];    // horizontal pass: (X, X^2) and (Y, Y^2) planes
  __shared__ __attribute__((aligned(16))) float sQ[kSH * kSingle];      // ... and the XY plane
  const int tid = threadIdx.x;
  int tx, ty, tile;
  if (!tile_of(ntx, nty, tx, ty, tile)) return;
  const int x0 = tx * kTW, y0 = ty * kTH;
  const float C1 = (0.01f * 1.0f) * (0.01f * 1.0f), C2 = (0.03f * 1.0f) * (0.03f * 1.0f);
  if (tid < 2 * kCols) {  // borders replicate the edge pixel (cuda/loss.cu:42-47, 100-101)
    const int img = tid >= kCols, e = tid - img * kCols, px = e / 3, c = e - px * 3;
    const int gx = min(max(x0 + px - kHalo, 0), W - 1);
    const float *src = (img ? gt : pred) + (size_t)gx * 3 + c;
    float *dst = &sR[(img * 3 + c) * kSH * kPitchF + px];
    float v[kSH];
#pragma unroll
    for (int r = 0; r < kSH; ++r) v[r] = src[(size_t)min(max(y0 + r - kHalo, 0), H - 1) * W * 3];
#pragma unroll
    for (int r = 0; r < kSH; ++r) dst[r * kPitchF] = v[r];
  }
  __syncthreads();
  const int hr = (tid >> 6) * 8 + (tid & 7), hq = (tid >> 3) & 7;  // horizontal pass: row hr, outputs 4hq .. 4hq+3
  const int vx = tid & (kTW - 1), vy = (tid >> 5) * 2;             // vertical pass: column vx, output rows vy, vy+1
  float k_mu[3][2], k_s1[3][2], k_s12[3][2];
  float loss = 0.0f;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    if (hr < kSH) {
      float X[14], Y[14], o[5][4];
      load_row14(&sR[(c * kSH + hr) * kPitchF + 4 * hq], X);
      load_row14(&sR[((3 + c) * kSH + hr) * kPitchF + 4 * hq], Y);
#pragma unroll
      for (int s = 0; s < 5; ++s)
#pragma unroll
        for (int j = 0; j < 4; ++j) o[s][j] = 0.0f;
#pragma unroll
      for (int k = 0; k < 14; ++k) {
        const float x = X[k], y = Y[k], xx = x * x, yy = y * y, xy = x * y;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (k - j < 0 || k - j > 10) continue;
          const float w = cGauss[k - j];
          o[0][j] = fmaf(w, x, o[0][j]); o[1][j] = fmaf(w, xx, o[1][j]); o[2][j] = fmaf(w, y, o[2][j]);
          o[3][j] = fmaf(w, yy, o[3][j]); o[4][j] = fmaf(w, xy, o[4][j]);
        }
      }
#pragma unroll
      for (int pr = 0; pr < 2; ++pr) {
        float4 *d = reinterpret_cast<float4 *>(&sP[(pr * kSH + hr) * kPair + 8 * hq]);
        d[0] = make_float4(o[2 * pr][0], o[2 * pr + 1][0], o[2 * pr][1], o[2 * pr + 1][1]);
        d[1] = make_float4(o[2 * pr][2], o[2 * pr + 1][2], o[2 * pr][3], o[2 * pr + 1][3]);
      }
      *reinterpret_cast<float4 *>(&sQ[hr * kSingle + 4 * hq]) = make_float4(o[4][0], o[4][1], o[4][2], o[4][3]);
    }
    __syncthreads();
    {  // vertical pass + SSIM (cuda/loss.cu:150-240)
      float o[5][2];
      column_pair(&sP[(0 * kSH + vy) * kPair + 2 * vx], o[0], o[1]);
      column_pair(&sP[(1 * kSH + vy) * kPair + 2 * vx], o[2], o[3]);
      column_single(&sQ[vy * kSingle + vx], o[4]);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const float mu1 = o[0][j], mu2 = o[2][j], mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2;
        const float s1 = o[1][j] - mu1_sq, s2 = o[3][j] - mu2_sq, s12 = o[4][j] - mu1 * mu2;
        const float A = mu1_sq + mu2_sq + C1, B = s1 + s2 + C2, Cc = 2.f * mu1 * mu2 + C1, D = 2.f * s12 + C2;
        // the reference's six divisions (cuda/loss.cu:214-236) share two reciprocals: 1/A and 1/B
        const float iA = __builtin_amdgcn_rcpf(A), iB = __builtin_amdgcn_rcpf(B), iAB = iA * iB;
        const float ssim = Cc * D * iAB;
        const int ti = (vy + j + kHalo) * kPitchF + vx + kHalo;
        const float l1 = fabsf(sR[c * kSH * kPitchF + ti] - sR[(3 + c) * kSH * kPitchF + ti]);
        const bool inside = x0 + vx < W && y0 + vy + j < H;
        loss += inside ? (1.0f - ssim_weight) * l1 + ssim_weight * (1.0f - ssim) : 0.0f;
        const float two_mu1_ssim = 2.f * mu1 * ssim;
        const float d_mu1 = 2.f * mu2 * (D - Cc) * iAB - two_mu1_ssim * iA + two_mu1_ssim * iB;
        k_mu[c][j] = -ssim_weight * d_mu1;
        k_s1[c][j] = ssim_weight * (ssim * iB);
        k_s12[c][j] = -ssim_weight * (2.f * Cc * iAB);
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int px = x0 + vx, py = y0 + vy + j;
    if (px < W && py < H) {
      const size_t id = ((size_t)py * W + px) * 3;
#pragma unroll
      for (int c = 0; c < 3; ++c) { dm_mu[id + c] = k_mu[c][j]; dm_s1[id + c] = k_s1[c][j]; dm_s12[id + c] = k_s12[c][j]; }
    }
  }
  loss = wave_sum(loss);
  if ((tid & 63) == 0) atomicAdd(&acc[(tile * 4 + (tid >> 6)) & (kSpread - 1)], loss);
}

__global__ __launch_bounds__(256) void loss_backward_kernel(int H, int W, int ntx, int nty, float ssim_weight,
                                                            const float *__restrict__ pred,
                                                            const float *__restrict__ gt,
                                                            const float *__restrict__ dm_mu,
                                                            const float *__restrict__ dm_s1,
                                                            const float *__restrict__ dm_s12,
                                                            float *__restrict__ image_grad,
                                                            float *__restrict__ next_acc) {
  __shared__ __attribute__((aligned(16))) float sD[9 * kSH * kPitchB];  // plane = channel * 3 + map, zero outside the image
  __shared__ __attribute__((aligned(16))) float sP[kSH * kPair];        // horizontal pass: maps (mu, s1) interleaved
  __shared__ __attribute__((aligned(16))) float sQ[kSH * kSingle];      // ... and s12
  const int tid = threadIdx.x;
  int tx, ty, tile;
  if (!tile_of(ntx, nty, tx, ty, tile)) return;
  if (tile == 0) next_acc[tid] = 0.0f;  // the spread counters of the NEXT fused_loss call (kSpread == 256 threads)
  const int x0 = tx * kTW, y0 = ty * kTH;
  const int vx = tid & (kTW - 1), vy = (tid >> 5) * 2;
  float p1[2][3], p2[2][3];  // the two pixels this thread finishes
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int px = x0 + vx, py = y0 + vy + j;
    const bool inside = px < W && py < H;
    const size_t id = inside ? ((size_t)py * W + px) * 3 : 0;
#pragma unroll
    for (int c = 0; c < 3; ++c) { p1[j][c] = pred[id + c]; p2[j][c] = gt[id + c]; }
  }
  unsigned rows_in = 0;  // bit r: halo row r lies inside the image (uniform)
#pragma unroll
  for (int r = 0; r < kSH; ++r) rows_in |= (unsigned)(y0 + r - kHalo >= 0 && y0 + r - kHalo < H) << r;
  for (int it = tid; it < 3 * kCols; it += 256) {  // adjoint convolution pads with zeros (cuda/loss.cu:49-55, 339-341)
    const int m = it / kCols, e = it - m * kCols, px = e / 3, c = e - px * 3;
    const int gx = x0 + px - kHalo;
    const bool in_x = gx >= 0 && gx < W;
    const unsigned keep = in_x ? rows_in : 0u;
    const float *src = (m == 0 ? dm_mu : m == 1 ? dm_s1 : dm_s12) + (in_x ? (size_t)gx * 3 + c : 0);
    float *dst = &sD[(c * 3 + m) * kSH * kPitchB + px];
    float v[kSH];
#pragma unroll
    for (int r = 0; r < kSH; ++r)  // always a valid address, the value is dropped outside: no branches around loads
      v[r] = src[(size_t)min(max(y0 + r - kHalo, 0), H - 1) * W * 3];
#pragma unroll
    for (int r = 0; r < kSH; ++r) dst[r * kPitchB] = (keep >> r & 1u) ? v[r] : 0.0f;
  }
  __syncthreads();
  // horizontal pass lanes: row bits (l0 l2 l1), quad bits (l3 l5 l4) -- the best this pitch allows for ds_read_b128
  const int hr = (tid >> 6) * 8 + ((tid & 1) | ((tid >> 1) & 2) | ((tid << 1) & 4));
  const int hq = ((tid >> 3) & 1) | ((tid >> 4) & 2) | ((tid >> 2) & 4);
  const float grad_scale = 1.0f / (float)(H * W * 3);
  float out[2][3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    if (hr < kSH) {
      float o[3][4];
#pragma unroll
      for (int m = 0; m < 3; ++m) {
        float D[14];
        load_row14(&sD[((c * 3 + m) * kSH + hr) * kPitchB + 4 * hq], D);
#pragma unroll
        for (int j = 0; j < 4; ++j) o[m][j] = 0.0f;
#pragma unroll
        for (int k = 0; k < 14; ++k)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (k - j >= 0 && k - j <= 10) o[m][j] = fmaf(cGauss[k - j], D[k], o[m][j]);
      }
      float4 *d = reinterpret_cast<float4 *>(&sP[hr * kPair + 8 * hq]);
      d[0] = make_float4(o[0][0], o[1][0], o[0][1], o[1][1]);
      d[1] = make_float4(o[0][2], o[1][2], o[0][3], o[1][3]);
      *reinterpret_cast<float4 *>(&sQ[hr * kSingle + 4 * hq]) = make_float4(o[2][0], o[2][1], o[2][2], o[2][3]);
    }
    __syncthreads();
    {
      float s[3][2];
      column_pair(&sP[vy * kPair + 2 * vx], s[0], s[1]);
      column_single(&sQ[vy * kSingle + vx], s[2]);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const float ssim_g = s[0][j] + (2.f * p1[j][c]) * s[1][j] + p2[j][c] * s[2][j];
        const float l1_g = (1.0f - ssim_weight) * ((p1[j][c] > p2[j][c]) ? 1.0f : -1.0f);
        out[j][c] = (ssim_g + l1_g) * grad_scale;
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int px = x0 + vx, py = y0 + vy + j;
    if (px < W && py < H) {
      const size_t id = ((size_t)py * W + px) * 3;
#pragma unroll
      for (int c = 0; c < 3; ++c) image_grad[id + c] = out[j][c];
    }
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
  float pv = *param, mv = *m, vv = *v;
  gs::adam_values(pv, mv, vv, g, lr, b1, b2, eps, bias1, bias2);  // (gs_math.h: shared with the backward that applies Adam itself)
  *param = pv;
  *m = mv;
  *v = vv;
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

// The SH group of a single-view step with its gradients rebuilt in place of read: thread i owns coefficient k = i % (n - 1)
// of row i / (n - 1), i.e. three consecutive elements (r, g, b) of the [rows, 3 (n - 1)] group -- neighbouring threads
// touch neighbouring 12-byte pieces of the parameter and moment rows.  A gradient is grad_precompute_rgb[row][c] *
// Y_{k+1}(direction of the row): gs::sh_bwd's own product (gs_math.h) on gs::view_dir / gs::sh_basis of the same inputs.
// The n - 1 threads of a row evaluate the same basis (the kernel is bound by its six parameter / moment streams).
template <int L>
__global__ __launch_bounds__(256) void optimizer_sh_factored_kernel(int rows, const int *__restrict__ c2g,
                                                                    float *__restrict__ sh, float *__restrict__ m,
                                                                    float *__restrict__ v, float lr, float b1, float b2,
                                                                    float eps, float bias1, float bias2,
                                                                    const float *__restrict__ xyz, float cx, float cy,
                                                                    float cz, const float *__restrict__ g_rgb) {
  constexpr int n = (L + 1) * (L + 1), kCoef = n - 1;
  const unsigned int i = blockIdx.x * 256u + threadIdx.x;
  if (i >= (unsigned int)rows * (unsigned int)kCoef) return;
  const unsigned int r = i / (unsigned int)kCoef, k = i - r * (unsigned int)kCoef;
  const long long row = c2g[r];
  float ux, uy, uz, len;
  gs::view_dir(xyz[3 * row], xyz[3 * row + 1], xyz[3 * row + 2], cx, cy, cz, ux, uy, uz, len);
  float Y[n];
  gs::sh_basis<L>(ux, uy, uz, Y);
  float yv = 0.0f;
#pragma unroll
  for (int q = 0; q < kCoef; ++q) yv = k == (unsigned int)q ? Y[q + 1] : yv;
  const long long o = (row * kCoef + k) * 3;
#pragma unroll
  for (int c = 0; c < 3; ++c)
    adam_update(&sh[o + c], &m[o + c], &v[o + c], g_rgb[3 * r + c] * yv, lr, b1, b2, eps, bias1, bias2);
}

// r05: the colour groups (band 0 and the SH coefficients) of a W-VIEW step, from what the split exchange delivers: every
// view's g_rgb[N,3] in global order + its camera position (rgb_all) and the number of views that saw a gaussian
// (common[., 11]).  Phase 1, one thread per gaussian: grad[k][c] = sum_r g_rgb^r[c] Y_k(dir^r) exactly as
// unpack_split_kernel (gs_fused.hip) forms it -- views in rank order, views with an all-zero g_rgb skipped, mul then add --
// W basis evaluations per gaussian, into the wave's LDS rows (odd pitch: a lane writing ITS row is conflict-free).
// Phase 2, the wave streams over its 64 consecutive parameter rows as ONE linear span (the rows of rgb[N,3] and of
// sh[N,3(n-1)] are contiguous in global order): parameter and both moments are read and written coalesced, the gradient
// comes from LDS.  Nothing of packed[N, 12 + 3 n] is written or read: 240 B per gaussian each way at SH degree 3.
template <int L>
__global__ __launch_bounds__(256) void optimizer_sh_views_kernel(int N, int world, const float *__restrict__ xyz,
                                                                 const float *__restrict__ rgb_all, size_t stride,
                                                                 const float *__restrict__ common,
                                                                 float *__restrict__ rgb, float *__restrict__ rgb_m,
                                                                 float *__restrict__ rgb_v, float lr_rgb,
                                                                 float *__restrict__ sh, float *__restrict__ sh_m,
                                                                 float *__restrict__ sh_v, float lr_sh, float b1, float b2,
                                                                 float eps, float bias1, float bias2) {
  constexpr int n = (L + 1) * (L + 1), kCols = 3 * n, kPitch = kCols | 1, kRest = 3 * (n - 1);
  __shared__ float s_g[256 * kPitch];
  const int lane = threadIdx.x & 63, wave_first = threadIdx.x - lane;
  const int iw = blockIdx.x * 256 + wave_first;  // first gaussian of this wave
  if (iw >= N) return;
  const int i = iw + lane;
  float *mine = s_g + (wave_first + lane) * kPitch;
  bool vis = false;
  if (i < N) {
    vis = common[(size_t)i * 12 + 11] > 0.0f;
    float acc[n][3];
#pragma unroll
    for (int k = 0; k < n; ++k) acc[k][0] = acc[k][1] = acc[k][2] = 0.0f;
    if (vis) {
      const float px = xyz[3 * i], py = xyz[3 * i + 1], pz = xyz[3 * i + 2];
      for (int r = 0; r < world; ++r) {
        const float *blk = rgb_all + (size_t)r * stride;
        const float g0 = blk[3 * (size_t)i], g1 = blk[3 * (size_t)i + 1], g2 = blk[3 * (size_t)i + 2];
        if (g0 == 0.0f && g1 == 0.0f && g2 == 0.0f) continue;
        const float *cp = blk + 3 * (size_t)N;
        float dx, dy, dz, len, Y[n];
        gs::view_dir(px, py, pz, cp[0], cp[1], cp[2], dx, dy, dz, len);
        gs::sh_basis<L>(dx, dy, dz, Y);
#pragma unroll
        for (int k = 0; k < n; ++k) { acc[k][0] += g0 * Y[k]; acc[k][1] += g1 * Y[k]; acc[k][2] += g2 * Y[k]; }
      }
    }
#pragma unroll
    for (int k = 0; k < n; ++k) { mine[3 * k] = acc[k][0]; mine[3 * k + 1] = acc[k][1]; mine[3 * k + 2] = acc[k][2]; }
  }
  const unsigned long long seen = __ballot(vis);
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  if (seen == 0ull) return;
  const int rows = min(64, N - iw);
  const float *wrows = s_g + wave_first * kPitch;
  {  // band 0: rgb[iw .. iw + rows), three floats per row
    float *p = rgb + (size_t)iw * 3, *m = rgb_m + (size_t)iw * 3, *v = rgb_v + (size_t)iw * 3;
    for (int e = lane; e < rows * 3; e += 64) {
      const int r = e / 3, c = e - r * 3;
      if ((seen >> r) & 1ull) adam_update(p + e, m + e, v + e, wrows[r * kPitch + c], lr_rgb, b1, b2, eps, bias1, bias2);
    }
  }
  if constexpr (kRest > 0) {
    float *p = sh + (size_t)iw * kRest, *m = sh_m + (size_t)iw * kRest, *v = sh_v + (size_t)iw * kRest;
#pragma unroll 5
    for (int e = lane; e < rows * kRest; e += 64) {
      const int r = e / kRest, c = e - r * kRest;
      if ((seen >> r) & 1ull) adam_update(p + e, m + e, v + e, wrows[r * kPitch + 3 + c], lr_sh, b1, b2, eps, bias1, bias2);
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
  gs::ScratchLock lock;  // scratch and the counters' state are process-wide; held through the read-back
  gs::DeviceBuffer &mu = gs::scratch(gs::SCR_LOSS_MU), &s1 = gs::scratch(gs::SCR_LOSS_S1),
                   &s12 = gs::scratch(gs::SCR_LOSS_S12), &acc = gs::scratch(gs::SCR_LOSS_ACC);
  int rc;
  if ((rc = mu.reserve(bytes)) || (rc = s1.reserve(bytes)) || (rc = s12.reserve(bytes)) ||
      (rc = acc.reserve(3 * kSpread * sizeof(float))))
    return rc;
  // the spread counters alternate between two sets: the backward kernel of one call clears the set of the next, which
  // saves a memset launch per training iteration (the first set of the buffer belongs to gsplat_compute_psnr).  What is
  // known about the counters' contents lives in `primed`: it names the stream and the allocation (generation: a
  // released and re-reserved buffer may come back at the same address) the last call left them clean on, and it is
  // void while a call is between its two launches -- a call that errors out in between leaves the next one to clear.
  struct Primed { bool ok = false; hipStream_t stream = nullptr; void *ptr = nullptr; unsigned long long gen = 0; int flip = 0; };
  static Primed primed;
  if (!primed.ok || primed.stream != st || primed.ptr != acc.ptr || primed.gen != acc.generation) {
    GS_HIP(hipMemsetAsync(acc.as<float>() + kSpread, 0, 2 * kSpread * sizeof(float), st));
    primed.stream = st; primed.ptr = acc.ptr; primed.gen = acc.generation; primed.flip = 0;
  }
  primed.ok = false;
  const int flip = primed.flip;
  float *cur = acc.as<float>() + (1 + flip) * kSpread, *next = acc.as<float>() + (2 - flip) * kSpread;
  const int ntx = (cols + kTW - 1) / kTW, nty = (rows + kTH - 1) / kTH;
  const dim3 grid((unsigned)(((ntx * nty + 7) / 8) * 8)), block(256);  // whole rounds of the eight XCDs: tile_of()
  loss_forward_kernel<<<grid, block, 0, st>>>(rows, cols, ntx, nty, ssim_weight, predicted_data, gt_data, cur,
                                              mu.as<float>(), s1.as<float>(), s12.as<float>());
  GS_LAUNCH_CHECK();
  loss_backward_kernel<<<grid, block, 0, st>>>(rows, cols, ntx, nty, ssim_weight, predicted_data, gt_data,
                                               mu.as<float>(), s1.as<float>(), s12.as<float>(), image_grad, next);
  GS_LAUNCH_CHECK();
  primed.flip = flip ^ 1;
  primed.ok = true;  // both launches are queued: `next` will be clean for the next call on this stream
  if (loss_out) {  // the reference returns the value, i.e. blocks (cuda/loss.cu:468-470)
    double total;
    if ((rc = read_spread_sum(cur, st, &total))) return rc;
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
  gs::ScratchLock lock;
  gs::DeviceBuffer &acc = gs::scratch(gs::SCR_LOSS_ACC);
  int rc = acc.reserve(3 * kSpread * sizeof(float));  // [0, kSpread): this function's; the rest: gsplat_fused_loss
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

int gsplat_optimizer_step_sh_factored(const int *compact_to_global, int num_culled, int l_max, float *sh, float *exp_avg,
                                      float *exp_avg_sq, float lr, float b1, float b2, float eps, float bias1,
                                      float bias2, const float *xyz, float cam_x, float cam_y, float cam_z,
                                      const float *grad_precompute_rgb, void *stream) {
  GS_REQUIRE(num_culled >= 0, "negative gaussian count");
  GS_REQUIRE(l_max >= 0 && l_max <= 3, "l_max must be 0..3");
  if (num_culled == 0 || l_max == 0) return GSPLAT_OK;  // no coefficients beyond band 0
  GS_REQUIRE_DEV(compact_to_global); GS_REQUIRE_DEV(sh); GS_REQUIRE_DEV(exp_avg); GS_REQUIRE_DEV(exp_avg_sq);
  GS_REQUIRE_DEV(xyz); GS_REQUIRE_DEV(grad_precompute_rgb);
  const long long per_row = (l_max + 1) * (l_max + 1) - 1;  // threads per row: one per coefficient (three channels each)
  GS_REQUIRE((long long)num_culled * per_row * 3 < (1ll << 31), "too many parameters for one launch");
  const unsigned int blocks = (unsigned int)(((long long)num_culled * per_row + 255) / 256);
  hipStream_t st = (hipStream_t)stream;
#define GS_SHF(LL)                                                                                                     \
  optimizer_sh_factored_kernel<LL><<<blocks, 256, 0, st>>>(num_culled, compact_to_global, sh, exp_avg, exp_avg_sq, lr, b1, \
                                                          b2, eps, bias1, bias2, xyz, cam_x, cam_y, cam_z, grad_precompute_rgb)
  switch (l_max) {
    case 1: GS_SHF(1); break;
    case 2: GS_SHF(2); break;
    default: GS_SHF(3); break;
  }
#undef GS_SHF
  GS_LAUNCH_CHECK();
  return GSPLAT_OK;
}

int gsplat_optimizer_step_sh_views(int l_max, int num_gaussians, int world_size, const float *xyz, const float *rgb_all,
                                   size_t rank_stride, const float *common, float *rgb, float *rgb_exp_avg,
                                   float *rgb_exp_avg_sq, float lr_rgb, float *sh, float *sh_exp_avg,
                                   float *sh_exp_avg_sq, float lr_sh, float b1, float b2, float eps, float bias1,
                                   float bias2, void *stream) {
  GS_REQUIRE(num_gaussians >= 0 && world_size >= 1, "bad sizes");
  GS_REQUIRE(l_max >= 0 && l_max <= 3, "l_max must be 0..3");
  if (num_gaussians == 0) return GSPLAT_OK;
  GS_REQUIRE_DEV(xyz); GS_REQUIRE_DEV(rgb_all); GS_REQUIRE_DEV(common);
  GS_REQUIRE_DEV(rgb); GS_REQUIRE_DEV(rgb_exp_avg); GS_REQUIRE_DEV(rgb_exp_avg_sq);
  if (l_max > 0) { GS_REQUIRE_DEV(sh); GS_REQUIRE_DEV(sh_exp_avg); GS_REQUIRE_DEV(sh_exp_avg_sq); }
  GS_REQUIRE(rank_stride >= 3 * (size_t)num_gaussians + 3, "rank_stride must cover [N,3] g_rgb + campos[3]");
  const unsigned int blocks = gs::div_up(num_gaussians, 256);
  hipStream_t st = (hipStream_t)stream;
#define GS_SHV(LL)                                                                                                     \
  optimizer_sh_views_kernel<LL><<<blocks, 256, 0, st>>>(num_gaussians, world_size, xyz, rgb_all, rank_stride, common, rgb, \
                                                       rgb_exp_avg, rgb_exp_avg_sq, lr_rgb, sh, sh_exp_avg, sh_exp_avg_sq, \
                                                       lr_sh, b1, b2, eps, bias1, bias2)
  switch (l_max) {
    case 0: GS_SHV(0); break;
    case 1: GS_SHV(1); break;
    case 2: GS_SHV(2); break;
    default: GS_SHV(3); break;
  }
#undef GS_SHV
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
